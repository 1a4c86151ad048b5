#!/bin/bash
# round-3 check: the attention / fused-QKV tests, the full-scale parity tests, the attention micro-bench and the bench line
set -u
TAG=${1:-r03u}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_gemm_w4_gpu.py -x -q -s -k "attention or qkv" > "$OUT/${TAG}_test_attn_qkv.log" 2>&1
grep -v amdgpu.ids "$OUT/${TAG}_test_attn_qkv.log" | tail -12
timeout 900 python -m pytest tests/test_fullscale_parity_gpu.py -x -q -s > "$OUT/${TAG}_test_fullscale.log" 2>&1
grep -v amdgpu.ids "$OUT/${TAG}_test_fullscale.log" | tail -8
timeout 300 python tools/attn_bench.py 4 2>/dev/null | grep -v amdgpu.ids | tail -4 | tee "$OUT/${TAG}_attn_bench.log"
timeout 900 python bench.py --no-fp8-lines 2>/dev/null | tail -1 | tee "$OUT/${TAG}_bench.json" | cut -c1-600
