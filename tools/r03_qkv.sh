#!/bin/bash
set -u
TAG=${1:-r03m}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gemm_w4_gpu.py tests/test_ops_gpu.py -x -q > "$OUT/${TAG}_test_gemm.log" 2>&1
tail -6 "$OUT/${TAG}_test_gemm.log"
timeout 600 python bench.py --no-cpu-baseline --no-fp8-lines > "$OUT/${TAG}_bench.json.log" 2>&1
tail -1 "$OUT/${TAG}_bench.json.log" | cut -c1-330
X2I_GEMM_PERSIST=0 timeout 600 python bench.py --no-cpu-baseline --no-fp8-lines > "$OUT/${TAG}_bench_nopersist.json.log" 2>&1
tail -1 "$OUT/${TAG}_bench_nopersist.json.log" | cut -c1-330
