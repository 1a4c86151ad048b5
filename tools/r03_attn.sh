#!/bin/bash
set -u
TAG=${1:-r03n}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" > "$OUT/${TAG}_test_attn.log" 2>&1
tail -25 "$OUT/${TAG}_test_attn.log"
timeout 300 python tools/attn_bench.py 4 > "$OUT/${TAG}_attn_bench.log" 2>&1
tail -12 "$OUT/${TAG}_attn_bench.log"
