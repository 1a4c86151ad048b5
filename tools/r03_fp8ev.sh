#!/bin/bash
set -u
TAG=${1:-r03k}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
timeout 1500 python -m pytest tests/test_fullscale_parity_gpu.py -x -q -s -k "four_step or full_depth_19" > "$OUT/${TAG}_fp8_evidence.log" 2>&1
grep -i "full-depth\|passed\|failed\|error" "$OUT/${TAG}_fp8_evidence.log" | tail -12
