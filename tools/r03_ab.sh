#!/bin/bash
set -u
TAG=${1:-r03j}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
timeout 600 python tools/gemm_sched_ab.py > "$OUT/${TAG}_gemm_sched_ab.log" 2>&1
tail -6 "$OUT/${TAG}_gemm_sched_ab.log"
timeout 900 python -m pytest tests/test_train_gpu.py -x -q -k "graphed or resumes or ping_pong or harness" > "$OUT/${TAG}_test_misc.log" 2>&1
tail -5 "$OUT/${TAG}_test_misc.log"
