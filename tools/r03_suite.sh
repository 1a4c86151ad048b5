#!/bin/bash
# full GPU suite + bench line (round-3 checkpoint)
set -u
TAG=${1:-r03d}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
python bench.py > "$OUT/${TAG}_bench_b4_1024.json.log" 2>&1
tail -2 "$OUT/${TAG}_bench_b4_1024.json.log" | cut -c1-1500
timeout 2400 python -m pytest tests -m gpu -x -q > "$OUT/${TAG}_gputest.log" 2>&1
tail -8 "$OUT/${TAG}_gputest.log"
