#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py configuration (GPU box):  tools/trace_bench.sh <tag> <bench.py arguments...>
# -> gpurun_out/<tag>_kernel_stats.csv (+ the bench line in gpurun_out/<tag>.json.log)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
D=$(mktemp -d)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -o s -- python "$R/bench.py" "$@" --no-cpu-baseline --no-fp8-lines --no-roofline > "$OUT/$TAG.json.log" 2>&1
cp "$D"/*/s_kernel_stats.csv "$OUT/${TAG}_kernel_stats.csv" 2>/dev/null || cp "$D"/s_kernel_stats.csv "$OUT/${TAG}_kernel_stats.csv"
rm -rf "$D"
