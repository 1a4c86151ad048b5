#!/bin/bash
# rocprofv3 --kernel-trace --stats of tools/vae_bench.py (GPU box):  tools/trace_vae.sh <tag>   (environment switches of x2i_amd/vae.py pass through)
# -> gpurun_out/<tag>_kernel_stats.csv
set -u
TAG=$1
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
D=$(mktemp -d)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -o s -- python "$R/tools/vae_bench.py" > "$OUT/$TAG.log" 2>&1
cp "$D"/*/s_kernel_stats.csv "$OUT/${TAG}_kernel_stats.csv" 2>/dev/null || cp "$D"/s_kernel_stats.csv "$OUT/${TAG}_kernel_stats.csv"
rm -rf "$D"
