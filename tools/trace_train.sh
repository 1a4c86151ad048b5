#!/bin/bash
# rocprofv3 --kernel-trace --stats of the distillation step (GPU box): tools/trace_train.sh <tag> [B] -> gpurun_out/<tag>_kernel_stats.csv
TAG=${1:-train}; B=${2:-1}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
D=$(mktemp -d)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -o s -- python "$R/tools/train_bench.py" $B 3 > "$R/gpurun_out/$TAG.log" 2>&1
cp "$D"/*/s_kernel_stats.csv "$R/gpurun_out/${TAG}_kernel_stats.csv" 2>/dev/null || cp "$D"/s_kernel_stats.csv "$R/gpurun_out/${TAG}_kernel_stats.csv"
rm -rf "$D"
