#!/bin/bash
# rocprofv3 --kernel-trace --stats of the attention backward passes one after the other (GPU box): tools/bwd_trace.sh [B] -> gpurun_out/bwd_trace_kernel_stats.csv
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
D=$(mktemp -d)
X2I_ATTN_BWD_OVERLAP=0 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -o s -- python "$R/tools/attn_bench.py" ${1:-2} bwd > "$R/gpurun_out/bwd_trace.log" 2>&1
cp "$D"/*/s_kernel_stats.csv "$R/gpurun_out/bwd_trace_kernel_stats.csv" 2>/dev/null || cp "$D"/s_kernel_stats.csv "$R/gpurun_out/bwd_trace_kernel_stats.csv"
head -7 "$R/gpurun_out/bwd_trace_kernel_stats.csv" | cut -c1-60,200-330
