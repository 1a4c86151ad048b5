#!/bin/bash
set -u
TAG=${1:-r03l}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-fp8-lines > "$OUT/stats.log" 2>&1
cp "$OUT"/stats/*/s_kernel_stats.csv "$OUT/${TAG}_bench_b4_1024_kernel_stats.csv" 2>/dev/null || cp "$OUT"/stats/s_kernel_stats.csv "$OUT/${TAG}_bench_b4_1024_kernel_stats.csv"
rm -rf "$OUT/stats"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats2" -o s -- python "$R/tools/roofline_probe.py" > "$OUT/${TAG}_roofline_probe.json.log" 2>&1
cp "$OUT"/stats2/*/s_kernel_stats.csv "$OUT/${TAG}_roofline_probe_kernel_stats.csv" 2>/dev/null || cp "$OUT"/stats2/s_kernel_stats.csv "$OUT/${TAG}_roofline_probe_kernel_stats.csv"
rm -rf "$OUT/stats2"
cut -c1-220 "$OUT/${TAG}_bench_b4_1024_kernel_stats.csv" | head -14
cut -c1-220 "$OUT/${TAG}_roofline_probe_kernel_stats.csv" | head -6
tail -1 "$OUT/${TAG}_roofline_probe.json.log" | cut -c1-400
bash "$R/tools/pmc_roofline.sh" "$OUT/pmc" > "$OUT/pmc.log" 2>&1
tail -30 "$OUT/pmc.log"
cp "$OUT/pmc/r03_pmc_roofline.json" "$OUT/"
cp "$OUT/pmc/summary.json" "$OUT/${TAG}_pmc_roofline_summary.json"
rm -rf "$OUT/pmc"
