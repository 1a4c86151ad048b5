#!/bin/bash
# GEMM tests, bench line, per-kernel stats of the bench
set -u
TAG=${1:-r03e}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gemm_w4_gpu.py tests/test_ops_gpu.py -x -q > "$OUT/${TAG}_test_gemm.log" 2>&1
tail -5 "$OUT/${TAG}_test_gemm.log"
python bench.py --no-cpu-baseline > "$OUT/${TAG}_bench_b4_1024.json.log" 2>&1
tail -1 "$OUT/${TAG}_bench_b4_1024.json.log" | cut -c1-700
X2I_GEMM_PERSIST=0 python bench.py --no-cpu-baseline > "$OUT/${TAG}_bench_b4_1024_nopersist.json.log" 2>&1
tail -1 "$OUT/${TAG}_bench_b4_1024_nopersist.json.log" | cut -c1-400
X2I_GEMM_W4=0 python bench.py --no-cpu-baseline > "$OUT/${TAG}_bench_b4_1024_8wave.json.log" 2>&1
tail -1 "$OUT/${TAG}_bench_b4_1024_8wave.json.log" | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/stats.log" 2>&1
cp "$OUT"/stats/*/s_kernel_stats.csv "$OUT/${TAG}_bench_b4_1024_kernel_stats.csv" 2>/dev/null || cp "$OUT"/stats/s_kernel_stats.csv "$OUT/${TAG}_bench_b4_1024_kernel_stats.csv"
rm -rf "$OUT/stats"
cut -c1-200 "$OUT/${TAG}_bench_b4_1024_kernel_stats.csv" | head -16
