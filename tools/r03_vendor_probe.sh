#!/bin/bash
# Round 3, first GPU call: which Tensile kernel does the vendor library run on the DiT shapes, and the whole-step eager comparator.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03a
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_vendor" -o s -- python "$R/tools/vendor_gemm_probe.py" 4 > "$OUT/r03a_vendor_gemm_probe.log" 2>&1
cp "$OUT"/stats_vendor/*/s_kernel_stats.csv "$OUT/r03a_vendor_gemm_probe_kernel_stats.csv" 2>/dev/null || cp "$OUT"/stats_vendor/s_kernel_stats.csv "$OUT/r03a_vendor_gemm_probe_kernel_stats.csv"
rm -rf "$OUT/stats_vendor"
cd "$R"
timeout 900 python tools/eager_gpu_baseline.py 4 > "$OUT/r03a_eager_gpu_baseline.log" 2>&1
tail -3 "$OUT/r03a_eager_gpu_baseline.log"
cat "$OUT/r03a_vendor_gemm_probe.log" | tail -12
cut -c1-260 "$OUT/r03a_vendor_gemm_probe_kernel_stats.csv" | head -30
