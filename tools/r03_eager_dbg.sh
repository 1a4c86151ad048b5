#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03g
mkdir -p "$OUT"
cd "$R"
run() { name=$1; shift; echo "== $name"; ( "$@" ) > "$OUT/r03g_$name.log" 2>&1; tail -2 "$OUT/r03g_$name.log" | cut -c1-400; }
run b4_1_1 timeout 300 python tools/eager_gpu_baseline.py 4 --blocks 1 1
run b4_2_0 timeout 300 python tools/eager_gpu_baseline.py 4 --blocks 2 0
run b4_0_2 timeout 300 python tools/eager_gpu_baseline.py 4 --blocks 0 2
run b2_full timeout 600 python tools/eager_gpu_baseline.py 2
run b4_full_rocblas env TORCH_BLAS_PREFER_HIPBLASLT=0 timeout 600 python tools/eager_gpu_baseline.py 4
