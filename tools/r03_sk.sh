#!/bin/bash
set -u
TAG=${1:-r03h}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
timeout 600 python -m pytest tests/test_gemm_w4_gpu.py -x -q > "$OUT/${TAG}_test_w4.log" 2>&1
tail -12 "$OUT/${TAG}_test_w4.log"
timeout 600 python tools/vendor_gemm_probe.py 4 > "$OUT/${TAG}_vendor_gemm_probe.log" 2>&1
grep shape "$OUT/${TAG}_vendor_gemm_probe.log" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-15s M=%6d N=%6d K=%6d vendor %7.1f x2i %7.1f w4-onetile %7.1f w8 %7.1f tile %d err %.1e' % (d['shape'], d['M'], d['N'], d['K'], d['vendor_TF_median'], d['x2i_TF_median'], d['x2i_w4_onetile_TF_median'], d['x2i_8wave_TF_median'], d['x2i_tile'], d['rel_l2_vs_vendor']))
"
timeout 600 python bench.py --no-cpu-baseline --no-fp8-lines > "$OUT/${TAG}_bench.json.log" 2>&1
tail -1 "$OUT/${TAG}_bench.json.log" | cut -c1-330
X2I_GEMM_STREAMK=0 timeout 600 python bench.py --no-cpu-baseline --no-fp8-lines > "$OUT/${TAG}_bench_nosk.json.log" 2>&1
tail -1 "$OUT/${TAG}_bench_nosk.json.log" | cut -c1-330
