#!/bin/bash
# first run of the 4-wave hand-scheduled GEMM: parity, then timing beside the 8-wave kernel and the vendor library
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r03c
mkdir -p "$OUT"
cd "$R"
timeout 900 python -m pytest tests/test_gemm_w4_gpu.py -x -q > "$OUT/r03c_test_w4.log" 2>&1
tail -15 "$OUT/r03c_test_w4.log"
timeout 600 python tools/vendor_gemm_probe.py 4 > "$OUT/r03c_vendor_gemm_probe.log" 2>&1
grep shape "$OUT/r03c_vendor_gemm_probe.log" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-15s M=%6d N=%6d K=%6d vendor %7.1f persist %7.1f w4 %7.1f w8 %7.1f tile %d err %.1e' % (d['shape'], d['M'], d['N'], d['K'], d['vendor_TF_median'], d['x2i_TF_median'], d['x2i_w4_onetile_TF_median'], d['x2i_8wave_TF_median'], d['x2i_tile'], d['rel_l2_vs_vendor']))
"
tail -3 "$OUT/r03c_vendor_gemm_probe.log"
