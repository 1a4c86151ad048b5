#!/bin/bash
# rocprofv3 PMC passes over tools/roofline_probe.py (exactly the two launches bench.py's `roofline` times): FETCH_SIZE / WRITE_SIZE /
# L2 hit rate / SQ busy counters, each in its OWN pass with --kernel-trace only.  Writes <out>/summary.json and
# <out>/pmc_roofline.json (copied to profiles/<tag>_pmc_roofline.json, which bench.py reads) (traffic per launch pair, FETCH_SIZE doubled per the gfx950 correction).
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=${1:-$R/gpurun_out/pmc_roofline}
case "$OUT" in /*) ;; *) OUT="$R/$OUT" ;; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  n=$(echo $c | tr " " "_" | cut -c1-40)
  timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$n" -o p -- python "$R/tools/roofline_probe.py" > "$OUT/$n.log" 2>&1
done
python "$R/tools/pmc_summarise.py" "$OUT" > "$OUT/summary.json"
python3 - "$OUT" <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + "/summary.json"))
ks = {k: v for k, v in d.items() if "gemm" in k}
tot = sum(v.get("hbm_read_bytes_corrected", 0.0) + v.get("hbm_write_bytes", 0.0) for v in ks.values())
out = {"traffic_bytes_per_pair": tot, "kernels": ks,
       "note": "rocprofv3 --pmc FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate passes over tools/roofline_probe.py; per-launch means "
               "summed over the GEMM kernels of the pair (bytes at the L2<->fabric boundary)"}
json.dump(out, open(sys.argv[1] + "/pmc_roofline.json", "w"), indent=1, sort_keys=True)
print(json.dumps({k: {c: round(x) for c, x in v.items()} for k, v in ks.items()}, indent=1)[:3000])
print("traffic_bytes_per_pair", tot)
PY
