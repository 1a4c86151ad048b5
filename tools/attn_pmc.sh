#!/bin/bash
# PMC passes over the attention kernels (variant list in tools/attn_bench.py); counters in their own passes, kernel-trace only
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=${1:-$R/gpurun_out/attn_pmc}
case "$OUT" in /*) ;; *) OUT="$R/$OUT" ;; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/p$i" -o p -- python "$R/tools/attn_bench.py" 4 > "$OUT/p$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/p_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "attn" not in n:
            continue
        key = ("w16" if "attn_w16" in n else "w4" if "attn_w4" in n else "attn16" if "attn16" in n else "pp" if "attn_pp" in n else "fwd4")
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
for k, e in out.items():
    if "SQ_WAVE_CYCLES" in e:
        w = e["SQ_WAVE_CYCLES"]
        e["frac_wait_any"] = e.get("SQ_WAIT_ANY", 0) / w
        e["frac_wait_inst"] = e.get("SQ_WAIT_INST_ANY", 0) / w
        e["frac_active"] = e.get("SQ_ACTIVE_INST_ANY", 0) / w
        # SQ_WAVE_CYCLES counts QUAD-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs.  One wave per SIMD
        # (the hand-scheduled kernel): SIMD cycles = 4 * wave quad-cycles; two waves per SIMD (ping-pong, 4-wave x 2 workgroups): 2 *
        waves_per_simd = 1 if k in ("w4", "w16") else 2
        e["mfma_busy_frac_of_simd_cycles"] = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4.0 * w / waves_per_simd)
print(json.dumps(out, indent=1, sort_keys=True))
PY
