#!/bin/bash
# Why attn_w4_kernel takes longer inside the step than alone: counters (own passes, --kernel-trace only) over tools/attn_in_sequence.py in both
# modes.  GRBM_GUI_ACTIVE / kernel duration = the clock the kernel ran at; SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = issue stalls.
#   tools/attn_clock_pmc.sh <outdir>   -> <outdir>/attn_clock.json
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=${1:-$R/gpurun_out/attn_clock}
case "$OUT" in /*) ;; *) OUT="$R/$OUT" ;; esac
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for mode in alone seq; do
  python "$R/tools/attn_in_sequence.py" $mode > "$OUT/$mode.events.log" 2>&1
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/$mode" -o p -- python "$R/tools/attn_in_sequence.py" $mode > "$OUT/$mode.pmc.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys
out = {}
for mode in ("alone", "seq"):
    cnt, dur = {}, []
    for f in glob.glob(f"{sys.argv[1]}/{mode}/*/p_counter_collection.csv") + glob.glob(f"{sys.argv[1]}/{mode}/p_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "attn_w4" not in r["Kernel_Name"] and "attn_w16" not in r["Kernel_Name"]:   # (whichever the product call launches)
                continue
            cnt.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "Start_Timestamp" in r:
                dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    e = {k: sum(v) / len(v) for k, v in cnt.items()}
    if dur:
        e["duration_ns_under_counters"] = sum(dur) / len(dur)
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs' GRBMs by rocprofv3 on this part when it exceeds 8 x the plausible clock: report both readings
        e["gui_active_cycles_per_ns"] = e.get("GRBM_GUI_ACTIVE", 0) / e["duration_ns_under_counters"]
    if "SQ_WAVE_CYCLES" in e:
        e["wait_inst_frac_of_wave_cycles"] = e.get("SQ_WAIT_INST_ANY", 0) / e["SQ_WAVE_CYCLES"]
        e["mfma_busy_frac_of_simd_cycles"] = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4.0 * e["SQ_WAVE_CYCLES"])
    e["events_line"] = open(f"{sys.argv[1]}/{mode}.events.log").read().strip().splitlines()[-1]
    out[mode] = e
json.dump(out, open(sys.argv[1] + "/attn_clock.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
