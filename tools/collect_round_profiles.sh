#!/bin/bash
# Everything kept under profiles/ for a round, in one GPU-box call:  tools/collect_round_profiles.sh r04x [quick]
# (kernel-trace --stats runs and PMC passes are separate rocprofv3 invocations; counters never share a run with tracing domains
# other than --kernel-trace).  "quick" = the bench line, its kernel trace, the roofline probe and its PMC passes only.
set -u
TAG=${1:-r04}
MODE=${2:-full}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_stats() {  # name, command...
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$name" -o s -- "$@" > "$OUT/stats_$name.log" 2>&1
  cp "$OUT/stats_$name"/*/s_kernel_stats.csv "$OUT/${TAG}_${name}_kernel_stats.csv" 2>/dev/null || cp "$OUT/stats_$name"/s_kernel_stats.csv "$OUT/${TAG}_${name}_kernel_stats.csv" 2>/dev/null
}
cd "$R"
python bench.py > "$OUT/${TAG}_bench_b4_1024.json.log" 2>&1
cd /tmp
# the timed job ONLY (1 warm-up + 2 timed passes = 12 denoise steps; no probe, no fp8 lines): bench.py's roofline.in_step reads this CSV
run_stats bench_b4_1024 python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-fp8-lines --no-roofline
echo '{"passes": 3, "command": "bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fp8-lines --no-roofline"}' > "$OUT/${TAG}_bench_b4_1024_kernel_stats.meta.json"
run_stats roofline_probe python "$R/tools/roofline_probe.py"
cp "$OUT/stats_roofline_probe.log" "$OUT/${TAG}_roofline_probe.json.log"
run_stats roofline_probe_fp8 python "$R/tools/roofline_probe.py" --fp8
cp "$OUT/stats_roofline_probe_fp8.log" "$OUT/${TAG}_roofline_probe_fp8.json.log"
bash "$R/tools/pmc_roofline.sh" "$OUT/pmc_roofline" > "$OUT/pmc_roofline.log" 2>&1
cp "$OUT/pmc_roofline/pmc_roofline.json" "$OUT/${TAG}_pmc_roofline.json"
bash "$R/tools/attn_pmc.sh" "$OUT/pmc_attn" > "$OUT/${TAG}_pmc_attn.json" 2> "$OUT/pmc_attn.err"
if [ "$MODE" = "quick" ]; then ls -la "$OUT" | head -40; exit 0; fi
cd "$R"
python bench.py --dtype fp8 --no-cpu-baseline > "$OUT/${TAG}_bench_b4_1024_fp8.json.log" 2>&1
python bench.py --dtype fp8 --fp8-mode all --no-cpu-baseline > "$OUT/${TAG}_bench_b4_1024_fp8_all.json.log" 2>&1
python bench.py --conditioning qwen7b --no-cpu-baseline --no-fp8-lines > "$OUT/${TAG}_bench_qwen7b_schnell.json.log" 2>&1
python bench.py --config 3 --no-cpu-baseline > "$OUT/${TAG}_bench_config3.json.log" 2>&1
python bench.py --config 5 --no-cpu-baseline --steps 2 --warmup 1 > "$OUT/${TAG}_bench_config5.json.log" 2>&1
for b in 1 2 8; do python bench.py --batch $b --no-cpu-baseline --no-fp8-lines > "$OUT/${TAG}_bench_batch$b.json.log" 2>&1; done
python bench.py --size 512 --batch 1 --no-cpu-baseline --no-fp8-lines > "$OUT/${TAG}_bench_512_batch1.json.log" 2>&1
python tools/microbench.py > "$OUT/${TAG}_microbench.log" 2>&1
python tools/fp8_bench.py > "$OUT/${TAG}_fp8_bench.log" 2>&1
python tools/conv_bench.py 4 > "$OUT/${TAG}_conv_bench.log" 2>&1
python tools/attn_bench.py 4 > "$OUT/${TAG}_attn_bench.log" 2>&1
python tools/attn_w16_clock.py 2 > "$OUT/${TAG}_attn_w16_clock.log" 2>&1
python tools/vae_bench.py > "$OUT/${TAG}_vae_bench.log" 2>&1
X2I_VAE_EPI_MOMENTS=0 X2I_VAE_UP_PHASES=0 python tools/vae_bench.py > "$OUT/${TAG}_vae_bench_round4_form.log" 2>&1
python tools/conv_probe.py > "$OUT/${TAG}_conv_probe.log" 2>&1
timeout 600 python tools/train_bench.py 1 2 > "$OUT/${TAG}_train_bench.log" 2>&1
python tools/fx_bench.py > "$OUT/${TAG}_fx_bench.log" 2>&1
X2I_LIB_VARIANT=ablate python tools/gemm_r2_probe.py --quick > "$OUT/${TAG}_gemm_r2_probe.log" 2>&1   # (measurement library since round 6)
python tools/vendor_probe.py 4 > "$OUT/${TAG}_vendor_probe.log" 2>&1
python tools/gemm_fabric_price.py > "$OUT/${TAG}_gemm_fabric_price.log" 2>&1
X2I_CONV_W4=0 python tools/conv_probe.py > "$OUT/${TAG}_conv_probe_conv_w4_0.log" 2>&1
X2I_CONV_W4=0 python tools/vae_bench.py > "$OUT/${TAG}_vae_bench_conv_w4_0.log" 2>&1
bash tools/attn_clock_pmc.sh "$OUT/attn_clock" > /dev/null 2>&1; cp "$OUT/attn_clock/attn_clock.json" "$OUT/${TAG}_attn_clock_alone_vs_in_sequence.json" 2>/dev/null; rm -rf "$OUT/attn_clock/alone" "$OUT/attn_clock/seq"
[ -x tools/ubench/bin/mfma_power ] && bash tools/clock_watch.sh "$OUT/${TAG}_mfma_power_clock.log" -- tools/ubench/bin/mfma_power 1 8 > "$OUT/${TAG}_mfma_power.log" 2>&1
cd /tmp
run_stats bench_b4_1024_fp8 python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --dtype fp8 --no-roofline
run_stats bench_config5 python "$R/bench.py" --config 5 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline
run_stats vae_b4 python "$R/tools/vae_bench.py"
bash "$R/tools/pmc_collect.sh" "$OUT/pmc" all > "$OUT/pmc.log" 2>&1
cp "$OUT/pmc/summary.json" "$OUT/${TAG}_pmc_gemm_attn.json"
bash "$R/tools/clock_watch.sh" "$OUT/${TAG}_clock_power_during_bench.log" -- python "$R/bench.py" --no-cpu-baseline --no-fp8-lines --steps 6 > /dev/null 2>&1
# soak: 50 bench passes (200 denoise steps; bench.py raises if any stream-K workspace carries the give-up marker)
python "$R/bench.py" --steps 50 --warmup 2 --no-cpu-baseline --no-fp8-lines --no-roofline > "$OUT/${TAG}_soak_50_passes.json.log" 2>&1
ls -la "$OUT" | head -60
