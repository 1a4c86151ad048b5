mkdir -p gpurun_out/r04v
for fx in 0 1; do
  X2I_GEMM_FX=$fx python bench.py --batch 1 --no-cpu-baseline --no-fp8-lines --no-roofline --steps 6 2>/dev/null | tail -1 > gpurun_out/r04v/r04v_bench_batch1_fx$fx.json.log
  X2I_GEMM_FX=$fx python bench.py --batch 1 --size 512 --no-cpu-baseline --no-fp8-lines --no-roofline --steps 10 2>/dev/null | tail -1 > gpurun_out/r04v/r04v_bench_512_batch1_fx$fx.json.log
done
for f in gpurun_out/r04v/*.json.log; do python - "$f" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split("/")[-1], round(d["value"],3), "img/s", round(d["ms_per_denoise_step"],2), "ms/step", round(d["model_frac_of_bf16_peak"],3))
PY
done
