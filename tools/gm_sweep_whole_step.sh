mkdir -p gpurun_out/r04f
for gm in 0 1 2 3 4 8 12 18; do
  X2I_GEMM_GM=$gm python bench.py --no-cpu-baseline --no-fp8-lines --no-roofline --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gm=$gm', round(d['value'],4), 'images/s', round(d['ms_per_denoise_step'],2), 'ms/step')"
done > gpurun_out/r04f/r04f_gm_sweep_whole_step.log 2>&1
X2I_GEMM_GM=0 python bench.py --no-cpu-baseline --no-fp8-lines --no-roofline --steps 4 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gm=0 again', round(d['value'],4), round(d['ms_per_denoise_step'],2))" >> gpurun_out/r04f/r04f_gm_sweep_whole_step.log 2>&1
cat gpurun_out/r04f/r04f_gm_sweep_whole_step.log
