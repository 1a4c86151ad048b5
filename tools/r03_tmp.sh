cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gemm_w4_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -8
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -5
for pair in 1 0 1 0; do
X2I_GEMM_PAIR=$pair timeout 900 python bench.py --no-fp8-lines --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pair=$pair', d['value'], d['ms_per_denoise_step'], d['roofline']['frac'])"
done
