cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gemm_w4_gpu.py tests/test_ops_gpu.py tests/test_model_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -4
python tools/gemm_unit_timeline.py 2>&1 | grep -v amdgpu | tail -7
for i in 1 2; do
timeout 900 python bench.py --no-fp8-lines --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_denoise_step'], d['roofline']['frac'])"
done
