cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout 900 python bench.py --no-fp8-lines --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_denoise_step'], d['roofline']['frac'])"
done
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['fp8_mlp']['images_s'], d['fp8_all']['images_s'])"
timeout 1200 python -m pytest tests/test_fullscale_parity_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | grep -v amdgpu | tail -3
