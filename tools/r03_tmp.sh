cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_fp8_gpu.py tests/test_ops_gpu.py -x -q -k "attention" 2>&1 | grep -v amdgpu.ids | tail -8
timeout 900 python bench.py 2>/dev/null | tail -1 | tee gpurun_out/r03x_bench.json | cut -c1-1500
