cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_fullsize_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
python tools/microbench.py 2>/dev/null | grep "skinny\|ln_modulate"
