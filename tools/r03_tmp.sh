cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fp8_gpu.py tests/test_model_gpu.py tests/test_train_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -4
python tools/microbench.py 2>/dev/null | grep "ln_modulate"
