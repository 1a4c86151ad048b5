#!/bin/bash
# stability soak of the round-3 kernels: a long bench run (200 denoise steps), the stream-K give-up marker afterwards, and the
# training step at B = 2 / 4 for the record
cd ${GRAFT_REPO_ROOT:-$PWD}
python - <<'PY'
import json, subprocess, sys
out = subprocess.run([sys.executable, "bench.py", "--steps", "50", "--warmup", "2", "--no-cpu-baseline", "--no-fp8-lines"], capture_output=True, text=True)
d = json.loads(out.stdout.strip().split("\n")[-1])
print("soak: 50 steps x 4 denoise steps:", round(d["value"], 3), "images/s", round(d["ms_per_denoise_step"], 2), "ms/step")
PY
python - <<'PY'
import torch
from x2i_amd import _lib, ops
A = torch.randn(18432, 3072, device="cuda").bfloat16(); W = (torch.randn(12288, 3072, device="cuda") * 0.02).bfloat16()
out = torch.empty(18432, 12288, device="cuda", dtype=torch.bfloat16)
for _ in range(500):
    ops.gemm(A, W, out=out, act=1)
torch.cuda.synchronize()
print("500 stream-K launches, gemm_sk_error =", _lib.get_option("gemm_sk_error"))
PY
timeout 900 python tools/train_bench.py 2 2 2>&1 | grep -v amdgpu | tail -3
timeout 900 python tools/train_bench.py 4 2 2>&1 | grep -v amdgpu | tail -3
