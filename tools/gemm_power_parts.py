#!/usr/bin/env python3
"""Tools-only (measurement library): package power and clock while the k-half-unit 256^2 GEMM runs with pieces of its main loop
removed (results wrong by design) -- which part of the loop the power cap is spent on.  One variant per invocation:
    tools/clock_watch.sh out.log -- python tools/gemm_power_parts.py <ablation code> [seconds]
codes: 0 full, 1 no ds_read, 4 no DMA, 5 neither, 7 MFMA only, 256 DMA from a cache-hot source."""
import os
import sys
import time
import torch
os.environ["X2I_LIB_VARIANT"] = "ablate"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import _lib, ops  # noqa: E402

abl = int(sys.argv[1])
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
M, N, K = 18432, 12288, 3072
A = torch.randn(M, K, device="cuda").bfloat16()
W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
_lib.set_option("gemm_tile", 256)
_lib.set_option("gemm_ablate", abl)
for _ in range(5):
    ops.gemm(A, W, out=out)
torch.cuda.synchronize()
t0 = time.time()
n = 0
while time.time() - t0 < secs:
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        ops.gemm(A, W, out=out)
    e.record()
    torch.cuda.synchronize()
    n += 1
    last = s.elapsed_time(e) / 50
print(f"ablation {abl}: {last * 1e3:8.1f} us per launch, {2.0 * M * N * K / last / 1e9:7.1f} TFLOP/s (last of {n} batches)")
