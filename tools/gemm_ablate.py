#!/usr/bin/env python3
"""Within-process A/B of the 256x256 GEMM main loop with pieces removed (results are wrong by design)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["X2I_LIB_VARIANT"] = "ablate"  # libx2i_hip_ablate.so: the only build that contains these kernels
from x2i_amd import _lib, ops

def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

_lib.set_option("gemm_tile", 256)
for (M, N, K) in [(16384, 3072, 12288), (16384, 9216, 3072), (18432, 21504, 3072)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for rnd in range(2):
        for abl, name in ((0, "full"), (1, "no ds_read"), (2, "no barrier/wait"), (4, "no DMA"), (3, "no ds_read+barrier"), (7, "MFMA only"), (8, "no epilogue"), (128, "no steady-state peel"), (0, "full"), (256, "DMA from cache-hot source"), (0, "full"), (256, "DMA from cache-hot source"), (0, "full")):
            _lib.set_option("gemm_ablate", abl)
            t = timeit(lambda: ops.gemm(A, W, out=out))
            print(f"M={M} N={N} K={K} round{rnd} {name:20s}: {t*1e3:7.3f} ms {2*M*N*K/t/1e12:7.1f} TF")
_lib.set_option("gemm_ablate", 0)
