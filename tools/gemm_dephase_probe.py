#!/usr/bin/env python3
"""Tools-only (measurement library): does de-phasing the workgroups of the persistent GEMM (one-time start offsets by XCD, or per
workgroup over one tile period) change the steady-state time of a launch?  On a power-capped part all CUs in the same phase draw
their peak power at the same moment; 30 back-to-back launches per variant, M = 18432, N = 12288, K = 3072."""
import os
import sys
import torch
os.environ["X2I_LIB_VARIANT"] = "ablate"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import ops  # noqa: E402

DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(0)
M, N, K = 18432, 12288, 3072
A = torch.randn((M, K), device=DEV, generator=g).bfloat16()
W = (torch.randn((N, K), device=DEV, generator=g) * 0.02).bfloat16()
C = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
dbg = torch.zeros((16 * 64,), device=DEV, dtype=torch.int64)


def timed(mode, iters=30):
    f = (lambda: ops.gemm(A, W, None, out=C, act2=mode, bias2=dbg.view(torch.float32))) if mode else (lambda: ops.gemm(A, W, None, out=C))
    for _ in range(5):
        f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for rnd in range(2):
    for mode, what in ((0, "product path"), (84, "start offsets by XCD, 2.5 us apart (0..17.5)"), (83, "start offsets by XCD, 5 us apart (0..35)"),
                       (81, "start offsets by XCD, 10 us apart (0..70)"), (85, "start offsets by XCD, 20 us apart (0..140)"),
                       (82, "start offsets by workgroup (0..80 us)")):
        print(f"round {rnd}: {what:48s} {timed(mode):8.1f} us", flush=True)
