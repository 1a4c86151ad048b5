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


def res_shapes():
    """The gated-residual launches (to_out: K = 3072; proj_out: K = 15360; batch 4 x 4608 rows, N = 3072): every workgroup reads 128 KiB of
    residual rows and writes 128 KiB in its epilogue, and with all K-loops in step the 256 epilogues fall into the same few microseconds."""
    Bz, S = 4, 4608
    for (Nn, Kk) in ((3072, 3072), (3072, 15360)):
        Ar = torch.randn((Bz, S, Kk), device=DEV, generator=g).bfloat16()
        Wr = (torch.randn((Nn, Kk), device=DEV, generator=g) * 0.02).bfloat16()
        X = torch.randn((Bz, S, Nn), device=DEV, generator=g).bfloat16()
        gate = torch.randn((Bz, Nn), device=DEV, generator=g) * 0.1

        def run(mode):
            kw = dict(act2=mode, bias2=dbg.view(torch.float32)) if mode else {}
            ops.gemm(Ar, Wr, None, out=X, M=S, batch=Bz, a_batch_stride=S * Kk, lda=Kk, c_batch_stride=S * Nn, ldc=Nn, res=X, res_batch_stride=S * Nn,
                     ldr=Nn, gate=gate, gate_batch_stride=Nn, **kw)

        def t_of(mode, iters=20):
            for _ in range(3):
                run(mode)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                run(mode)
            e.record()
            torch.cuda.synchronize()
            return s.elapsed_time(e) / iters * 1e3
        for rnd in range(3):
            print(f"gated residual N={Nn} K={Kk} round {rnd}: " + "  ".join(f"{what} {t_of(mode):7.1f} us" for mode, what in
                  ((0, "in step"), (84, "2.5 us/XCD"), (83, "5 us/XCD"), (81, "10 us/XCD"), (85, "20 us/XCD"))), flush=True)


if "--res" in sys.argv:
    res_shapes()
    sys.exit(0)
for rnd in range(2):
    for mode, what in ((0, "product path"), (84, "start offsets by XCD, 2.5 us apart (0..17.5)"), (83, "start offsets by XCD, 5 us apart (0..35)"),
                       (81, "start offsets by XCD, 10 us apart (0..70)"), (85, "start offsets by XCD, 20 us apart (0..140)"),
                       (82, "start offsets by workgroup (0..80 us)")):
        print(f"round {rnd}: {what:48s} {timed(mode):8.1f} us", flush=True)
