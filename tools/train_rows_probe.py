#!/usr/bin/env python3
"""Tools-only: ln_mod_bwd / gate_bwd of the training step (S = 4608 rows per sample, D = 3072, R = 8) in the two forms of option
train_rows_wg -- a workgroup per row group with a thread per eight columns (1) and a wave per row (0) -- HIP events, median of rounds."""
import os
import statistics
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import _lib, ops  # noqa: E402

DEV = "cuda"
S, D, R = 4608, 3072, 8


def timed(fn, rep=20):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rep):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / rep * 1e3


for B in (1, 2, 4):
    g = torch.Generator(device=DEV).manual_seed(B)
    x, dy, dx = (torch.randn((B, S, D), device=DEV, generator=g).bfloat16() for _ in range(3))
    t = torch.randn((B, S, D), device=DEV, generator=g).bfloat16()
    sc, gate = torch.randn((B, D), device=DEV, generator=g) * 0.3, torch.randn((B, D), device=DEV, generator=g)
    nw = S // R
    part = torch.empty((B, nw, 2, D), device=DEV)
    dT = torch.empty_like(dx)
    outs = {}
    for form in (0, 1):
        _lib.set_option("train_rows_wg", form)
        d1 = dx.clone()
        ops.ln_mod_bwd(x, dy, sc, d1, d1, part, B=B, S=S, D=D, R=R, mult_bs=D)
        p1 = part.clone()
        ops.gate_bwd(dx, t, gate, None, dT, part, B=B, S=S, D=D, R=R, gate_bs=D)
        outs[form] = (d1, p1.sum(1), dT.clone(), part.view(B, -1)[:, :nw * D].view(B, nw, D).sum(1).clone())
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())  # noqa: E731
    print(f"B={B}: forms agree to rel-L2  dx {rel(outs[1][0], outs[0][0]):.2e}  d(scale|shift) sums {rel(outs[1][1], outs[0][1]):.2e}  "
          f"dT {rel(outs[1][2], outs[0][2]):.2e}  d(gate) sums {rel(outs[1][3], outs[0][3]):.2e}")
    for name, fn, mb in (("ln_mod_bwd", lambda: ops.ln_mod_bwd(x, dy, sc, dx, dT, part, B=B, S=S, D=D, R=R, mult_bs=D), 4 * B * S * D * 2 / 1e6),
                         ("gate_bwd  ", lambda: ops.gate_bwd(dx, t, gate, None, dT, part, B=B, S=S, D=D, R=R, gate_bs=D), 3 * B * S * D * 2 / 1e6)):
        ts = {0: [], 1: []}
        for r in range(5):
            for form in (0, 1):
                _lib.set_option("train_rows_wg", form)
                ts[form].append(timed(fn))
        m0, m1 = statistics.median(ts[0]), statistics.median(ts[1])
        print(f"  {name} B={B} ({mb:.0f} MB):  wave per row {m0:7.1f} us ({mb / m0:5.2f} TB/s)   workgroup per row group {m1:7.1f} us ({mb / m1:5.2f} TB/s)")
_lib.set_option("train_rows_wg", 1)
# second stage of the column sums: 576 partial rows x 3072 columns per call at batch 1 (432 calls per training step)
for B in (1, 4):
    nw = S // R
    part = torch.randn((B, nw, 2, D), device=DEV)
    out = torch.zeros((B, 2 * D), device=DEV)
    t = statistics.median(timed(lambda: ops.reduce_rows(part, out, np_=nw, len_=D, nz=B, in_zs=nw * 2 * D, in_ps=2 * D, out_zs=2 * D, accumulate=True)) for _ in range(5))
    ref = part[:, :, 0].sum(1)
    out.zero_()
    ops.reduce_rows(part, out, np_=nw, len_=D, nz=B, in_zs=nw * 2 * D, in_ps=2 * D, out_zs=2 * D, accumulate=True)
    print(f"  reduce_rows B={B}: {nw} partial rows x {D}: {t:6.1f} us ({B * nw * D * 4 / t / 1e6:5.2f} TB/s)   rel-L2 vs torch.sum {float((out[:, :D] - ref).norm() / ref.norm()):.1e}")
