#!/usr/bin/env python3
"""Tile-patch height (option gemm_gm) for the deep-K, narrow-N launches alone (single-block proj_out M = 18432, N = 3072, K = 15360 and
ff.net.2 K = 12288; bf16 and e4m3): these are the launches whose K-loop waits for operand delivery (1.47 us per K-tile against 1.18 with
cache-hot operands, tools/gemm_unit_timeline.py --fp8 --hot)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import _lib, ops  # noqa: E402

DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(0)


def timeit(fn, iters=12):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for (M, N, K) in ((18432, 3072, 15360), (16384, 3072, 12288), (18432, 12288, 3072), (18432, 9216, 3072)):
    A = torch.randn((M, K), device=DEV, generator=g).bfloat16()
    W = (torch.randn((N, K), device=DEV, generator=g) * 0.02).bfloat16()
    b = torch.randn((N,), device=DEV, generator=g).bfloat16()
    X = torch.randn((M, N), device=DEV, generator=g).bfloat16()
    gate = torch.randn((1, N), device=DEV, generator=g)
    A8, sa = ops.quantize_rows_fp8(A)
    W8, sw = ops.quantize_rows_fp8(W)
    fl = 2.0 * M * N * K
    row = []
    for gm in (0, 1, 2, 3, 4, 6, 8, 12):
        _lib.set_option("gemm_gm", gm)
        t16 = timeit(lambda: ops.gemm(A, W, b, out=X, res=X, gate=gate))
        t8 = timeit(lambda: ops.gemm_fp8(A8, W8, b, out=X, a_scale=sa, w_scale=sw, res=X, gate=gate))
        row.append((gm, t16, t8))
    _lib.set_option("gemm_gm", 0)
    print(f"M={M} N={N} K={K} (gated residual):")
    for gm, t16, t8 in row:
        print(f"   gm={gm:2d}: bf16 {t16:7.1f} us {fl / t16 / 1e6:6.0f} TF | e4m3 {t8:7.1f} us {fl / t8 / 1e6:6.0f} TF")
