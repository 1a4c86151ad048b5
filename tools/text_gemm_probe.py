#!/usr/bin/env python3
"""Text-stream linears of a double block when they are NOT grouped with the image stream (e4m3 configurations: the image stream runs on
e4m3, the 512-row text stream stays bf16): 128^2 tiles vs the persistent 256^2 kernel (option gemm_tile), per-GPU batch 4 / 2 / 1."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import _lib, ops  # noqa: E402

DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(0)


def timeit(fn, iters=30):
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


for B in (4, 2, 1):
    St, S, D = 512, 4608, 3072
    for name, N, K, kind in (("ff_context.0 (GELU)", 4 * D, D, "gelu"), ("ff_context.2 (gated res)", D, 4 * D, "res"), ("to_add_out (gated res)", D, D, "res")):
        A = torch.randn((B, S, K), device=DEV, generator=g).bfloat16()
        W = (torch.randn((N, K), device=DEV, generator=g) * 0.02).bfloat16()
        b = torch.randn((N,), device=DEV, generator=g).bfloat16()
        X = torch.randn((B, S, N), device=DEV, generator=g).bfloat16()
        gate = torch.randn((B, N), device=DEV, generator=g)
        if kind == "gelu":
            fn = lambda: ops.gemm(A, W, b, out=X, M=St, batch=B, a_batch_stride=S * K, lda=K, c_batch_stride=S * N, ldc=N, act=1)  # noqa: E731
        else:
            fn = lambda: ops.gemm(A, W, b, out=X, M=St, batch=B, a_batch_stride=S * K, lda=K, c_batch_stride=S * N, ldc=N, res=X, res_batch_stride=S * N,  # noqa: E731
                                  ldr=N, gate=gate, gate_batch_stride=N)
        res = {}
        for tile in (0, 128, 256):
            _lib.set_option("gemm_tile", tile)
            res[tile] = (timeit(fn), _lib.get_option("last_gemm_tile"))
        _lib.set_option("gemm_tile", 0)
        fl = 2.0 * B * St * N * K
        print(f"B={B} {name:26s}: " + " | ".join(f"tile={t}: {res[t][0]:7.1f} us {fl / res[t][0] / 1e6:6.0f} TF [{res[t][1]}]" for t in (0, 128, 256)))
