#!/usr/bin/env python3
"""Small-batch GEMM probe: the gated-residual linears whose batch item has fewer 256^2 tiles than the chip has CUs, with the parallel
split + fix-up (option gemm_fx = 1, the default) and with whole tiles (0), at per-GPU batch 1 / 2 / 4; HIP-event timing."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import _lib, ops  # noqa: E402

DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(0)


def timeit(fn, iters=20):
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


def single(B, S, N, K):
    A = torch.randn((B, S, K), device=DEV, generator=g).bfloat16()
    W = (torch.randn((N, K), device=DEV, generator=g) * 0.02).bfloat16()
    b = torch.randn((N,), device=DEV, generator=g).bfloat16()
    X = torch.randn((B, S, N), device=DEV, generator=g).bfloat16()
    gate = torch.randn((B, N), device=DEV, generator=g)
    return lambda: ops.gemm(A, W, b, out=X, M=S, batch=B, a_batch_stride=S * K, lda=K, c_batch_stride=S * N, ldc=N, res=X, res_batch_stride=S * N, ldr=N,
                            gate=gate, gate_batch_stride=N), 2.0 * B * S * N * K


def pair(B, Si, St, N, K):
    mk = []
    fl = 0.0
    for S in (Si, St):
        A = torch.randn((B, S, K), device=DEV, generator=g).bfloat16()
        W = (torch.randn((N, K), device=DEV, generator=g) * 0.02).bfloat16()
        b = torch.randn((N,), device=DEV, generator=g).bfloat16()
        X = torch.randn((B, S, N), device=DEV, generator=g).bfloat16()
        gate = torch.randn((B, N), device=DEV, generator=g)
        mk.append(dict(A=A, W=W, bias=b, out=X, M=S, batch=B, a_batch_stride=S * K, lda=K, c_batch_stride=S * N, ldc=N, res=X, res_batch_stride=S * N, ldr=N,
                       gate=gate, gate_batch_stride=N))
        fl += 2.0 * B * S * N * K
    return lambda: ops.gemm_pair(mk[0], mk[1]), fl


for name, build in (("proj_out 1024^2 (S=4608, K=15360)", lambda B: single(B, 4608, 3072, 15360)),
                    ("ff.2 pair 1024^2 (4096 + 512 rows, K=12288)", lambda B: pair(B, 4096, 512, 3072, 12288)),
                    ("proj_out 512^2 (S=1536, K=15360)", lambda B: single(B, 1536, 3072, 15360)),
                    ("ff.2 pair 512^2 (1024 + 512 rows, K=12288)", lambda B: pair(B, 1024, 512, 3072, 12288))):
    for B in (1, 2, 4):
        fn, fl = build(B)
        res = {}
        for fx in (0, 1):
            _lib.set_option("gemm_fx", fx)
            res[fx] = timeit(fn)
            tile = _lib.get_option("last_gemm_tile")
        print(f"{name:46s} B={B}: whole tiles {res[0]:8.1f} us ({fl / res[0] / 1e6:6.0f} TF) | split + fix-up {res[1]:8.1f} us ({fl / res[1] / 1e6:6.0f} TF)"
              f"  x{res[0] / res[1]:.3f}  [last_gemm_tile {tile}]")
ops.streamk_check(sync=True)
