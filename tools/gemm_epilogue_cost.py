#!/usr/bin/env python3
"""Tools-only: what the epilogue kinds of the persistent GEMM cost on the roofline pair's GELU shape (M = 18432, N = 12288, K = 3072)
and on shallower / deeper K -- plain bias, ReLU (one instruction), tanh-GELU (seven), SiLU, gated residual."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import ops  # noqa: E402

DEV = "cuda"


def timed(f, iters=20):
    for _ in range(3):
        f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    M = 18432
    for N, K in ((12288, 3072), (3072, 3072), (12288, 1536), (12288, 6144)):
        A = torch.randn((M, K), device=DEV, generator=g).bfloat16()
        W = (torch.randn((N, K), device=DEV, generator=g) * 0.02).bfloat16()
        b = torch.randn((N,), device=DEV, generator=g).bfloat16()
        C = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
        R = torch.randn((M, N), device=DEV, generator=g).bfloat16()
        gate = torch.randn((1, N), device=DEV, generator=g)
        fl = 2.0 * M * N * K
        kinds = (("bias", {}), ("relu", dict(act=4)), ("gelu_tanh", dict(act=ops.ACT_GELU_TANH)), ("silu", dict(act=ops.ACT_SILU)),
                 ("gated residual", dict(res=R, gate=gate, gate_batch_stride=0)))
        if os.environ.get("X2I_LIB_VARIANT") == "ablate":   # measurement-only library: the same launches without their epilogue / stores
            kinds += (("bias, NO epilogue", dict(act2=77)), ("bias, no global stores", dict(act2=78)), ("gelu, no global stores", dict(act=ops.ACT_GELU_TANH, act2=78)))
        for name, kw in kinds:
            ms = timed(lambda: ops.gemm(A, W, b, out=C, **kw))
            print(f"N={N:6d} K={K:5d} {name:16s} {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
