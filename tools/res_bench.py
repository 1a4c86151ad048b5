#!/usr/bin/env python3
"""Cost of the gated-residual epilogue on the DiT shapes (run on the GPU box): same GEMM with and without gate/res."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import ops  # noqa: E402
from tools.microbench import timeit, rnd  # noqa: E402


def main():
    B, D, Si, S = 4, 3072, 4096, 4608
    for (M, N, K, name) in [(B * Si, D, D, "attn_out"), (B * Si, D, 4 * D, "ff_out"), (B * S, D, 5 * D, "single_out")]:
        A, W, b = rnd(M, K), rnd(N, K, scale=0.02), rnd(N)
        out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
        res = rnd(M, N)
        gate = torch.randn(N, device="cuda", dtype=torch.float32)
        t0 = timeit(lambda: ops.gemm(A, W, b, out=out))
        t1 = timeit(lambda: ops.gemm(A, W, b, out=out, gate=gate, res=res))
        t2 = timeit(lambda: ops.gemm(A, W, b, out=res, gate=gate, res=res))
        f = 2 * M * N * K / 1e12
        print(f"{name:11s} plain {t0*1e6:8.1f} us {f/t0:7.1f} TF | gate+res {t1*1e6:8.1f} us {f/t1:7.1f} TF | in-place {t2*1e6:8.1f} us {f/t2:7.1f} TF")


if __name__ == "__main__":
    main()
