#!/usr/bin/env python3
"""Race screen for the pipelined GEMM kernels (run on the GPU box): the 256^2 full-line kernel, its k-half-unit predecessor and
the 128^2 kernel accumulate in the same order, so their outputs must be bit-identical; repeat over shapes and launches with other
work in flight between them."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import ops  # noqa: E402


def run(tile, lform, A, W, b, **kw):
    os.environ["X2I_GEMM_TILE"] = tile
    os.environ["X2I_GEMM_LFORM"] = lform
    return ops.gemm(A, W, b, **kw)


def main():
    torch.manual_seed(0)
    bad = 0
    shapes = [(4096, 3072, 3072), (4608, 3072, 15360), (2304, 9216, 3072), (1280, 768, 64), (777, 520, 320), (8192, 12288, 3072)]
    noise = torch.randn(64 << 20, device="cuda")
    for it in range(int(os.environ.get("X2I_STRESS_ITERS", "6"))):
        for (M, N, K) in shapes:
            A = torch.randn(M, K, device="cuda").bfloat16()
            W = (torch.randn(N, K, device="cuda") * 0.03).bfloat16()
            b = torch.randn(N, device="cuda").bfloat16()
            res = torch.randn(M, N, device="cuda").bfloat16()
            gate = torch.randn(1, N, device="cuda")
            ref = run("128", "1", A, W, b, act=1)
            for rep in range(3):
                noise.mul_(1.0001)  # unrelated traffic between launches
                for tile, lf in (("256", "1"), ("256", "0")):
                    out = run(tile, lf, A, W, b, act=1)
                    if not torch.equal(out, ref):
                        bad += 1
                        print("MISMATCH", M, N, K, tile, lf, float((out.float() - ref.float()).abs().max()))
            r128 = res.clone()
            run("128", "1", A, W, b, out=r128, res=r128, gate=gate)
            r256 = res.clone()
            run("256", "1", A, W, b, out=r256, res=r256, gate=gate)
            if not torch.equal(r128, r256):
                bad += 1
                print("MISMATCH gated", M, N, K)
    print("stress_gemm: %d mismatches" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
