#!/usr/bin/env python3
"""fp8 (e4m3, MX K=128 MFMA) vs bf16 GEMM timings at the DiT MLP shapes (run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import ops  # noqa: E402


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    B = int(os.environ.get("X2I_B", "4"))
    for (M, N, K, name) in [(B * 4096, 12288, 3072, "ff.0 (gelu, e4m3 out)"), (B * 4096, 3072, 12288, "ff.2 (gated res)"),
                            (B * 4608, 12288, 3072, "proj_mlp (gelu, e4m3 out)"), (B * 4608, 3072, 15360, "proj_out (gated res)"),
                            (B * 4608, 9216, 3072, "qkv-shaped (plain)")]:
        A = torch.randn(M, K, device="cuda").bfloat16()
        W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
        b = torch.randn(N, device="cuda").bfloat16()
        A8, sa = ops.quantize_rows_fp8(A)
        W8, sw = ops.quantize_rows_fp8(W)
        gelu, resid = "gelu" in name, "res" in name
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        out8 = torch.empty(M, N, device="cuda", dtype=ops.FP8)
        gate = torch.randn(1, N, device="cuda")
        if gelu:
            f8 = lambda: ops.gemm_fp8(A8, W8, b, out=out8, a_scale=sa, w_scale=sw, act=1, out_fp8=True)  # noqa: E731
            f16 = lambda: ops.gemm(A, W, b, out=out, act=1)  # noqa: E731
        elif resid:
            f8 = lambda: ops.gemm_fp8(A8, W8, b, out=out, a_scale=sa, w_scale=sw, res=out, gate=gate)  # noqa: E731
            f16 = lambda: ops.gemm(A, W, b, out=out, res=out, gate=gate)  # noqa: E731
        else:
            f8 = lambda: ops.gemm_fp8(A8, W8, b, out=out, a_scale=sa, w_scale=sw)  # noqa: E731
            f16 = lambda: ops.gemm(A, W, b, out=out)  # noqa: E731
        for rnd in range(2):
            t16, t8 = timeit(f16), timeit(f8)
            fl = 2 * M * N * K
            print(f"{name:28s} M={M:6d} N={N:6d} K={K:6d}: bf16 {t16*1e3:7.3f} ms {fl/t16/1e12:7.1f} TF | fp8 {t8*1e3:7.3f} ms {fl/t8/1e12:7.1f} TF"
                  f" | x{t16/t8:.2f}")
        del A, W, A8, W8, out, out8


if __name__ == "__main__":
    main()
