#!/usr/bin/env python3
"""Tools-only numerical study (runs on the CPU): does MX-style block scaling (one E8M0 scale per 32 K-elements, what
`v_mfma_scale_f32_16x16x128_f8f6f4` applies for free) lower the error of the e4m3 GEMMs below the per-row x per-channel scaling the
product uses?  Operands with the statistics of the DiT linears: A = LayerNorm+modulate rows (unit-variance
rows with a per-channel (1 + scale) spread, a few outlier channels), W ~ N(0, 0.02^2); error = rel-L2 of the dequantised product
against the fp32 product of the unquantised operands."""
import torch


def q_e4m3(x):
    return x.clamp(-448, 448).to(torch.float8_e4m3fn).float()


def quant_rows(x):                      # product path: one fp32 scale per row (amax / 448)
    s = x.abs().amax(-1, keepdim=True).clamp_min(1e-12) / 448.0
    return q_e4m3(x / s) * s


def quant_mx(x, block=32, two_level=False):
    """E8M0 (power-of-two) scale per `block` consecutive K elements; two_level: on top of the fp32 row scale."""
    M, K = x.shape
    xb = x.view(M, K // block, block)
    row = (x.abs().amax(-1, keepdim=True).clamp_min(1e-12) / 448.0).view(M, 1, 1) if two_level else torch.ones(M, 1, 1)
    amax = (xb / row).abs().amax(-1, keepdim=True).clamp_min(1e-30)
    e = torch.ceil(torch.log2(amax / 448.0))     # smallest power of two with amax / 2^e <= 448
    s = torch.exp2(e) * row
    return (q_e4m3(xb / s) * s).view(M, K)


def rel(a, b):
    return float((a - b).norm() / b.norm())


def main():
    torch.manual_seed(0)
    M, N, K = 1024, 1024, 3072
    for name, outliers in (("LN+modulate rows", 0), ("with 8 outlier channels x30", 8), ("with 8 outlier channels x300", -8)):
        A = torch.randn(M, K) * (1 + 0.3 * torch.randn(K)) + 0.1 * torch.randn(K)
        if outliers:
            idx = torch.randperm(K)[:abs(outliers)]
            A[:, idx] *= 30.0 if outliers > 0 else 300.0
        W = torch.randn(N, K) * 0.02
        ref = A @ W.t()
        res = {}
        res["row x channel (product)"] = rel(quant_rows(A) @ quant_rows(W).t(), ref)
        res["MX e8m0 per 32"] = rel(quant_mx(A) @ quant_mx(W).t(), ref)
        res["row fp32 x MX e8m0 per 32"] = rel(quant_mx(A, two_level=True) @ quant_mx(W, two_level=True).t(), ref)
        res["bf16 operands"] = rel(A.bfloat16().float() @ W.bfloat16().float().t(), ref)
        print(name + ": " + ", ".join(f"{k} {v:.3e}" for k, v in res.items()))


if __name__ == "__main__":
    main()
