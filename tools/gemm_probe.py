#!/usr/bin/env python3
"""Launch a handful of GEMM / attention kernels at FLUX shapes (for rocprofv3 --pmc passes)."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import ops

B, D, H, S = 4, 3072, 24, 4608
def rnd(*s, scale=1.0):
    return (torch.randn(*s, device="cuda") * scale).bfloat16()
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "gemm"):
    for (M, N, K, act) in [(B * S, 4 * D, D, 1), (B * S, D, 5 * D, 0), (B * 4096, 3 * D, D, 0)]:
        A, W, b = rnd(M, K), rnd(N, K, scale=0.02), rnd(N)
        out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
        for _ in range(3):
            ops.gemm(A, W, b, out=out, act=act)
        torch.cuda.synchronize()
if which in ("all", "fp8"):
    for (M, N, K, gelu) in [(B * S, 4 * D, D, True), (B * S, D, 5 * D, False)]:
        A8, sa = ops.quantize_rows_fp8(rnd(M, K))
        W8, sw = ops.quantize_rows_fp8(rnd(N, K, scale=0.02))
        b = rnd(N)
        out = torch.empty((M, N), device="cuda", dtype=ops.FP8 if gelu else torch.bfloat16)
        gate = torch.randn(1, N, device="cuda")
        for _ in range(3):
            if gelu:
                ops.gemm_fp8(A8, W8, b, out=out, a_scale=sa, w_scale=sw, act=1, out_fp8=True)
            else:
                ops.gemm_fp8(A8, W8, b, out=out, w_scale=sw, res=out, gate=gate)
        torch.cuda.synchronize()
if which in ("all", "proj"):
    for (C, Hh) in ((37, 2048), (29, 3584)):
        x = rnd(B, C, 512, Hh, scale=3.0)
        w, bb = torch.randn(C, 25, device="cuda").bfloat16().float(), torch.randn(1, device="cuda")
        table = ops.proj_conv5x5_pack(w)
        for _ in range(3):
            ops.proj_conv5x5(x, w, bb)
            ops.proj_conv5x5_packed(x, table, bb)
        torch.cuda.synchronize()
        del x
if which in ("all", "attn"):
    Spad = ops.pad128(S)
    Q, K_, VT = rnd(B, H, Spad, 128), rnd(B, H, Spad, 128), rnd(B, H, 128, Spad)
    O = torch.empty((B, S, D), device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        # the PRODUCT call: Q carries softmax_scale * log2(e) (x2i_qkv_desc.q_scale) and the kernel gets scale = ln 2 -> attn_w16_kernel on a span-permuted V^T (attn_w4_kernel with X2I_ATTN_W16=0)
        ops.attention((Q.float() * (math.log2(math.e) / math.sqrt(128))).bfloat16(), K_, VT, O, B, H, S, Spad, D, S * D, math.log(2.0),
                      vt_perm=ops.attention_prefers_vt_perm(H, S, math.log(2.0)))
    torch.cuda.synchronize()
if which == "conv":
    # ControlNeXt ResnetBlock conv2 (3x3, 128 -> 128 at 512^2) with / without the residual, and the 256-wide one
    for (h, cin, cout, res) in [(512, 128, 128, True), (512, 128, 128, False), (256, 256, 256, True)]:
        x, wt, b = rnd(B, h, h, cin), rnd(cout, 9 * cin, scale=0.02), rnd(cout)
        r = rnd(B, h, h, cout) if res else None
        out = torch.empty((B, h, h, cout), device="cuda", dtype=torch.bfloat16)
        for _ in range(3):
            ops.conv2d_nhwc(x, wt, b, h, h, cin, cout, 3, 3, 1, 1, res=r, out=out)
        torch.cuda.synchronize()
if which == "vaeconv":
    # the VAE decoder's big convolutions (B = 4): <= 128 output channels at the image resolution (csrc/gemm512c.hip) and a 256-wide one (csrc/gemm256c.hip)
    for (h, cin, cout) in [(1024, 256, 128), (1024, 128, 128), (512, 512, 256)]:
        x, wt, b = rnd(B, h, h, cin), rnd(cout, 9 * cin, scale=0.02), rnd(cout)
        out = torch.empty((B, h, h, cout), device="cuda", dtype=torch.bfloat16)
        for _ in range(3):
            ops.conv2d_nhwc(x, wt, b, h, h, cin, cout, 3, 3, 1, 1, out=out)
        torch.cuda.synchronize()
        del x, out
