#!/usr/bin/env python3
"""Tools-only: attn_w4_kernel (variant 9) against attn_w16_kernel (variant 12) at B = 4, 24 heads, S = 4608, each sustained for ~1.5 s with
the clock / socket power it ran at (amdsmi), alternating -- does the 16x16x32 MFMA shape buy clock at the power cap?"""
import math
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from x2i_amd import _lib, ops  # noqa: E402

B, H, S, D = 4, 24, 4608, 3072
Spad = ops.pad128(S)
perm = torch.tensor([16 * ((kk >> 2) & 1) + 4 * (kk >> 3) + (kk & 3) for kk in range(32)], device="cuda")
Q = (torch.randn(B, H, Spad, 128, device="cuda") * (math.log2(math.e) / math.sqrt(128))).bfloat16()
K_, VT = torch.randn(B, H, Spad, 128, device="cuda").bfloat16(), torch.randn(B, H, 128, Spad, device="cuda").bfloat16()
VTP = VT.view(B, H, 128, Spad // 32, 32)[..., perm].reshape(B, H, 128, Spad).contiguous()
O = torch.empty((B, S, 5 * D), device="cuda", dtype=torch.bfloat16)
fl = 4.0 * B * H * S * S * 128


def run(var):
    def f():
        _lib.set_option("attn_variant", var)
        ops.attention(Q, K_, VTP if var == 12 else VT, O, B, H, S, Spad, 5 * D, S * 5 * D, math.log(2.0))
        _lib.set_option("attn_variant", 0)
    return f


for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for var, name in ((9, "w4  (32x32x16)"), (12, "w16 (16x16x32)")):
        if "w16only" in sys.argv and var != 12:
            continue
        times, clk = bench._interleaved_probe([run(var)], 6, 300)
        t = sorted(times[0])[len(times[0]) // 2]
        sc, pw = clk.get("sclk_mhz") or {}, clk.get("socket_power_w") or {}
        cyc = t * (sc.get("median") or 0) * 1e6 / 7 / 72    # 1728 workgroups = 7 rounds on 256 CUs, 72 key tiles each
        print(f"{name}: {t * 1e6:7.1f} us  {fl / t / 1e12:7.1f} TFLOP/s   sclk {sc.get('median')} MHz   power {pw.get('median')} W   ~{cyc:.0f} cycles per key tile", flush=True)
