#!/usr/bin/env python3
"""Tools-only: the hand-scheduled 16x16x32 attention kernel (attn_variant = 12, V^T pre-permuted) against the 4-wave kernel on small shapes."""
import math
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import _lib, ops  # noqa: E402

perm = torch.tensor([16 * ((kk >> 2) & 1) + 4 * (kk >> 3) + (kk & 3) for kk in range(32)], device="cuda")
shapes = [(1, 1, 256), (1, 1, 64), (1, 1, 128), (1, 2, 192), (2, 3, 700), (1, 2, 1152), (1, 24, 4608)] if len(sys.argv) < 2 else [tuple(int(x) for x in sys.argv[1:4])]
for B, H, S in shapes:
    g = torch.Generator(device="cuda").manual_seed(S)
    Spad = ops.pad128(S)
    D = H * 128
    Q = torch.randn((B, H, Spad, 128), device="cuda", generator=g).bfloat16()
    K = torch.randn((B, H, Spad, 128), device="cuda", generator=g).bfloat16()
    VT = torch.randn((B, H, 128, Spad), device="cuda", generator=g).bfloat16()
    VTP = VT.view(B, H, 128, Spad // 32, 32)[..., perm].reshape(B, H, 128, Spad).contiguous()
    outs, lses = {}, {}
    for var in (4, 12):
        _lib.set_option("attn_variant", var)
        O = torch.zeros((B, S, D), device="cuda", dtype=torch.bfloat16)
        lse = torch.zeros((B, H, Spad), device="cuda")
        ops.attention_lse(Q, K, VTP if var == 12 else VT, O, lse, B, H, S, Spad, D, S * D, 1 / math.sqrt(128))
        torch.cuda.synchronize()
        outs[var], lses[var] = O.float(), lse
    _lib.set_option("attn_variant", 0)
    d = outs[12] - outs[4]
    rel = float(d.norm() / outs[4].norm())
    dl = float((lses[12][..., :S] - lses[4][..., :S]).abs().max())
    print(f"B={B} H={H} S={S}: rel-L2 {rel:.3e}  max |lse diff| {dl:.3e}  finite {bool(torch.isfinite(outs[12]).all())}", flush=True)
    if rel > 1e-2:
        bad = (d.abs() > 0.05).nonzero()
        print("   first bad (b, row, col):", bad[:6].tolist(), " count", bad.shape[0])
