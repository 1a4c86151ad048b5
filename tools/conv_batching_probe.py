#!/usr/bin/env python3
"""Tools-only: what ONE launch over the 19 ControlNeXt nets' batch (19 x B samples) would buy over 19 launches of B samples, for the
timestep-dependent convolutions of a net at 1024^2 (shapes of lightcontrol.forward_nhwc) and its GroupNorm passes.  Same weights for every
sample here -- the question is launch size, not arithmetic.     python tools/conv_batching_probe.py [B]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import ops  # noqa: E402

DEV = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NETS = 19


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


g = torch.Generator(device=DEV).manual_seed(0)
shapes = [("down0 o conv2 (5x5 s2, 128 -> 128, 512^2)", 512, 128, 128, 5, 2, 2, True),
          ("res1.conv1 (3x3, 128 -> 256, 256^2)", 256, 128, 256, 3, 1, 1, False),
          ("down1 o shortcut (3x3 s2, 128 -> 256, 256^2)", 256, 128, 256, 3, 2, 1, True),
          ("down1 o conv2 (5x5 s2, 256 -> 256, 256^2)", 256, 256, 256, 5, 2, 2, True),
          ("mid conv (3x3, 256 -> 256, 128^2)", 128, 256, 256, 3, 1, 1, False),
          ("final (2x2 s2, 256 -> 3072, 128^2)", 128, 256, 3072, 2, 2, 0, False)]
tot1 = totn = 0.0
for name, hw, ci, co, k, s, p, has_res in shapes:
    x = torch.randn((NETS * B, hw, hw, ci), device=DEV, generator=g).bfloat16()
    w = (torch.randn((co, k * k * ci), device=DEV, generator=g) * 0.02).bfloat16()
    b = torch.randn((co,), device=DEV, generator=g).bfloat16()
    oh = (hw + 2 * p - k) // s + 1
    out = torch.empty((NETS * B, oh, oh, co), device=DEV, dtype=torch.bfloat16)
    res = torch.randn((NETS * B, oh, oh, co), device=DEV, generator=g).bfloat16() if has_res else None

    def per_net():
        for n in range(NETS):
            sl = slice(n * B, (n + 1) * B)
            ops.conv2d_nhwc(x[sl], w, b, hw, hw, ci, co, k, k, s, p, out=out[sl], res=None if res is None else res[sl])

    def batched():
        ops.conv2d_nhwc(x, w, b, hw, hw, ci, co, k, k, s, p, out=out, res=res)
    t1, tn = timeit(per_net), timeit(batched)
    fl = 2.0 * NETS * B * oh * oh * co * k * k * ci
    tot1 += t1
    totn += tn
    print(f"{name:48s} 19 launches {t1 / 1e3:7.3f} ms ({fl / t1 / 1e6:6.0f} TF)   one launch {tn / 1e3:7.3f} ms ({fl / tn / 1e6:6.0f} TF)   x{t1 / tn:.2f}", flush=True)
# GroupNorm (statistics + apply) on the two tensor sizes that carry most of the bytes
for name, hw, c, G in (("GroupNorm 256^2 x 128, 4 groups + SiLU", 256, 128, 4), ("GroupNorm 256^2 x 256, 8 groups + SiLU", 256, 256, 8), ("GroupNorm 128^2 x 256, 8 groups", 128, 256, 8)):
    x = torch.randn((NETS * B, hw, hw, c), device=DEV, generator=g).bfloat16()
    gw, gb = torch.ones(c, device=DEV, dtype=torch.bfloat16), torch.zeros(c, device=DEV, dtype=torch.bfloat16)

    def per_net():
        for n in range(NETS):
            ops.groupnorm_nhwc(x[n * B:(n + 1) * B], gw, gb, G, 1e-6, act=ops.ACT_SILU)

    def batched():
        ops.groupnorm_nhwc(x, gw, gb, G, 1e-6, act=ops.ACT_SILU)
    t1, tn = timeit(per_net), timeit(batched)
    tot1 += t1
    totn += tn
    print(f"{name:48s} 19 launches {t1 / 1e3:7.3f} ms   one launch {tn / 1e3:7.3f} ms   x{t1 / tn:.2f}", flush=True)
print(f"sum: {tot1 / 1e3:.2f} ms as 19 launches each, {totn / 1e3:.2f} ms batched")
