#!/usr/bin/env python3
"""Tools-only: gn_apply (x2i_groupnorm_nhwc_from_moments_bf16) on the VAE decode's largest tensors with and without an activation: HBM-bound
(5.1-5.6 TB/s of read + write whatever the activation); four 16-byte loads in flight per thread instead of two measured 4.3-4.7 TB/s and was
not adopted."""
import sys, torch
sys.path.insert(0, "/root/repo")
import bench
from x2i_amd import ops
for (B, HW, C, G) in [(4, 1024 * 1024, 128, 32), (4, 512 * 512, 256, 32), (4, 1024 * 1024, 256, 32)]:
    x = torch.randn(B, HW, C, device="cuda").bfloat16()
    w = torch.ones(C, device="cuda").bfloat16(); b = torch.zeros(C, device="cuda").bfloat16()
    y = torch.empty_like(x)
    mom = ops.groupnorm_moments(x)
    for act, name in ((0, "none"), (3, "silu"), (4, "relu")):
        fn = lambda: ops.groupnorm_nhwc_from_moments(x, mom, w, b, G, 1e-6, act=act, out=y)
        t, _ = bench._interleaved_probe([fn], 3, 10)
        t = sorted(t[0])[len(t[0]) // 2]
        print(f"B={B} HW={HW} C={C} act={name}: {t*1e6:7.1f} us  {2*x.numel()*2/t/1e12:5.2f} TB/s (read + write)", flush=True)
