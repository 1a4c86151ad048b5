#!/usr/bin/env python3
"""ControlNeXt / VAE convolution shapes through the implicit-GEMM path (run on the GPU box): achieved TFLOP/s per shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import ops  # noqa: E402
from tools.microbench import timeit, rnd  # noqa: E402

B = 4
SHAPES = [  # h, w, cin, cout, k, stride, pad, residual, name
    (512, 512, 128, 128, 3, 1, 1, True, "res0.conv2"),
    (512, 512, 128, 128, 3, 2, 1, False, "down0"),
    (256, 256, 128, 256, 3, 1, 1, False, "res1.conv1"),
    (256, 256, 128, 256, 1, 1, 0, False, "res1.shortcut"),
    (256, 256, 256, 256, 3, 1, 1, True, "res1.conv2"),
    (256, 256, 256, 256, 3, 2, 1, False, "down1"),
    (128, 128, 256, 256, 3, 1, 1, False, "mid"),
    (128, 128, 256, 3072, 2, 2, 0, False, "out 2x2s2"),
    (512, 512, 256, 256, 3, 1, 1, False, "vae up2 res"),
    (1024, 1024, 128, 128, 3, 1, 1, False, "vae up3 res"),
]


def main():
    global B
    B = int(os.environ.get("X2I_B", "4"))
    extra = [(128, 128, 128, 128, 3, 1, 1, True, "res0.conv2@128"), (128, 128, 128, 128, 3, 1, 1, False, "res0 nores@128"),
             (512, 512, 128, 128, 3, 1, 1, False, "res0 nores")]
    for (h, w, cin, cout, k, s, pad, res, name) in extra + SHAPES + extra:
        x = rnd(B, h, w, cin)
        wt = rnd(cout, k * k * cin, scale=0.02)
        b = rnd(cout)
        oh, ow = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
        r = rnd(B, oh, ow, cout) if res else None
        out = torch.empty((B, oh, ow, cout), device="cuda", dtype=torch.bfloat16)
        t = timeit(lambda: ops.conv2d_nhwc(x, wt, b, h, w, cin, cout, k, k, s, pad, res=r, out=out))
        fl = 2.0 * B * oh * ow * cout * k * k * cin
        print(f"conv {name:14s} {h}x{w} {cin}->{cout} k{k}s{s}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TFLOP/s  ({fl/1e9:.0f} GFLOP)")


if __name__ == "__main__":
    main()
