#!/usr/bin/env python3
"""Tools-only: the VAE decoder's convolution shapes one by one (B = 4) -- time, TFLOP/s and effective HBM rate of x2i_conv2d_nhwc_bf16 with the plain,
the residual and the moments-writing epilogue.  X2I_LIB_VARIANT=<name> runs another build of the library (A/B against an older commit)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from x2i_amd import ops  # noqa: E402

DEV = "cuda"
B = int(os.environ.get("X2I_B", "4"))
old = "--old-abi" in sys.argv      # a library from before x2i_conv_desc.moments: no moments runs


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    shapes = [(128, 128, 1024), (256, 128, 1024), (256, 256, 512), (512, 256, 512), (512, 512, 256), (512, 512, 128)]
    for Cin, Cout, HW in shapes:
        x = torch.randn((B, HW, HW, Cin), device=DEV, generator=g).bfloat16()
        w = (torch.randn((Cout, 9 * Cin), device=DEV, generator=g) * 0.02).bfloat16()
        b = torch.randn((Cout,), device=DEV, generator=g).bfloat16()
        r = torch.randn((B, HW, HW, Cout), device=DEV, generator=g).bfloat16()
        y = torch.empty((B, HW, HW, Cout), device=DEV, dtype=torch.bfloat16)
        mom = torch.empty((B, Cout, 2), device=DEV)
        fl = 2.0 * B * HW * HW * Cout * 9 * Cin
        for name, kw in (("plain", {}), ("residual", dict(res=r)), ("plain + moments", dict(moments=mom)), ("residual + moments", dict(res=r, moments=mom))):
            if old and "moments" in kw:
                continue
            fn = lambda: ops.conv2d_nhwc(x, w, b, HW, HW, Cin, Cout, 3, 3, 1, 1, out=y, **kw)   # noqa: E731
            times, _ = bench._interleaved_probe([fn], 4, 10)
            t = sorted(times[0])[len(times[0]) // 2]
            byts = 2.0 * B * HW * HW * (Cin + Cout * (2 if "res" in kw else 1))
            print(f"Cin={Cin:4d} Cout={Cout:4d} {HW}^2 {name:20s} {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TFLOP/s  {byts / t / 1e12:5.2f} TB/s of tensors", flush=True)


if __name__ == "__main__":
    main()
