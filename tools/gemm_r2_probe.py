#!/usr/bin/env python3
"""Tools-only: the "two residents" GEMM form (gemm_r2.hip, option gemm_r2) against the persistent 256^2 kernel on the model's shapes --
bit-equality first, then interleaved timing with the clock / power the part ran at.  With X2I_LIB_VARIANT=ablate also the K-loops alone
(no epilogue, act2 = 77) and the r2 loop with its memory streams removed (wrong results by design): which stream costs the clock.
    python tools/gemm_r2_probe.py [--quick]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from x2i_amd import _lib, ops  # noqa: E402

DEV = "cuda"
ABL = os.environ.get("X2I_LIB_VARIANT") == "ablate"


def main():
    quick = "--quick" in sys.argv
    g = torch.Generator(device=DEV).manual_seed(0)
    shapes = [(18432, 12288, 3072, "gelu"), (18432, 12288, 3072, "bias"), (18432, 3072, 3072, "res"), (18432, 9216, 3072, "bias"),
              (18432, 3072, 15360, "res"), (18432, 3072, 12288, "res"), (4608, 12288, 3072, "gelu"), (4608, 3072, 15360, "res")]
    if quick:
        shapes = shapes[:3]
    if "--small" in sys.argv:   # the launches of a 512^2 batch-1 step (S = 1536 = 512 text + 1024 image rows) and of a 1024^2 batch-1 step
        shapes = [(1536, 3072, 15360, "res"), (1536, 12288, 3072, "gelu"), (1536, 9216, 3072, "bias"), (1024, 3072, 12288, "res"), (512, 3072, 12288, "res"),
                  (1536, 3072, 3072, "res"), (1024, 12288, 3072, "gelu"), (4608, 3072, 15360, "res"), (4608, 3072, 3072, "res"), (4608, 12288, 3072, "gelu")]
    for M, N, K, kind in shapes:
        A = torch.randn((M, K), device=DEV, generator=g).bfloat16()
        W = (torch.randn((N, K), device=DEV, generator=g) * 0.02).bfloat16()
        b = torch.randn((N,), device=DEV, generator=g).bfloat16()
        R = torch.randn((M, N), device=DEV, generator=g).bfloat16()
        gate = torch.randn((1, N), device=DEV, generator=g)
        kw = {"gelu": dict(act=ops.ACT_GELU_TANH), "bias": {}, "res": dict(res=R, gate=gate, gate_batch_stride=0)}[kind]
        C0 = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
        C1 = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
        fl = 2.0 * M * N * K

        def run(r2, out, extra=None, abl=0):
            def f():
                _lib.set_option("gemm_r2", r2)
                if ABL:
                    _lib.set_option("gemm_ablate", abl)
                ops.gemm(A, W, b, out=out, **dict(kw, **(extra or {})))
                if ABL:
                    _lib.set_option("gemm_ablate", 0)
                _lib.set_option("gemm_r2", 0)
            return f

        run(0, C0)()
        run(1, C1)()
        torch.cuda.synchronize()
        tile = _lib.get_option("last_gemm_tile")
        same = torch.equal(C0, C1)
        fns = [("persistent 256^2", run(0, C0)), ("two residents", run(1, C1))]
        if ABL and kind == "bias":
            fns += [("256^2, no epilogue", run(0, C0, dict(act2=77))), ("r2, no epilogue", run(1, C1, dict(act2=77)))]
            fns += [(f"r2 loop, {nm}", run(1, C1, dict(act2=77), abl=v)) for v, nm in ((1, "no A loads"), (2, "no W DMA"), (3, "no loads"), (7, "MFMA only"))]
        print(f"M={M} N={N} K={K} {kind}: r2 took tile code {tile}, bit-identical to the persistent kernel: {same}", flush=True)
        for name, fn in fns:   # one at a time: the clock / power record belongs to that kernel alone
            times, clk = bench._interleaved_probe([fn], 6, 120)
            t = sorted(times[0])[len(times[0]) // 2]
            sc, pw = clk.get("sclk_mhz") or {}, clk.get("socket_power_w") or {}
            print(f"    {name:22s} {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TFLOP/s   sclk {sc.get('median')} MHz  power {pw.get('median')} W", flush=True)
        times, _ = bench._interleaved_probe([fns[0][1], fns[1][1]], 6, 8)   # ... and interleaved, for the ratio
        m = [sorted(t)[len(t) // 2] for t in times]
        print(f"    interleaved: persistent {m[0] * 1e6:.1f} us, two residents {m[1] * 1e6:.1f} us, ratio {m[1] / m[0]:.3f}", flush=True)


if __name__ == "__main__":
    main()
