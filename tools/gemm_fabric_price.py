#!/usr/bin/env python3
"""Tools-only (measurement library): what the L2 <-> fabric traffic of the persistent GEMM COSTS at the power cap.  The roofline
pair's launches as the product issues them, and the same launches with every workgroup's operand panels folded onto (a) one tile's panels -- they
stay in each XCD's L2, fabric fetch ~ 0 -- and (b) a 4 x 8 tile patch -- out of L2, inside the Infinity Cache: the product's fabric traffic without
its HBM traffic.  Same MFMA stream, same LDS traffic, same epilogue and stores; interleaved rounds; sclk / socket power beside every figure.

    X2I_LIB_VARIANT=ablate python tools/gemm_fabric_price.py
"""
import os
import sys

os.environ.setdefault("X2I_LIB_VARIANT", "ablate")
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import ClockPowerSampler  # noqa: E402
from x2i_amd import ops  # noqa: E402

DEV = "cuda"


def main():
    g = torch.Generator(device=DEV).manual_seed(0)
    M = 18432
    for (N, K, act, name) in ((12288, 3072, 1, "proj_mlp + GELU"), (3072, 15360, 0, "proj_out")):
        A = torch.randn((M, K), device=DEV, generator=g).bfloat16()
        W = (torch.randn((N, K), device=DEV, generator=g) * 0.02).bfloat16()
        b = torch.randn((N,), device=DEV, generator=g).bfloat16()
        C = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
        fl = 2.0 * M * N * K
        variants = ((0, "product launch"), (76, "panels folded onto a 4 x 8 tile patch (Infinity Cache)"), (75, "panels folded onto one tile (L2)"))
        res = {v: [] for v, _ in variants}
        pw = {v: [] for v, _ in variants}
        for v, _ in variants:
            for _ in range(3):
                ops.gemm(A, W, b, out=C, act=act, act2=v)
        torch.cuda.synchronize()
        for rnd in range(5):
            for v, _ in variants:
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with ClockPowerSampler(torch.cuda.current_device()) as smp:
                    s.record()
                    for _ in range(40):
                        ops.gemm(A, W, b, out=C, act=act, act2=v)
                    e.record()
                    torch.cuda.synchronize()
                res[v].append(s.elapsed_time(e) / 40 * 1e-3)
                pw[v] += smp.samples
        print(f"== M={M} N={N} K={K} {name}: algorithmic operand bytes {2.0 * (M * K + N * K) / 1e6:.0f} MB, output {2.0 * M * N / 1e6:.0f} MB")
        for v, what in variants:
            t = sorted(res[v])[len(res[v]) // 2]
            clk = sorted(c for c, _ in pw[v] if c is not None)
            pws = sorted(p for _, p in pw[v] if p is not None)
            print(f"  {what:58s} {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TFLOP/s   sclk {clk[len(clk) // 2] if clk else None} MHz   "
                  f"power {pws[len(pws) // 2] if pws else None} W   rounds: " + " ".join(f"{x * 1e6:.0f}" for x in res[v]))


if __name__ == "__main__":
    main()
