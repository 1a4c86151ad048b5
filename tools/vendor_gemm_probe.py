#!/usr/bin/env python3
"""Tools-only probe (never imported by the product): the vendor GEMM (torch F.linear -> hipBLASLt) beside x2i_gemm_bf16 on the
DiT linear shapes, same random operands, interleaved rounds in one process (guide rule 24).  Run it under
`rocprofv3 --kernel-trace --stats` to learn WHICH Tensile kernel the vendor library picks for each shape: the kernel name spells
macro tile, MFMA shape, waves, DirectToLds, prefetch depth and stream-K.

    python tools/vendor_gemm_probe.py [B]          # prints one JSON line per shape
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import _lib, ops  # noqa: E402

DEV = "cuda"


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=DEV, dtype=torch.float32) * scale).to(torch.bfloat16)


def time_rounds(fns, rounds=5, iters=8):
    """fns: dict name -> callable; interleaved rounds, returns name -> (median_s, min_s)."""
    for f in fns.values():
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    res = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(iters):
                f()
            e.record()
            torch.cuda.synchronize()
            res[k].append(s.elapsed_time(e) / iters * 1e-3)
    return {k: (sorted(v)[len(v) // 2], min(v)) for k, v in res.items()}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    D, Si, St = 3072, 4096, 512
    S = Si + St
    shapes = [(B * Si, 3 * D, D, "qkv_img"), (B * Si, D, D, "attn_out"), (B * Si, 4 * D, D, "ff_in"), (B * Si, D, 4 * D, "ff_out"),
              (B * S, 7 * D, D, "single_in"), (B * S, D, 5 * D, "single_out"), (B * S, 4 * D, D, "probe_proj_mlp"),
              (B * St, 3 * D, D, "qkv_txt")]
    print(f"device: {torch.cuda.get_device_name(0)}  B={B}", flush=True)
    for (M, N, K, name) in shapes:
        A, W, b = rnd(M, K), rnd(N, K, scale=0.02), rnd(N)
        out = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
        def x2i(w4, persist=1):
            _lib.set_option("gemm_w4", w4)
            _lib.set_option("gemm_persist", persist)
            ops.gemm(A, W, b, out=out)

        fns = {"vendor": lambda: F.linear(A, W, b), "x2i": lambda: x2i(1), "x2i_w4_onetile": lambda: x2i(1, 0), "x2i_8wave": lambda: x2i(0)}
        r = time_rounds(fns)
        x2i(1, 1)
        fl = 2.0 * M * N * K
        ref = F.linear(A, W, b).float()
        err = ((out.float() - ref).norm() / ref.norm()).item()
        print(json.dumps({"shape": name, "M": M, "N": N, "K": K,
                          "vendor_TF_median": round(fl / r["vendor"][0] / 1e12, 1), "vendor_TF_best": round(fl / r["vendor"][1] / 1e12, 1),
                          "x2i_TF_median": round(fl / r["x2i"][0] / 1e12, 1), "x2i_TF_best": round(fl / r["x2i"][1] / 1e12, 1),
                          "x2i_w4_onetile_TF_median": round(fl / r["x2i_w4_onetile"][0] / 1e12, 1),
                          "x2i_8wave_TF_median": round(fl / r["x2i_8wave"][0] / 1e12, 1),
                          "x2i_tile": _lib.get_option("last_gemm_tile"), "rel_l2_vs_vendor": err}), flush=True)
        del A, W, out, ref


if __name__ == "__main__":
    main()
