#!/usr/bin/env python3
"""Tools-only A/B of the 4-wave K-loop schedules (measurement library: X2I_LIB_VARIANT=ablate; option gemm_w4 = 1 product schedule,
2 = non-temporal A pieces, 3 = non-temporal W pieces, 4 = one barrier for both operands of the next tile), one-tile-per-workgroup
kernel, plain epilogue, interleaved rounds in one process."""
import json
import os
import sys

import torch

os.environ["X2I_LIB_VARIANT"] = "ablate"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import _lib, ops  # noqa: E402
from tools.vendor_gemm_probe import rnd, time_rounds  # noqa: E402


def main():
    _lib.set_option("gemm_persist", 0)
    for (M, N, K, name) in [(16384, 12288, 3072, "ff_in"), (16384, 3072, 12288, "ff_out"), (16384, 9216, 3072, "qkv_img"), (18432, 21504, 3072, "single_in")]:
        A, W, b = rnd(M, K), rnd(N, K, scale=0.02), rnd(N)
        out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)

        def run(v):
            _lib.set_option("gemm_w4", v)
            ops.gemm(A, W, b, out=out)
        ref = None
        res = {}
        for v in (1, 2, 3, 4):
            run(v)
            if ref is None:
                ref = out.clone()
            assert torch.equal(out, ref), v
        r = time_rounds({f"v{v}": (lambda v=v: run(v)) for v in (1, 2, 3, 4)}, rounds=7, iters=8)
        fl = 2.0 * M * N * K
        print(json.dumps({"shape": name, **{k: round(fl / t[0] / 1e12, 1) for k, t in r.items()}}), flush=True)
    _lib.set_option("gemm_w4", 1)


if __name__ == "__main__":
    main()
