#!/usr/bin/env python3
"""Tools-only (measurement library): where a small-batch gated-residual launch spends its time with the parallel split + fix-up (gemm_fx = 1)
and with whole tiles -- per-workgroup K-loop timestamps (100 MHz) of the first 16 workgroups against the launch's duration.
    python tools/fx_timeline.py [M N K]      (default: single-block proj_out of a 512^2 batch-1 step, 1536 x 3072 x 15360)"""
import os
import sys
import torch
os.environ["X2I_LIB_VARIANT"] = "ablate"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import _lib, ops  # noqa: E402

DEV = "cuda"
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (1536, 3072, 15360)
g = torch.Generator(device=DEV).manual_seed(0)
A = torch.randn((M, K), device=DEV, generator=g).bfloat16()
W = (torch.randn((N, K), device=DEV, generator=g) * 0.02).bfloat16()
R = torch.randn((M, N), device=DEV, generator=g).bfloat16()
C = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
gate = torch.randn((1, N), device=DEV, generator=g)
dbg = torch.zeros((16 * 64,), device=DEV, dtype=torch.int64)
for fx in (0, 1):
    _lib.set_option("gemm_fx", fx)
    _lib.set_option("gemm_tile", 256 if fx == 0 else 0)   # (whole tiles: force the persistent 256^2 kernel, the timestamps live there)
    for mode in (0, 80):
        kw = dict(res=R, gate=gate, gate_batch_stride=0)
        if mode:
            kw.update(act2=mode, bias2=dbg.view(torch.float32))
        for _ in range(3):
            ops.gemm(A, W, None, out=C, **kw)
        dbg.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm(A, W, None, out=C, **kw)
        e1.record()
        torch.cuda.synchronize()
        print(f"== gemm_fx = {fx}, tile code {_lib.get_option('last_gemm_tile')}, {'with timestamps' if mode else 'plain'}: launch {e0.elapsed_time(e1) * 1e3:.1f} us", flush=True)
        if mode:
            t = dbg.view(16, 32, 2).cpu()
            starts = [int(t[w, 0, 0]) for w in range(16) if int(t[w, 0, 0]) > 0]
            t00 = min(starts) if starts else 0
            for w in range(16):
                row = t[w]
                n = int((row[:, 0] > 0).sum())
                if n == 0:
                    continue
                print(f"   wg {w:2d}: " + "  ".join(f"[{(int(row[i, 0]) - t00) / 100:6.1f} .. {(int(row[i, 1]) - t00) / 100:6.1f}]" for i in range(n)))
_lib.set_option("gemm_fx", 0)
_lib.set_option("gemm_tile", 0)
