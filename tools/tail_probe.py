#!/usr/bin/env python3
"""Peeled-tail vs one-launch timing for the single-block GEMM shapes (option gemm_split_tail)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import _lib, ops

def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

D = 3072
for B in (1, 2, 4):
    for (M, N, K, name, act) in [(B * 4608, 4 * D, D, "s_mlp", 1), (B * 4608, D, 5 * D, "s_out", 0), (B * 4608, 3 * D, D, "s_qkv", 0), (B * 4096, D, D, "attn_out", 0),
                                 (B * 4096, 3 * D, D, "qkv_img", 0)]:
        A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        r = []
        for split in (1, 0, 1, 0):
            _lib.set_option("gemm_split_tail", split)
            r.append(timeit(lambda: ops.gemm(A, W, b, out=out, act=act)))
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        print(f"B={B} {name:9s} tiles={tiles:5d} rounds={tiles/256:6.3f}  split {min(r[0], r[2]):7.1f} us  whole {min(r[1], r[3]):7.1f} us")
_lib.set_option("gemm_split_tail", 1)
