#!/usr/bin/env python3
"""A/B the bf16 GEMM at the DiT shapes (B=4) -- prints TFLOP/s per shape; run once per library variant."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import ops

def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

B, D = 4, 3072
tot_f, tot_t = 0, 0
for (M, N, K, name, act) in [(B * 4096, 3 * D, D, "qkv_img", 0), (B * 4096, D, D, "attn_out", 0), (B * 4096, 4 * D, D, "ff_in", 1),
                             (B * 4096, D, 4 * D, "ff_out", 0), (B * 4608, 3 * D, D, "s_qkv", 0), (B * 4608, 4 * D, D, "s_mlp", 1),
                             (B * 4608, D, 5 * D, "s_out", 0)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t = min(timeit(lambda: ops.gemm(A, W, b, out=out, act=act)) for _ in range(2))
    tot_f += 2 * M * N * K; tot_t += t
    print(f"{name:9s} {2*M*N*K/t/1e12:7.1f} TF", end=" |")
print(f" ALL {tot_f/tot_t/1e12:7.1f} TF")
