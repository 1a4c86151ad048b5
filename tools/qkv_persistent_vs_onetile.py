#!/usr/bin/env python3
"""Tools-only: the fused QKV projection at the single-block shape through the persistent kernel, the one-tile kernel and the two-step
form (plain GEMM + x2i_qkv_split_bf16): counts of differing elements (all three must agree bit for bit), with the first offenders."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import ops, _lib
DEV = "cuda"
B, St, Si = 4, 0, 4608
H, Kd = 24, 3072
D, S = H * 128, St + Si
Spad = ops.pad128(S)
gen = torch.Generator(device=DEV).manual_seed(3)
W = (torch.randn((3 * D, Kd), device=DEV, generator=gen) * 0.02).bfloat16()
bias = (torch.randn((3 * D,), device=DEV, generator=gen) * 0.5).bfloat16()
X = torch.randn((B, S, Kd), device=DEV, generator=gen).bfloat16()
nq, nk = ((1 + 0.2 * torch.randn((128,), device=DEV, generator=gen)).bfloat16() for _ in range(2))
ang = torch.randn((S, 64), device=DEV, generator=gen) * 3
cos, sin = torch.cos(ang).repeat_interleave(2, 1).contiguous(), torch.sin(ang).repeat_interleave(2, 1).contiguous()
def run():
    Q, K, VT = (torch.zeros((B, H, Spad, 128), device=DEV, dtype=torch.bfloat16), torch.zeros((B, H, Spad, 128), device=DEV, dtype=torch.bfloat16),
                torch.zeros((B, H, 128, Spad), device=DEV, dtype=torch.bfloat16))
    ops.gemm_qkv(X, W, bias, Q, K, VT, nq, nk, cos, sin, M=B * S, H=H, Spad=Spad, tok_off=0, rows_per_sample=S)
    return Q, K, VT
_lib.set_option("gemm_persist", 1)
a = run()
_lib.set_option("gemm_persist", 0)
c = run()
for name, x, y in zip("Q K VT".split(), a, c):
    d = (x != y)
    print(name, "differing", int(d.sum()), "of", d.numel())
    if d.any():
        idx = d.nonzero()[:6].tolist()
        print("  first", idx, [(float(x[tuple(i)]), float(y[tuple(i)])) for i in idx[:6]])
        if name != "VT":
            bad_tok = d.any(-1).nonzero()
            print("  tokens affected:", bad_tok[:8].tolist(), "dims per token", d.sum(-1).float()[d.any(-1)][:8].tolist())
# two-step reference: plain GEMM then x2i_qkv_split_bf16
QKV = torch.empty((B * S, 3 * D), device=DEV, dtype=torch.bfloat16)
ops.gemm(X.view(B * S, Kd), W, bias, out=QKV)
Q2, K2, V2 = (torch.zeros((B, H, Spad, 128), device=DEV, dtype=torch.bfloat16), torch.zeros((B, H, Spad, 128), device=DEV, dtype=torch.bfloat16),
              torch.zeros((B, H, 128, Spad), device=DEV, dtype=torch.bfloat16))
ops.qkv_split(None, QKV, 3 * D, 3 * D, B, S, 0, H, None, None, nq, nk, cos, sin, Q2, K2, V2, Spad)
for nm, t in (("persistent", a), ("one-tile", c)):
    print(nm, "vs two-step:", [int((x != y).sum()) for x, y in zip(t, (Q2, K2, V2))])
