"""debug: attention variant 9 vs fp32 pieces"""
import math, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import _lib, ops
DEV = "cuda"
def run(Q, K, VT, B, H, S, var=9):
    Spad = ops.pad128(S)
    O = torch.zeros((B, S, H * 128), device=DEV, dtype=torch.bfloat16)
    _lib.set_option("attn_variant", var)
    ops.attention(Q, K, VT, O, B, H, S, Spad, H * 128, S * H * 128, 1 / math.sqrt(128))
    torch.cuda.synchronize()
    return O.float()
def rel(a, b):
    return ((a - b).norm() / b.norm()).item()
S = 128
Spad = 128
g = torch.Generator(device=DEV).manual_seed(1)
Q = torch.randn((1, 1, Spad, 128), device=DEV, generator=g).bfloat16()
K = torch.randn((1, 1, Spad, 128), device=DEV, generator=g).bfloat16()
VT = torch.randn((1, 1, 128, Spad), device=DEV, generator=g).bfloat16()
O = run(Q, K, VT, 1, 1, S)[0]
q, k, v = Q[0, 0].float(), K[0, 0].float(), VT[0, 0].float().t()
sc = q @ k.t() / math.sqrt(128)
full = torch.softmax(sc, -1) @ v
print("full", rel(O, full))
p = torch.softmax(sc, -1)
t0 = p[:, :64] @ v[:64]
t1 = p[:, 64:] @ v[64:]
print("vs tile0 part only", rel(O, t0), " vs tile1 part only", rel(O, t1))
print("O - t0 vs t1:", rel(O - t0, t1), "  O - t1 vs t0:", rel(O - t1, t0))
# unnormalised variants
e = torch.exp(sc - sc.max(-1, keepdim=True).values)
l0, l1 = e[:, :64].sum(-1, keepdim=True), e[:, 64:].sum(-1, keepdim=True)
print("softmax within tile0 only", rel(O, (e[:, :64] @ v[:64]) / l0), " tile1 only", rel(O, (e[:, 64:] @ v[64:]) / l1))
print("(e0 v0 + e1 v1)/l0", rel(O, (e @ v) / l0), " /l1", rel(O, (e @ v) / l1))
for r0 in range(0, 128, 32):
    print("rows", r0, rel(O[r0:r0 + 32], full[r0:r0 + 32]), [round(rel(O[r0:r0+32, c:c+32], full[r0:r0+32, c:c+32]), 3) for c in range(0, 128, 32)])
# V-side check: make V constant per tile
VT2 = VT.clone(); VT2[:, :, :, :64] = 1.0; VT2[:, :, :, 64:] = 3.0
O2 = run(Q, K, VT2, 1, 1, S)[0]
w1 = p[:, 64:].sum(-1)
print("expected 1 + 2*w1 (first rows):", (1 + 2 * w1)[:6].tolist(), " got:", O2[:6, 0].tolist(), O2[:6, 77].tolist())
VTa = VT.clone(); VTa[:, :, :, 64:] = 0
VTb = VT.clone(); VTb[:, :, :, :64] = 0
Oa = run(Q, K, VTa, 1, 1, S)[0]
Ob = run(Q, K, VTb, 1, 1, S)[0]
print("tile0 contribution:", rel(Oa, t0), [round(rel(Oa[:, c:c+32], t0[:, c:c+32]), 3) for c in range(0, 128, 32)])
print("tile1 contribution:", rel(Ob, t1), [round(rel(Ob[:, c:c+32], t1[:, c:c+32]), 3) for c in range(0, 128, 32)])
# per-key weights of tile 1: V^T = identity-like: dim d = key (64 + d) for d < 64
VTc = torch.zeros_like(VT)
for d in range(64):
    VTc[0, 0, d, 64 + d] = 1.0
    VTc[0, 0, 64 + d, d] = 1.0
Oc = run(Q, K, VTc, 1, 1, S)[0]
print("weights tile1 (dims 0..63 = keys 64..127):", rel(Oc[:, :64], p[:, 64:]), " tile0 (dims 64..127 = keys 0..63):", rel(Oc[:, 64:], p[:, :64]))
row = 5
print("row 5 tile1 weights exp:", [round(x, 3) for x in p[row, 64:80].tolist()])
print("row 5 tile1 weights got:", [round(x, 3) for x in Oc[row, :16].tolist()])
print("row 5 tile0 weights exp:", [round(x, 3) for x in p[row, :16].tolist()])
print("row 5 tile0 weights got:", [round(x, 3) for x in Oc[row, 64:80].tolist()])

print("tile1 weight error per 16 keys:", [round(rel(Oc[:, c:c+16], p[:, 64+c:64+c+16]), 3) for c in range(0, 64, 16)])
print("tile0 weight error per 16 keys:", [round(rel(Oc[:, 64+c:64+c+16], p[:, c:c+16]), 3) for c in range(0, 64, 16)])
