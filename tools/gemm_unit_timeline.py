#!/usr/bin/env python3
"""Tools-only (measurement library): per-unit K-loop timestamps of the persistent GEMM's first 16 workgroups (100 MHz clock) on
M = 18432, N = 12288, K = 3072 without epilogue: how long a K-loop statement runs and how long the matrix pipe waits between two."""
import os
import sys
import torch
os.environ["X2I_LIB_VARIANT"] = "ablate"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import ops  # noqa: E402

DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(0)
M, N, K = 18432, 12288, 3072
A = torch.randn((M, K), device=DEV, generator=g).bfloat16()
W = (torch.randn((N, K), device=DEV, generator=g) * 0.02).bfloat16()
C = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
dbg = torch.zeros((16 * 64,), device=DEV, dtype=torch.int64)
R = torch.randn((M, N), device=DEV, generator=g).bfloat16()
gate = torch.randn((1, N), device=DEV, generator=g)
for mode, what, kw in ((79, "no epilogue", {}), (80, "bias epilogue with its stores", {}), (80, "gated residual epilogue", dict(res=R, gate=gate, gate_batch_stride=0))):
    for it in range(4):
        dbg.zero_()
        ops.gemm(A, W, None, out=C, act2=mode, bias2=dbg.view(torch.float32), **kw)
    torch.cuda.synchronize()
    t = dbg.view(16, 32, 2).cpu()
    print(f"== {what}")
    for w in (0, 1, 2):
        row = t[w]
        n = int((row[:, 0] > 0).sum())
        t0 = int(row[0, 0])
        dur = [(int(row[i, 1]) - int(row[i, 0])) / 100 for i in range(n)]
        gap = [(int(row[i + 1, 0]) - int(row[i, 1])) / 100 for i in range(n - 1)]
        print(f"wg {w}: {n} units; K-loop us: " + " ".join(f"{d:.1f}" for d in dur))
        print("        gaps us: " + " ".join(f"{d:.2f}" for d in gap) + f"   total {(int(row[n - 1, 1]) - t0) / 100:.1f} us")

# the fused QKV epilogue (single-block geometry: one GEMM over B*S rows, N = 3 * 24 * 128)
H, Sj = 24, 4608
Nq = 3 * H * 128
Wq = (torch.randn((Nq, K), device=DEV, generator=g) * 0.02).bfloat16()
bq = torch.randn((Nq,), device=DEV, generator=g).bfloat16()
Q, Kk = (torch.zeros((4, H, Sj, 128), device=DEV, dtype=torch.bfloat16) for _ in range(2))
VT = torch.zeros((4, H, 128, Sj), device=DEV, dtype=torch.bfloat16)
nq, nk = (torch.ones((128,), device=DEV, dtype=torch.bfloat16) for _ in range(2))
ang = torch.randn((Sj, 64), device=DEV, generator=g)
cos, sin = torch.cos(ang).repeat_interleave(2, 1).contiguous(), torch.sin(ang).repeat_interleave(2, 1).contiguous()
for it in range(4):
    dbg.zero_()
    ops.gemm_qkv(A, Wq, bq, Q, Kk, VT, nq, nk, cos, sin, M=M, H=H, Spad=Sj, tok_off=0, rows_per_sample=Sj, _act2=80, _bias2=dbg.view(torch.float32))
torch.cuda.synchronize()
t = dbg.view(16, 32, 2).cpu()
print("== fused QKV epilogue (tiles of the q / k / v sections in the XCD order)")
for w in (0, 1, 2):
    row = t[w]
    n = int((row[:, 0] > 0).sum())
    dur = [(int(row[i, 1]) - int(row[i, 0])) / 100 for i in range(n)]
    gap = [(int(row[i + 1, 0]) - int(row[i, 1])) / 100 for i in range(n - 1)]
    print(f"wg {w}: {n} units; K-loop us: " + " ".join(f"{d:.1f}" for d in dur))
    print("        gaps us: " + " ".join(f"{d:.2f}" for d in gap))
