#!/usr/bin/env python3
"""Tools-only (measurement library): per-unit K-loop timestamps of the persistent GEMM's first 16 workgroups (100 MHz clock) on
M = 18432, N = 12288, K = 3072 without epilogue: how long a K-loop statement runs and how long the matrix pipe waits between two."""
import os
import sys
import torch
os.environ["X2I_LIB_VARIANT"] = os.environ.get("X2I_TIMELINE_LIB", "ablate")   # (another measurement build: X2I_TIMELINE_LIB=<name>)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import ops  # noqa: E402

DEV = "cuda"
g = torch.Generator(device=DEV).manual_seed(0)


def show(t, nwg=3):
    for w in range(nwg):
        row = t[w]
        n = int((row[:, 0] > 0).sum())
        t0 = int(row[0, 0])
        dur = [(int(row[i, 1]) - int(row[i, 0])) / 100 for i in range(n)]
        gap = [(int(row[i + 1, 0]) - int(row[i, 1])) / 100 for i in range(n - 1)]
        print(f"wg {w}: {n} units; K-loop us: " + " ".join(f"{d:.1f}" for d in dur))
        print("        gaps us: " + " ".join(f"{d:.2f}" for d in gap) + f"   total {(int(row[n - 1, 1]) - t0) / 100:.1f} us")


if "--fp8" in sys.argv:
    # the persistent e4m3 kernel (gen_gemm256f8.py): the two roofline launches
    shapes = ((18432, 12288, 3072, "gelu_e4m3"), (18432, 3072, 15360, "res"), (18432, 12288, 3072, "plain"))
    if "--split" in sys.argv:   # what the GELU and what the e4m3 output path add to the plain epilogue, separately
        shapes = ((18432, 12288, 3072, "plain_e4m3"), (18432, 12288, 3072, "gelu_bf16"), (18432, 12288, 3072, "gelu_e4m3"), (18432, 12288, 3072, "plain"))
    if "--hot" in sys.argv:   # one round of tiles whose operands fit the caches: what the K-loop takes when nothing has to come from HBM
        shapes = ((4096, 4096, 15360, "plain"), (4096, 4096, 3072, "plain"), (2048, 8192, 15360, "plain"))
    for (M, N, K, kind) in shapes:
        A8, sa = ops.quantize_rows_fp8(torch.randn((M, K), device=DEV, generator=g).bfloat16())
        W8, sw = ops.quantize_rows_fp8((torch.randn((N, K), device=DEV, generator=g) * 0.02).bfloat16())
        dbg = torch.zeros((16 * 64,), device=DEV, dtype=torch.int64)
        out = torch.empty((M, N), device=DEV, dtype=ops.FP8 if kind.endswith("e4m3") else torch.bfloat16)
        gate = torch.randn((1, N), device=DEV, generator=g)
        for mode, what in ((79, "no epilogue"), (80, "with its epilogue")):
            for it in range(4):
                dbg.zero_()
                if kind.endswith("e4m3"):
                    ops.gemm_fp8(A8, W8, None, out=out, a_scale=sa, w_scale=sw, act=1 if kind == "gelu_e4m3" else 0, out_fp8=True, _act2=mode,
                                 _bias2=dbg.view(torch.float32))
                elif kind == "gelu_bf16":
                    ops.gemm_fp8(A8, W8, None, out=out, a_scale=sa, w_scale=sw, act=1, _act2=mode, _bias2=dbg.view(torch.float32))
                elif kind == "res":
                    ops.gemm_fp8(A8, W8, None, out=out, w_scale=sw, res=out, gate=gate, _act2=mode, _bias2=dbg.view(torch.float32))
                else:
                    ops.gemm_fp8(A8, W8, None, out=out, a_scale=sa, w_scale=sw, _act2=mode, _bias2=dbg.view(torch.float32))
            torch.cuda.synchronize()
            print(f"== e4m3 M={M} N={N} K={K} {kind}: {what}   ({K // 128} K-tiles per tile; 100 % matrix pipe = {K // 128 * 2048 / 2.4e3:.1f} us at 2.4 GHz)")
            show(dbg.view(16, 32, 2).cpu())
    sys.exit(0)

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
# (round 6: the split of this epilogue into its parts is tools/qkv_parts.py on PRODUCT-flag builds -- this build's timestamps and spills run the
# q / k epilogue at 2.5x the product's time, profiles/r04ad_*, r06a_qkv_parts_measurement_build.log)
PARTS = ((80, "product epilogue"), (79, "no epilogue at all"))


def launch_us(mode, iters=10):
    f = lambda: ops.gemm_qkv(A, Wq, bq, Q, Kk, VT, nq, nk, cos, sin, M=M, H=H, Spad=Sj, tok_off=0, rows_per_sample=Sj, vt_perm=True, _act2=mode, _bias2=dbg.view(torch.float32))  # noqa: E731
    for _ in range(3):
        f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for mode, what in PARTS:
    us = launch_us(mode)
    dbg.zero_()
    ops.gemm_qkv(A, Wq, bq, Q, Kk, VT, nq, nk, cos, sin, M=M, H=H, Spad=Sj, tok_off=0, rows_per_sample=Sj, vt_perm=True, _act2=mode, _bias2=dbg.view(torch.float32))
    torch.cuda.synchronize()
    t = dbg.view(16, 32, 2).cpu()
    print(f"== fused QKV epilogue, {what}: {us:.1f} us per launch (tiles of the q / k / v sections in the XCD order)")
    gaps = []
    for w in range(16):
        row = t[w]
        n = int((row[:, 0] > 0).sum())
        dur = [(int(row[i, 1]) - int(row[i, 0])) / 100 for i in range(n)]
        gap = [(int(row[i + 1, 0]) - int(row[i, 1])) / 100 for i in range(n - 1)]
        gaps += gap
        if w < 3:
            print(f"wg {w}: {n} units; K-loop us: " + " ".join(f"{d:.1f}" for d in dur))
            print("        gaps us: " + " ".join(f"{d:.2f}" for d in gap))
    if gaps:
        gs = sorted(gaps)
        print(f"   mean gap {sum(gaps) / len(gaps):.2f} us, median {gs[len(gs) // 2]:.2f}, min {gs[0]:.2f}, max {gs[-1]:.2f} over {len(gaps)} gaps of 16 workgroups")
