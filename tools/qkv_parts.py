#!/usr/bin/env python3
"""Tools-only: what the fused-QKV epilogue's microseconds ARE.  Times the single-block fused-QKV launch (M = B * 4608,
N = 9216, K = 3072, span-permuted V^T) on the product library and on the part-ablated product-flag builds of tools/qkv_parts_build.sh, one
process per library (the library is chosen at import), `rounds` interleaved rounds.

    bash tools/qkv_parts_build.sh 86 87 88 89 90 91 92 && python tools/qkv_parts.py
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = (("", "product"), ("qkv86", "no cos / sin loads"), ("qkv87", "no 16-lane RMS reduction"), ("qkv88", "no Q / K stores"),
            ("qkv89", "no V^T stores"), ("qkv90", "q / k tiles parked only"), ("qkv91", "v tiles parked only"), ("qkv92", "no epilogue at all"))

if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import torch
    sys.path.insert(0, ROOT)
    from x2i_amd import ops
    DEV = "cuda"
    g = torch.Generator(device=DEV).manual_seed(0)
    B, H, Sj, K = 4, 24, 4608, 3072
    M, Nq = B * Sj, 3 * H * 128
    A = torch.randn((M, K), device=DEV, generator=g).bfloat16()
    Wq = (torch.randn((Nq, K), device=DEV, generator=g) * 0.02).bfloat16()
    bq = torch.randn((Nq,), device=DEV, generator=g).bfloat16()
    Q, Kk = (torch.zeros((B, H, Sj, 128), device=DEV, dtype=torch.bfloat16) for _ in range(2))
    VT = torch.zeros((B, H, 128, Sj), device=DEV, dtype=torch.bfloat16)
    nq, nk = (torch.ones((128,), device=DEV, dtype=torch.bfloat16) for _ in range(2))
    ang = torch.randn((Sj, 64), device=DEV, generator=g)
    cos, sin = torch.cos(ang).repeat_interleave(2, 1).contiguous(), torch.sin(ang).repeat_interleave(2, 1).contiguous()
    C = torch.empty((M, Nq), device=DEV, dtype=torch.bfloat16)

    def f():
        ops.gemm_qkv(A, Wq, bq, Q, Kk, VT, nq, nk, cos, sin, M=M, H=H, Spad=Sj, tok_off=0, rows_per_sample=Sj, vt_perm=True)

    def plain():
        ops.gemm(A, Wq, bq, out=C)
    pairs = ops.rope_pairs(cos, sin)

    def fpair():   # the pair-form RoPE table (x2i_qkv_desc.sin == NULL): what flux.py passes since round 6
        ops.gemm_qkv(A, Wq, bq, Q, Kk, VT, nq, nk, pairs, None, M=M, H=H, Spad=Sj, tok_off=0, rows_per_sample=Sj, vt_perm=True)
    res = []
    for fn in (f, plain, fpair):
        for _ in range(5):
            fn()
        ts = []
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                fn()
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 10 * 1e3)
        res.append(sorted(ts)[2])
    print("RESULT %.1f %.1f %.1f" % (res[0], res[1], res[2]))
    sys.exit(0)

rounds = 3
acc = {v: [] for v, _ in VARIANTS}
for r in range(rounds):
    for v, _ in VARIANTS:
        if v and not os.path.exists(os.path.join(ROOT, "x2i_amd", "libx2i_hip_%s.so" % v)):
            continue
        env = dict(os.environ, X2I_LIB_VARIANT=v)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, capture_output=True, text=True).stdout
        line = [l for l in out.splitlines() if l.startswith("RESULT")]
        if line:
            acc[v].append(tuple(float(x) for x in line[0].split()[1:]))
print("fused-QKV launch M=18432 N=9216 K=3072 (B=4, span-permuted V^T); median of 5 x 10 launches per process, %d processes per library" % rounds)
print("%-28s %10s %14s %14s" % ("library", "fused us", "plain-bias us", "fused, pair-form table us"))
for v, name in VARIANTS:
    if acc[v]:
        a = sorted(x[0] for x in acc[v])[len(acc[v]) // 2]
        b = sorted(x[1] for x in acc[v])[len(acc[v]) // 2]
        c = sorted(x[2] for x in acc[v])[len(acc[v]) // 2]
        print("%-28s %10.1f %14.1f %14.1f   %s" % (name, a, b, c, " ".join("%.1f/%.1f" % (x[0], x[2]) for x in acc[v])))
