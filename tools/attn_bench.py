"""Attention kernel forms at the DiT shape (B, 24 heads, S = 4608): python tools/attn_bench.py [B]"""
import math
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from x2i_amd import _lib, ops  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    H, S, D = 24, 4608, 3072
    Spad = ops.pad128(S)
    rnd = lambda *sh: torch.randn(sh, device="cuda").bfloat16()  # noqa: E731
    Q, K_, VT = rnd(B, H, Spad, 128), rnd(B, H, Spad, 128), rnd(B, H, 128, Spad)
    O = torch.empty((B, S, D), device="cuda", dtype=torch.bfloat16)
    names = {4: "4-wave", 5: "ping-pong", 7: "ping-pong, DMA in vector phase", 8: "ping-pong, 4-deep rings + fragment prefetch",
             9: "hand-scheduled, one wave per SIMD (attention_w4.hip)", 10: "4-wave organisation on 16x16x32 MFMAs (attention16.hip, A/B)"}
    ref = None
    perm = torch.tensor([16 * ((kk >> 2) & 1) + 4 * (kk >> 3) + (kk & 3) for kk in range(32)], device="cuda")
    VTP = VT.view(B, H, 128, Spad // 32, 32)[..., perm].reshape(B, H, 128, Spad).contiguous()   # variant 11: keys permuted within 32-key spans
    names[11] = "... with V^T pre-permuted (one 16-byte fragment read)"
    names[12] = "hand-scheduled, one wave per SIMD, 16x16x32 MFMAs (attention_w16.hip, V^T pre-permuted)"
    for var in (4, 10, 11, 12):   # the 16x16x32 A/B kernel against its 32x32x16 partner: same values up to rounding?
        _lib.set_option("attn_variant", var)
        ops.attention(Q, K_, VTP if var >= 11 else VT, O, B, H, S, Spad, D, S * D, 1 / math.sqrt(128))
        torch.cuda.synchronize()
        if ref is None:
            ref = O.float().clone()
        else:
            print(f"variant {var} vs variant 4: rel-L2 {float((O.float() - ref).norm() / ref.norm()):.3e}")
    for var in (9, 12, 8, 4, 10, 11, 9, 12, 8, 4, 9, 12):
        _lib.set_option("attn_variant", var)
        t = timeit(lambda: ops.attention(Q, K_, VTP if var >= 11 else VT, O, B, H, S, Spad, D, S * D, 1 / math.sqrt(128)))
        print(f"attention[{names[var]}] B={B}: {t*1e3:8.3f} ms  {4*B*H*S*S*128/t/1e12:8.1f} TFLOP/s")
    _lib.set_option("attn_variant", 0)


if __name__ == "__main__" and not (len(sys.argv) > 2 and sys.argv[2] == "bwd"):
    main()


def backward_bench():
    """fused attention backward (statistics + dQ + dK/dV passes) at the DiT shape: python tools/attn_bench.py B bwd"""
    B = int(sys.argv[1])
    H, S = 24, 4608
    Spad = ops.pad128(S)
    rnd = lambda *sh: torch.randn(sh, device="cuda").bfloat16()  # noqa: E731
    Q, K_, V = rnd(B, H, Spad, 128), rnd(B, H, Spad, 128), rnd(B, H, Spad, 128)
    QT, KT = ops.transpose(Q.view(B * H, Spad, 128)), ops.transpose(K_.view(B * H, Spad, 128))
    dOh = rnd(B * H, Spad, 128)
    dOT = ops.transpose(dOh)
    Dv = torch.randn((B, H, Spad), device="cuda") * 0.1
    L = torch.empty((B, H, Spad), device="cuda")
    dQ, dK, dV = (torch.empty((B, H, Spad, 128), device="cuda", dtype=torch.bfloat16) for _ in range(3))
    t = timeit(lambda: ops.attention_bwd(Q, K_, V, QT, KT, dOh, dOT, L, Dv, dQ, dK, dV, B, H, S, Spad, 1 / math.sqrt(128)))
    fl = (1 + 3 + 4) * 2 * B * H * S * S * 128
    print(f"attention backward (3 passes, 8 MFMA groups) B={B}: {t*1e3:8.3f} ms  {fl/t/1e12:8.1f} TFLOP/s")
    # without the statistics pass (the training step hands over the forward kernel's statistics), per option set; passes one after the other
    # (attn_bwd_overlap = 0) so that each pass's own time shows
    fl7 = (3 + 4) * 2 * B * H * S * S * 128
    abl = bool(_lib.load().x2i_is_ablation_build())   # the round-2 kernels (attn_bwd_pipe = 0) live in the measurement library: X2I_LIB_VARIANT=ablate
    for opts in (({}, {"attn_bwd_pipe": 0}, {"attn_bwd_overlap": 0}, {"attn_bwd_overlap": 0, "attn_bwd_pipe": 0}, {}) if abl else ({}, {"attn_bwd_overlap": 0}, {})):
        old = {k: _lib.set_option(k, v) for k, v in opts.items()}
        t = timeit(lambda: ops.attention_bwd(Q, K_, V, QT, KT, dOh, dOT, L, Dv, dQ, dK, dV, B, H, S, Spad, 1 / math.sqrt(128), have_lse=True))
        for k, v in old.items():
            _lib.set_option(k, v)
        print(f"attention backward (dQ + dK/dV passes, 7 MFMA groups) B={B} {opts}: {t*1e3:8.3f} ms  {fl7/t/1e12:8.1f} TFLOP/s")


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[2] == "bwd":
    backward_bench()
