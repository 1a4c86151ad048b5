"""Stream-K attention (x2i_attention_vp_ws_bf16) against whole items at the DiT shape (24 heads, S = 4608): python tools/attn_sk_bench.py [B ...]"""
import math
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from x2i_amd import _lib, ops  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def cut_sweep(B):
    """measurement library only (X2I_LIB_VARIANT=ablate): the cut tile c of the last round's items, swept"""
    H, S = 24, 4608
    Spad, D = ops.pad128(S), H * 128
    rnd = lambda *sh: torch.randn(sh, device="cuda").bfloat16()  # noqa: E731
    Q, K, VT = (rnd(B, H, Spad, 128).float() * (1.4426950408889634 / math.sqrt(128))).bfloat16(), rnd(B, H, Spad, 128), rnd(B, H, 128, Spad)
    O = torch.empty((B, S, D), device="cuda", dtype=torch.bfloat16)
    for c in (0, 28, 32, 36, 40, 44, 48, 52, 56, 60, 64):
        _lib.set_option("attn_ablate", 100 + c if c else 0)
        ts = sorted(timeit(lambda: ops.attention(Q, K, VT, O, B, H, S, Spad, D, S * D, math.log(2.0), vt_perm=True)) for _ in range(3))
        print(f"B={B} cut at tile {c if c else 'auto'}: {ts[1] * 1e6:8.1f} us")
    _lib.set_option("attn_ablate", 0)


def main():
    H, S = 24, 4608
    if "--cut" in sys.argv:
        for a in [x for x in sys.argv[1:] if x != "--cut"]:
            cut_sweep(int(a))
        return
    for a in (sys.argv[1:] or ["1", "2", "4", "8"]):
        B = int(a)
        Spad, D = ops.pad128(S), H * 128
        rnd = lambda *sh: torch.randn(sh, device="cuda").bfloat16()  # noqa: E731
        Q, K, VT = (rnd(B, H, Spad, 128).float() * (1.4426950408889634 / math.sqrt(128))).bfloat16(), rnd(B, H, Spad, 128), rnd(B, H, 128, Spad)
        O = torch.empty((B, S, D), device="cuda", dtype=torch.bfloat16)
        res = {0: [], 1: []}
        for rnd_ in range(3):
            for sk in (0, 1):
                _lib.set_option("attn_streamk", sk)
                res[sk].append(timeit(lambda: ops.attention(Q, K, VT, O, B, H, S, Spad, D, S * D, math.log(2.0), vt_perm=True)))
        _lib.set_option("attn_streamk", 1)
        t0, t1 = sorted(res[0])[1], sorted(res[1])[1]
        fl = 4.0 * B * H * S * S * 128
        items = B * H * ((S + 255) // 256)
        print(f"B={B}: {items} items = {items / 256:.3f} rounds   whole items {t0 * 1e6:8.1f} us ({fl / t0 / 1e12:6.1f} TF)   stream-K {t1 * 1e6:8.1f} us ({fl / t1 / 1e12:6.1f} TF)   "
              f"x{t0 / t1:.3f}")
    ops.streamk_check(sync=True)


if __name__ == "__main__":
    main()
