"""What a work item of the attention kernel costs beyond its key tiles: 1728 items (6.75 rounds of one-workgroup-per-CU blocks) of 16 / 32 / 64 / 72 key
tiles each, whole items (attn_streamk = 0); time per round = a + b * tiles.   python tools/attn_item_cost.py"""
import math
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from x2i_amd import _lib, ops  # noqa: E402
from tools.attn_sk_bench import timeit  # noqa: E402


def main():
    for form in ("whole items, one workgroup each", "persistent, whole items"):
        _lib.set_option("attn_streamk", 0 if form.startswith("whole") else 1)
        print("==", form)
        with ops.streamk_scope(None):
            one_form()
    _lib.set_option("attn_streamk", 1)


def one_form():
    pts = []
    for (B, H, S) in ((16, 27, 1024), (8, 27, 2048), (4, 27, 4096), (4, 24, 4608)):
        Spad, D = ops.pad128(S), H * 128
        rnd = lambda *sh: torch.randn(sh, device="cuda").bfloat16()  # noqa: E731
        Q, K, VT = (rnd(B, H, Spad, 128).float() * (1.4426950408889634 / math.sqrt(128))).bfloat16(), rnd(B, H, Spad, 128), rnd(B, H, 128, Spad)
        O = torch.empty((B, S, D), device="cuda", dtype=torch.bfloat16)
        t = sorted(timeit(lambda: ops.attention(Q, K, VT, O, B, H, S, Spad, D, S * D, math.log(2.0), vt_perm=True)) for _ in range(3))[1]
        items, nt = B * H * ((S + 255) // 256), (S + 63) // 64
        rounds = math.ceil(items / 256)
        pts.append((nt, t / rounds))
        print(f"B={B} H={H} S={S}: {items} items of {nt} tiles, {t * 1e6:8.1f} us = {t / rounds * 1e6:7.2f} us per round")
    (n0, t0), (n1, t1) = pts[0], pts[2]
    b = (t1 - t0) / (n1 - n0)
    a = t0 - b * n0
    print(f"per round: {a * 1e6:.2f} us + {b * 1e6:.3f} us per key tile  ->  an item's fixed cost = {a / b:.1f} key tiles")


if __name__ == "__main__":
    main()
