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
    names = {4: "4-wave", 5: "ping-pong", 7: "ping-pong, DMA in vector phase", 8: "ping-pong, 4-deep rings + fragment prefetch"}
    for var in (5, 8, 7, 4, 5, 8):
        _lib.set_option("attn_variant", var)
        t = timeit(lambda: ops.attention(Q, K_, VT, O, B, H, S, Spad, D, S * D, 1 / math.sqrt(128)))
        print(f"attention[{names[var]}] B={B}: {t*1e3:8.3f} ms  {4*B*H*S*S*128/t/1e12:8.1f} TFLOP/s")
    _lib.set_option("attn_variant", 0)


if __name__ == "__main__":
    main()
