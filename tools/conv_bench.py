"""Projector conv5x5: VALU form vs matrix-core (Toeplitz MFMA) form, HIP-event timing.  python tools/conv_bench.py [B]"""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from x2i_amd import ops  # noqa: E402


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
    for (C, H) in ((37, 2048), (29, 3584)):
        x = torch.randn(B, C, 512, H, device="cuda").bfloat16()
        w, bb = torch.randn(C, 25, device="cuda"), torch.randn(1, device="cuda")
        table = ops.proj_conv5x5_pack(w)
        t = timeit(lambda: ops.proj_conv5x5(x, w, bb))
        print(f"proj_conv5x5[VALU] B={B} C={C} H={H}: {t*1e6:8.1f} us  {x.numel()*2/t/1e9:8.1f} GB/s (input)")
        for v, name in ((0, "automatic"), (3, "two row blocks per wave"), (2, "register-pipelined"), (1, "plain 2-layer stages")):
            with ops.option("conv5_variant", v):
                t = timeit(lambda: ops.proj_conv5x5_packed(x, table, bb))
            print(f"proj_conv5x5[MFMA, {name}] B={B} C={C} H={H}: {t*1e6:8.1f} us  {x.numel()*2/t/1e9:8.1f} GB/s (input)")
        del x


if __name__ == "__main__":
    main()
