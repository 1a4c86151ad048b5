#!/usr/bin/env python3
"""Per-kernel timings at FLUX 1024^2 shapes (run on the GPU box).  Prints one line per kernel with achieved
TFLOP/s or GB/s -- the numbers quoted in DESIGN.md's kernel table come from here + rocprofv3."""
import math
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ABLATE = "--ablate" in sys.argv  # also time the attention ablation variants (needs the measurement-only library)
if ABLATE:
    os.environ["X2I_LIB_VARIANT"] = "ablate"
from x2i_amd import _lib, ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def rnd(*shape, scale=1.0, dtype=torch.bfloat16):
    return (torch.randn(*shape, device=DEV, dtype=torch.float32) * scale).to(dtype)


def main():
    B = int(os.environ.get("X2I_B", "4"))
    D, H = 3072, 24
    Si, St = 4096, 512
    S = Si + St
    print(f"device: {torch.cuda.get_device_name(0)}  B={B}")
    for (M, N, K, name) in [(B * Si, 3 * D, D, "qkv_img"), (B * Si, D, D, "attn_out"), (B * Si, 4 * D, D, "ff_in(gelu)"),
                            (B * Si, D, 4 * D, "ff_out"), (B * S, 7 * D, D, "single_in"), (B * S, D, 5 * D, "single_out"),
                            (B * St, 3 * D, D, "qkv_txt")]:
        A, W, b = rnd(M, K), rnd(N, K, scale=0.02), rnd(N)
        out = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
        act = 1 if "gelu" in name else 0
        for tile in ("128", "256"):
            _lib.set_option("gemm_tile", int(tile))
            t = timeit(lambda: ops.gemm(A, W, b, out=out, act=act))
            print(f"gemm[{tile}] {name:12s} M={M:6d} N={N:6d} K={K:6d}: {t*1e3:8.3f} ms  {2*M*N*K/t/1e12:8.1f} TFLOP/s")
        _lib.set_option("gemm_tile", 0)
        del A, W, out
    # attention
    Spad = ops.pad128(S)
    Q, K_ = rnd(B, H, Spad, 128), rnd(B, H, Spad, 128)
    VT = rnd(B, H, 128, Spad)
    O = torch.empty((B, S, D), device=DEV, dtype=torch.bfloat16)
    for var, nm in (("4", "4-wave thr8"), ("1", "nw8 lockstep thr8"), ("2", "nw4 thr0"), ("5", "ping-pong thr8"), ("7", "ping-pong, DMA in vector phase"), ("5", "ping-pong thr8"), ("7", "ping-pong, DMA in vector phase"), ("4", "4-wave thr8")):
        _lib.set_option("attn_variant", int(var))
        t = timeit(lambda: ops.attention(Q, K_, VT, O, B, H, S, Spad, D, S * D, 1 / math.sqrt(128)))
        print(f"attention[{nm}] B={B} H={H} S={S}: {t*1e3:8.3f} ms  {4*B*H*S*S*128/t/1e12:8.1f} TFLOP/s")
    _lib.set_option("attn_variant", 0)
    for abl, nm in (("0", "full"), ("1", "no softmax"), ("2", "no barrier/wait"), ("4", "no DMA"), ("7", "MFMA+LDS reads only"), ("0", "full"),
                    ("32", "branchy loop (no peel)"), ("0", "full"), ("32", "branchy loop (no peel)"), ("0", "full")):
        if not ABLATE:
            break
        _lib.set_option("attn_ablate", int(abl))
        t = timeit(lambda: ops.attention(Q, K_, VT, O, B, H, S, Spad, D, S * D, 1 / math.sqrt(128)))
        print(f"attention-ablate[{nm}]: {t*1e3:8.3f} ms  {4*B*H*S*S*128/t/1e12:8.1f} TFLOP/s")
    if ABLATE:
        _lib.set_option("attn_ablate", 0)
    # qkv split
    qkv = rnd(B * S, 3 * D)
    nw = rnd(128)
    cos, sin = torch.randn(S, 128, device=DEV), torch.randn(S, 128, device=DEV)
    t = timeit(lambda: ops.qkv_split(None, qkv, 3 * D, 3 * D, B, S, 0, H, None, None, nw, nw, cos, sin, Q, K_, VT, Spad))
    print(f"qkv_split: {t*1e3:8.3f} ms  {2*B*S*3*D*2/t/1e9:8.1f} GB/s (read+write)")
    # ln modulate
    X = rnd(B, S, D)
    Y = torch.empty_like(X)
    mod = torch.randn(B, 2 * D, device=DEV)
    t = timeit(lambda: ops.ln_modulate(X, Y, B, S, D, 0, None, None, mod, mod[:, D:], 2 * D))
    print(f"ln_modulate: {t*1e3:8.3f} ms  {2*B*S*D*2/t/1e9:8.1f} GB/s (read+write)")
    # modulation table GEMV
    Ntot = 19 * 12 * D + 38 * 3 * D + 2 * D
    Wm, bm = rnd(Ntot, D, scale=0.02), rnd(Ntot)
    temb = torch.randn(B, D, device=DEV)
    out = torch.empty((B, Ntot), device=DEV)
    t = timeit(lambda: ops.skinny_linear(temb, Wm, bm, out=out, act_in=3))
    print(f"modulation skinny N={Ntot}: {t*1e3:8.3f} ms  {Ntot*D*2/t/1e9:8.1f} GB/s (weights)")
    del Wm
    # projector conv
    for (C, Hh) in ((37, 2048), (29, 3584)):
        x = rnd(B, C, 512, Hh)
        w, bb = torch.randn(C, 25, device=DEV), torch.randn(1, device=DEV)
        t = timeit(lambda: ops.proj_conv5x5(x, w, bb))
        print(f"proj_conv5x5[VALU] C={C} H={Hh}: {t*1e3:8.3f} ms  {x.numel()*2/t/1e9:8.1f} GB/s (input)")
        table = ops.proj_conv5x5_pack(w)
        t = timeit(lambda: ops.proj_conv5x5_packed(x, table, bb))
        print(f"proj_conv5x5[MFMA] C={C} H={Hh}: {t*1e3:8.3f} ms  {x.numel()*2/t/1e9:8.1f} GB/s (input)")
        del x


if __name__ == "__main__":
    main()
