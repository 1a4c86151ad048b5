#!/usr/bin/env python3
"""The sampling path's attention kernel (attn_w16_kernel, or attn_w4_kernel with X2I_ATTN_W16=0) at the single-block geometry (B = 4, 24 heads, S = 4608), launched `alone` (back to back) or `seq` (every launch between the
two roofline GEMM launches, the step's order) -- the two estimators of bench.py's roofline_attention, one per process so that a counter pass
(tools/attn_clock_pmc.sh: GRBM_GUI_ACTIVE / duration = the clock the kernel ran at) sees one of them at a time.
    python tools/attn_in_sequence.py alone|seq [launches]"""
import math
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from x2i_amd import ops  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "seq"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
B, H, S, D = 4, 24, 4608, 3072
Spad = ops.pad128(S)
Q = (torch.randn(B, H, Spad, 128, device="cuda") * (math.log2(math.e) / math.sqrt(128))).bfloat16()
K_, VT = torch.randn(B, H, Spad, 128, device="cuda").bfloat16(), torch.randn(B, H, 128, Spad, device="cuda").bfloat16()
CAT = torch.empty((B, S, 5 * D), device="cuda", dtype=torch.bfloat16)
A0 = torch.randn(B * S, D, device="cuda").bfloat16()
W0 = (torch.randn(4 * D, D, device="cuda") * 0.02).bfloat16()
W1 = (torch.randn(D, 5 * D, device="cuda") * 0.02).bfloat16()
X = torch.empty(B * S, D, device="cuda", dtype=torch.bfloat16)
VP = ops.attention_prefers_vt_perm(H, S, math.log(2.0))   # (random V^T: the key order does not change the work)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
for i in range(-3, n):
    if mode == "seq":
        ops.gemm(A0, W0, None, out=CAT, act=1, ldc=5 * D, c_offset=D)
    if i >= 0:
        ev[i][0].record()
    ops.attention(Q, K_, VT, CAT, B, H, S, Spad, 5 * D, S * 5 * D, math.log(2.0), vt_perm=VP)
    if i >= 0:
        ev[i][1].record()
    if mode == "seq":
        ops.gemm(CAT, W1, None, out=X)
torch.cuda.synchronize()
t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
print(f"attention {mode}: median {t[len(t) // 2]:.1f} us (min {t[0]:.1f}, max {t[-1]:.1f}) over {n} launches; {4.0 * B * H * S * S * 128 / t[len(t) // 2] / 1e6:.1f} TFLOP/s")
