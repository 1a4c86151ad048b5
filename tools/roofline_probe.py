#!/usr/bin/env python3
"""Run ONLY bench.py's roofline measurement (the two single-block GEMM shapes, HIP-event timed on the launch stream) so that
`rocprofv3 --kernel-trace --stats -- python tools/roofline_probe.py` gives an average kernel duration that can be compared
directly with the `roofline.achieved` the bench prints (profiles/r01d_roofline_probe_*)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

r = bench.gemm_roofline_fp8(4) if "--fp8" in sys.argv else bench.gemm_roofline(4)
D, S, B = 3072, 4608, 4
fl = 2.0 * B * S * (4 * D * D + 5 * D * D)
r["implied_avg_kernel_us"] = fl / (r["achieved"] * 1e12) / 2 * 1e6  # mean over the two shapes
print(json.dumps(r))
