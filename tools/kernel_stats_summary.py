#!/usr/bin/env python3
"""Per-denoise-step view of a rocprofv3 kernel_stats.csv of bench.py (steps = calls of euler_kernel): tools/kernel_stats_summary.py <csv> [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = max([int(r["Calls"]) for r in rows if "euler_kernel" in r["Name"]] or [1])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# {sys.argv[1]}: {steps} denoise steps, {tot / 1e6 / steps:.3f} ms of kernel time per step (incl. one-time set-up kernels)")
for r in rows[:top]:
    nm = r["Name"].replace("(anonymous namespace)::", "").replace("x2i_gemm::", "").replace("void ", "")
    print(f'{nm[:100]:100s} {int(r["Calls"]) / steps:6.1f}/step {float(r["TotalDurationNs"]) / 1e6 / steps:7.3f} ms/step  avg {float(r["AverageNs"]) / 1e3:8.1f} us {float(r["TotalDurationNs"]) / tot * 100:5.1f}%')
