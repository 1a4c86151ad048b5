#!/usr/bin/env python3
"""DESIGN.md section 4's table from a committed rocprofv3 --kernel-trace --stats CSV of the bench job (B = 4, 1024^2):
    python tools/kernel_table.py profiles/<tag>_bench_b4_1024_kernel_stats.csv
One row per kernel family that takes >= 0.2 % of the timed job: launches per denoise step, mean launch time, ms per denoise step, share, and the
algorithmic rate (FLOPs of the launch shapes of flux.py at B = 4 / the mean time) as a fraction of the 2.5 PF dense bf16 peak."""
import csv
import re
import sys

PEAK = 2.5e15
M1, M2, D = 18432, 16384 + 2048, 3072          # single-stream rows; image + text rows of a grouped double-block launch
FAM = [  # (regex on the demangled name, label, file, serves, FLOP per launch or None)
    (r"gemm256p_kernel<1, false, false, false, false,", "`gemm256p_kernel<GELU>`", "gemm256p.hip", "single blocks: proj_mlp + GELU (N = 12288, K = 3072)", 2.0 * M1 * 4 * D * D),
    (r"gemm256p_kernel<1, false, false, false, true,", "`gemm256p_kernel<GELU, PAIR>`", "gemm256p.hip", "double blocks: ff.net.0 + ff_context.net.0, one grouped launch", 2.0 * M2 * 4 * D * D),
    (r"gemm256p_kernel<0, true, false, false, false,", "`gemm256p_kernel<RES>`", "gemm256p.hip", "single blocks: proj_out + gated residual (N = 3072, K = 15360)", 2.0 * M1 * D * 5 * D),
    (r"gemm256p_kernel<0, true, false, false, true,", "`gemm256p_kernel<RES, PAIR>`", "gemm256p.hip", "double blocks: to_out + to_add_out (K = 3072) and ff.net.2 pairs (K = 12288), gated residual", 0.5 * (2.0 * M2 * D * D + 2.0 * M2 * D * 4 * D)),
    (r"gemm256p_kernel<0, false, false, true, false,", "`gemm256p_kernel<QKV>`", "gemm256p.hip", "single blocks: to_q / k / v with RMSNorm + RoPE + head split + V^T in the epilogue (N = 9216)", 2.0 * M1 * 3 * D * D),
    (r"gemm256p_kernel<0, false, false, true, true,", "`gemm256p_kernel<QKV, PAIR>`", "gemm256p.hip", "double blocks: image + text QKV projections, one grouped launch", 2.0 * M2 * 3 * D * D),
    (r"attn_w16_kernel", "`attn_w16_kernel`", "attention_w16.hip + gen_attn_w16.py", "joint self-attention, 24 heads, S = 4608, d = 128", 4.0 * 4 * 24 * 4608 * 4608 * 128),
    (r"ln_rows_kernel", "`ln_rows_kernel`", "elementwise.hip", "LayerNorm (no affine) + AdaLN modulate", None),
    (r"skinny_linear_kernel", "`skinny_linear_kernel`", "elementwise.hip", "AdaLN modulation table (27 % of the parameters), embedders", None),
    (r"gemm256p_kernel<0, false, false, false, false,", "`gemm256p_kernel<bias>`", "gemm256p.hip", "x_embedder, context_embedder, proj_out of the model, projector linears", None),
    (r"gemm_bf16_kernel|gemm256w_bf16_kernel", "`gemm_bf16_kernel` / `gemm256w_bf16_kernel`", "gemm128.hip / gemm256w.hip", "small and text-stream launches the persistent form does not take", None),
    (r"euler_kernel|sinusoid|seq_mean|conv5x5|layer_mean|ln_kernel", "projector / scheduler kernels", "projector.hip, proj_conv_mfma.hip, elementwise.hip", "layer-fusion conv, pooled mean, Euler step, Timesteps", None),
]


def main(path):
    rows = list(csv.DictReader(open(path)))
    steps = next((int(r["Calls"]) for r in rows if "euler_kernel" in r["Name"]), 0)
    setup = ("distribution_elementwise", "vectorized_elementwise", "rocclr", "elementwise_kernel", "CatArray", "fill", "Fill")
    job = [r for r in rows if not any(s in r["Name"] for s in setup)]
    total = sum(float(r["TotalDurationNs"]) for r in job)
    print("| kernel | file | serves | launches / step | mean µs | ms / step | share | rate |")
    print("|---|---|---|---|---|---|---|---|")
    seen = 0.0
    for rx, label, f, what, fl in FAM:
        rs = [r for r in job if re.search(rx, r["Name"])]
        if not rs:
            continue
        t = sum(float(r["TotalDurationNs"]) for r in rs)
        n = sum(int(r["Calls"]) for r in rs)
        seen += t
        if t / total < 0.002:
            continue
        rate = "%.2f of 2.5 PF (%.0f TF)" % (fl / (t / n * 1e-9) / PEAK, fl / (t / n * 1e-9) / 1e12) if fl else "HBM-bound" if "ln_rows" in rx or "skinny" in rx else "—"
        print("| %s | %s | %s | %.1f | %.1f | %.2f | %.1f %% | %s |" % (label, f, what, n / steps, t / n * 1e-3, t * 1e-6 / steps, 100 * t / total, rate))
    print("| everything else | | | | | %.2f | %.1f %% | |" % ((total - seen) * 1e-6 / steps, 100 * (total - seen) / total))
    print()
    print("(%d denoise steps in the trace; %.1f ms of kernel time per denoise step)" % (steps, total * 1e-6 / steps))


if __name__ == "__main__":
    main(sys.argv[1])
