#!/usr/bin/env python3
"""bench.py -- images/s and ms/denoise-step of the X2I sampling hot path on MI355X.

One "step" (driver contract) = one pass of the hot path over one batch of synthetic input:
    alignment projector (MLLM hidden states -> prompt_embeds, pooled) + N-step FLUX denoising loop + unpack,
i.e. everything between the MLLM and the VAE for a batch of images.  Workload at N=1: BASELINE.json configs[1]
(QwenVL2.5-3B conditioning: C=37, H=2048, S_txt=512; shuttle-3/FLUX-schnell architecture, 1024x1024, 4 steps,
batch 4), random-init weights, synthetic inputs already resident in HBM.  N>1: weak scaling, the batch axis is
sharded (batch 4 per rank), one all-gather of the final packed latents over RCCL.

`python bench.py --gpus N` launches the N ranks ITSELF when it is not already running under a launcher (WORLD_SIZE unset): it
re-executes this file under `python -m torch.distributed.run --standalone --nnodes=1 --nproc-per-node N` on 127.0.0.1; under a
launcher (torchrun / the driver's torch.distributed.run command) it asserts WORLD_SIZE == --gpus.  The printed n_gpus is
dist.get_world_size(), never the flag.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel = bf16 MFMA GEMM, measured live with HIP
events on the launch stream) and "cpu_baseline" (the CPU oracle timed on the host cores on a bounded sample).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16 = 2.5e15  # dense MFMA bf16, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_FP8 = 5.0e15   # dense MFMA fp8 (MX-scaled K=128), same guide
PEAK_HBM = 8.0e12


def flops_per_denoise_step(B, St, Si, D=3072, L=19, Ls=38, Kj=4096, Cin=64, pooled=768):
    """SURVEY.md Appendix C (2*M*N*K per GEMM, 4*S^2*D per attention)."""
    S = St + Si
    emb = 2 * B * Si * Cin * D + 2 * B * St * Kj * D + 2 * B * (256 * D + D * D + pooled * D + D * D)
    dbl = L * (B * S * 2 * (3 * D * D + D * D + 8 * D * D) + 2 * 2 * B * D * 6 * D + 4 * B * S * S * D)
    sgl = Ls * (B * S * 2 * (3 * D * D + 4 * D * D + 5 * D * D) + 2 * B * D * 3 * D + 4 * B * S * S * D)
    out = 2 * B * D * 2 * D + 2 * B * Si * D * Cin
    return emb + dbl + sgl + out


class ClockPowerSampler:
    """sclk / socket power of the bench GPU sampled by a host thread (amdsmi, ~100 Hz) while a probe runs, so that a roofline figure
    carries the power state it was measured in (the part is power-capped: DESIGN.md section 4)."""

    def __init__(self, index=0, period=0.01):
        self.samples, self.period, self._stop, self._thr, self._h = [], period, False, None, None
        try:
            import amdsmi
            self._smi = amdsmi
            try:
                amdsmi.amdsmi_init()
            except Exception:  # noqa: BLE001  (already initialised)
                pass
            hs = amdsmi.amdsmi_get_processor_handles()
            self._h = hs[index] if index < len(hs) else None
        except Exception:  # noqa: BLE001
            self._h = None

    def _read(self):
        smi, h = self._smi, self._h
        clk = pw = None
        try:
            clk = float(smi.amdsmi_get_clock_info(h, smi.AmdSmiClkType.GFX)["clk"])
        except Exception:  # noqa: BLE001
            pass
        try:
            pi = smi.amdsmi_get_power_info(h)
            for k in ("current_socket_power", "average_socket_power", "socket_power"):
                if k in pi and isinstance(pi[k], (int, float)) and pi[k] > 0:
                    pw = float(pi[k])
                    break
        except Exception:  # noqa: BLE001
            pass
        return clk, pw

    def __enter__(self):
        if self._h is not None:
            import threading

            def loop():
                while not self._stop:
                    self.samples.append(self._read())
                    time.sleep(self.period)
            self._thr = threading.Thread(target=loop, daemon=True)
            self._thr.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._thr is not None:
            self._thr.join(timeout=2)

    def summary(self):
        def stats(v):
            v = sorted(x for x in v if x is not None)
            return None if not v else dict(min=v[0], median=v[len(v) // 2], max=v[-1])
        return dict(n=len(self.samples), sclk_mhz=stats([c for c, _ in self.samples]), socket_power_w=stats([p for _, p in self.samples]),
                    source="amdsmi, host thread, %.0f ms period" % (self.period * 1e3) if self._h is not None else "unavailable")


def _interleaved_probe(fns, rounds, per_round):
    """fns: the launches of the roofline pair.  `rounds` rounds; in each, every shape is launched `per_round` times back to back between
    its own pair of HIP events on the launch stream, the shapes alternating (so a slow clock phase hits both, not one).  Returns
    per-shape lists of per-launch seconds (one entry per round) plus the clock / power record of the whole probe."""
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(rounds)] for _ in fns]
    for fn in fns:
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    with ClockPowerSampler(torch.cuda.current_device()) as smp:
        for r in range(rounds):
            for i, fn in enumerate(fns):
                ev[i][r][0].record()
                for _ in range(per_round):
                    fn()
                ev[i][r][1].record()
        torch.cuda.synchronize()
    times = [[e0.elapsed_time(e1) * 1e-3 / per_round for (e0, e1) in row] for row in ev]
    return times, smp.summary()


def _in_step_gemm_rate(B):
    """GEMM FLOPs of the timed job / summed GEMM kernel time, from the COMMITTED rocprofv3 --kernel-trace --stats of this bench
    (profiles/<tag>_bench_b4_1024_kernel_stats.csv: TotalDurationNs over every gemm* kernel; the number of denoise steps in the trace
    = the calls of euler_kernel -- 1 eager graph warm-up pass + 1 warm-up + 2 timed replays = 16): the rate of the dominant kernel
    INSIDE the step, which cannot drift with the state of a probe.  B = 4 only."""
    import csv
    import glob
    if B != 4:
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[4-9]*_bench_b4_1024_kernel_stats.csv")))
    if not files:
        return None
    path = files[-1]
    t_ns, calls, steps = 0.0, 0, 0
    with open(path) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Name", "")
            if "gemm" in name and "fp8" not in name and "skinny" not in name:
                t_ns += float(row["TotalDurationNs"])
                calls += int(row["Calls"])
            if "euler_kernel" in name:
                steps = int(row["Calls"])   # one scheduler step per denoise step: counts the eager graph warm-up pass too
    D, St, Si = 3072, 512, 4096
    S = St + Si
    gemm_fl = (2 * B * Si * 64 * D + 2 * B * St * 4096 * D + 19 * B * S * 2 * 12 * D * D + 38 * B * S * 2 * 12 * D * D + 2 * B * Si * D * 64)
    gemm_fl += 2 * B * St * (2048 * 4096 + 4096 * 4096 + 4096 * 768) / 4     # the projector's three linears, once per 4-step pass
    if t_ns <= 0 or steps <= 0:
        return None
    rate = gemm_fl * steps / (t_ns * 1e-9)
    return dict(achieved=rate / 1e12, frac=rate / PEAK_BF16, unit="TFLOP/s", gemm_ms_per_denoise_step=t_ns * 1e-6 / steps, gemm_launches=calls,
                measured="committed profile (builder's box, not this run)", source=os.path.relpath(path, ROOT), note="all MFMA GEMM launches of %d denoise steps (2*M*N*K of every nn.Linear of the DiT, SURVEY.md "
                "Appendix C) / their summed rocprofv3 kernel time" % steps)


_PIPE_ALONE = None


def matrix_pipe_alone_live(variants=((1, "bf16 16x16x32, operands rotating"), (3, "bf16 16x16x32, one operand held x8"), (2, "e4m3 16x16x128")),
                           launches=6, iters=300000):
    """What the power management lets the matrix pipe ALONE do on THIS box, measured live: the register-resident MFMA
    streams of tools/ubench/mfma_power.hip -- one wave per SIMD on every CU, random operands and accumulators in registers, no memory
    traffic in the loop -- launched from the measurement library tools/ubench/libx2i_ubench.so (never the product .so), `launches` launches
    of ~0.13 s per variant (sustained: the first launch of a variant is dropped), HIP events on the launch stream, sclk / socket power
    sampled beside them.  Returns None when the library is absent (the bench line then carries the committed-profile figure)."""
    import ctypes
    global _PIPE_ALONE
    if _PIPE_ALONE is not None:
        return _PIPE_ALONE
    path = os.path.join(ROOT, "tools", "ubench", "libx2i_ubench.so")
    if not os.path.exists(path):
        return None
    lib = ctypes.CDLL(path)
    lib.x2i_ubench_data_bytes.restype = ctypes.c_longlong
    lib.x2i_ubench_flop_per_iter.restype = ctypes.c_double
    lib.x2i_ubench_flop_per_iter.argtypes = [ctypes.c_int]
    lib.x2i_ubench_mfma_pipe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    n = int(lib.x2i_ubench_data_bytes())
    data = (torch.rand(n // 2, device="cuda") * 4 - 2).bfloat16()     # random bf16 in [-2, 2): the stand-alone tool's operands
    stream = torch.cuda.current_stream().cuda_stream
    out = {}
    for var, name in variants:
        fl = float(lib.x2i_ubench_flop_per_iter(var))
        it = iters // 2 if var == 2 else iters
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
        with ClockPowerSampler(torch.cuda.current_device()) as smp:
            for e0, e1 in ev:
                e0.record()
                rc = lib.x2i_ubench_mfma_pipe(var, it, ctypes.c_void_p(data.data_ptr()), ctypes.c_void_p(stream))
                e1.record()
                if rc:
                    return None
            torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) * 1e-3 for e0, e1 in ev[1:])
        t = ts[len(ts) // 2]
        out[var] = dict(stream=name, achieved=fl * it / t / 1e12, unit="TFLOP/s", ms_per_launch=t * 1e3, clock_power=smp.summary())
    del data
    bf = out[1]
    res = dict(achieved=bf["achieved"], frac_of_peak=bf["achieved"] * 1e12 / PEAK_BF16, unit="TFLOP/s", measured="live (this run; tools/ubench/libx2i_ubench.so, HIP events)",
               clock_power=bf["clock_power"], operand_held=out.get(3), e4m3=out.get(2),
               note="v_mfma_f32_16x16x32_bf16 on random operands held in registers, one wave per SIMD on every CU, no memory traffic, %d launches of "
                    "%.0f ms (median of all but the first): what the power management lets the matrix pipe alone do on this box" % (launches, bf["ms_per_launch"]))
    if out.get(2):
        res["e4m3"]["frac_of_peak"] = out[2]["achieved"] * 1e12 / PEAK_FP8
    _PIPE_ALONE = res
    return res


def gemm_roofline(B, rounds=6, per_round=8):
    """Dominant kernel: the bf16 MFMA GEMM.  Times the two largest launch shapes of a single-stream block -- proj_mlp + bias + GELU
    (M=B*4608, N=12288, K=3072) and proj_out + bias (N=3072, K=15360; in the model this launch also adds the gated residual) -- with
    HIP events on the launch stream; achieved = algorithmic FLOP / time.  (Same two launches since round 1, for comparability.)
    Round 4: the two shapes are INTERLEAVED over `rounds` rounds x `per_round` launches; frac = the MEDIAN round, `frac_best` the fastest,
    and the sclk / socket power sampled during the probe ride along, so that a box-to-box difference is explained by the record
    itself (driver 0.51 vs builder 0.58 with equal whole-step rates)."""
    from x2i_amd import ops
    D, S = 3072, 4608
    B = min(B, 8)   # (the probe's flattened A operand must stay below the kernels' 2 GB operand limit: 8 x 4608 rows x 15360 x 2 B = 1.1 GB)
    shapes = ((B * S, 4 * D, D, 1), (B * S, D, 5 * D, 0))
    bufs, fns = [], []
    for (M, N, K, act) in shapes:
        A = torch.randn(M, K, device="cuda").bfloat16()
        W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
        bias = torch.randn(N, device="cuda").bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        bufs.append((A, W, bias, out))
        fns.append(lambda A=A, W=W, bias=bias, out=out, act=act: ops.gemm(A, W, bias, out=out, act=act))
    times, power = _interleaved_probe(fns, rounds, per_round)
    del bufs, fns
    fl = [2.0 * M * N * K for (M, N, K, _) in shapes]
    pair = sorted(times[0][r] + times[1][r] for r in range(rounds))      # per-round time of the pair
    t_med, t_min, t_max = pair[len(pair) // 2], pair[0], pair[-1]
    flt = sum(fl)
    # HBM-side traffic of the same two launches from the committed rocprofv3 PMC passes over tools/roofline_probe.py (tools/pmc_roofline.sh:
    # FETCH_SIZE doubled per the gfx950 correction, + WRITE_SIZE, separate passes; per-launch means summed over every GEMM kernel the
    # pair launches; since round 3 that is ONE persistent 256^2 kernel per GEMM, its last round cut along K).  Only valid for the profiled batch (B=4 -> M=18432).
    traffic, src = None, None
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[4-9]*_pmc_roofline.json"))) or [os.path.join(ROOT, "profiles", "r03_pmc_roofline.json")]
    pj = cands[-1]
    if B == 4 and os.path.exists(pj):
        try:
            d = json.load(open(pj))
            traffic = float(d["traffic_bytes_per_pair"])
            src = "%s (%s)" % (os.path.relpath(pj, ROOT), d.get("note", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE"))
        except (KeyError, ValueError):
            traffic = None
    alg_bytes = sum(2.0 * (M * K + N * K + M * N) for (M, N, K, _) in shapes)
    pipe = matrix_pipe_alone_live() or dict(achieved=1941.0, frac_of_peak=0.776, unit="TFLOP/s", measured="committed profile (builder's box, not this run)",
                                            source="profiles/r05o_mfma_power_sustained_fixed_ubench.log",
                                            note="v_mfma_f32_16x16x32_bf16 on random operands held in registers, no memory traffic, sustained "
                                                 "(tools/ubench/libx2i_ubench.so was not built: committed figure)")
    return dict(bound="mfma", achieved=flt / t_med / 1e12, peak=PEAK_BF16 / 1e12, unit="TFLOP/s", frac=flt / t_med / PEAK_BF16,
                frac_best=flt / t_min / PEAK_BF16, frac_worst=flt / t_max / PEAK_BF16,
                probe=dict(rounds=rounds, launches_per_round=per_round, order="interleaved", statistic="median round (frac), fastest (frac_best)",
                           us_per_launch=[[round(t * 1e6, 1) for t in row] for row in times], clock_power=power),
                measured="live (HIP events on the launch stream, this run)",
                matrix_pipe_alone=pipe, frac_of_pipe_alone=(flt / t_med / 1e12) / pipe["achieved"],
                in_step=_in_step_gemm_rate(B),
                traffic=traffic, traffic_measured="committed PMC profile (builder's box, not this run)" if traffic is not None else None,
                traffic_source=src, algorithmic_bytes=alg_bytes, kernel="gemm256p_kernel (bf16 instantiations)",
                shapes="M=%d: N=12288,K=3072 (+GELU) + N=3072,K=15360 (single-block proj_mlp / proj_out shapes, 1.39 + 1.74 TFLOP; one persistent "
                       "256^2 launch each, last round cut along K)" % (B * S))


def attention_roofline(B, rounds=6, per_round=8):
    """Second kernel of the step (21 % of its time): the attention kernel the step uses -- `attn_w16_kernel` on a span-permuted V^T where
    x2i_attention_prefers_vt_perm says so (flux.py asks the same question), else `attn_w4_kernel` -- at the single-block geometry (24 heads,
    S = 4608, output into the [S, 5D] concatenation buffer, exp2-domain scale as flux.py passes it), with the other kernel's numbers
    beside it (`other`).  Two live estimators, both HIP events on the launch stream:
    `alone` = back-to-back launches (what tools/attn_bench.py reports), and `in_sequence` = every attention launch timed on its own
    between the two roofline GEMM launches, i.e. in the power / clock state the step leaves it in.  frac = the in-sequence median."""
    from x2i_amd import ops
    H, S, D = 24, 4608, 3072
    B = min(B, 8)
    Spad = ops.pad128(S)
    Q = (torch.randn(B, H, Spad, 128, device="cuda") * (math.log2(math.e) / math.sqrt(128))).bfloat16()
    K_, VT = torch.randn(B, H, Spad, 128, device="cuda").bfloat16(), torch.randn(B, H, 128, Spad, device="cuda").bfloat16()
    CAT = torch.empty((B, S, 5 * D), device="cuda", dtype=torch.bfloat16)
    A0 = torch.randn(B * S, D, device="cuda").bfloat16()
    W0 = (torch.randn(4 * D, D, device="cuda") * 0.02).bfloat16()
    W1 = (torch.randn(D, 5 * D, device="cuda") * 0.02).bfloat16()
    X = torch.empty(B * S, D, device="cuda", dtype=torch.bfloat16)

    vp = ops.attention_prefers_vt_perm(H, S, math.log(2.0))   # (random V^T: the key order does not change the work)

    def attn():
        ops.attention(Q, K_, VT, CAT, B, H, S, Spad, 5 * D, S * 5 * D, math.log(2.0), vt_perm=vp)

    def attn_other():
        ops.attention(Q, K_, VT, CAT, B, H, S, Spad, 5 * D, S * 5 * D, math.log(2.0), vt_perm=not vp)

    def g0():
        ops.gemm(A0, W0, None, out=CAT, act=1, ldc=5 * D, c_offset=D)     # proj_mlp + GELU into the concatenation buffer, as the block does

    def g1():
        ops.gemm(CAT, W1, None, out=X)                                   # proj_out over [attention | MLP]
    fl = 4.0 * B * H * S * S * 128
    names = {True: "attn_w16_kernel", False: "attn_w4_kernel"}

    def in_sequence(fn):
        for _ in range(2):
            g0(); fn(); g1()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(rounds * per_round)]
        with ClockPowerSampler(torch.cuda.current_device()) as smp:
            for e0, e1 in ev:
                g0()
                e0.record()
                fn()
                e1.record()
                g1()
            torch.cuda.synchronize()
        return sorted(e0.elapsed_time(e1) * 1e-3 for e0, e1 in ev), smp
    times, power = _interleaved_probe([attn], rounds, per_round)
    alone = sorted(times[0])
    seq, smp = in_sequence(attn)
    t_seq, t_alone = seq[len(seq) // 2], alone[len(alone) // 2]
    to, po = _interleaved_probe([attn_other], rounds, per_round)   # the A/B partner, same box, same minute
    so, smo = in_sequence(attn_other)
    to = sorted(to[0])
    other = dict(kernel=names[not vp], us_alone=to[len(to) // 2] * 1e6, us_in_sequence=so[len(so) // 2] * 1e6, clock_power_alone=po,
                 clock_power_in_sequence=smo.summary())
    return dict(bound="mfma", kernel=names[vp], other=other, achieved=fl / t_seq / 1e12, peak=PEAK_BF16 / 1e12, unit="TFLOP/s", frac=fl / t_seq / PEAK_BF16,
                frac_alone=fl / t_alone / PEAK_BF16, us_in_sequence=t_seq * 1e6, us_alone=t_alone * 1e6, us_in_sequence_min_max=[seq[0] * 1e6, seq[-1] * 1e6],
                measured="live (HIP events on the launch stream, this run)", flop_per_launch=fl,
                clock_power_alone=power, clock_power_in_sequence=smp.summary(),
                note="in_sequence: each launch between the two roofline GEMM launches (the step's order); alone: back-to-back attention launches. "
                     "The difference is the clock the power-capped part runs the kernel at behind a GEMM (DESIGN.md section 4, round 5)",
                shapes="B=%d, 24 heads, S=4608 (512 text + 4096 image tokens), head dim 128; 4*B*H*S^2*128 FLOP" % B)


def vae_decode_line(B, images_s_denoise, passes=3):
    """The step right after the sampling path (row N1), BESIDE the headline and never inside it: ms per 1024^2 image of the HIP VAE decoder
    (random-init FLUX VAE), its algorithmic FLOPs and fraction of the bf16 peak, and the images/s the job would have with the decode
    included.  HIP-event timed."""
    from x2i_amd.vae import AutoencoderKL, decode_flops
    vae = AutoencoderKL(device="cuda").init_random_(0)
    z = torch.randn(B, 16, 128, 128, device="cuda").bfloat16()
    for _ in range(2):
        vae.decode(z, return_dict=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(passes):
        vae.decode(z, return_dict=False)
    e1.record()
    torch.cuda.synchronize()
    ms_img = e0.elapsed_time(e1) / passes / B
    fl = decode_flops(vae.config, 128, 128)
    fl_exec = decode_flops(vae.config, 128, 128, up_phases=vae.up_phases)
    form = {0: "one 3x3 conv, x2 upsampling in the gather", 1: "two 3x2 column phases", 2: "four 2x2 phases"}[vae.up_phases]
    moments = vae.epilogue_moments
    del vae
    return dict(ms_per_image=ms_img, flop_per_image=fl, flop_executed_per_image=fl_exec, achieved=fl / (ms_img * 1e-3) / 1e12, unit="TFLOP/s",
                frac=fl / (ms_img * 1e-3) / PEAK_BF16, frac_executed=fl_exec / (ms_img * 1e-3) / PEAK_BF16,
                images_s_incl_decode=1.0 / (1.0 / images_s_denoise + ms_img * 1e-3), measured="live (HIP events, this run)", batch=B,
                upsample_conv_form=form, groupnorm_statistics="conv epilogue moments" if moments else "statistics pass per GroupNorm",
                note="FLUX VAE decoder at 1024^2, random-init weights; outside the timed region of `value` (BASELINE metric = projector + denoise loop). "
                     "flop_per_image / frac: the reference formulation's FLOPs (a 3x3 conv on the doubled image per Upsample2D); "
                     "flop_executed_per_image: what the phase form of those convs executes")


def gemm_roofline_fp8(B, rounds=6, per_round=8):
    """--dtype fp8: the same two launches on the e4m3 kernel (MX-scaled K=128 MFMA), priced against the 5 PF dense fp8 peak.
    proj_mlp + GELU writes e4m3, proj_out carries the gated residual, exactly as the fp8 model issues them.  Same interleaved
    median-of-rounds estimator as gemm_roofline()."""
    from x2i_amd import ops
    D, S = 3072, 4608
    B = min(B, 8)   # (the probe's flattened A operand must stay below the kernels' 2 GB operand limit)
    shapes = ((B * S, 4 * D, D, True), (B * S, D, 5 * D, False))
    fns, keep = [], []
    for (M, N, K, gelu) in shapes:
        A8, sa = ops.quantize_rows_fp8(torch.randn(M, K, device="cuda").bfloat16())
        W8, sw = ops.quantize_rows_fp8((torch.randn(N, K, device="cuda") * 0.02).bfloat16())
        bias = torch.randn(N, device="cuda").bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=ops.FP8 if gelu else torch.bfloat16)
        gate = torch.randn(1, N, device="cuda")
        keep.append((A8, sa, W8, sw, bias, out, gate))
        if gelu:
            fns.append(lambda A8=A8, W8=W8, bias=bias, out=out, sa=sa, sw=sw: ops.gemm_fp8(A8, W8, bias, out=out, a_scale=sa, w_scale=sw, act=1, out_fp8=True))
        else:
            fns.append(lambda A8=A8, W8=W8, bias=bias, out=out, sw=sw, gate=gate: ops.gemm_fp8(A8, W8, bias, out=out, w_scale=sw, res=out, gate=gate))
    times, power = _interleaved_probe(fns, rounds, per_round)
    kernel = "gemm256_fp8_kernel"
    try:
        from x2i_amd import _lib
        if _lib.get_option("last_gemm_tile") in (8256, 9256):
            kernel = "gemm256p_fp8_kernel"
    except Exception:  # noqa: BLE001
        pass
    del fns, keep
    fl = sum(2.0 * M * N * K for (M, N, K, _) in shapes)
    pair = sorted(times[0][r] + times[1][r] for r in range(rounds))
    t_med, t_min = pair[len(pair) // 2], pair[0]
    alg_bytes = (B * S * D + 4 * D * D + B * S * 4 * D) + (B * S * 5 * D + 5 * D * D + 2 * 2 * B * S * D)
    pipe = matrix_pipe_alone_live()
    p8 = pipe.get("e4m3") if pipe else None
    return dict(bound="mfma", achieved=fl / t_med / 1e12, peak=PEAK_FP8 / 1e12, unit="TFLOP/s", frac=fl / t_med / PEAK_FP8,
                matrix_pipe_alone=p8, frac_of_pipe_alone=(fl / t_med / 1e12) / p8["achieved"] if p8 else None,
                frac_best=fl / t_min / PEAK_FP8, measured="live (HIP events on the launch stream, this run)", traffic=None, algorithmic_bytes=float(alg_bytes), kernel=kernel,
                probe=dict(rounds=rounds, launches_per_round=per_round, order="interleaved", clock_power=power,
                           us_per_launch=[[round(t * 1e6, 1) for t in row] for row in times]),
                shapes="M=%d: N=12288,K=3072 (+GELU, e4m3 out) + N=3072,K=15360 (gated residual), e4m3 operands" % (B * S))


def _cpu_threads(phys):
    """Thread count of the CPU baseline: physical_cores / 2, pinned (a per-box sweep made the figure box-dependent; torch's
    CPU GEMM peaks near half the physical cores on the 2-socket hosts of this pool), capped by the cores this process may use."""
    n_max = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n = max(1, min(n_max, (phys // 2) if phys else max(1, n_max // 4)))
    torch.set_num_threads(n)
    return n, n_max


def _physical_cores():
    try:
        import subprocess
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {l.split(":")[0].strip(): l.split(":", 1)[1].strip() for l in out.splitlines() if ":" in l}
        return int(kv["Core(s) per socket"]) * int(kv["Socket(s)"]), kv.get("Model name", "?")
    except Exception:  # noqa: BLE001
        return None, "?"


def cpu_baseline(reps=3, budget_s=100.0):
    """The CPU oracle (restated reference path, fp32, torch eager) on the host cores.  MEASURED: one whole denoise step at 512 x 512,
    batch 1 -- 19 double-stream + 38 single-stream FLUX blocks at full width on 512 text + 1024 image tokens (21.5 TFLOP; SURVEY.md
    section 8(d)); the 57 blocks share one double-block and one single-block weight set (same arithmetic and bytes per block; generating
    11.9 B random fp32 parameters on the host would take longer than the measurement).  Timed `reps` times (fewer when the time budget
    is spent), value = the median.  EXTRAPOLATED and labelled so: the 1024 x 1024
    step the GPU line is quoted on (74.4 TFLOP: 1 + 1 blocks timed on 512 + 4096 tokens, x19 / x38)."""
    from oracle import flux as OF
    from oracle import primitives as P
    from oracle import sampler as OS
    phys, cpu_name = _physical_cores()
    threads, logical = _cpu_threads(phys)
    cfg = dict(OF.DEFAULT_CFG)
    cfg.update(num_layers=1, num_single_layers=1)
    sd = OF.random_flux_state_dict(cfg, seed=0)
    St = 512

    def blocks(h2, n_double, n_single):
        ids = torch.cat([torch.zeros(St, 3), OS.prepare_latent_image_ids(h2, h2)], 0)
        rot = P.flux_pos_embed(ids)
        hid, enc, temb = torch.randn(1, h2 * h2, 3072), torch.randn(1, St, 3072), torch.randn(1, 3072)
        with torch.no_grad():
            t0 = time.time()
            for _ in range(n_double):
                enc, hid = OF.double_block(sd, "transformer_blocks.0", hid, enc, temb, rot, 24)
            t_d = time.time() - t0
            j = torch.cat([enc, hid], 1)
            t0 = time.time()
            for _ in range(n_single):
                j = OF.single_block(sd, "single_transformer_blocks.0", j, temb, rot, 24)
            t_s = time.time() - t0
        return t_d, t_s

    steps, t_start = [], time.time()
    for r in range(reps):                    # measured: whole 512^2 steps
        d512, s512 = blocks(32, 19, 38)
        steps.append(d512 + s512)
        if time.time() - t_start + steps[-1] > budget_s:
            break
    st = sorted(steps)
    step512 = st[len(st) // 2]
    d1k, s1k = blocks(64, 1, 1)              # 1 + 1 blocks at 1024^2, extrapolated below
    step1k = 19 * d1k + 38 * s1k
    return dict(value=1.0 / (4 * step512), unit="images/s (512x512, 4 steps, batch 1: MEASURED whole step)", cores=threads, kind="port",
                physical_cores=phys, logical_cpus=logical, cpu=cpu_name, repeats=len(steps), step_seconds=[round(x, 2) for x in steps],
                sample="CPU oracle fp32 (torch eager, %d threads = physical cores / 2, pinned; %s physical cores, %d logical CPUs visible): "
                       "19 double + 38 single FLUX blocks at D=3072 on 512 txt + 1024 img tokens, batch 1, timed %d x, median = %.1f s per "
                       "512x512 denoise step (21.5 TFLOP); 4 steps per image" % (threads, phys, logical, len(steps), step512),
                ms_per_denoise_step=step512 * 1e3,
                extrapolated_1024=dict(value=1.0 / (4 * step1k), unit="images/s", ms_per_denoise_step=step1k * 1e3,
                                       note="1 double + 1 single block timed on 512 txt + 4096 img tokens (%.2f s + %.2f s), x19 / x38: "
                                            "the workload of the GPU line; an extrapolation, not a measurement" % (d1k, s1k)))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def relaunch_under_torchrun(n, argv):
    """`python bench.py --gpus N` with no launcher around it: become the launcher.  One process per GPU over RCCL on this node,
    rendezvous on 127.0.0.1 (the container hostname may not resolve).  Returns the launcher's exit code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what the host driver supports (RCCL needs it)
    env["X2I_BENCH_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def stub_workload(args, rank):
    """--selftest-launcher: a CPU stand-in for the hot path (a few matmuls on a [B, 4096, 64] 'latent'), so that the launcher, the
    gloo / RCCL rendezvous, the all-gather of the final latents, the barrier-bracketed max-over-ranks timing and the JSON contract can be
    exercised without a GPU (tests/test_dist_cpu.py).  Its line says so in `metric` and `data`; it is never a bench figure."""
    g = torch.Generator().manual_seed(100 + rank)
    lat0 = torch.randn((args.batch, (args.size // 16) ** 2, 64), generator=g)
    w = torch.randn((64, 64), generator=torch.Generator().manual_seed(1234)) / 8   # the same "weights" on every rank

    def one_pass():
        lat = lat0
        for _ in range(args.denoise_steps):
            lat = lat - 0.25 * torch.tanh(lat @ w)
        return lat
    return one_pass


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU")
    ap.add_argument("--denoise-steps", type=int, default=4)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="BASELINE.json config (1-based): 2 = Qwen-3B/shuttle-3 (default, the bench line), 3 = MiniCPM projector, "
                         "4 = InternVL-4B projector, 5 = LightControl edit branch (FLUX.1-dev schedule, 20 steps, 19 ControlNeXt)")
    ap.add_argument("--conditioning", default=None, choices=["qwen3b", "qwen7b", "minicpm", "internvl4b", "internvl1b"],
                    help="override the MLLM whose hidden-state geometry / projector factory feeds the DiT; `--conditioning qwen7b` with the "
                         "default config is the north-star's named target (QwenVL-2.5-7B -> FLUX-schnell)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8"],
                    help="bf16 (default, the headline arithmetic = the reference's) or fp8: the MLP GEMMs (72 %% of the GEMM FLOPs) on "
                         "e4m3 MFMA (FluxTransformer2DModel.enable_fp8), reported as a separate line with its own tolerance")
    ap.add_argument("--fp8-mode", default="mlp", choices=["mlp", "all"],
                    help="with --dtype fp8: 'mlp' = the MLP GEMMs only; 'all' = also the image-stream / single-block QKV and attention "
                         "output projections (97 %% of the GEMM FLOPs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-fp8-lines", action="store_true", help="skip the extra fp8_mlp / fp8_all measurements attached to the bf16 line")
    ap.add_argument("--no-roofline", action="store_true",
                    help="skip the live roofline probe (for rocprofv3 --kernel-trace runs whose GEMM total must contain the timed job only)")
    ap.add_argument("--rccl-selftest", action="store_true",
                    help="initialise the RCCL process group and run the all-gather / barrier / max-over-ranks path even at one rank (under a "
                         "launcher with --nproc-per-node 1, or plain): exercises the N > 1 code path on a one-GPU box")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU stand-in workload over gloo: exercises the launcher / collective / timing / JSON path without a GPU (tests only)")
    args = ap.parse_args(argv)

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    launched = "WORLD_SIZE" in os.environ
    if not args.selftest_launcher and torch.cuda.is_available() and args.gpus > torch.cuda.device_count():
        raise SystemExit("bench.py: --gpus %d but this node shows %d GPU(s); refusing to start ranks that have no device of their own"
                         % (args.gpus, torch.cuda.device_count()))
    if args.gpus > 1 and not launched:
        sys.exit(relaunch_under_torchrun(args.gpus, sys.argv[1:] if argv is None else list(argv)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; refusing to print a line whose n_gpus differs from "
                         "the request" % (args.gpus, world))
    stub = args.selftest_launcher
    dist = None
    if stub:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    use_dist = world > 1 or args.rccl_selftest
    if use_dist:
        import datetime
        import torch.distributed as dist
        if not launched:   # plain `python bench.py --rccl-selftest`: a one-rank group on 127.0.0.1
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if stub:
            dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=10))
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(minutes=30))   # "nccl" IS RCCL on ROCm
        world = dist.get_world_size()   # what is reported below is the group's size, not the flag

    def sync():
        if not stub:
            torch.cuda.synchronize()

    B, N = args.batch, args.denoise_steps
    St = 512
    kind = args.conditioning or {2: "qwen3b", 3: "minicpm", 4: "internvl4b", 5: "qwen7b"}[args.config]
    ops = model = None
    if stub:
        C = Hm = 0
        inner = stub_workload(args, rank)
        noise = inner()
    else:
        from x2i_amd import ops
        from x2i_amd.flux import FluxTransformer2DModel
        from x2i_amd.pipeline import FluxPipeline, FlowMatchEulerDiscreteScheduler
        from x2i_amd.infer.harness import PROJECTORS, HIDDEN
        make, C, pkw = PROJECTORS[kind]
        Hm = HIDDEN[kind]
        proj = make(in_channels=C, device=dev, **pkw).init_random_(seed=7)
        hint = None
        if args.config == 5:
            from x2i_amd.lightcontrol import ControlNeXtModel, FluxTransformer2DModel as LCFlux
            if N == 4:
                N = 20
            model = LCFlux(guidance_embeds=True, device=dev).init_random_(seed=1234)  # every rank holds the SAME weight replica
            nets = []
            for i in range(19):
                net = ControlNeXtModel(device=dev)
                gw = torch.Generator(device="cpu").manual_seed(500 + i)
                for name, prm in net.named_parameters():
                    if prm.dim() > 1:  # conv / linear weights: N(0, 1/fan_in)
                        v = torch.randn(prm.shape, generator=gw) / prm[0].numel() ** 0.5
                    elif name.endswith("weight"):  # GroupNorm scale
                        v = 1.0 + 0.1 * torch.randn(prm.shape, generator=gw)
                    else:
                        v = 0.02 * torch.randn(prm.shape, generator=gw)
                    prm.data.copy_(v)
                nets.append(net)
            pipe = FluxPipeline(model, FlowMatchEulerDiscreteScheduler(shift=3.0, use_dynamic_shifting=True), control_nets=nets)
            hint = (torch.rand((B, 3, args.size, args.size), device=dev) * 2 - 1).bfloat16()
        else:
            model = FluxTransformer2DModel(device=dev).init_random_(seed=1234)  # every rank holds the SAME weight replica
            pipe = FluxPipeline(model, FlowMatchEulerDiscreteScheduler())  # schnell / shuttle-3 schedule: shift 1.0
        if args.dtype == "fp8":
            model.enable_fp8(args.fp8_mode)
        g = torch.Generator(device=dev).manual_seed(100 + rank)
        mllm_hidden = (torch.randn((B, C, St, Hm), device=dev, generator=g) * 3.0).bfloat16()
        noise = torch.randn((B, (args.size // 16) ** 2, 64), device=dev, generator=g).bfloat16()

        def inner():
            pooled, embeds = proj(mllm_hidden)
            return pipe(prompt_embeds=embeds, pooled_prompt_embeds=pooled, num_inference_steps=N, guidance_scale=3.5,
                        height=args.size, width=args.size, output_type="latent", latents=noise, guided_hint=hint,
                        use_graph=not args.no_graph).images
    gathered = [torch.empty_like(noise) for _ in range(world)] if use_dist else None

    def one_pass():
        lat = inner()
        if use_dist:
            dist.all_gather(gathered, lat)  # one all-gather of the final packed latents (RCCL over xGMI on the GPU path)
            return gathered
        return lat if stub else FluxPipeline._unpack_latents(lat, args.size, args.size, 16)

    # graph policy of FluxPipeline: a shape's first pass runs eagerly, its second is captured; both happen here, in front of the W warm-ups
    setup_passes = 0 if (args.no_graph or stub) else 2
    for _ in range(setup_passes + args.warmup):
        one_pass()
    sync()
    if use_dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    sync()
    if use_dist:
        dist.barrier()
    sync()
    dt_local = dt = time.perf_counter() - t0
    rank_ms = [dt_local / args.steps * 1e3]
    if use_dist:
        tt = torch.tensor([dt_local], device=dev, dtype=torch.float64)
        allt = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        rank_ms = [float(x.item()) / args.steps * 1e3 for x in allt]
        dt = max(float(x.item()) for x in allt)          # MAX over ranks
    if ops is not None:
        ops.streamk_check(sync=True)   # outside the timed region: a stream-K segment that gave up = undefined results = no line

    if rank == 0:
        ms_pass = dt / args.steps * 1e3
        images_s = world * B * args.steps / dt
        Si = (args.size // 16) ** 2
        fl = flops_per_denoise_step(B, St, Si)
        line = {
            "metric": "images/sec, FLUX-schnell %dx%d %d-step (projector + denoise loop), whole job" % (args.size, args.size, N),
            "value": images_s, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "setup_passes": setup_passes,
            "ms_per_step": ms_pass, "ms_per_denoise_step": ms_pass / N, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic (random-init weights, random MLLM hidden states, seeded noise)",
            "rccl_ranks": world if use_dist else 0, "rank_ms_per_step": [round(x, 3) for x in rank_ms],
            "launcher": ("self (bench.py re-executed under torch.distributed.run)" if os.environ.get("X2I_BENCH_LAUNCHED") else
                         "external (WORLD_SIZE set by the caller)") if world > 1 else "none",
            "config": {"workload": "BASELINE configs[%d]: %s conditioning (C=%d,H=%d,S_txt=512) -> projector -> %s DiT %dx%d, %d steps"
                                   % (args.config - 1, kind, C, Hm,
                                      "FLUX.1-dev + 19 ControlNeXt (LightControl)" if args.config == 5 else "shuttle-3/FLUX-schnell",
                                      args.size, args.size, N),
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": "batch-sharded x%d" % world,
                       "graph": not args.no_graph},
            "model_tflops_per_gpu": fl * N / (ms_pass * 1e-3) / 1e12,
            "model_frac_of_bf16_peak": fl * N / (ms_pass * 1e-3) / PEAK_BF16,
        }
        if stub:
            line.update({"metric": "LAUNCHER SELF-TEST (CPU stand-in workload, gloo) -- not a benchmark figure", "dtype": "f32",
                         "data": "launcher self-test: no GPU work", "model_tflops_per_gpu": None, "model_frac_of_bf16_peak": None,
                         "config": {"workload": "launcher self-test", "batch_per_gpu": B, "global_batch": B * world,
                                    "parallelism": "batch-sharded x%d" % world},
                         "gathered_shape": [world] + list(noise.shape) if world > 1 else list(noise.shape)})
            print(json.dumps(line), flush=True)
        else:
            if args.config == 5:
                line["metric"] = "images/sec, LightControl FLUX.1-dev %dx%d %d-step (projector + denoise loop), whole job" % (args.size, args.size, N)
                line["model_tflops_per_gpu"] = (fl + 8.30e12 * B) * N / (ms_pass * 1e-3) / 1e12  # + 19 x 436.8 GFLOP per image-step
                line["model_frac_of_bf16_peak"] = line["model_tflops_per_gpu"] * 1e12 / PEAK_BF16
            if args.dtype == "fp8":
                line["dtype"] = "fp8"
                line["dtype_detail"] = ("e4m3 (OCP) operands with fp32 accumulation for ff.net.0/ff.net.2 (image stream) and the single blocks' "
                                        "proj_mlp/proj_out = 72% of the GEMM FLOPs" +
                                        ("; plus image-stream / single-block to_q|k|v and to_out / to_add_out = 97%" if args.fp8_mode == "all" else "") +
                                        "; everything else bf16 as in the headline run; stated drift (~6e-2 on the final latents: outside the 5e-2 proposed for the path) in "
                                        "tests/test_fp8_gpu.py")
                line["metric"] += " [fp8 MLP GEMMs]" if args.fp8_mode == "mlp" else " [fp8 MLP + attention-projection GEMMs]"
                line["roofline"] = None if args.no_roofline else gemm_roofline_fp8(B)
            else:
                line["roofline"] = None if args.no_roofline else gemm_roofline(B)
                if not args.no_roofline and args.size == 1024:
                    line["roofline_attention"] = attention_roofline(B)
                    if args.config == 2 and world == 1:
                        line["vae"] = vae_decode_line(B, images_s)
            if args.dtype == "bf16" and world == 1 and args.config == 2 and not args.no_fp8_lines:
                # the opt-in e4m3 configurations in the same driver-timed record (the headline above stays bf16 = the reference's arithmetic):
                # 1 warm-up + 3 timed passes each; stated tolerances in tests/test_fp8_gpu.py / test_fullscale_parity_gpu.py
                for mode in ("mlp", "all"):
                    model.enable_fp8(mode)
                    for _ in range(max(1, setup_passes)):   # (eager pass, then the capture: see setup_passes)
                        one_pass()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(3):
                        one_pass()
                    torch.cuda.synchronize()
                    ms8 = (time.perf_counter() - t1) / 3 * 1e3
                    line["fp8_" + mode] = {"images_s": B * 1e3 / ms8, "ms_per_denoise_step": ms8 / N, "passes": 3,
                                           "model_tflops_per_gpu": fl * N / (ms8 * 1e-3) / 1e12,
                                           "gemm_flops_on_e4m3": 0.72 if mode == "mlp" else 0.97}
                r8 = gemm_roofline_fp8(B)
                line["fp8_mlp"]["roofline"] = line["fp8_all"]["roofline"] = {k: r8[k] for k in ("bound", "achieved", "peak", "unit", "frac", "frac_best", "frac_of_pipe_alone", "kernel", "shapes")}
                model.enable_fp8(None)
                ops.streamk_check(sync=True)
            if not args.no_cpu_baseline and world == 1 and args.config == 2:
                line["cpu_baseline"] = cpu_baseline()
            print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()  # rank 0 is still timing the roofline kernels: leave the job together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
