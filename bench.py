#!/usr/bin/env python3
"""bench.py -- images/s and ms/denoise-step of the X2I sampling hot path on MI355X.

One "step" (driver contract) = one pass of the hot path over one batch of synthetic input:
    alignment projector (MLLM hidden states -> prompt_embeds, pooled) + N-step FLUX denoising loop + unpack,
i.e. everything between the MLLM and the VAE for a batch of images.  Workload at N=1: BASELINE.json configs[1]
(QwenVL2.5-3B conditioning: C=37, H=2048, S_txt=512; shuttle-3/FLUX-schnell architecture, 1024x1024, 4 steps,
batch 4), random-init weights, synthetic inputs already resident in HBM.  N>1: weak scaling, the batch axis is
sharded (batch 4 per rank), one all-gather of the final packed latents over RCCL.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (dominant kernel = bf16 MFMA GEMM, measured live with HIP
events on the launch stream) and "cpu_baseline" (the CPU oracle timed on the host cores on a bounded sample).
"""
import argparse
import json
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


def gemm_roofline(B, iters=10):
    """Dominant kernel: the bf16 MFMA GEMM.  Times the two largest launch shapes of a single-stream block -- proj_mlp + bias + GELU
    (M=B*4608, N=12288, K=3072) and proj_out + bias (N=3072, K=15360; in the model this launch also adds the gated residual) -- with
    HIP events on the launch stream; achieved = algorithmic FLOP / time.  (Same two launches since round 1, for comparability.)"""
    from x2i_amd import ops
    D, S = 3072, 4608
    B = min(B, 8)   # (the probe's flattened A operand must stay below the kernels' 2 GB operand limit: 8 x 4608 rows x 15360 x 2 B = 1.1 GB)
    res = []
    for (M, N, K, act) in ((B * S, 4 * D, D, 1), (B * S, D, 5 * D, 0)):
        A = torch.randn(M, K, device="cuda").bfloat16()
        W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
        bias = torch.randn(N, device="cuda").bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(3):
            ops.gemm(A, W, bias, out=out, act=act)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(iters):
            ops.gemm(A, W, bias, out=out, act=act)
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) / iters * 1e-3
        res.append((2.0 * M * N * K, t))
        del A, W, out
    fl = sum(r[0] for r in res)
    tt = sum(r[1] for r in res)
    # HBM-side traffic of the same two launches from the committed rocprofv3 PMC passes over tools/roofline_probe.py (tools/pmc_roofline.sh:
    # FETCH_SIZE doubled per the gfx950 correction, + WRITE_SIZE, separate passes; per-launch means summed over every GEMM kernel the
    # pair launches; since round 3 that is ONE persistent 256^2 kernel per GEMM, its last round cut along K).  Only valid for the profiled batch (B=4 -> M=18432).
    traffic, src = None, None
    pj = os.path.join(ROOT, "profiles", "r03_pmc_roofline.json")
    if B == 4 and os.path.exists(pj):
        try:
            d = json.load(open(pj))
            traffic = float(d["traffic_bytes_per_pair"])
            src = "profiles/r03_pmc_roofline.json (%s)" % d.get("note", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE")
        except (KeyError, ValueError):
            traffic = None
    alg_bytes = sum(2.0 * (M * K + N * K + M * N) for (M, N, K) in ((B * S, 4 * D, D), (B * S, D, 5 * D)))
    return dict(bound="mfma", achieved=fl / tt / 1e12, peak=PEAK_BF16 / 1e12, unit="TFLOP/s", frac=fl / tt / PEAK_BF16,
                traffic=traffic, traffic_source=src, algorithmic_bytes=alg_bytes, kernel="gemm256p_bf16_kernel",
                shapes="M=%d: N=12288,K=3072 (+GELU) + N=3072,K=15360 (single-block proj_mlp / proj_out shapes, 1.39 + 1.74 TFLOP; one persistent "
                       "256^2 launch each, last round cut along K)" % (B * S))


def gemm_roofline_fp8(B, iters=10):
    """--dtype fp8: the same two launches on the e4m3 kernel (gemm256_fp8_kernel: MX-scaled K=128 MFMA), priced against the 5 PF
    dense fp8 peak.  proj_mlp + GELU writes e4m3, proj_out carries the gated residual, exactly as the fp8 model issues them."""
    from x2i_amd import ops
    D, S = 3072, 4608
    B = min(B, 8)   # (the probe's flattened A operand must stay below the kernels' 2 GB operand limit: 8 x 4608 rows x 15360 x 2 B = 1.1 GB)
    res = []
    for (M, N, K, gelu) in ((B * S, 4 * D, D, True), (B * S, D, 5 * D, False)):
        A8, sa = ops.quantize_rows_fp8(torch.randn(M, K, device="cuda").bfloat16())
        W8, sw = ops.quantize_rows_fp8((torch.randn(N, K, device="cuda") * 0.02).bfloat16())
        bias = torch.randn(N, device="cuda").bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=ops.FP8 if gelu else torch.bfloat16)
        gate = torch.randn(1, N, device="cuda")
        if gelu:
            fn = lambda: ops.gemm_fp8(A8, W8, bias, out=out, a_scale=sa, w_scale=sw, act=1, out_fp8=True)  # noqa: E731
        else:
            fn = lambda: ops.gemm_fp8(A8, W8, bias, out=out, w_scale=sw, res=out, gate=gate)  # noqa: E731
        for _ in range(3):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        res.append((2.0 * M * N * K, s.elapsed_time(e) / iters * 1e-3))
        del A8, W8, out
    fl, tt = sum(r[0] for r in res), sum(r[1] for r in res)
    alg_bytes = (B * S * D + 4 * D * D + B * S * 4 * D) + (B * S * 5 * D + 5 * D * D + 2 * 2 * B * S * D)
    traffic, src = None, None
    pj = os.path.join(ROOT, "profiles", "r02d_pmc_gemm_attn.json")
    if B == 4 and os.path.exists(pj):
        d = json.load(open(pj))
        try:
            traffic = sum(d[k]["hbm_read_bytes_corrected"] + d[k]["hbm_write_bytes"]
                          for k in ("gemm256_fp8_kernel<1, false, true> grid=1769472", "gemm256_fp8_kernel<0, true, false> grid=442368"))
            src = "profiles/r02d_pmc_gemm_attn.json (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE, separate passes; L2<->fabric bytes of both launches)"
        except KeyError:
            traffic = None
    return dict(bound="mfma", achieved=fl / tt / 1e12, peak=PEAK_FP8 / 1e12, unit="TFLOP/s", frac=fl / tt / PEAK_FP8, traffic=traffic,
                traffic_source=src, algorithmic_bytes=float(alg_bytes), kernel="gemm256_fp8_kernel",
                shapes="M=%d: N=12288,K=3072 (+GELU, e4m3 out) + N=3072,K=15360 (gated residual), e4m3 operands" % (B * S))


def _pick_cpu_threads():
    """torch's CPU GEMM peaks well below the logical core count in this container (cgroup quota / SMT): probe a few thread
    counts on a short matmul and keep the fastest."""
    a, b = torch.randn(1024, 3072), torch.randn(3072, 3072)
    n_max = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    best, best_t = 1, 1e9
    for n in (8, 16, 32, 64, 128, 256):
        if n > n_max:
            break
        torch.set_num_threads(n)
        torch.nn.functional.linear(a, b)
        t0 = time.time()
        for _ in range(3):
            torch.nn.functional.linear(a, b)
        dt = time.time() - t0
        if dt < best_t:
            best, best_t = n, dt
    torch.set_num_threads(best)
    return best, n_max


def _physical_cores():
    try:
        import subprocess
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {l.split(":")[0].strip(): l.split(":", 1)[1].strip() for l in out.splitlines() if ":" in l}
        return int(kv["Core(s) per socket"]) * int(kv["Socket(s)"]), kv.get("Model name", "?")
    except Exception:  # noqa: BLE001
        return None, "?"


def cpu_baseline():
    """The CPU oracle (restated reference path, fp32, torch eager) on the host cores.  MEASURED: one whole denoise step at 512 x 512,
    batch 1 -- 19 double-stream + 38 single-stream FLUX blocks at full width on 512 text + 1024 image tokens (21.5 TFLOP; SURVEY.md
    section 8(d)); the 57 blocks share one double-block and one single-block weight set (same arithmetic and bytes per block; generating
    11.9 B random fp32 parameters on the host would take longer than the measurement).  EXTRAPOLATED and labelled so: the 1024 x 1024
    step the GPU line is quoted on (74.4 TFLOP: 1 + 1 blocks timed on 512 + 4096 tokens, x19 / x38)."""
    from oracle import flux as OF
    from oracle import primitives as P
    from oracle import sampler as OS
    threads, logical = _pick_cpu_threads()
    phys, cpu_name = _physical_cores()
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

    d512, s512 = blocks(32, 19, 38)          # measured: a whole 512^2 step
    step512 = d512 + s512
    d1k, s1k = blocks(64, 1, 1)              # 1 + 1 blocks at 1024^2, extrapolated below
    step1k = 19 * d1k + 38 * s1k
    return dict(value=1.0 / (4 * step512), unit="images/s (512x512, 4 steps, batch 1: MEASURED whole step)", cores=threads, kind="port",
                physical_cores=phys, logical_cpus=logical, cpu=cpu_name,
                sample="CPU oracle fp32 (torch eager, %d threads = fastest of a thread sweep; %s physical cores, %d logical CPUs visible): "
                       "19 double + 38 single FLUX blocks at D=3072 on 512 txt + 1024 img tokens, batch 1, timed once = %.1f s per "
                       "512x512 denoise step (21.5 TFLOP); 4 steps per image" % (threads, phys, logical, step512),
                ms_per_denoise_step=step512 * 1e3,
                extrapolated_1024=dict(value=1.0 / (4 * step1k), unit="images/s", ms_per_denoise_step=step1k * 1e3,
                                       note="1 double + 1 single block timed on 512 txt + 4096 img tokens (%.2f s + %.2f s), x19 / x38: "
                                            "the workload of the GPU line; an extrapolation, not a measurement" % (d1k, s1k)))


def main():
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
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8"],
                    help="bf16 (default, the headline arithmetic = the reference's) or fp8: the MLP GEMMs (72 %% of the GEMM FLOPs) on "
                         "e4m3 MFMA (FluxTransformer2DModel.enable_fp8), reported as a separate line with its own tolerance")
    ap.add_argument("--fp8-mode", default="mlp", choices=["mlp", "all"],
                    help="with --dtype fp8: 'mlp' = the MLP GEMMs only; 'all' = also the image-stream / single-block QKV and attention "
                         "output projections (97 %% of the GEMM FLOPs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-fp8-lines", action="store_true", help="skip the extra fp8_mlp / fp8_all measurements attached to the bf16 line")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from x2i_amd.flux import FluxTransformer2DModel
    from x2i_amd.pipeline import FluxPipeline, FlowMatchEulerDiscreteScheduler
    from x2i_amd.proj import create_proj3_qwen3b

    from x2i_amd.infer.harness import PROJECTORS, HIDDEN
    B, N = args.batch, args.denoise_steps
    St = 512
    kind = {2: "qwen3b", 3: "minicpm", 4: "internvl4b", 5: "qwen7b"}[args.config]
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
    gathered = [torch.empty_like(noise) for _ in range(world)] if world > 1 else None

    def one_pass():
        pooled, embeds = proj(mllm_hidden)
        lat = pipe(prompt_embeds=embeds, pooled_prompt_embeds=pooled, num_inference_steps=N, guidance_scale=3.5,
                   height=args.size, width=args.size, output_type="latent", latents=noise, guided_hint=hint,
                   use_graph=not args.no_graph).images
        if world > 1:
            dist.all_gather(gathered, lat)  # one RCCL all-gather of the final packed latents
            return gathered
        return FluxPipeline._unpack_latents(lat, args.size, args.size, 16)

    for _ in range(args.warmup):
        one_pass()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        ms_pass = dt / args.steps * 1e3
        images_s = world * B * args.steps / dt
        Si = (args.size // 16) ** 2
        fl = flops_per_denoise_step(B, St, Si)
        line = {
            "metric": "images/sec, FLUX-schnell 1024x1024 4-step (projector + denoise loop), whole job",
            "value": images_s, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_pass, "ms_per_denoise_step": ms_pass / N, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic (random-init weights, random MLLM hidden states, seeded noise)",
            "config": {"workload": "BASELINE configs[%d]: %s conditioning (C=%d,H=%d,S_txt=512) -> projector -> %s DiT %dx%d, %d steps"
                                   % (args.config - 1, kind, C, Hm,
                                      "FLUX.1-dev + 19 ControlNeXt (LightControl)" if args.config == 5 else "shuttle-3/FLUX-schnell",
                                      args.size, args.size, N),
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": "batch-sharded x%d" % world,
                       "graph": not args.no_graph},
            "model_tflops_per_gpu": fl * N / (ms_pass * 1e-3) / 1e12,
            "model_frac_of_bf16_peak": fl * N / (ms_pass * 1e-3) / PEAK_BF16,
        }
        if args.config == 5:
            line["metric"] = "images/sec, LightControl FLUX.1-dev 1024x1024 %d-step (projector + denoise loop), whole job" % N
            line["model_tflops_per_gpu"] = (fl + 8.30e12 * B) * N / (ms_pass * 1e-3) / 1e12  # + 19 x 436.8 GFLOP per image-step
            line["model_frac_of_bf16_peak"] = line["model_tflops_per_gpu"] * 1e12 / PEAK_BF16
        if args.dtype == "fp8":
            line["dtype"] = "fp8"
            line["dtype_detail"] = ("e4m3 (OCP) operands with fp32 accumulation for ff.net.0/ff.net.2 (image stream) and the single blocks' "
                                    "proj_mlp/proj_out = 72% of the GEMM FLOPs" +
                                    ("; plus image-stream / single-block to_q|k|v and to_out / to_add_out = 97%" if args.fp8_mode == "all" else "") +
                                    "; everything else bf16 as in the headline run; stated tolerance vs the fp32 oracle in "
                                    "tests/test_fp8_gpu.py")
            line["metric"] += " [fp8 MLP GEMMs]" if args.fp8_mode == "mlp" else " [fp8 MLP + attention-projection GEMMs]"
            line["roofline"] = gemm_roofline_fp8(B)
        else:
            line["roofline"] = gemm_roofline(B)
        if args.dtype == "bf16" and world == 1 and args.config == 2 and not args.no_fp8_lines:
            # the opt-in e4m3 configurations in the same driver-timed record (the headline above stays bf16 = the reference's arithmetic):
            # 1 warm-up + 3 timed passes each; stated tolerances in tests/test_fp8_gpu.py / test_fullscale_parity_gpu.py
            for mode in ("mlp", "all"):
                model.enable_fp8(mode)
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
            line["fp8_mlp"]["roofline"] = line["fp8_all"]["roofline"] = {k: r8[k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "shapes")}
            model.enable_fp8(None)
        if not args.no_cpu_baseline and world == 1 and args.config == 2:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()  # rank 0 is still timing the roofline kernels: leave the job together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
