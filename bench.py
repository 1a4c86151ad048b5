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
    """Dominant kernel: the bf16 MFMA GEMM.  Times the single-block proj_out-shaped GEMM (M=B*4608, N=3072, K=15360)
    and the fused-in GEMM (N=21504, K=3072) with HIP events on the launch stream; achieved = algorithmic FLOP / time."""
    from x2i_amd import ops
    D, S = 3072, 4608
    res = []
    for (M, N, K) in ((B * S, 7 * D, D), (B * S, D, 5 * D)):
        A = torch.randn(M, K, device="cuda").bfloat16()
        W = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
        bias = torch.randn(N, device="cuda").bfloat16()
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        for _ in range(3):
            ops.gemm(A, W, bias, out=out)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(iters):
            ops.gemm(A, W, bias, out=out)
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) / iters * 1e-3
        res.append((2.0 * M * N * K, t))
        del A, W, out
    fl = sum(r[0] for r in res)
    tt = sum(r[1] for r in res)
    return dict(bound="mfma", achieved=fl / tt / 1e12, peak=PEAK_BF16 / 1e12, unit="TFLOP/s", frac=fl / tt / PEAK_BF16,
                traffic=None, kernel="gemm_bf16_kernel", shapes="M=%d: N=21504,K=3072 + N=3072,K=15360" % (B * S))


def cpu_baseline(budget_s=25.0):
    """The CPU oracle (restated reference path, fp32, torch eager) on the host cores: one double block + one single block
    at full width on a bounded token sample, extrapolated linearly in block count to one denoise step of one image."""
    from oracle import flux as OF
    from oracle import primitives as P
    from oracle import sampler as OS
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    cfg = dict(OF.DEFAULT_CFG)
    cfg.update(num_layers=1, num_single_layers=1)
    sd = OF.random_flux_state_dict(cfg, seed=0)
    St, h2 = 512, 32  # 512 text + 1024 image tokens (= the 512x512 configuration of BASELINE configs[0])
    ids = torch.cat([torch.zeros(St, 3), OS.prepare_latent_image_ids(h2, h2)], 0)
    rot = P.flux_pos_embed(ids)
    hid, enc, temb = torch.randn(1, h2 * h2, 3072), torch.randn(1, St, 3072), torch.randn(1, 3072)
    with torch.no_grad():
        t0 = time.time()
        e, h = OF.double_block(sd, "transformer_blocks.0", hid, enc, temb, rot, 24)
        t_d = time.time() - t0
        j = torch.cat([e, h], 1)
        t0 = time.time()
        OF.single_block(sd, "single_transformer_blocks.0", j, temb, rot, 24)
        t_s = time.time() - t0
    step_512 = 19 * t_d + 38 * t_s  # seconds per denoise step, 512x512, batch 1
    f512 = flops_per_denoise_step(1, 512, 1024)
    f1024 = flops_per_denoise_step(1, 512, 4096)
    step_1024 = step_512 * f1024 / f512  # FLOP-proportional extrapolation to the 1024x1024 workload
    return dict(value=1.0 / (4 * step_1024), unit="images/s", cores=cores, kind="port",
                sample="CPU oracle fp32: 1 double + 1 single FLUX block at D=3072, 512 txt + 1024 img tokens, batch 1 "
                       "(%.2fs + %.2fs); x19 / x38 blocks -> %.1f s/step at 512^2; scaled by FLOPs (x%.2f) to 1024^2; 4 steps"
                       % (t_d, t_s, step_512, f1024 / f512),
                ms_per_denoise_step_512=step_512 * 1e3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU")
    ap.add_argument("--denoise-steps", type=int, default=4)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
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

    B, N = args.batch, args.denoise_steps
    St, C, Hm = 512, 37, 2048
    model = FluxTransformer2DModel(device=dev).init_random_(seed=1234 + rank)
    proj = create_proj3_qwen3b(in_channels=C, use_t5=False, use_scale=False, use_cnn=True, device=dev).init_random_(seed=7)
    pipe = FluxPipeline(model, FlowMatchEulerDiscreteScheduler())  # schnell / shuttle-3 schedule: shift 1.0
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    mllm_hidden = (torch.randn((B, C, St, Hm), device=dev, generator=g) * 3.0).bfloat16()
    noise = torch.randn((B, (args.size // 16) ** 2, 64), device=dev, generator=g).bfloat16()
    gathered = [torch.empty_like(noise) for _ in range(world)] if world > 1 else None

    def one_pass():
        pooled, embeds = proj(mllm_hidden)
        lat = pipe(prompt_embeds=embeds, pooled_prompt_embeds=pooled, num_inference_steps=N, guidance_scale=3.5,
                   height=args.size, width=args.size, output_type="latent", latents=noise,
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
            "config": {"workload": "BASELINE configs[1]: QwenVL2.5-3B conditioning (C=37,H=2048,S_txt=512) -> projector -> "
                                   "shuttle-3/FLUX-schnell DiT %dx%d, %d steps" % (args.size, args.size, N),
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": "batch-sharded x%d" % world,
                       "graph": not args.no_graph},
            "model_tflops_per_gpu": fl * N / (ms_pass * 1e-3) / 1e12,
            "model_frac_of_bf16_peak": fl * N / (ms_pass * 1e-3) / PEAK_BF16,
        }
        line["roofline"] = gemm_roofline(B)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
