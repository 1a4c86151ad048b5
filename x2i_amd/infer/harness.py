"""Shared sampling harness (Row H): MLLM hidden states -> projector -> FluxPipeline -> latents [-> VAE -> image files].

Re-authored counterpart of infer/inference_qwenvl.py (and its MiniCPM / InternVL / multi-turn siblings); nothing is
copied from them.  What is kept from the reference's call surface:
  * CLI flags: --qwen_size / --internvl_size / --minicpm_path, --flux_path, --num_steps (4), --num_gen_imgs (1),
    --task, --use_answer (infer/inference_qwenvl.py:27-37, inference_minicpm.py:29-35, inference_internvl.py)
  * hidden-state stacking contract: every layer's hidden state of the PROMPT pass stacked to [B, C, S, H]
    (Qwen: torch.cat(hidden_states[0]).unsqueeze(0) for B=1, inference_qwenvl.py:121-132; batched form
    torch.stack(hs, dim=1), inference_minicpm.py:116-118); --use_answer concatenates the generated-token states (:125-129)
  * projector factory per MLLM size and the "module." prefix strip of the checkpoint loader (:77-94)
  * generate(): pipeline kwargs, vae_scale_factor = 2 ** len(vae.config.block_out_channels), _unpack_latents,
    latents / scaling_factor + shift_factor, vae.decode, postprocess (:183-217) -- and ALL images of the batch are saved
    (the reference saves only image[0])
New in this build: batches of prompts per call, torchrun batch sharding with one all-gather of the final latents
(x2i_amd.dist), and --synthetic (random weights + synthetic MLLM hidden states) so the harness runs with no checkpoints.
The MLLM encoders and the VAE stay on stock PyTorch-ROCm (out of scope / "next" N1, N2).
"""
import argparse
import json
import os

import torch

from .. import dist as xdist
from .. import ops
from .. import proj as xproj
from ..flux import FluxTransformer2DModel
from ..pipeline import FlowMatchEulerDiscreteScheduler, FluxPipeline

# (factory, in_channels C = n_layers + 1, kwargs) per conditioning model -- infer/inference_qwenvl.py:79-82,
# inference_internvl.py:75-78, inference_minicpm.py:78
PROJECTORS = {
    "qwen3b": (xproj.create_proj3_qwen3b, 37, dict(use_t5=False, use_scale=False, use_cnn=True)),
    "qwen7b": (xproj.create_proj3_qwen7b, 29, dict(use_t5=False, use_scale=False, use_cnn=True)),
    "internvl1b": (xproj.create_proj_internvl1b, 25, dict(use_t5=False, use_scale=True)),
    "internvl4b": (xproj.create_proj_internvl4b, 37, dict(use_t5=False, use_scale=False)),
    "minicpm": (xproj.create_proj_minicpm, 29, dict(use_t5=False, use_scale=False, use_cnn=True)),
}
HIDDEN = {"qwen3b": 2048, "qwen7b": 3584, "internvl1b": 896, "internvl4b": 2048, "minicpm": 3584}
TASKS = ("all", "text2image", "image2image", "imagetext2image", "video2image", "audio2image", "x2image")


def build_parser(family):
    p = argparse.ArgumentParser("Inference (x2i_amd, %s)" % family, add_help=True)
    if family == "qwenvl":
        p.add_argument("--qwen_size", type=str, default="7b", choices=["3b", "7b"])
        p.add_argument("--qwen_path", type=str, default=None)
    elif family == "internvl":
        p.add_argument("--internvl_size", type=str, default="4b", choices=["1b", "4b"])
        p.add_argument("--internvl_path", type=str, default=None)
    elif family == "minicpm":
        p.add_argument("--minicpm_path", type=str, default="openbmb/MiniCPM-o-2_6")
    p.add_argument("--proj_path", type=str, default=None, help="projector checkpoint (diffusion_pytorch_model.bin)")
    p.add_argument("--flux_path", type=str, default="shuttleai/shuttle-3-diffusion")
    p.add_argument("--use_answer", type=bool, default=False)  # type=bool kept as in the reference (any non-empty string -> True)
    p.add_argument("--num_steps", type=int, default=4)
    p.add_argument("--num_gen_imgs", type=int, default=1)
    p.add_argument("--task", type=str, default="all", choices=TASKS)
    p.add_argument("--height", type=int, default=1024)
    p.add_argument("--width", type=int, default=1024)
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--outputs", type=str, default=None)
    p.add_argument("--assets", type=str, default="./data", help="directory holding image/ video/ audio/ demo assets")
    p.add_argument("--batch", type=int, default=1,
                   help="prompts per pipeline call and rank (the reference always uses 1); under torchrun a task's prompts are packed "
                        "batch x world_size at a time and sharded over the ranks")
    p.add_argument("--synthetic", action="store_true", help="no checkpoints: random weights, synthetic MLLM hidden states")
    p.add_argument("--no_graph", action="store_true")
    p.add_argument("--full_generate", action="store_true",
                   help="run the MLLM's 128-token generate() as the reference does instead of the single prefill forward")
    p.add_argument("--decode", action="store_true", help="with --synthetic: also run the (random-weight) VAE decoder and save images")
    return p


def stack_hidden_states(hidden_states, use_answer=False):
    """HF generate(..., output_hidden_states=True).hidden_states -> [B, C, S, H].

    hidden_states[0] is the tuple over layers of the prompt pass [B, S, H]; hidden_states[1:] the per-generated-token
    tuples [B, 1, H].  use_answer=True keeps only the generated-token states, concatenated along S
    (infer/inference_qwenvl.py:125-129)."""
    if use_answer:
        steps = [torch.stack(tuple(h), dim=1) for h in hidden_states[1:]]  # each [B, C, 1, H]
        return torch.cat(steps, dim=2)
    return torch.stack(tuple(hidden_states[0]), dim=1)


def strip_module_prefix(state_dict):
    """Checkpoints are saved from DDP-wrapped modules (train/train_qwenvl.py:641-647): drop 'module.'."""
    return {k.replace("module.", ""): v for k, v in state_dict.items()}


def load_projector(kind, path=None, device="cuda", seed=0):
    """Factory per MLLM size; with a checkpoint path the constructor arguments come from the file itself (flat .bin with
    optional "module." prefix, or the ComfyUI {"config","state_dict"} packaging) and are checked against `kind`."""
    make, C, kw = PROJECTORS[kind]
    if path is None:
        return make(in_channels=C, device=device, **kw).init_random_(seed).eval()
    from ..checkpoints import load_projector_checkpoint
    proj = load_projector_checkpoint(path, device, in_channels=C)
    got_c = proj.cha_scale.shape[1] if proj.use_scale else (proj.conv.weight.shape[1] if proj.use_cnn else C)
    if got_c != C or proj.mlp.layernorm.weight.shape[0] != HIDDEN[kind]:
        raise ValueError("projector checkpoint %s (C=%d, H=%d) does not match --%s (C=%d, H=%d)"
                         % (path, got_c, proj.mlp.layernorm.weight.shape[0], kind, C, HIDDEN[kind]))
    return proj


def load_pipeline(flux_path, device="cuda", synthetic=False, seed=0):
    """FluxPipeline without text encoders / VAE, as infer/inference_qwenvl.py:72-73 builds it; the scheduler constants are
    read from the checkpoint's scheduler_config.json, never hard-coded."""
    if synthetic:
        tr = FluxTransformer2DModel(device=device).init_random_(seed)
        return FluxPipeline(tr, FlowMatchEulerDiscreteScheduler())
    from ..checkpoints import load_pipeline_dir
    tr, sched = load_pipeline_dir(flux_path, device)
    return FluxPipeline(tr, sched)


def load_vae(flux_path, device, synthetic=False):
    """VAE decoder on the HIP path (x2i_amd.vae): `vae/` of the diffusers pipeline directory."""
    from ..vae import AutoencoderKL
    if synthetic:
        return AutoencoderKL(device=device).init_random_(3)
    return AutoencoderKL.from_pretrained(flux_path, subfolder="vae", device=device)


class SyntheticConditioner:
    """Stands in for the MLLM: seeded hidden states of the right [B, C, S, H] shape and activation scale."""

    def __init__(self, kind, device, seq_len=512):
        self.kind, self.device, self.seq_len = kind, device, seq_len
        self.calls = 0

    def __call__(self, videos=None, images=None, audios=None, text_prompt=None, batch=1):
        g = torch.Generator(device=self.device).manual_seed(1000 + self.calls)
        self.calls += 1
        C, H = PROJECTORS[self.kind][1], HIDDEN[self.kind]
        return (torch.randn((batch, C, self.seq_len, H), device=self.device, generator=g) * 3.0).to(torch.bfloat16)


class Harness:
    def __init__(self, args, kind, conditioner, device="cuda"):
        self.args, self.kind, self.device = args, kind, torch.device(device)
        self.rank, self.world = xdist.init_from_env(device=self.device if self.device.type == "cuda" else None)
        self.proj = load_projector(kind, None if args.synthetic else args.proj_path, device)
        self.pipeline = load_pipeline(args.flux_path, device, args.synthetic)
        self.vae = load_vae(args.flux_path, device, synthetic=True) if (args.synthetic and args.decode) else (
            None if args.synthetic else load_vae(args.flux_path, device))
        self.conditioner = conditioner
        self.outputs = args.outputs or "./outputs_%s" % kind

    @torch.no_grad()
    def embeds(self, **inputs):
        """MLLM hidden states [B,C,S,H] -> (pooled_prompt_embeds, prompt_embeds) (infer/inference_qwenvl.py:178-180)."""
        return self.proj(self.conditioner(**inputs))

    @torch.no_grad()
    def generate(self, pooled, embeds, subdir, filename, seed=None, height=None, width=None):
        a = self.args
        height, width = height or a.height, width or a.width
        out_dir = os.path.join(self.outputs, subdir)
        os.makedirs(out_dir, exist_ok=True)
        gen = None
        if seed is not None:
            gen = torch.Generator(self.device).manual_seed(seed)
        B = embeds.shape[0]
        if self.world > 1:
            # every rank draws the same global noise, samples its slice, and receives all latents back
            C = self.pipeline.transformer.config.in_channels // 4
            lat, _ = self.pipeline.prepare_latents(B, C, height, width, embeds.dtype, self.device, gen)
            latents = xdist.sample_sharded(self.pipeline, embeds, pooled, latents=lat, num_inference_steps=a.num_steps,
                                           guidance_scale=3.5, height=height, width=width, output_type="latent",
                                           use_graph=not a.no_graph)
        else:
            latents = self.pipeline(prompt_embeds=embeds, pooled_prompt_embeds=pooled, num_inference_steps=a.num_steps,
                                    guidance_scale=3.5, height=height, width=width, output_type="latent", generator=gen,
                                    use_graph=not a.no_graph).images
        ops.streamk_check(sync=True)   # the images are about to leave the GPU: a stream-K give-up marker (undefined results) is fatal here
        if self.rank != 0:
            return latents
        if self.vae is None:
            torch.save(latents.cpu(), os.path.join(out_dir, filename + "_latents.pt"))
            return latents
        vsf = 2 ** len(self.vae.config.block_out_channels)  # :209
        x = FluxPipeline._unpack_latents(latents, height, width, vsf)
        x = (x / self.vae.config.scaling_factor) + self.vae.config.shift_factor  # :213
        image = self.vae.decode(x, return_dict=False)[0]
        img = ((image.float() / 2 + 0.5).clamp(0, 1) * 255).round().to(torch.uint8).permute(0, 2, 3, 1).cpu().numpy()
        from PIL import Image
        for i in range(img.shape[0]):  # the reference saves image[0] only; a batched build saves them all
            Image.fromarray(img[i]).save(os.path.join(out_dir, "%s%s.jpg" % (filename, "" if B == 1 else "_b%d" % i)))
        return latents

    # ---- batched real prompts (new in this build; the reference calls the pipeline with B = 1)
    @torch.no_grad()
    def condition_jobs(self, jobs):
        """MLLM + projector for a list of jobs -> [(pooled [1,768], prompt_embeds [1,S,4096])], one entry per job.  Conditioners
        take one prompt at a time (their processors build one multimodal message); S may differ per job (MiniCPM inputs are
        unpadded, infer/inference_minicpm.py:160-177), which is why batching happens AFTER this step."""
        out = []
        for job in jobs:
            pooled, embeds = self.embeds(**job)
            for b in range(embeds.shape[0]):
                out.append((pooled[b:b + 1], embeds[b:b + 1]))
        return out

    @staticmethod
    def group_by_length(conds):
        """Indices of `conds` grouped by text length S, in first-seen order: samples of equal S ride in one pipeline call.
        (Zero-padding a shorter prompt is NOT equivalent: FLUX's joint attention has no mask, padded text rows would be attended.)"""
        groups = {}
        for i, (_, e) in enumerate(conds):
            groups.setdefault(e.shape[1], []).append(i)
        return list(groups.values())

    @torch.no_grad()
    def generate_jobs(self, jobs, subdir, filenames, seed=None):
        """Up to --batch prompts in one sampling call per text length; under torchrun rank r conditions and samples ONLY its
        shard of the prompts (x2i_amd.dist.shard_range) and one all-gather returns every rank the full set of latents."""
        a = self.args
        n = len(jobs)
        lo, hi = xdist.shard_range(n, self.rank, self.world)
        C = self.pipeline.transformer.config.in_channels // 4
        gen = torch.Generator(self.device).manual_seed(seed) if seed is not None else None
        # the GLOBAL noise is drawn identically on every rank and sliced, so results do not depend on the number of ranks
        noise, _ = self.pipeline.prepare_latents(n, C, a.height, a.width, torch.bfloat16, self.device, gen)
        conds = self.condition_jobs(jobs[lo:hi])
        local = torch.empty((hi - lo,) + tuple(noise.shape[1:]), device=self.device, dtype=noise.dtype)
        for idx in self.group_by_length(conds):
            pooled = torch.cat([conds[i][0] for i in idx], 0)
            embeds = torch.cat([conds[i][1] for i in idx], 0)
            lat = self.pipeline(prompt_embeds=embeds, pooled_prompt_embeds=pooled, num_inference_steps=a.num_steps, guidance_scale=3.5,
                                height=a.height, width=a.width, output_type="latent",
                                latents=noise[[lo + i for i in idx]], use_graph=not a.no_graph).images
            local[idx] = lat
        latents = xdist.all_gather_batch(local, n) if self.world > 1 else local
        ops.streamk_check(sync=True)
        if self.rank == 0:
            self.save_outputs(latents, subdir, filenames, a.height, a.width)
        return latents

    def save_outputs(self, latents, subdir, filenames, height, width):
        out_dir = os.path.join(self.outputs, subdir)
        os.makedirs(out_dir, exist_ok=True)
        if self.vae is None:
            for i, fn in enumerate(filenames):
                torch.save(latents[i:i + 1].cpu(), os.path.join(out_dir, fn + "_latents.pt"))
            return
        vsf = 2 ** len(self.vae.config.block_out_channels)
        x = FluxPipeline._unpack_latents(latents, height, width, vsf)
        x = (x / self.vae.config.scaling_factor) + self.vae.config.shift_factor
        from ..pipeline import VaeImageProcessor
        images = VaeImageProcessor(vae_scale_factor=vsf).postprocess(self.vae.decode(x, return_dict=False)[0], output_type="pil")
        for im, fn in zip(images, filenames):
            im.save(os.path.join(out_dir, fn + ".jpg"))

    def run_tasks(self, tasks):
        """tasks: {name: [dict(filename=..., **conditioner_inputs)]}; runs the ones selected by --task.  --synthetic keeps the
        one-job-per-call form with a synthetic batch of --batch samples; with a real MLLM the jobs of a task are packed --batch at a
        time into one sampling call (generate_jobs)."""
        a = self.args
        for name, jobs in tasks.items():
            if a.task not in ("all", name):
                continue
            for i in range(a.num_gen_imgs):
                if a.synthetic:
                    for job in jobs:
                        job = dict(job)
                        fn = job.pop("filename")
                        pooled, embeds = self.embeds(batch=a.batch, **job)
                        self.generate(pooled, embeds, name, "%s_%d" % (fn, i), seed=a.seed)
                    continue
                bs = max(1, a.batch) * self.world
                for j0 in range(0, len(jobs), bs):
                    chunk = [dict(j) for j in jobs[j0:j0 + bs]]
                    fns = ["%s_%d" % (j.pop("filename"), i) for j in chunk]
                    self.generate_jobs(chunk, name, fns, seed=a.seed)


def asset(args, *parts):
    return os.path.join(args.assets, *parts)
