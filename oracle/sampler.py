"""Oracle restatement of the sampling loop: diffusers 0.31.0 FluxPipeline.__call__
(as driven by infer/inference_qwenvl.py:188-207) + FlowMatchEulerDiscreteScheduler.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED for the
third-party parts; pack/unpack/ids/calculate_shift are pinned against the
reference's in-repo copies (train/train_qwenvl.py:216-246,
lightcontrol/train_lightcontrol.py:403-410) by tests/golden fixtures.
"""
import math

import numpy as np
import torch

from . import flux as OF

# scheduler/scheduler_config.json of the public checkpoints (not readable offline; restated)
SCHEDULER_SCHNELL = dict(num_train_timesteps=1000, shift=1.0, use_dynamic_shifting=False,
                         base_shift=0.5, max_shift=1.15, base_image_seq_len=256, max_image_seq_len=4096)
SCHEDULER_DEV = dict(num_train_timesteps=1000, shift=3.0, use_dynamic_shifting=True,
                     base_shift=0.5, max_shift=1.15, base_image_seq_len=256, max_image_seq_len=4096)


def pack_latents(latents):
    """FluxPipeline._pack_latents: [B,C,h,w] -> [B,(h/2)(w/2),4C] (copy: train/train_qwenvl.py:229-234)."""
    B, C, h, w = latents.shape
    x = latents.view(B, C, h // 2, 2, w // 2, 2).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B, (h // 2) * (w // 2), C * 4)


def unpack_latents(latents, height, width, vae_scale_factor):
    """FluxPipeline._unpack_latents (0.31.0) (copy: lightcontrol/train_lightcontrol.py:403-410)."""
    B, _, ch = latents.shape
    h = height // vae_scale_factor
    w = width // vae_scale_factor
    x = latents.view(B, h, w, ch // 4, 2, 2).permute(0, 3, 1, 4, 2, 5)
    return x.reshape(B, ch // 4, h * 2, w * 2)


def prepare_latent_image_ids(h2, w2, dtype=torch.float32):
    """FluxPipeline._prepare_latent_image_ids(batch, h2, w2) -> [h2*w2, 3] (copy: train/train_qwenvl.py:216-227)."""
    ids = torch.zeros(h2, w2, 3)
    ids[..., 1] = ids[..., 1] + torch.arange(h2)[:, None]
    ids[..., 2] = ids[..., 2] + torch.arange(w2)[None, :]
    return ids.reshape(h2 * w2, 3).to(dtype)


def calculate_shift(image_seq_len, base_seq_len=256, max_seq_len=4096, base_shift=0.5, max_shift=1.16):
    """copy: train/train_qwenvl.py:236-246."""
    m = (max_shift - base_shift) / (max_seq_len - base_seq_len)
    b = base_shift - m * base_seq_len
    return image_seq_len * m + b


def flow_match_sigmas(num_steps, sched_cfg, image_seq_len):
    """sigmas = linspace(1, 1/N, N) -> FlowMatchEulerDiscreteScheduler.set_timesteps(sigmas=, mu=).

    Returns (timesteps fp32 [N], sigmas fp32 [N+1] with trailing 0).
    """
    sig = np.linspace(1.0, 1.0 / num_steps, num_steps)
    if sched_cfg["use_dynamic_shifting"]:
        mu = calculate_shift(image_seq_len, sched_cfg["base_image_seq_len"], sched_cfg["max_image_seq_len"],
                             sched_cfg["base_shift"], sched_cfg["max_shift"])
        sig = math.exp(mu) / (math.exp(mu) + (1.0 / sig - 1.0) ** 1.0)
    else:
        s = sched_cfg["shift"]
        sig = s * sig / (1 + (s - 1) * sig)
    sig = torch.from_numpy(np.asarray(sig)).to(torch.float32)
    timesteps = sig * sched_cfg["num_train_timesteps"]
    return timesteps, torch.cat([sig, torch.zeros(1)])


def euler_step(sample, model_output, sigma, sigma_next):
    """FlowMatchEulerDiscreteScheduler.step: fp32 add, stored back in the model dtype."""
    prev = sample.to(torch.float32) + (sigma_next - sigma) * model_output
    return prev.to(model_output.dtype)


@torch.no_grad()
def sample_latents(sd, cfg, prompt_embeds, pooled, height, width, num_steps, sched_cfg=SCHEDULER_SCHNELL,
                   guidance_scale=3.5, latents=None, generator=None, guided_hint=None, control_sds=()):
    """FluxPipeline.__call__(prompt_embeds=, pooled_prompt_embeds=, ..., output_type="latent").images

    Returns packed latents [B, (H/16)(W/16), 64] in prompt_embeds.dtype.
    """
    B = prompt_embeds.shape[0]
    dtype = prompt_embeds.dtype
    txt_ids = torch.zeros(prompt_embeds.shape[1], 3, dtype=dtype)
    C = cfg["in_channels"] // 4
    h = 2 * (int(height) // 16)
    w = 2 * (int(width) // 16)
    if latents is None:
        latents = pack_latents(torch.randn((B, C, h, w), generator=generator, dtype=dtype))
    else:
        latents = latents.to(dtype)
    img_ids = prepare_latent_image_ids(h // 2, w // 2, dtype)
    timesteps, sigmas = flow_match_sigmas(num_steps, sched_cfg, latents.shape[1])
    guidance = None
    if cfg["guidance_embeds"]:
        guidance = torch.full([1], guidance_scale, dtype=torch.float32).expand(B)
    for i, t in enumerate(timesteps):
        timestep = t.expand(B).to(latents.dtype)
        noise = OF.flux_forward(sd, cfg, latents, prompt_embeds, pooled, timestep / 1000, img_ids, txt_ids,
                                guidance=guidance, guided_hint=guided_hint, control_sds=control_sds)
        latents = euler_step(latents, noise, sigmas[i], sigmas[i + 1])
    return latents
