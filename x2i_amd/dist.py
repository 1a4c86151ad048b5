"""Batch sharding of the sampler over the GPUs of one node (one process per GPU, torch.distributed; backend "nccl"
is RCCL over xGMI on ROCm, "gloo" in CPU tests).

The sampling path has no cross-sample operation, so ranks never exchange data inside the
denoising loop: rank r owns samples [lo_r, hi_r), every rank holds a full weight replica, and the ONLY collective is
one all-gather of the final packed latents [B/N, S_img, 64] bf16 (512 KiB per image at 1024^2) -- latency-bound on
xGMI, so it is issued as a single fused all_gather rather than a ring of chunks.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, device=None):
    """torchrun-style init (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_range(n, rank, world):
    """Contiguous split of n samples; the first n % world ranks get one extra."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(t, rank, world, dim=0):
    lo, hi = shard_range(t.shape[dim], rank, world)
    return t.narrow(dim, lo, hi - lo)


def all_gather_batch(local, total, group=None):
    """Gather per-rank batch shards (possibly uneven, sizes from shard_range) into the full [total, ...] tensor on every
    rank with ONE all_gather (shards are padded to the largest shard)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [shard_range(total, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))], 0)
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous(), group=group)
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], 0)


def sample_sharded(pipeline, prompt_embeds, pooled_prompt_embeds, latents=None, guided_hint=None, group=None, **kw):
    """Run `pipeline(...)` on this rank's slice of the batch and return the FULL batch of packed latents on every rank.
    All ranks pass the same global tensors (or at least tensors of the global batch size)."""
    if not dist.is_initialized():
        return pipeline(prompt_embeds=prompt_embeds, pooled_prompt_embeds=pooled_prompt_embeds, latents=latents,
                        guided_hint=guided_hint, **kw).images
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    total = prompt_embeds.shape[0]
    lo, hi = shard_range(total, rank, world)
    if latents is None:
        # draw the GLOBAL noise on every rank (same generator state everywhere) and slice it: forwarding `generator=` to the
        # local call would give sample j of every rank the same noise and a result that differs from the unsharded run
        gen = kw.pop("generator", None)
        tr = pipeline.transformer
        latents, _ = pipeline.prepare_latents(total, tr.config.in_channels // 4, kw.get("height") or 1024, kw.get("width") or 1024,
                                              prompt_embeds.dtype, tr.device, gen)
    else:
        kw.pop("generator", None)  # unused once latents are given (as in the pipeline itself)
    if hi > lo:
        local = pipeline(prompt_embeds=prompt_embeds[lo:hi], pooled_prompt_embeds=pooled_prompt_embeds[lo:hi],
                         latents=None if latents is None else latents[lo:hi],
                         guided_hint=None if guided_hint is None else guided_hint[lo:hi], **kw).images
    else:  # more ranks than samples: contribute an empty shard
        local = latents.new_zeros((0,) + tuple(latents.shape[1:]))
    return all_gather_batch(local, total, group)
