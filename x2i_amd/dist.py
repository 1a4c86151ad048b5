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


def init_from_env(backend=None, device=None, timeout_hours=24.0):
    """torchrun-style init (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  Returns (rank, world).  The collective timeout is long on
    purpose: ranks with nothing to do (the teacher ranks of a synthetic training run) wait in barrier_and_destroy() for the WHOLE run,
    and the default watchdog (10 min RCCL / 30 min gloo) would abort them -- and with them the job."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        import datetime
        kw = {"timeout": datetime.timedelta(hours=timeout_hours)}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def barrier_and_destroy():
    """Leave the job together: world barrier, then destroy_process_group on every rank (a rank that returns early -- e.g. a teacher rank
    with nothing to do -- may host the rendezvous store the others still need for their first collective)."""
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


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


# ---------------------------------------------------------------------------------------------------------------------
# Training ranks / teacher ranks (row N4; reference core/pipeline/train_and_infer.py:31-122).  The reference dedicates the first rank
# of every `infer group` of a node to the frozen teacher pipeline: the group's other ranks (the trainers) send it their prompts, it runs
# the teacher and hands every trainer its slice of the teacher tensors; the trainers form the data-parallel group of the projector.
# On MI355X the 8 GPUs of a node are fully connected point to point, so both exchanges are issued as direct sends / receives between the
# teacher rank and each trainer (one xGMI link each, all in flight together) instead of a rooted collective.
class TeacherStudentGroups:
    """Rank layout of one job: `local_infer_world_size` teacher ranks per node, each serving a contiguous group of
    ceil(local_world_size / local_infer_world_size) ranks whose first member is the teacher.  Attributes: infer_ranks (this rank's group),
    infer_rank (its teacher), is_infer_rank, train_ranks (all trainers of the job), infer_pg / train_pg (process groups; every rank must
    construct this object, as with any dist.new_group)."""

    def __init__(self, rank, world_size, local_world_size, local_infer_world_size=1, backend=None):
        if world_size % local_world_size:
            raise ValueError("world_size must be a multiple of local_world_size")
        per = -(-local_world_size // local_infer_world_size)
        if per < 2:
            raise ValueError("every teacher rank needs at least one trainer: ceil(local_world_size / local_infer_world_size) must be > 1")
        self.rank, self.world_size = rank, world_size
        groups = []
        for node in range(world_size // local_world_size):
            ranks = list(range(node * local_world_size, (node + 1) * local_world_size))
            for g in range(local_infer_world_size):
                grp = ranks[g * per:(g + 1) * per]
                if len(grp) < 2:
                    raise ValueError(f"infer group {grp} has no trainer (local_world_size={local_world_size}, local_infer_world_size={local_infer_world_size})")
                groups.append(grp)
        self.groups = groups
        teachers = [g[0] for g in groups]
        self.train_ranks = sorted(set(range(world_size)) - set(teachers))
        self.infer_pg, self.infer_ranks = None, None
        for grp in groups:   # every rank creates every group, in the same order
            pg = dist.new_group(ranks=grp, backend=backend)
            if rank in grp:
                self.infer_pg, self.infer_ranks = pg, grp
        self.train_pg = dist.new_group(ranks=self.train_ranks, backend=backend)
        self.infer_rank = self.infer_ranks[0]
        self.is_infer_rank = rank == self.infer_rank

    @property
    def trainers(self):
        return self.infer_ranks[1:]


def _wait_all(reqs):
    for r in reqs:
        r.wait()


def send_to_infer_device(data, groups):
    """Trainers -> teacher (reference send_to_infer_device): every trainer passes its tensor; the teacher passes a tensor of the per-rank
    shape (only shape / dtype / device are used) and receives the trainers' tensors concatenated along dim 0 in rank order; trainers get
    None."""
    if groups.is_infer_rank:
        n = len(groups.trainers)
        out = torch.empty((n * data.shape[0],) + tuple(data.shape[1:]), dtype=data.dtype, device=data.device)
        _wait_all([dist.irecv(c, src=r) for c, r in zip(out.chunk(n, dim=0), groups.trainers)])
        return out
    dist.isend(data.contiguous(), dst=groups.infer_rank).wait()
    return None


def receive_from_infer_device(data, groups):
    """Teacher -> trainers (reference receive_from_infer_device): the teacher passes the concatenation (dim 0, trainer order) of what each
    trainer is to get; a trainer passes a tensor of its slice's shape to receive into.  Returns the trainer's slice (None on the teacher)."""
    if groups.is_infer_rank:
        n = len(groups.trainers)
        if data.shape[0] % n:
            raise ValueError(f"receive_from_infer_device: dim 0 ({data.shape[0]}) is not a multiple of the {n} trainers")
        _wait_all([dist.isend(c.contiguous(), dst=r) for c, r in zip(data.chunk(n, dim=0), groups.trainers)])
        return None
    dist.irecv(data, src=groups.infer_rank).wait()
    return data
