#!/usr/bin/env python3
"""Can RCCL run two ranks on ONE device (a one-GPU box is all this environment offers)?  torchrun --nproc-per-node 2 this file."""
import datetime
import os
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
try:
    dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=60))
    x = torch.full((4, 8), float(rank), device=dev, dtype=torch.bfloat16)
    out = [torch.empty_like(x) for _ in range(dist.get_world_size())]
    dist.all_gather(out, x)
    torch.cuda.synchronize()
    print(f"rank {rank}: all_gather over RCCL with {dist.get_world_size()} ranks on one device ok:", [float(o[0, 0]) for o in out], flush=True)
    dist.barrier()
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    print(f"rank {rank}: RCCL refused two ranks on one device: {type(e).__name__}: {str(e)[:300]}", flush=True)
