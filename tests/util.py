"""Shared helpers for the test-suite (fixture loading, error metrics)."""
import json
import os

import torch
from safetensors.torch import load_file

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))


def golden(name):
    return load_file(os.path.join(GOLDEN, name + ".safetensors")), MANIFEST[name]


def seeded(shape, seed, scale=1.0):
    return scale * torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def rel_l2(a, b):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_abs(a, b):
    return float((a.detach().float().cpu() - b.detach().float().cpu()).abs().max())


def span_permute(VT):
    """V^T [..., Spad] in the "span-permuted" key order of x2i_qkv_desc.vt_perm (include/x2i.h): within every 32-key span, position kk holds
    key 16 ((kk >> 2) & 1) + 4 (kk >> 3) + (kk & 3)."""
    perm = torch.tensor([16 * ((kk >> 2) & 1) + 4 * (kk >> 3) + (kk & 3) for kk in range(32)], device=VT.device)
    return VT.reshape(*VT.shape[:-1], VT.shape[-1] // 32, 32)[..., perm].reshape(VT.shape).contiguous()
