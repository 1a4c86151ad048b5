"""On-disk formats of the hot path's weights, validated offline on key names and shapes:

  * projector: flat `torch.save(state_dict)` "diffusion_pytorch_model.bin", keys optionally prefixed "module."
    (train/train_qwenvl.py:641-647; loader infer/inference_qwenvl.py:84-91), or the ComfyUI packaging
    {"config": {...Proj7Exp kwargs...}, "state_dict": {...}} (x2i_comfyui/model.py:33-39,90-97)
  * FLUX transformer: diffusers directory (config.json + *.safetensors shards, key names SURVEY.md Appendix B)
  * scheduler: scheduler/scheduler_config.json (never hard-coded: shuttle-3 vs schnell vs dev differ, Appendix E)
  * ControlNeXt: ONE state dict saved from an nn.ModuleList of 19 models -> keys "{i}.<name>", possibly "module."-prefixed
    (lightcontrol/train_lightcontrol.py:517-522,785-791)
"""
import json
import os
import re

import torch

from . import proj as xproj
from .flux import FluxTransformer2DModel
from .pipeline import FlowMatchEulerDiscreteScheduler

_PROJ_KW = ("in_channels", "kernel_size", "input_dim", "output_dim0", "output_dim1", "num_layers", "num_heads", "norm_eps",
            "head_dim", "use_t5", "use_scale", "use_cnn")


def _strip(sd):
    return {re.sub(r"^(module\.)+", "", k): v for k, v in sd.items()}


def projector_config_from_state_dict(sd, in_channels=None):
    """Recover the Proj7Exp constructor arguments from tensor shapes (the flat .bin carries no config).  The plain layer-mean
    fusion (use_scale=False, use_cnn=False; utils/proj.py:70-71) has no parameter that records the layer count: pass
    `in_channels` (it only documents the expected C; the mean accepts any)."""
    H = sd["mlp.layernorm.weight"].shape[0]
    cfg = dict(kernel_size=5, input_dim=H, output_dim0=sd["mlp.fc.1.weight"].shape[0],
               output_dim1=sd["mlp.projector.2.weight"].shape[0], norm_eps=1e-6, use_t5=False)
    if "cha_scale" in sd:
        cfg.update(in_channels=sd["cha_scale"].shape[1], use_scale=True, use_cnn=False)
    elif "conv.weight" in sd:
        cfg.update(in_channels=sd["conv.weight"].shape[1], use_scale=False, use_cnn=True, kernel_size=sd["conv.weight"].shape[-1])
    elif in_channels is not None:
        cfg.update(in_channels=int(in_channels), use_scale=False, use_cnn=False)
    else:
        raise KeyError("projector checkpoint has neither conv.weight nor cha_scale (plain layer-mean fusion): pass in_channels=")
    if any(k.startswith("t5stack.") for k in sd):
        raise NotImplementedError("checkpoint contains a T5Stack (use_t5=True): dead path in the reference, unsupported here")
    return cfg


def load_projector_checkpoint(path, device="cuda", in_channels=None):
    """-> x2i_amd.proj.Proj7Exp in eval mode, whichever of the two packagings `path` holds."""
    blob = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(blob, dict) and "state_dict" in blob and "config" in blob:
        sd = _strip(blob["state_dict"])
        cfg = {k: blob["config"][k] for k in _PROJ_KW if k in blob["config"]}
    else:
        sd = _strip(blob)
        cfg = projector_config_from_state_dict(sd, in_channels)
    proj = xproj.Proj7Exp(device=device, **cfg)
    proj.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()}, strict=True)
    return proj.eval()


def load_projector_state_dict(proj, path):
    """Load a projector `.bin` (optionally `module.`-prefixed / ComfyUI-wrapped) into an existing module, strictly (resume path of the
    distillation harness; train/train_qwenvl.py:404-409)."""
    sd = torch.load(path, map_location="cpu")
    if isinstance(sd, dict) and "state_dict" in sd and "config" in sd:
        sd = sd["state_dict"]
    proj.load_state_dict(_strip(sd), strict=True)
    return proj


def save_projector_checkpoint(proj, path, comfyui=False, config=None):
    sd = {k: v.detach().cpu() for k, v in proj.state_dict().items()}
    torch.save({"config": config, "state_dict": sd} if comfyui else sd, path)


def load_control_nets(path, device="cuda"):
    """-> list of ControlNeXtModel from a ModuleList state dict (keys "{i}.<name>")."""
    from .lightcontrol import ControlNeXtModel
    sd = _strip(torch.load(path, map_location="cpu", weights_only=True))
    groups = {}
    for k, v in sd.items():
        m = re.match(r"^(\d+)\.(.+)$", k)
        if not m:
            raise KeyError("unexpected ControlNeXt key %r (expected '<index>.<name>')" % k)
        groups.setdefault(int(m.group(1)), {})[m.group(2)] = v
    nets = []
    for i in range(len(groups)):
        if i not in groups:
            raise KeyError("ControlNeXt checkpoint is missing model %d" % i)
        out_ch = groups[i]["mid_convs.1.weight"].shape[0]
        net = ControlNeXtModel(device=device, control_out_channels=out_ch)
        net.load_state_dict({k: v.to(torch.bfloat16) for k, v in groups[i].items()}, strict=True)
        nets.append(net)
    return nets


def save_control_nets(nets, path):
    sd = {}
    for i, n in enumerate(nets):
        for k, v in n.state_dict().items():
            sd["%d.%s" % (i, k)] = v.detach().cpu()
    torch.save(sd, path)


def save_transformer(model, path, subfolder="transformer", max_shard_bytes=2 << 30):
    """Write config.json + safetensors shards with the diffusers key names (for round-trip tests and synthetic runs)."""
    from safetensors.torch import save_file
    d = os.path.join(path, subfolder) if subfolder else path
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "config.json"), "w") as fh:
        cfg = dict(model.config)
        cfg["axes_dims_rope"] = list(cfg["axes_dims_rope"])
        cfg["_class_name"] = "FluxTransformer2DModel"
        json.dump(cfg, fh, indent=1)
    shard, size, idx = {}, 0, 1
    items = list(model.state_dict().items())
    for n, (k, v) in enumerate(items):
        t = v.detach().cpu().contiguous()
        shard[k] = t
        size += t.numel() * t.element_size()
        if size >= max_shard_bytes or n == len(items) - 1:
            save_file(shard, os.path.join(d, "diffusion_pytorch_model-%05d.safetensors" % idx))
            shard, size, idx = {}, 0, idx + 1


def load_pipeline_dir(path, device="cuda"):
    """(FluxTransformer2DModel, FlowMatchEulerDiscreteScheduler) from a diffusers pipeline directory."""
    tr = FluxTransformer2DModel.from_pretrained(path, subfolder="transformer", device=device)
    with open(os.path.join(path, "scheduler", "scheduler_config.json")) as fh:
        sched = FlowMatchEulerDiscreteScheduler.from_config(json.load(fh))
    return tr, sched
