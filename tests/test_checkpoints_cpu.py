"""N3: the on-disk formats round-trip through our loaders (key names / shapes as the reference writes them)."""
import json
import os

import pytest
import torch

from oracle import flux as OF
from oracle import projector as OP
from x2i_amd import checkpoints as CK


@pytest.mark.parametrize("kind,prefix,comfy", [("qwen3b", "", False), ("internvl1b", "module.", False), ("qwen7b", "", True)])
def test_projector_formats(tmp_path, kind, prefix, comfy):
    sd = OP.random_proj_state_dict(kind, seed=1)
    path = str(tmp_path / "diffusion_pytorch_model.bin")
    if comfy:  # x2i_comfyui/model.py:90-97
        cfg = dict(in_channels=29, kernel_size=5, input_dim=3584, output_dim0=768, output_dim1=4096, num_layers=2, num_heads=28,
                   norm_eps=1e-6, head_dim=128, use_t5=False, use_scale=False, use_cnn=True)
        torch.save({"config": cfg, "state_dict": sd}, path)
    else:
        torch.save({prefix + k: v for k, v in sd.items()}, path)
    proj = CK.load_projector_checkpoint(path, device="cpu")
    got = proj.state_dict()
    assert set(got) == set(sd)
    for k in sd:
        assert torch.equal(got[k], sd[k].bfloat16()), k
    assert proj.use_scale == (kind == "internvl1b") and not proj.training


def test_projector_rejects_t5_checkpoints(tmp_path):
    sd = OP.random_proj_state_dict("qwen3b", seed=1)
    sd["t5stack.block.0.layer.0.SelfAttention.q.weight"] = torch.zeros(4, 4)
    p = str(tmp_path / "p.bin")
    torch.save(sd, p)
    with pytest.raises(NotImplementedError):
        CK.load_projector_checkpoint(p, device="cpu")


def test_control_net_modulelist_format(tmp_path):
    sds = [OF.random_controlnext_state_dict(seed=i, out_channels=256) for i in range(3)]
    flat = {"module.%d.%s" % (i, k): v for i, sd in enumerate(sds) for k, v in sd.items()}  # DeepSpeed/DDP-wrapped ModuleList
    p = str(tmp_path / "diffusion_pytorch_model.bin")
    torch.save(flat, p)
    nets = CK.load_control_nets(p, device="cpu")
    assert len(nets) == 3
    for n, sd in zip(nets, sds):
        got = n.state_dict()
        assert set(got) == set(sd)
        assert torch.equal(got["mid_convs.1.weight"], sd["mid_convs.1.weight"].bfloat16())
    CK.save_control_nets(nets, str(tmp_path / "again.bin"))
    again = CK.load_control_nets(str(tmp_path / "again.bin"), device="cpu")
    assert torch.equal(again[2].state_dict()["embedding.0.weight"], nets[2].state_dict()["embedding.0.weight"])


def test_transformer_directory_roundtrip(tmp_path):
    from x2i_amd.flux import FluxTransformer2DModel
    cfg = dict(OF.DEFAULT_CFG, num_layers=1, num_single_layers=2, num_attention_heads=2, joint_attention_dim=64,
               pooled_projection_dim=32, guidance_embeds=True)
    sd = {k: v.bfloat16() for k, v in OF.random_flux_state_dict(cfg, seed=4).items()}
    m = FluxTransformer2DModel(**cfg, device="cpu")
    m.load_state_dict(sd, strict=True)
    root = str(tmp_path / "flux")
    CK.save_transformer(m, root, max_shard_bytes=1 << 20)  # force several shards
    os.makedirs(os.path.join(root, "scheduler"))
    json.dump(dict(OF_sched := dict(num_train_timesteps=1000, shift=3.0, use_dynamic_shifting=True, base_shift=0.5, max_shift=1.15,
                                    base_image_seq_len=256, max_image_seq_len=4096, _class_name="FlowMatchEulerDiscreteScheduler")),
              open(os.path.join(root, "scheduler", "scheduler_config.json"), "w"))
    assert len([f for f in os.listdir(os.path.join(root, "transformer")) if f.endswith(".safetensors")]) > 1
    tr, sched = CK.load_pipeline_dir(root, device="cpu")
    assert tr.config.guidance_embeds and tr.config.num_single_layers == 2
    assert sched.config.use_dynamic_shifting and sched.config.shift == 3.0
    got = tr.state_dict()
    for k in sd:
        assert torch.equal(got[k], sd[k]), k
    # a missing or an extra key is an error, not a silent partial load
    os.remove(sorted(p for p in (os.path.join(root, "transformer", f) for f in os.listdir(os.path.join(root, "transformer")))
                     if p.endswith(".safetensors"))[0])
    with pytest.raises(KeyError):
        FluxTransformer2DModel.from_pretrained(root, subfolder="transformer", device="cpu")
