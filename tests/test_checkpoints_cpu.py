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


def test_pipeline_from_pretrained_is_the_reference_call(tmp_path):
    """infer/inference_qwenvl.py:72-75, byte for byte (only the import line differs), on a synthetic diffusers directory."""
    from x2i_amd.flux import FluxTransformer2DModel
    from x2i_amd.pipeline import FluxPipeline, VaeImageProcessor
    from x2i_amd.vae import AutoencoderKL
    cfg = dict(OF.DEFAULT_CFG, num_layers=1, num_single_layers=1, num_attention_heads=2, joint_attention_dim=64, pooled_projection_dim=32)
    m = FluxTransformer2DModel(**cfg, device="cpu")
    m.load_state_dict({k: v.bfloat16() for k, v in OF.random_flux_state_dict(cfg, seed=6).items()}, strict=True)
    flux_path = str(tmp_path / "shuttle")
    CK.save_transformer(m, flux_path)
    os.makedirs(os.path.join(flux_path, "scheduler"))
    json.dump(dict(_class_name="FlowMatchEulerDiscreteScheduler", num_train_timesteps=1000, shift=1.0, use_dynamic_shifting=False),
              open(os.path.join(flux_path, "scheduler", "scheduler_config.json"), "w"))
    # a VAE directory: config.json + decoder.* safetensors
    from safetensors.torch import save_file
    vae0 = AutoencoderKL(block_out_channels=(64, 64), layers_per_block=1, norm_num_groups=16, device="cpu").init_random_(1)
    os.makedirs(os.path.join(flux_path, "vae"))
    json.dump(dict(latent_channels=16, out_channels=3, block_out_channels=[64, 64], layers_per_block=1, norm_num_groups=16,
                   scaling_factor=0.3611, shift_factor=0.1159), open(os.path.join(flux_path, "vae", "config.json"), "w"))
    save_file({k: v.detach().contiguous() for k, v in vae0.state_dict().items()},
              os.path.join(flux_path, "vae", "diffusion_pytorch_model.safetensors"))
    device, dtype = "cpu", torch.bfloat16

    pipeline = FluxPipeline.from_pretrained(flux_path, text_encoder=None, text_encoder_2=None,
        tokenizer=None, tokenizer_2=None, vae=None, revision="refs/pr/1", torch_dtype=dtype).to(device)
    vae = AutoencoderKL.from_pretrained(flux_path, revision="refs/pr/1", subfolder="vae", torch_dtype=dtype).to(device)

    assert isinstance(pipeline, FluxPipeline) and pipeline.vae is None
    assert pipeline.scheduler.config.shift == 1.0 and not pipeline.scheduler.config.use_dynamic_shifting
    got = pipeline.transformer.state_dict()
    for k, v in m.state_dict().items():
        assert torch.equal(got[k], v), k
    assert 2 ** len(vae.config.block_out_channels) == 4 and vae.config.scaling_factor == 0.3611
    assert torch.equal(vae.state_dict()["decoder.conv_in.weight"], vae0.state_dict()["decoder.conv_in.weight"])
    with pytest.raises(ValueError):
        FluxPipeline.from_pretrained(flux_path, text_encoder=object())
    with pytest.raises(ValueError):
        FluxPipeline.from_pretrained(flux_path, torch_dtype=torch.float16)
    # image post-processing (:210,216)
    img = VaeImageProcessor(vae_scale_factor=16).postprocess(torch.tensor([[[[-1.0, 0.0], [1.0, 3.0]]] * 3]), output_type="pil")
    assert len(img) == 1 and img[0].size == (2, 2) and img[0].getpixel((0, 0)) == (0, 0, 0) and img[0].getpixel((1, 1)) == (255, 255, 255)
    assert img[0].getpixel((1, 0)) == (128, 128, 128)


def test_projector_mean_fusion_checkpoint(tmp_path):
    """Proj7Exp(use_scale=False, use_cnn=False): plain layer mean (utils/proj.py:70-71) -- no parameter records the layer count."""
    sd = {k: v for k, v in OP.random_proj_state_dict("internvl1b", seed=2).items() if not k.startswith(("conv.", "cha_scale"))}
    p = str(tmp_path / "mean.bin")
    torch.save(sd, p)
    with pytest.raises(KeyError):
        CK.load_projector_checkpoint(p, device="cpu")
    proj = CK.load_projector_checkpoint(p, device="cpu", in_channels=25)
    assert not proj.use_scale and not proj.use_cnn and set(proj.state_dict()) == set(sd)


def test_packed_conv_weights_follow_parameter_updates():
    """_Conv.packed() caches the [Cout,ky,kx,Cin] repack: load_state_dict / in-place updates after the first use must refresh it."""
    from x2i_amd.lightcontrol import ControlNeXtModel
    from x2i_amd.vae import AutoencoderKL
    net = ControlNeXtModel(device="cpu", control_out_channels=64)
    sd0 = {k: v.bfloat16() for k, v in OF.random_controlnext_state_dict(seed=0, out_channels=64).items()}
    sd1 = {k: v.bfloat16() for k, v in OF.random_controlnext_state_dict(seed=1, out_channels=64).items()}
    net.load_state_dict(sd0, strict=True)
    c = net.embedding[3]
    a = c.packed().clone()
    assert torch.equal(a, sd0["embedding.3.weight"].permute(0, 2, 3, 1).reshape(64, -1))
    net.load_state_dict(sd1, strict=True)
    assert torch.equal(c.packed(), sd1["embedding.3.weight"].permute(0, 2, 3, 1).reshape(64, -1))
    c.weight.mul_(2)  # in-place on the parameter itself (requires_grad=False) bumps its version counter
    assert torch.equal(c.packed(), (sd1["embedding.3.weight"] * 2).permute(0, 2, 3, 1).reshape(64, -1))
    vae = AutoencoderKL(block_out_channels=(64, 64), layers_per_block=1, norm_num_groups=16, device="cpu").init_random_(1)
    ci = vae.decoder.conv_in
    w64, _ = ci.packed(cin_pad=64)
    assert w64.shape == (64, 9 * 64)
    w16, _ = ci.packed()  # different padding request after the first call is honoured, not served from the cache
    assert w16.shape == (64, 9 * 16)
    vae.init_random_(2)
    assert torch.equal(ci.packed()[0], ci.weight.permute(0, 2, 3, 1).reshape(64, -1))
