"""The oracle replayed against fixtures produced by the reference's own code
(tests/golden/make_golden.py).  CPU only.  Tolerance: fp32 round-off (the
oracle and the reference run the same torch ops, possibly in another order)."""
import pytest
import torch

from oracle import flux as OF
from oracle import projector as OP
from oracle import sampler as OS
from oracle import primitives as P
from tests.util import golden, seeded, rel_l2

TOL = 2e-5


@pytest.mark.parametrize("name", ["proj_qwen3b", "proj_qwen7b", "proj_internvl1b", "proj_internvl4b", "proj_minicpm",
                                  "proj_internvl1b_mean"])
def test_projector_matches_reference(name):
    t, meta = golden(name)
    sd = OP.random_proj_state_dict(meta["kind"], seed=meta["weight_seed"],
                                   use_scale=True if "drop" in meta else None)
    for k in meta.get("drop", []):
        sd.pop(k)
    x = seeded(meta["input_shape"], meta["input_seed"], meta["input_scale"])
    x1, x2 = OP.proj7exp(sd, x)
    assert x1.shape == t["x1"].shape and x2.shape == t["x2"].shape
    assert rel_l2(x1, t["x1"]) < TOL
    assert rel_l2(x2, t["x2"]) < TOL


@pytest.mark.parametrize("cls", ["MLP", "MLP2", "MLP_plus"])
def test_legacy_mlp_matches_reference(cls):
    t, meta = golden("legacy_" + cls)
    sd = {k[3:]: v for k, v in t.items() if k.startswith("sd.")}
    x1, x2 = OP.legacy_mlp(sd, t["x"], eps=meta["eps"])
    assert rel_l2(x1, t["x1"]) < TOL and rel_l2(x2, t["x2"]) < TOL


def test_legacy_proj_front_stage():
    t, meta = golden("legacy_Proj_pre")
    sd = {k[3:]: v for k, v in t.items() if k.startswith("sd.")}
    assert rel_l2(OP.legacy_proj_pre(sd, t["x"], meta["eps"]), t["pre"]) < TOL


def test_pipeline_helpers_match_reference_copies():
    t, meta = golden("helpers")
    assert torch.equal(OS.pack_latents(t["lat"]), t["packed"])
    assert torch.equal(OS.unpack_latents(t["packed"], 64, 96, 16), t["unpacked"])
    assert torch.equal(OS.unpack_latents(OS.pack_latents(t["lat"]), 64, 96, 16), t["lat"])
    assert torch.equal(OS.prepare_latent_image_ids(4, 6), t["ids"])  # ref copy halves (8,12) internally
    got = torch.tensor([OS.calculate_shift(n) for n in meta["seq_lens"]], dtype=torch.float64)
    assert torch.allclose(got, t["shifts"], rtol=0, atol=1e-12)
    got = torch.tensor([OS.calculate_shift(n, 256, 4096, 0.5, 1.15) for n in meta["seq_lens"]], dtype=torch.float64)
    assert torch.allclose(got, t["shifts115"], rtol=0, atol=1e-12)


def test_flux_tiny_schnell_composition():
    t, meta = golden("flux_tiny_schnell")
    cfg = meta["cfg"]
    sd = OF.random_flux_state_dict(cfg, seed=meta["weight_seed"], std=meta["weight_std"])
    out = OF.flux_forward(sd, cfg, t["hidden"], t["enc"], t["pooled"], t["timestep"], t["img_ids"], t["txt_ids"])
    assert rel_l2(out, t["out"]) < TOL


def test_flux_tiny_dev_lightcontrol_composition():
    t, meta = golden("flux_tiny_dev_control")
    cfg = meta["cfg"]
    sd = OF.random_flux_state_dict(cfg, seed=meta["weight_seed"], std=meta["weight_std"])
    csds = [OF.random_controlnext_state_dict(seed=s, out_channels=meta["control_out_channels"])
            for s in meta["control_seeds"]]
    out = OF.flux_forward(sd, cfg, t["hidden"], t["enc"], t["pooled"], t["timestep"], t["img_ids"], t["txt_ids"],
                          guidance=t["guidance"], guided_hint=t["hint"], control_sds=csds)
    assert rel_l2(out, t["out"]) < TOL
    # the control branch must matter (guards against a silently skipped injection)
    out0 = OF.flux_forward(sd, cfg, t["hidden"], t["enc"], t["pooled"], t["timestep"], t["img_ids"], t["txt_ids"],
                           guidance=t["guidance"])
    assert rel_l2(out0, t["out"]) > 1e-3


def test_full_width_blocks():
    t, meta = golden("flux_full_width_blocks")
    cfg = dict(OF.DEFAULT_CFG)
    cfg.update(num_layers=1, num_single_layers=1)
    sd = OF.random_flux_state_dict(cfg, seed=meta["weight_seed"], std=meta["weight_std"])
    ids = torch.cat([torch.zeros(meta["St"], 3), OS.prepare_latent_image_ids(meta["h2"], meta["w2"])], 0)
    rotary = P.flux_pos_embed(ids)
    enc, hid = OF.double_block(sd, "transformer_blocks.0", t["hidden"], t["enc"], t["temb"], rotary, 24)
    assert rel_l2(enc, t["enc_out"]) < TOL and rel_l2(hid, t["hidden_out"]) < TOL
    joint = torch.cat([t["enc"], t["hidden"]], 1)
    sgl = OF.single_block(sd, "single_transformer_blocks.0", joint, t["temb"], rotary, 24)
    assert rel_l2(sgl, t["single_out"]) < TOL


def test_controlnext_full():
    t, meta = golden("controlnext_full")
    sd = OF.random_controlnext_state_dict(seed=meta["weight_seed"])
    o = OF.controlnext_forward(sd, "", t["hint"], t["timestep"])
    assert o["scale"] == meta["scale"]
    assert rel_l2(o["out"], t["out"]) < TOL


@pytest.mark.parametrize("cls", ["Proj", "Proj2", "Proj3"])
def test_legacy_proj_full_forward_matches_reference(cls):
    """model_internvl/proj.py:149-211 run by the reference itself (T5Stack as installed: transformers 5.15) vs the oracle wiring."""
    t, meta = golden("legacy_%s_full" % cls)
    sd = {k[3:]: v.float() for k, v in t.items() if k.startswith("sd.")}
    cfg = meta["cfg"]
    sd["t5stack.embed_tokens.weight"] = torch.zeros(32128, cfg["input_dim"])  # unused with inputs_embeds=; not stored in the fixture
    x1, x2 = OP.legacy_proj(sd, t["x"], cfg, t5_first=(cls == "Proj3"))
    assert rel_l2(x2, t["x2"]) < TOL and rel_l2(x1, t["x1"]) < TOL


def test_legacy_transformer_proj_matches_reference():
    t, meta = golden("legacy_Transformer_proj")
    sd = {k[3:]: v.float() for k, v in t.items() if k.startswith("sd.")}
    x1, x2 = OP.transformer_proj(sd, t["x"], meta["d_model"], meta["n_heads"], meta["num_layers"])
    assert rel_l2(x2, t["x2"]) < TOL and rel_l2(x1, t["x1"]) < TOL


@pytest.mark.parametrize("cls", ["Proj", "Proj2", "Proj3"])
def test_legacy_proj_classes_mirror_the_reference_parameter_tree(cls):
    """x2i_amd.proj.Proj / Proj2 / Proj3: same constructor arguments and state-dict keys as the reference classes (CPU, no kernels)."""
    import x2i_amd.proj as XP
    t, meta = golden("legacy_%s_full" % cls)
    want = {k[3:]: tuple(v.shape) for k, v in t.items() if k.startswith("sd.")}
    want["t5stack.embed_tokens.weight"] = (32128, meta["cfg"]["input_dim"])
    m = getattr(XP, cls)(device="cpu", **meta["cfg"])
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == want
    m2 = XP.Transformer_proj(64, 2, 32, 48, num_layers=2, device="cpu")
    t2, _ = golden("legacy_Transformer_proj")
    assert {k: tuple(v.shape) for k, v in m2.state_dict().items()} == {k[3:]: tuple(v.shape) for k, v in t2.items() if k.startswith("sd.")}


def test_distillation_loss_restatement_vs_reference_statements():
    """x2i_amd.distill.kd_attention_loss (the torch restatement the GPU parity tests check the HIP kernel against) reproduces the loss value
    and the autograd gradients obtained by executing the reference's own statements (train/train_qwenvl.py normalize + the two kl_div
    loops; tests/golden/make_golden.py gen_distill)."""
    import os
    from safetensors.torch import load_file
    from x2i_amd.distill import kd_attention_loss
    g = load_file(os.path.join(os.path.dirname(__file__), "golden", "distill_loss.safetensors"))
    st = [g[f"student{i}"].clone().requires_grad_(True) for i in range(3)]
    loss = kd_attention_loss([g[f"teacher{i}"] for i in range(3)], st, temperature=3.0)
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    for i in range(3):
        assert torch.allclose(st[i].grad, g[f"grad{i}"], rtol=1e-4, atol=1e-7), i
