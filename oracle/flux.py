"""Oracle restatement of the FLUX DiT composition + ControlNeXt hint encoder.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PINNED: the functions here are
replayed against fixtures produced by the reference's own
lightcontrol/lightcontrol_flux.py (imported under a primitive shim by
tests/golden/make_golden.py).

Functional form over a flat diffusers-format state dict; `cfg` is a dict with
the constructor arguments of the reference model (lightcontrol_flux.py:230-242).
"""
import torch

from . import primitives as P

DEFAULT_CFG = dict(
    patch_size=1,
    in_channels=64,
    num_layers=19,
    num_single_layers=38,
    attention_head_dim=128,
    num_attention_heads=24,
    joint_attention_dim=4096,
    pooled_projection_dim=768,
    guidance_embeds=False,
    axes_dims_rope=(16, 56, 56),
)


def double_block(sd, prefix, hidden, enc, temb, rotary, heads, taps=None):
    """FluxTransformerBlock.forward -- lightcontrol_flux.py:159-204.  `taps`: optional [img_list, txt_list, single_list]; the attention
    module's outputs are appended as a forward hook on `block.attn` would see them (train/train_qwenvl.py:206-214)."""
    n_h, gate_msa, shift_mlp, scale_mlp, gate_mlp = P.ada_layer_norm_zero(sd, prefix + ".norm1", hidden, temb)  # :166
    n_e, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = P.ada_layer_norm_zero(
        sd, prefix + ".norm1_context", enc, temb
    )  # :168-170
    attn_img, attn_txt = P.flux_attention(sd, prefix + ".attn", n_h, heads, rotary, encoder_hidden=n_e)  # :173-177
    if taps is not None:
        taps[0].append(attn_img)
        taps[1].append(attn_txt)
    hidden = hidden + gate_msa.unsqueeze(1) * attn_img  # :180-181
    n_h = P.layer_norm_plain(hidden) * (1 + scale_mlp[:, None]) + shift_mlp[:, None]  # :183-184
    hidden = hidden + gate_mlp.unsqueeze(1) * P.feed_forward(sd, prefix + ".ff", n_h)  # :186-189
    enc = enc + c_gate_msa.unsqueeze(1) * attn_txt  # :193-194
    n_e = P.layer_norm_plain(enc) * (1 + c_scale_mlp[:, None]) + c_shift_mlp[:, None]  # :196-197
    enc = enc + c_gate_mlp.unsqueeze(1) * P.feed_forward(sd, prefix + ".ff_context", n_e)  # :199-200
    if enc.dtype == torch.float16:
        enc = enc.clip(-65504, 65504)  # :201-202
    return enc, hidden  # :204


def single_block(sd, prefix, hidden, temb, rotary, heads, taps=None):
    """FluxSingleTransformerBlock.forward -- lightcontrol_flux.py:82-104."""
    residual = hidden
    n_h, gate = P.ada_layer_norm_zero_single(sd, prefix + ".norm", hidden, temb)  # :89
    mlp = torch.nn.functional.gelu(P.linear(sd, prefix + ".proj_mlp", n_h), approximate="tanh")  # :90
    attn = P.flux_attention(sd, prefix + ".attn", n_h, heads, rotary)  # :92-95
    if taps is not None:
        taps[2].append(attn)
    cat = torch.cat([attn, mlp], dim=2)  # :97
    hidden = residual + gate.unsqueeze(1) * P.linear(sd, prefix + ".proj_out", cat)  # :98-100
    if hidden.dtype == torch.float16:
        hidden = hidden.clip(-65504, 65504)
    return hidden


def controlnext_forward(sd, prefix, sample, timestep):
    """ControlNeXtModel.forward -- lightcontrol_flux.py:708-749 (modules :590-668).

    `timestep` is the already x1000-rescaled value the transformer passes (:505).
    Returns {"out": [B,3072,H/16,W/16], "scale": 1.0}.
    """
    p = prefix
    B = sample.shape[0]
    t = timestep
    if not torch.is_tensor(t):
        t = torch.tensor([t], dtype=torch.float64 if isinstance(t, float) else torch.int64)
    elif t.dim() == 0:
        t = t[None]
    t = t.expand(B)
    t_emb = P.timesteps_proj(t, 128).to(sample.dtype)  # :730-733
    emb = P.timestep_embedding(sd, p + "time_embedding", t_emb)  # :735
    # embedding (:593-603): conv s2, GN(2), ReLU, conv, GN(2), ReLU, conv, GN(2), ReLU
    x = P.conv2d(sd, p + "embedding.0", sample, stride=2, padding=1)
    x = torch.relu(P.group_norm(sd, p + "embedding.1", x, 2, 1e-5))
    x = P.conv2d(sd, p + "embedding.3", x, padding=1)
    x = torch.relu(P.group_norm(sd, p + "embedding.4", x, 2, 1e-5))
    x = P.conv2d(sd, p + "embedding.6", x, padding=1)
    x = torch.relu(P.group_norm(sd, p + "embedding.7", x, 2, 1e-5))
    groups = (4, 8)
    for i in range(2):  # :741-743
        x = P.resnet_block2d(sd, p + f"down_res.{i}", x, emb, groups[i])
        x = P.downsample2d(sd, p + f"down_sample.{i}", x)
    # mid_convs[0] (:632-653): conv, ReLU, GN(8), conv, GN(8); residual add (:744)
    h = P.conv2d(sd, p + "mid_convs.0.0", x, padding=1)
    h = torch.relu(h)
    h = P.group_norm(sd, p + "mid_convs.0.2", h, 8, 1e-5)
    h = P.conv2d(sd, p + "mid_convs.0.3", h, padding=1)
    h = P.group_norm(sd, p + "mid_convs.0.4", h, 8, 1e-5)
    x = h + x
    x = P.conv2d(sd, p + "mid_convs.1", x, stride=2)  # :745  k=2, s=2
    return {"out": x, "scale": 1.0}


def flux_forward(
    sd,
    cfg,
    hidden_states,
    encoder_hidden_states,
    pooled_projections,
    timestep,
    img_ids,
    txt_ids,
    guidance=None,
    guided_hint=None,
    control_sds=(),
    taps=None,
):
    """FluxTransformer2DModel.forward -- lightcontrol_flux.py:390-553.

    `control_sds`: sequence of ControlNeXt state dicts (one per leading double
    block), the functional stand-in for `control_nets` (:401, :504-507).
    Returns the bare tensor [B, S_img, in_channels] (what return_dict=False gives, :549-550).
    """
    heads = cfg["num_attention_heads"]
    hidden = P.linear(sd, "x_embedder", hidden_states)  # :445
    timestep = timestep.to(hidden.dtype) * 1000  # :447
    if guidance is not None:
        guidance = guidance.to(hidden.dtype) * 1000  # :449
    temb = P.combined_time_text_embed(sd, "time_text_embed", timestep, pooled_projections, guidance)  # :452-456
    enc = P.linear(sd, "context_embedder", encoder_hidden_states)  # :457
    if txt_ids.ndim == 3:
        txt_ids = txt_ids[0]
    if img_ids.ndim == 3:
        img_ids = img_ids[0]
    ids = torch.cat((txt_ids, img_ids), dim=0)  # :471
    rotary = P.flux_pos_embed(ids, tuple(cfg["axes_dims_rope"]))  # :472
    for i in range(cfg["num_layers"]):  # :474-507
        enc, hidden = double_block(sd, f"transformer_blocks.{i}", hidden, enc, temb, rotary, heads, taps)
        if i < len(control_sds):
            control = controlnext_forward(control_sds[i], "", guided_hint, timestep)  # :505
            out = control["out"].flatten(2).transpose(1, 2).to(hidden.dtype)  # :506
            hidden = hidden + out * control["scale"]  # :507
    n_txt = enc.shape[1]
    hidden = torch.cat([enc, hidden], dim=1)  # :510
    for i in range(cfg["num_single_layers"]):  # :512-538
        hidden = single_block(sd, f"single_transformer_blocks.{i}", hidden, temb, rotary, heads, taps)
    hidden = hidden[:, n_txt:, ...]  # :540
    hidden = P.ada_layer_norm_continuous(sd, "norm_out", hidden, temb)  # :542
    return P.linear(sd, "proj_out", hidden)  # :543


# ----------------------------------------------------------------------------
# random-weight state dicts (shapes: SURVEY.md Appendix B; lightcontrol_flux.py:243-282, :590-668)
# ----------------------------------------------------------------------------
def flux_param_shapes(cfg):
    D = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    hd = cfg["attention_head_dim"]
    shapes = {}

    def lin(name, out_f, in_f, bias=True):
        shapes[name + ".weight"] = (out_f, in_f)
        if bias:
            shapes[name + ".bias"] = (out_f,)

    lin("x_embedder", D, cfg["in_channels"])
    lin("context_embedder", D, cfg["joint_attention_dim"])
    lin("time_text_embed.timestep_embedder.linear_1", D, 256)
    lin("time_text_embed.timestep_embedder.linear_2", D, D)
    if cfg["guidance_embeds"]:
        lin("time_text_embed.guidance_embedder.linear_1", D, 256)
        lin("time_text_embed.guidance_embedder.linear_2", D, D)
    lin("time_text_embed.text_embedder.linear_1", D, cfg["pooled_projection_dim"])
    lin("time_text_embed.text_embedder.linear_2", D, D)
    for i in range(cfg["num_layers"]):
        p = f"transformer_blocks.{i}"
        lin(p + ".norm1.linear", 6 * D, D)
        lin(p + ".norm1_context.linear", 6 * D, D)
        for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
            lin(p + ".attn." + n, D, D)
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            shapes[p + ".attn." + n + ".weight"] = (hd,)
        lin(p + ".ff.net.0.proj", 4 * D, D)
        lin(p + ".ff.net.2", D, 4 * D)
        lin(p + ".ff_context.net.0.proj", 4 * D, D)
        lin(p + ".ff_context.net.2", D, 4 * D)
    for i in range(cfg["num_single_layers"]):
        p = f"single_transformer_blocks.{i}"
        lin(p + ".norm.linear", 3 * D, D)
        lin(p + ".proj_mlp", 4 * D, D)
        lin(p + ".proj_out", D, 5 * D)
        for n in ("to_q", "to_k", "to_v"):
            lin(p + ".attn." + n, D, D)
        for n in ("norm_q", "norm_k"):
            shapes[p + ".attn." + n + ".weight"] = (hd,)
    lin("norm_out.linear", 2 * D, D)
    lin("proj_out", cfg["patch_size"] ** 2 * cfg["in_channels"], D)
    return shapes


def random_flux_state_dict(cfg, seed=0, std=0.02, dtype=torch.float32, device="cpu", gate_std=None):
    """Seeded N(0, std) weights (norm weights ~ 1 + N(0, 0.1)); for tests / bench only."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd = {}
    for name, shape in flux_param_shapes(cfg).items():
        if ".norm_" in name and name.endswith(".weight") and len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = std * torch.randn(shape, generator=g)
        else:
            t = std * torch.randn(shape, generator=g)
        sd[name] = t.to(device=device, dtype=dtype)
    return sd


def controlnext_param_shapes():
    s = {}

    def conv(name, co, ci, k):
        s[name + ".weight"] = (co, ci, k, k)
        s[name + ".bias"] = (co,)

    def gn(name, c):
        s[name + ".weight"] = (c,)
        s[name + ".bias"] = (c,)

    s["time_embedding.linear_1.weight"] = (256, 128)
    s["time_embedding.linear_1.bias"] = (256,)
    s["time_embedding.linear_2.weight"] = (256, 256)
    s["time_embedding.linear_2.bias"] = (256,)
    conv("embedding.0", 64, 3, 3)
    gn("embedding.1", 64)
    conv("embedding.3", 64, 64, 3)
    gn("embedding.4", 64)
    conv("embedding.6", 128, 64, 3)
    gn("embedding.7", 128)
    for i, (ci, co) in enumerate(((128, 128), (128, 256))):
        p = f"down_res.{i}"
        gn(p + ".norm1", ci)
        conv(p + ".conv1", co, ci, 3)
        s[p + ".time_emb_proj.weight"] = (co, 256)
        s[p + ".time_emb_proj.bias"] = (co,)
        gn(p + ".norm2", co)
        conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".conv_shortcut", co, ci, 1)
        conv(f"down_sample.{i}.conv", co, co, 3)
    conv("mid_convs.0.0", 256, 256, 3)
    gn("mid_convs.0.2", 256)
    conv("mid_convs.0.3", 256, 256, 3)
    gn("mid_convs.0.4", 256)
    conv("mid_convs.1", 3072, 256, 2)
    return s


def random_controlnext_state_dict(seed=0, dtype=torch.float32, device="cpu", out_channels=3072):
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd = {}
    for name, shape in controlnext_param_shapes().items():
        if name.startswith("mid_convs.1"):
            shape = (out_channels,) + tuple(shape[1:])
        if len(shape) == 1 and (".norm" in name or name.startswith("embedding.") or name.startswith("mid_convs.0.")) and name.endswith(".weight") and len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            t = torch.randn(shape, generator=g) / (fan_in ** 0.5)
        elif len(shape) == 2:
            t = torch.randn(shape, generator=g) / (shape[1] ** 0.5)
        else:
            t = 0.02 * torch.randn(shape, generator=g)
        sd[name] = t.to(device=device, dtype=dtype)
    return sd
