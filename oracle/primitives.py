"""Functional restatement of the diffusers==0.31.0 primitives the X2I hot path uses.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED for this file:
diffusers is a third-party dependency pinned at 0.31.0 by the reference
(requirements.txt:3) whose source is neither under /root/reference nor
installed here.  Each function names the diffusers symbol it restates and the
reference call site that fixes its constructor arguments.

Convention: every function takes a flat state dict `sd` (diffusers key names,
see SURVEY.md Appendix B) and a key prefix; computation runs in the dtype of
the tensors it is handed (fp32 for the oracle proper, bf16 to emulate the
reference's eager bf16 path), with exactly the fp32 promotions diffusers makes.
"""
import math

import torch
import torch.nn.functional as F


def _w(sd, key):
    return sd[key]


def linear(sd, prefix, x):
    """nn.Linear: y = x W^T + b (bias optional)."""
    return F.linear(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"))


# ----------------------------------------------------------------------------
# time / text embedding  (diffusers.models.embeddings)
# ----------------------------------------------------------------------------
def timesteps_proj(t, dim, flip_sin_to_cos=True, downscale_freq_shift=0.0, scale=1.0, max_period=10000):
    """`Timesteps(dim, flip_sin_to_cos, downscale_freq_shift)` == get_timestep_embedding.

    Call sites: CombinedTimestep*Embeddings uses Timesteps(256, True, 0);
    ControlNeXt uses Timesteps(128, True, 0) (lightcontrol_flux.py:590).
    Always returns fp32.
    """
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=t.device)
    exponent = exponent / (half - downscale_freq_shift)
    freqs = torch.exp(exponent)
    ang = t[:, None].float() * freqs[None, :]
    ang = scale * ang
    emb = torch.cat([torch.sin(ang), torch.cos(ang)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


def timestep_embedding(sd, prefix, x):
    """`TimestepEmbedding(in, out, act_fn="silu")`: linear_2(silu(linear_1(x)))."""
    return linear(sd, prefix + ".linear_2", F.silu(linear(sd, prefix + ".linear_1", x)))


def text_projection(sd, prefix, x):
    """`PixArtAlphaTextProjection(in, hidden, act_fn="silu")`: linear_2(silu(linear_1(x)))."""
    return linear(sd, prefix + ".linear_2", F.silu(linear(sd, prefix + ".linear_1", x)))


def combined_time_text_embed(sd, prefix, timestep, pooled, guidance=None):
    """`CombinedTimestepTextProjEmbeddings` / `CombinedTimestepGuidanceTextProjEmbeddings`.

    Constructed at lightcontrol_flux.py:249-254; called at :452-456 with the
    x1000-rescaled timestep (and guidance).
    """
    tproj = timesteps_proj(timestep, 256)
    temb = timestep_embedding(sd, prefix + ".timestep_embedder", tproj.to(pooled.dtype))
    if guidance is not None:
        gproj = timesteps_proj(guidance, 256)
        temb = temb + timestep_embedding(sd, prefix + ".guidance_embedder", gproj.to(pooled.dtype))
    return temb + text_projection(sd, prefix + ".text_embedder", pooled)


# ----------------------------------------------------------------------------
# norms  (diffusers.models.normalization)
# ----------------------------------------------------------------------------
def layer_norm_plain(x, eps=1e-6):
    """nn.LayerNorm(D, elementwise_affine=False, eps=1e-6) (lightcontrol_flux.py:149,152)."""
    return F.layer_norm(x, (x.shape[-1],), None, None, eps)


def rms_norm(x, weight, eps=1e-6):
    """diffusers `RMSNorm(dim, eps, elementwise_affine=True)` as used for qk_norm="rms_norm"."""
    in_dtype = x.dtype
    var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    x = x * torch.rsqrt(var + eps)
    if weight is not None:
        if weight.dtype in (torch.float16, torch.bfloat16):
            x = x.to(weight.dtype)
        x = x * weight
    else:
        x = x.to(in_dtype)
    return x


def ada_layer_norm_zero(sd, prefix, x, emb):
    """`AdaLayerNormZero(D)` (lightcontrol_flux.py:125,127; used :166-170).

    Returns (x_mod, gate_msa, shift_mlp, scale_mlp, gate_mlp); chunk order is
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp.
    """
    e = linear(sd, prefix + ".linear", F.silu(emb))
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = e.chunk(6, dim=1)
    x = layer_norm_plain(x) * (1 + scale_msa[:, None]) + shift_msa[:, None]
    return x, gate_msa, shift_mlp, scale_mlp, gate_mlp


def ada_layer_norm_zero_single(sd, prefix, x, emb):
    """`AdaLayerNormZeroSingle(D)` (lightcontrol_flux.py:63; used :89): shift, scale, gate."""
    e = linear(sd, prefix + ".linear", F.silu(emb))
    shift, scale, gate = e.chunk(3, dim=1)
    x = layer_norm_plain(x) * (1 + scale[:, None]) + shift[:, None]
    return x, gate


def ada_layer_norm_continuous(sd, prefix, x, cond):
    """`AdaLayerNormContinuous(D, D, elementwise_affine=False, eps=1e-6)` (:281, used :542).

    NOTE the chunk order: scale FIRST, then shift.
    """
    e = linear(sd, prefix + ".linear", F.silu(cond).to(x.dtype))
    scale, shift = torch.chunk(e, 2, dim=1)
    return layer_norm_plain(x) * (1 + scale)[:, None, :] + shift[:, None, :]


# ----------------------------------------------------------------------------
# RoPE  (diffusers.models.embeddings.FluxPosEmbed / apply_rotary_emb)
# ----------------------------------------------------------------------------
def flux_pos_embed(ids, axes_dim=(16, 56, 56), theta=10000):
    """`FluxPosEmbed(theta=10000, axes_dim)(ids)` -> (cos, sin) each [S, sum(axes_dim)] fp32.

    Frequencies in float64, angle = outer(pos, freq), each value repeated twice
    (repeat_interleave) so that adjacent pairs share one angle.
    """
    pos = ids.float()
    cos_out, sin_out = [], []
    for i, d in enumerate(axes_dim):
        freqs = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float64, device=ids.device)[: d // 2] / d))
        ang = torch.outer(pos[:, i].to(torch.float64), freqs)
        cos_out.append(ang.cos().repeat_interleave(2, dim=1).float())
        sin_out.append(ang.sin().repeat_interleave(2, dim=1).float())
    return torch.cat(cos_out, dim=-1), torch.cat(sin_out, dim=-1)


def apply_rotary_emb(x, rotary):
    """`apply_rotary_emb(x, (cos, sin))`, use_real=True, use_real_unbind_dim=-1.

    x: [B, heads, S, D]; rotates ADJACENT pairs (x0,x1)->(x0 c - x1 s, x1 c + x0 s).
    """
    cos, sin = rotary
    cos = cos[None, None]
    sin = sin[None, None]
    xr, xi = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    x_rot = torch.stack([-xi, xr], dim=-1).flatten(3)
    return (x.float() * cos + x_rot.float() * sin).to(x.dtype)


# ----------------------------------------------------------------------------
# attention  (diffusers Attention + FluxAttnProcessor2_0)
# ----------------------------------------------------------------------------
def flux_attention(sd, prefix, hidden, heads, rotary=None, encoder_hidden=None, eps=1e-6):
    """`Attention(..., qk_norm="rms_norm", processor=FluxAttnProcessor2_0())`.

    Double-stream form (ctor lightcontrol_flux.py:135-147): returns
    (img_out, txt_out) after to_out[0] / to_add_out.  Single-stream form
    (pre_only=True, :69-80): returns the un-projected joint sequence.
    Text tokens come FIRST in the joint sequence.
    """
    B = hidden.shape[0]

    def heads_view(t):
        return t.view(B, -1, heads, t.shape[-1] // heads).transpose(1, 2)

    q = heads_view(linear(sd, prefix + ".to_q", hidden))
    k = heads_view(linear(sd, prefix + ".to_k", hidden))
    v = heads_view(linear(sd, prefix + ".to_v", hidden))
    q = rms_norm(q, sd[prefix + ".norm_q.weight"], eps)
    k = rms_norm(k, sd[prefix + ".norm_k.weight"], eps)
    if encoder_hidden is not None:
        eq = heads_view(linear(sd, prefix + ".add_q_proj", encoder_hidden))
        ek = heads_view(linear(sd, prefix + ".add_k_proj", encoder_hidden))
        ev = heads_view(linear(sd, prefix + ".add_v_proj", encoder_hidden))
        eq = rms_norm(eq, sd[prefix + ".norm_added_q.weight"], eps)
        ek = rms_norm(ek, sd[prefix + ".norm_added_k.weight"], eps)
        q = torch.cat([eq, q], dim=2)
        k = torch.cat([ek, k], dim=2)
        v = torch.cat([ev, v], dim=2)
    if rotary is not None:
        q = apply_rotary_emb(q, rotary)
        k = apply_rotary_emb(k, rotary)
    o = F.scaled_dot_product_attention(q, k, v, dropout_p=0.0, is_causal=False)
    o = o.transpose(1, 2).reshape(B, -1, q.shape[1] * q.shape[-1]).to(q.dtype)
    if encoder_hidden is not None:
        n_txt = encoder_hidden.shape[1]
        o_txt, o_img = o[:, :n_txt], o[:, n_txt:]
        o_img = linear(sd, prefix + ".to_out.0", o_img)
        o_txt = linear(sd, prefix + ".to_add_out", o_txt)
        return o_img, o_txt
    return o


def feed_forward(sd, prefix, x):
    """`FeedForward(dim, dim_out=dim, activation_fn="gelu-approximate")`: net.0.proj, gelu(tanh), net.2."""
    h = F.gelu(linear(sd, prefix + ".net.0.proj", x), approximate="tanh")
    return linear(sd, prefix + ".net.2", h)


# ----------------------------------------------------------------------------
# ControlNeXt pieces  (diffusers.models.resnet)
# ----------------------------------------------------------------------------
def conv2d(sd, prefix, x, stride=1, padding=0):
    return F.conv2d(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"), stride=stride, padding=padding)


def group_norm(sd, prefix, x, groups, eps):
    return F.group_norm(x, groups, sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def resnet_block2d(sd, prefix, x, temb, groups, eps=1e-6):
    """`ResnetBlock2D(in, out, temb_channels=256, groups=g)` defaults: eps 1e-6, swish,
    time_embedding_norm="default", output_scale_factor=1; 1x1 conv_shortcut iff in != out."""
    h = F.silu(group_norm(sd, prefix + ".norm1", x, groups, eps))
    h = conv2d(sd, prefix + ".conv1", h, padding=1)
    t = linear(sd, prefix + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = h + t
    h = F.silu(group_norm(sd, prefix + ".norm2", h, groups, eps))
    h = conv2d(sd, prefix + ".conv2", h, padding=1)
    if (prefix + ".conv_shortcut.weight") in sd:
        x = conv2d(sd, prefix + ".conv_shortcut", x)
    return x + h


def downsample2d(sd, prefix, x):
    """`Downsample2D(ch, use_conv=True, out_channels, padding=1, name="op")`: Conv2d(3, stride 2, pad 1), key `conv`."""
    return conv2d(sd, prefix + ".conv", x, stride=2, padding=1)
