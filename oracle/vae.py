"""Oracle restatement of the VAE decode that follows the sampling path:
diffusers==0.31.0 `AutoencoderKL.decode` for the FLUX VAE, as called at infer/inference_qwenvl.py:213-214.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: diffusers is third-party (requirements.txt:3), absent from
/root/reference and from this image; the architecture below is the published FLUX `vae/config.json`
(block_out_channels [128,256,512,512], layers_per_block 2, norm_num_groups 32, latent_channels 16, mid-block attention,
no quant / post-quant conv, scaling_factor 0.3611, shift_factor 0.1159) and diffusers' Decoder / UpDecoderBlock2D /
ResnetBlock2D(temb=None) / Attention(heads=1, residual_connection=True) / Upsample2D(nearest x2 + conv) semantics.
"""
import torch
import torch.nn.functional as F

FLUX_VAE_CFG = dict(latent_channels=16, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                    norm_num_groups=32, scaling_factor=0.3611, shift_factor=0.1159)


def _conv(sd, p, x, stride=1, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def _gn(sd, p, x, groups, eps=1e-6):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


def resnet(sd, p, x, groups):
    h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x, groups)), padding=1)
    h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h, groups)), padding=1)
    if (p + ".conv_shortcut.weight") in sd:
        x = _conv(sd, p + ".conv_shortcut", x)
    return x + h


def mid_attention(sd, p, x, groups):
    B, C, H, W = x.shape
    res = x
    h = x.view(B, C, H * W)
    h = F.group_norm(h, groups, sd[p + ".group_norm.weight"], sd[p + ".group_norm.bias"], 1e-6).transpose(1, 2)
    q = F.linear(h, sd[p + ".to_q.weight"], sd[p + ".to_q.bias"])
    k = F.linear(h, sd[p + ".to_k.weight"], sd[p + ".to_k.bias"])
    v = F.linear(h, sd[p + ".to_v.weight"], sd[p + ".to_v.bias"])
    o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]  # one head of width C
    o = F.linear(o, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return o.transpose(1, 2).reshape(B, C, H, W) + res


def vae_decode(sd, z, cfg=FLUX_VAE_CFG):
    """z: [B, latent_channels, h, w] (caller already applied z / scaling_factor + shift_factor) -> image [B,3,8h,8w]."""
    G = cfg["norm_num_groups"]
    rev = list(reversed(cfg["block_out_channels"]))
    x = _conv(sd, "decoder.conv_in", z, padding=1)
    x = resnet(sd, "decoder.mid_block.resnets.0", x, G)
    x = mid_attention(sd, "decoder.mid_block.attentions.0", x, G)
    x = resnet(sd, "decoder.mid_block.resnets.1", x, G)
    for i in range(len(rev)):
        for j in range(cfg["layers_per_block"] + 1):
            x = resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", x, G)
        if i != len(rev) - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = _conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", x, padding=1)
    x = F.silu(_gn(sd, "decoder.conv_norm_out", x, G))
    return _conv(sd, "decoder.conv_out", x, padding=1)


def vae_decoder_param_shapes(cfg=FLUX_VAE_CFG):
    s = {}

    def conv(n, co, ci, k):
        s[n + ".weight"] = (co, ci, k, k)
        s[n + ".bias"] = (co,)

    def vec(n, c):
        s[n + ".weight"] = (c,)
        s[n + ".bias"] = (c,)

    def lin(n, o, i):
        s[n + ".weight"] = (o, i)
        s[n + ".bias"] = (o,)

    def res(n, ci, co):
        vec(n + ".norm1", ci)
        conv(n + ".conv1", co, ci, 3)
        vec(n + ".norm2", co)
        conv(n + ".conv2", co, co, 3)
        if ci != co:
            conv(n + ".conv_shortcut", co, ci, 1)

    rev = list(reversed(cfg["block_out_channels"]))
    top = rev[0]
    conv("decoder.conv_in", top, cfg["latent_channels"], 3)
    res("decoder.mid_block.resnets.0", top, top)
    a = "decoder.mid_block.attentions.0"
    vec(a + ".group_norm", top)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        lin(a + "." + n, top, top)
    res("decoder.mid_block.resnets.1", top, top)
    prev = top
    for i, co in enumerate(rev):
        for j in range(cfg["layers_per_block"] + 1):
            res(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else co, co)
        prev = co
        if i != len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", co, co, 3)
    vec("decoder.conv_norm_out", rev[-1])
    conv("decoder.conv_out", cfg["out_channels"], rev[-1], 3)
    return s


def random_vae_decoder_state_dict(cfg=FLUX_VAE_CFG, seed=0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for n, shp in vae_decoder_param_shapes(cfg).items():
        if len(shp) == 4:
            t = torch.randn(shp, generator=g) / (shp[1] * shp[2] * shp[3]) ** 0.5
        elif len(shp) == 2:
            t = torch.randn(shp, generator=g) / shp[1] ** 0.5
        elif n.endswith("weight"):
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        else:
            t = 0.02 * torch.randn(shp, generator=g)
        sd[n] = t.to(dtype)
    return sd
