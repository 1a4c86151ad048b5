"""FLUX VAE decoder on the HIP path -- the step right after the sampling path:
`image = vae.decode(latents / scaling_factor + shift_factor, return_dict=False)[0]` (infer/inference_qwenvl.py:213-214).

Interface of diffusers' AutoencoderKL as the reference uses it: `.config.block_out_channels / scaling_factor /
shift_factor`, `.decode(z, return_dict=False)[0]`, diffusers state-dict keys (`decoder.*`; encoder / quant keys of a
full checkpoint are ignored).  Activations are NHWC bf16; every conv is the implicit-GEMM MFMA kernel (the x2
nearest-neighbour upsampling of Upsample2D is fused into the gather), GroupNorm(32)+SiLU and residual adds are fused
around them, the single-head 512-wide mid-block attention runs as two GEMMs around a row-softmax kernel
(S = q k^T is materialised: 512 MiB per 1024^2 image, once per image).
"""
import math
import os

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_NONE, ACT_SILU


def _p(*shape, device):
    return nn.Parameter(torch.empty(shape, device=device, dtype=torch.bfloat16), requires_grad=False)


class _Conv(nn.Module):
    def __init__(self, cin, cout, k, device):
        super().__init__()
        self.weight = _p(cout, cin, k, k, device=device)
        self.bias = _p(cout, device=device)
        self._packed = None
        self._phases = None

    def packed(self, cin_pad=None, cout_pad=None):
        """[Cout, ky, kx, Cin] bf16; optionally zero-padded along Cin (16 -> 64) or Cout."""
        key = (self.weight._version, self.bias._version, self.weight.data_ptr(), cin_pad, cout_pad)
        if self._packed is None or self._packed[2] != key:  # load_state_dict / in-place updates / other pads invalidate
            w = self.weight.permute(0, 2, 3, 1)
            b = self.bias
            if cin_pad and cin_pad > w.shape[3]:
                w = torch.nn.functional.pad(w, (0, cin_pad - w.shape[3]))
            if cout_pad and cout_pad > w.shape[0]:
                w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, cout_pad - w.shape[0]))
                b = torch.nn.functional.pad(b, (0, cout_pad - b.shape[0]))
            self._packed = (w.reshape(w.shape[0], -1).contiguous(), b.contiguous(), key)
        return self._packed[0], self._packed[1]

    def packed_up_phases(self, rows=False):
        """Upsample2D's conv (nearest x2, then 3 x 3, padding 1) in PHASE form: output column 2x + px of the doubled grid sees the source
        columns {x - 1, x} (px = 0) or {x, x + 1} (px = 1) only -- two of the three taps of every filter row read the same source pixel --
        and likewise output row 2y + py the source rows {y - 1, y} / {y, y + 1}.  A phase is a small convolution on the UN-doubled image
        whose coinciding taps' weights are added (in f32, rounded to bf16 once); no doubled tensor exists.
        rows = False: the two column phases, 3 x 2 taps with the rows still doubled in the gather (x2i_conv_desc.up = 2), 6/9 of the
        multiply-adds: returns (w_px0, w_px1, bias), bf16 [Cout, 3 * 2 * Cin] in (ky, kx, ci) order.
        rows = True: all four (py, px) phases, 2 x 2 taps, 4/9 of the multiply-adds (the phases interleave through ldc = 2 Cout and
        x2i_conv_desc.out_row_pitch): returns ([[w00, w01], [w10, w11]], bias), bf16 [Cout, 2 * 2 * Cin]."""
        key = (self.weight._version, self.bias._version, self.weight.data_ptr(), "up_phases", rows)
        if getattr(self, "_phases", None) is None or self._phases[-1] != key:
            w = self.weight.float().permute(0, 2, 3, 1)                         # [Cout, ky, kx, Cin]
            cols = [torch.stack([w[:, :, 0], w[:, :, 1] + w[:, :, 2]], 2),       # px = 0: source columns x - 1, x
                    torch.stack([w[:, :, 0] + w[:, :, 1], w[:, :, 2]], 2)]       # px = 1: source columns x, x + 1
            pk = lambda t: t.to(torch.bfloat16).reshape(t.shape[0], -1).contiguous()   # noqa: E731
            if rows:
                both = [[torch.stack([c[:, 0], c[:, 1] + c[:, 2]], 1) for c in cols],    # py = 0: source rows y - 1, y
                        [torch.stack([c[:, 0] + c[:, 1], c[:, 2]], 1) for c in cols]]    # py = 1: source rows y, y + 1
                self._phases = ([[pk(t) for t in r] for r in both], self.bias.contiguous(), key)
            else:
                self._phases = (pk(cols[0]), pk(cols[1]), self.bias.contiguous(), key)
        return self._phases[:-1]


class _Vec(nn.Module):
    def __init__(self, c, device):
        super().__init__()
        self.weight = _p(c, device=device)
        self.bias = _p(c, device=device)


class _Lin(nn.Module):
    def __init__(self, i, o, device):
        super().__init__()
        self.weight = _p(o, i, device=device)
        self.bias = _p(o, device=device)


class _Seq(nn.Module):
    def __init__(self, items):
        super().__init__()
        for k, v in items.items():
            self.add_module(str(k), v)

    def __getitem__(self, i):
        return self._modules[str(i)]

    def __len__(self):
        return len(self._modules)


class _Res(nn.Module):
    def __init__(self, cin, cout, device):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.norm1 = _Vec(cin, device)
        self.conv1 = _Conv(cin, cout, 3, device)
        self.norm2 = _Vec(cout, device)
        self.conv2 = _Conv(cout, cout, 3, device)
        if cin != cout:
            self.conv_shortcut = _Conv(cin, cout, 1, device)

    def run(self, x, H, W, G, mom=None, want=False):
        """mom: channel moments of x left by the conv that produced it (None: a statistics pass reads x); want: return (y, moments of y).
        With moments the GroupNorms here read their input once (normalise) instead of twice (x2i_conv_desc.moments)."""
        n = _gn(x, mom, self.norm1, G)
        w, b = self.conv1.packed()
        m1 = _mom(x, self.cout) if want else None
        h = ops.conv2d_nhwc(n, w, b, H, W, self.cin, self.cout, 3, 3, 1, 1, moments=m1)
        n = _gn(h, m1, self.norm2, G)
        if hasattr(self, "conv_shortcut"):
            w, b = self.conv_shortcut.packed()
            x = ops.conv2d_nhwc(x, w, b, H, W, self.cin, self.cout, 1, 1, 1, 0)
        w, b = self.conv2.packed()
        m2 = _mom(x, self.cout) if want else None
        y = ops.conv2d_nhwc(n, w, b, H, W, self.cout, self.cout, 3, 3, 1, 1, res=x, moments=m2)
        return (y, m2) if want else y


def _mom(x, c):
    return torch.empty((x.shape[0], c, 2), device=x.device, dtype=torch.float32)


def _gn(x, mom, norm, G, act=ACT_SILU):
    if mom is None:
        return ops.groupnorm_nhwc(x, norm.weight, norm.bias, G, 1e-6, act=act)
    return ops.groupnorm_nhwc_from_moments(x, mom, norm.weight, norm.bias, G, 1e-6, act=act)


class _Attn(nn.Module):
    def __init__(self, c, device):
        super().__init__()
        self.c = c
        self.group_norm = _Vec(c, device)
        self.to_q, self.to_k, self.to_v = _Lin(c, c, device), _Lin(c, c, device), _Lin(c, c, device)
        self.to_out = _Seq({0: _Lin(c, c, device)})

    def run(self, x, H, W, G, mom=None):
        B, C, T = x.shape[0], self.c, H * W
        h = _gn(x, mom, self.group_norm, G, act=ACT_NONE)  # [B,H,W,C] == [B,T,C]
        q = ops.gemm(h, self.to_q.weight, self.to_q.bias, M=B * T)
        k = ops.gemm(h, self.to_k.weight, self.to_k.bias, M=B * T)
        # V^T[c][t] = sum_j Wv[c][j] h[t][j]  (one GEMM per image with h as the "weight"); the bias of to_v is added after
        # P V instead: softmax rows sum to one, so P (V + 1 b^T) = P V + b^T
        vt = torch.empty((B, C, T), device=x.device, dtype=torch.bfloat16)
        ops.gemm(self.to_v.weight, h, None, out=vt, M=C, N=T, K=C, batch=B, a_batch_stride=0, lda=C, c_batch_stride=C * T, ldc=T,
                 w_batch_stride=T * C)
        s = torch.empty((B, T, T), device=x.device, dtype=torch.bfloat16)
        ops.gemm(q, k, None, out=s, M=T, N=T, K=C, batch=B, a_batch_stride=T * C, lda=C, c_batch_stride=T * T, ldc=T,
                 w_batch_stride=T * C)
        ops.softmax_rows_(s, 1.0 / math.sqrt(C))
        o = torch.empty((B, T, C), device=x.device, dtype=torch.bfloat16)
        ops.gemm(s, vt, self.to_v.bias, out=o, M=T, N=C, K=T, batch=B, a_batch_stride=T * T, lda=T, c_batch_stride=T * C, ldc=C,
                 w_batch_stride=C * T)
        out = torch.empty_like(x)
        ops.gemm(o, self.to_out[0].weight, self.to_out[0].bias, out=out, M=B * T, res=x, ldr=C)  # + residual
        return out


class _Up(nn.Module):
    def __init__(self, c, device):
        super().__init__()
        self.conv = _Conv(c, c, 3, device)


class _UpBlock(nn.Module):
    def __init__(self, cin, cout, n, upsample, device):
        super().__init__()
        self.resnets = _Seq({j: _Res(cin if j == 0 else cout, cout, device) for j in range(n)})
        if upsample:
            self.upsamplers = _Seq({0: _Up(cout, device)})


class _Mid(nn.Module):
    def __init__(self, c, device):
        super().__init__()
        self.resnets = _Seq({0: _Res(c, c, device), 1: _Res(c, c, device)})
        self.attentions = _Seq({0: _Attn(c, device)})


class _Decoder(nn.Module):
    def __init__(self, cfg, device):
        super().__init__()
        rev = list(reversed(cfg.block_out_channels))
        self.conv_in = _Conv(cfg.latent_channels, rev[0], 3, device)
        self.mid_block = _Mid(rev[0], device)
        blocks, prev = {}, rev[0]
        for i, co in enumerate(rev):
            blocks[i] = _UpBlock(prev, co, cfg.layers_per_block + 1, i != len(rev) - 1, device)
            prev = co
        self.up_blocks = _Seq(blocks)
        self.conv_norm_out = _Vec(rev[-1], device)
        self.conv_out = _Conv(rev[-1], cfg.out_channels, 3, device)


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def decode_flops(cfg, H, W, up_phases=0):
    """Algorithmic FLOPs of one `decode` of an [*, latent_channels, H, W] latent (2 * Cin * Cout * k^2 per output pixel of every
    convolution, 2 M N K of the mid-block attention's linears and its two T x T products), un-padded channel counts.
    up_phases = 1 / 2: the FLOPs the HIP path EXECUTES when Upsample2D's convs run in column-phase / four-phase form (6/9, 4/9 of theirs)."""
    rev = list(reversed(cfg.block_out_channels))
    conv = lambda ci, co, k, h, w: 2.0 * ci * co * k * k * h * w   # noqa: E731
    res = lambda ci, co, h, w: conv(ci, co, 3, h, w) + conv(co, co, 3, h, w) + (conv(ci, co, 1, h, w) if ci != co else 0.0)   # noqa: E731
    fl = conv(cfg.latent_channels, rev[0], 3, H, W)
    c, T = rev[0], H * W
    fl += 2 * res(c, c, H, W) + 4 * 2.0 * T * c * c + 2 * 2.0 * T * T * c
    prev = rev[0]
    for i, co in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            fl += res(prev if j == 0 else co, co, H, W)
        if i != len(rev) - 1:
            H, W = 2 * H, 2 * W
            fl += conv(co, co, 3, H, W) * {0: 1.0, 1: 6.0 / 9.0, 2: 4.0 / 9.0}[up_phases]
        prev = co
    return fl + conv(rev[-1], cfg.out_channels, 3, H, W)


class AutoencoderKL(nn.Module):
    """Decoder half of diffusers' AutoencoderKL with the FLUX configuration as defaults."""

    def __init__(self, latent_channels=16, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 norm_num_groups=32, scaling_factor=0.3611, shift_factor=0.1159, device="cuda"):
        super().__init__()
        self.config = _Cfg(latent_channels=latent_channels, out_channels=out_channels, block_out_channels=tuple(block_out_channels),
                           layers_per_block=layers_per_block, norm_num_groups=norm_num_groups, scaling_factor=scaling_factor,
                           shift_factor=shift_factor)
        if any(c % 64 for c in block_out_channels):
            raise ValueError("x2i_amd VAE: block_out_channels must be multiples of 64 (implicit-GEMM conv)")
        self.decoder = _Decoder(self.config, device)
        # X2I_VAE_UP_PHASES (read once, here): how Upsample2D's conv runs -- 2 (product): four 2 x 2 phase convolutions on the un-doubled image;
        # 1: two 3 x 2 column phases; 0: ONE 3 x 3 conv with the x2 upsampling in its gather.  Same result within the tolerance of one more
        # bf16 rounding of summed weights (_Conv.packed_up_phases; tests/test_vae_gpu.py)
        self.up_phases = int(os.environ.get("X2I_VAE_UP_PHASES", "2"))
        if self.up_phases not in (0, 1, 2):
            raise ValueError("X2I_VAE_UP_PHASES must be 0 (one 3x3 conv, upsampling in the gather), 1 (two column phases) or 2 (four phases), got %d"
                             % self.up_phases)
        # X2I_VAE_EPI_MOMENTS=0: every GroupNorm takes its statistics in a pass of its own (A/B); default: from the producing conv's epilogue
        self.epilogue_moments = os.environ.get("X2I_VAE_EPI_MOMENTS", "1") != "0"
        # X2I_VAE_NARROW_CONV_OUT=0: conv_out as an implicit GEMM on the 128-column tile kernel (A/B); default: the narrow-output MFMA kernel
        self.narrow_conv_out = os.environ.get("X2I_VAE_NARROW_CONV_OUT", "1") != "0"

    def _apply(self, fn, recurse=True):
        r = super()._apply(fn, recurse)
        for m in self.modules():
            if isinstance(m, _Conv):
                m._packed = None
                m._phases = None
        return r

    def load_state_dict(self, sd, strict=True):
        """Accepts a full AutoencoderKL checkpoint: encoder.* / quant_conv.* / post_quant_conv.* keys are ignored."""
        if any(k.startswith("post_quant_conv.") for k in sd) and self.config.get("use_post_quant_conv", False):
            raise NotImplementedError("x2i_amd VAE: use_post_quant_conv=True is not supported (the FLUX VAE has none)")
        dec = {k: v for k, v in sd.items() if k.startswith("decoder.")}
        return super().load_state_dict(dec, strict=strict)

    @torch.no_grad()
    def decode(self, z, return_dict=True):
        cfg, d = self.config, self.decoder
        G = cfg.norm_num_groups
        B, Cz, H, W = z.shape
        x = torch.zeros((B, H, W, 64), device=z.device, dtype=torch.bfloat16)  # latent channels zero-padded to one K-step
        x[..., :Cz] = z.to(torch.bfloat16).permute(0, 2, 3, 1)
        rev = list(reversed(cfg.block_out_channels))
        em = self.epilogue_moments   # the convs' epilogues leave the channel moments the next GroupNorm needs (no statistics pass over the tensor)
        w, b = d.conv_in.packed(cin_pad=64)
        mom = _mom(x, rev[0]) if em else None
        x = ops.conv2d_nhwc(x, w, b, H, W, 64, rev[0], 3, 3, 1, 1, moments=mom)
        x, mom = d.mid_block.resnets[0].run(x, H, W, G, mom, True) if em else (d.mid_block.resnets[0].run(x, H, W, G), None)
        x = d.mid_block.attentions[0].run(x, H, W, G, mom)      # (its output comes from a plain GEMM: the next norm1 takes its own statistics)
        x, mom = d.mid_block.resnets[1].run(x, H, W, G, None, True) if em else (d.mid_block.resnets[1].run(x, H, W, G), None)
        for i, co in enumerate(rev):
            blk = d.up_blocks[i]
            for j in range(len(blk.resnets)):
                x, mom = blk.resnets[j].run(x, H, W, G, mom, True) if em else (blk.resnets[j].run(x, H, W, G), None)
            if hasattr(blk, "upsamplers"):
                y = torch.empty((B, 2 * H, 2 * W, co), device=x.device, dtype=torch.bfloat16) if self.up_phases else None
                mom = _mom(x, co) if em else None
                if self.up_phases == 2:
                    # F.interpolate(nearest, x2) + conv as four 2 x 2 phase convolutions on the un-doubled image (4/9 of the work): phase
                    # (py, px) writes the pixels (2y + py, 2x + px): column stride 2 Cout, row pitch two full rows
                    wp, b = blk.upsamplers[0].conv.packed_up_phases(rows=True)
                    for py in (0, 1):
                        for px in (0, 1):
                            ops.conv2d_nhwc(x, wp[py][px], b, H, W, co, co, 2, 2, 1, 1 - py, pad_w=1 - px, out_h=H, out_w=W, out=y, ldc=2 * co,
                                            out_row_pitch=4 * W * co, c_offset=(py * 2 * W + px) * co, c_batch_stride=4 * H * W * co,
                                            moments=mom, moments_accumulate=(py, px) != (0, 0))
                    x = y
                elif self.up_phases:
                    # ... as two 3 x 2 column-phase convolutions, the rows doubled in the gather (6/9 of the work; A/B form)
                    w0, w1, b = blk.upsamplers[0].conv.packed_up_phases()
                    for px, wp in ((0, w0), (1, w1)):
                        ops.conv2d_nhwc(x, wp, b, H, W, co, co, 3, 2, 1, 1, up=2, pad_w=1 - px, out_w=W, out=y, ldc=2 * co, c_offset=px * co,
                                        c_batch_stride=4 * H * W * co, moments=mom, moments_accumulate=px == 1)
                    x = y
                else:
                    w, b = blk.upsamplers[0].conv.packed()
                    x = ops.conv2d_nhwc(x, w, b, H, W, co, co, 3, 3, 1, 1, up=True, moments=mom)  # F.interpolate(nearest, x2) + conv, fused gather
                H, W = 2 * H, 2 * W
        n = _gn(x, mom, d.conv_norm_out, G)
        if self.narrow_conv_out and cfg.out_channels <= 4 and rev[-1] in (32, 64, 96, 128):
            w, b = d.conv_out.packed()             # three output channels in the rows of the 16 x 16 x 32 MFMA (x2i_conv3x3_narrow_bf16)
            y = ops.conv3x3_narrow(n, w, b, cfg.out_channels)
        else:
            w, b = d.conv_out.packed(cout_pad=8)  # implicit GEMM, 3 -> 8 output channels so that rows are 16-byte aligned (A/B form)
            y = ops.conv2d_nhwc(n, w, b, H, W, rev[-1], 8, 3, 3, 1, 1)
        img = y[..., :cfg.out_channels].permute(0, 3, 1, 2).contiguous()
        if not return_dict:
            return (img,)
        return _Cfg(sample=img)

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=torch.bfloat16, device=None, **kw):
        """diffusers directory layout: <path>/<subfolder>/config.json + *.safetensors (decoder.* keys are used)."""
        import glob
        import json
        import os

        from safetensors import safe_open
        if device is None:  # `.from_pretrained(...).to(device)` call chains: land on the visible HIP device, else the CPU
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, "config.json")) as fh:
            c = json.load(fh)
        keys = ("latent_channels", "out_channels", "block_out_channels", "layers_per_block", "norm_num_groups", "scaling_factor",
                "shift_factor")
        if c.get("use_post_quant_conv", False):
            raise NotImplementedError("x2i_amd VAE: config has use_post_quant_conv=true; the decoder here starts at conv_in (FLUX)")
        vae = cls(**{k: c[k] for k in keys if k in c}, device=device)
        own = dict(vae.named_parameters())
        seen = set()
        for shard in sorted(glob.glob(os.path.join(d, "*.safetensors"))):
            with safe_open(shard, framework="pt", device="cpu") as sf:
                for k in sf.keys():
                    if k in own:
                        own[k].copy_(sf.get_tensor(k))  # (not .data.copy_: keeps the version counter honest)
                        seen.add(k)
        missing = set(own) - seen
        if missing:
            raise KeyError("VAE checkpoint is missing decoder keys: %s ..." % sorted(missing)[:6])
        return vae.eval()

    @torch.no_grad()
    def init_random_(self, seed=0):
        g = torch.Generator(device="cpu").manual_seed(seed)
        for n, p in self.named_parameters():
            if p.dim() >= 2:
                v = torch.randn(p.shape, generator=g) / p[0].numel() ** 0.5
            elif n.endswith("weight"):
                v = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
            else:
                v = 0.02 * torch.randn(p.shape, generator=g)
            p.copy_(v)
        return self
