"""LightControl (ControlNeXt) instruction-edit branch on the HIP path.

Mirrors lightcontrol/lightcontrol_flux.py: `ControlNeXtModel` (:575-749) with the reference's parameter names, and a
`FluxTransformer2DModel` whose forward takes `guided_hint=` / `control_nets=` and, with return_dict=False, returns the
BARE tensor (:549-550; train_lightcontrol.py:732-746 relies on it).  The reference ships no LightControl inference
script; `LightControlSampler` is this build's counterpart of the per-step call in train_lightcontrol.py:732-743 plus the
FLUX.1-dev Euler schedule (SURVEY.md "Row L").

Activations are NHWC bf16; convolutions with Cin >= 64 are implicit GEMMs on the MFMA kernel (weights repacked once to
[Cout][ky][kx][Cin]); GroupNorm / ReLU / SiLU / time-embedding adds / residuals are fused around them.
Everything up to the first ResnetBlock's conv1 does not depend on the timestep and is cached per hint image
(`prepare_hint`): 135 of the 437 GFLOP per model per image are paid once per sample instead of once per step.

Round 4: the two places where the reference chains LINEAR maps without a nonlinearity between them are evaluated as one
convolution each (`_Composed`): ResnetBlock2D's `conv2(...) + shortcut(x)` is followed directly by Downsample2D's 3x3 stride-2 conv
(lightcontrol_flux.py:741-743), so   down(conv2(n) + b2 + sc(x))  =  [down o conv2](n)  +  [down o sc](x)  +  down(bias fields),
where down o conv2 is a 5x5 stride-2 convolution (25 taps on the quarter-size output grid instead of 9 on the full-size grid plus 9 on
the quarter-size one: 1.8x fewer FLOPs, and the full-size tensor between the two is never written), down o sc a 3x3 stride-2 one
(res 1) or timestep-independent and cached per hint (res 0, identity shortcut), and the bias terms constant maps.  Zero padding is
reproduced exactly: the only outputs where `down`'s padding of the INTERMEDIATE differs from padding the input are the first output row
and column, and those are corrected by three small launches (a 1x5, a 5x1 and a 1x1 convolution on the first input row / column / pixel)
that take back the terms the padded intermediate would not have contributed.  Not bit-identical to the chained form (the intermediate's
bf16 rounding is gone, the composed taps are rounded to bf16 once); same tolerance against the fp32 oracle.
"""
import torch
import torch.nn as nn

from . import ops
from .flux import FluxTransformer2DModel as _BaseFlux
from .ops import ACT_NONE, ACT_RELU, ACT_SILU


def _param(*shape, device, dtype=torch.bfloat16):
    return nn.Parameter(torch.empty(shape, device=device, dtype=dtype), requires_grad=False)


class _Conv(nn.Module):
    def __init__(self, cin, cout, k, device):
        super().__init__()
        self.weight = _param(cout, cin, k, k, device=device)
        self.bias = _param(cout, device=device)
        self.k = k
        self._packed = None

    def packed(self):
        """[Cout, ky, kx, Cin] bf16, K-contiguous for the implicit GEMM (repacked once; reference layout is OIHW)."""
        key = (self.weight._version, self.weight.data_ptr())
        if self._packed is None or self._packed[1] != key:  # load_state_dict / in-place updates / moves invalidate the copy
            self._packed = (self.weight.permute(0, 2, 3, 1).reshape(self.weight.shape[0], -1).contiguous(), key)
        return self._packed[0]


def _compose_taps(wd, wc):
    """Weights of (conv with taps wd) o (conv with taps wc), stride of the first factor applied outside: wd bf16 [Co, KH1, KW1, Mid] and
    wc bf16 [Mid, KH2, KW2, Ci] (both in the packed (ky, kx, c) order) -> bf16 [Co, (KH1 + KH2 - 1) * (KW1 + KW2 - 1) * Ci] with
    W[co, ky + jy, kx + jx, ci] = sum_mid wd[co, ky, kx, mid] * wc[mid, jy, jx, ci].  ONE MFMA GEMM against a shifted, zero-filled
    copy of wc (the copy is index plumbing; the arithmetic is fp32 accumulation of exact bf16 products, rounded to bf16 once)."""
    co, kh1, kw1, mid = wd.shape
    _, kh2, kw2, ci = wc.shape
    kh, kw = kh1 + kh2 - 1, kw1 + kw2 - 1
    wt = torch.zeros((kh, kw, ci, kh1, kw1, mid), device=wd.device, dtype=torch.bfloat16)
    wct = wc.permute(1, 2, 3, 0)  # [jy, jx, ci, mid]
    for ky in range(kh1):
        for kx in range(kw1):
            for jy in range(kh2):
                for jx in range(kw2):
                    wt[ky + jy, kx + jx, :, ky, kx, :] = wct[jy, jx]
    c = ops.gemm(wd.reshape(co, kh1 * kw1 * mid).contiguous(), wt.reshape(kh * kw * ci, kh1 * kw1 * mid), None, out_f32=True)
    return ops.to_bf16(c)


class _Composed:
    """Per-net composed weights of one `ResnetBlock2D.conv2 (+ shortcut) -> Downsample2D.conv` chain (see the module docstring):
    w5 = down o conv2 (5x5, stride 2, pad 2), the three first-row / first-column / first-pixel corrections (stored NEGATED where they are
    subtracted: the conv epilogue adds), wsc = down o conv_shortcut (3x3 stride 2) when the block has one, and the constant map of the
    bias terms on the output grid."""

    def __init__(self, res, down, oh, ow):
        cd, cc = down.conv, res.conv2
        wd = cd.packed().view(cd.weight.shape[0], 3, 3, -1)          # [co, ky, kx, mid]
        wc = cc.packed().view(cc.weight.shape[0], 3, 3, -1)          # [mid, jy, jx, ci]
        self.co, self.mid, self.ci = wd.shape[0], wd.shape[3], wc.shape[3]
        self.w5 = _compose_taps(wd, wc)
        # the intermediate's zero padding: output row 0 must not see `down`'s ky = 0 taps (they sit on padding), but the composed 5x5
        # applies them to conv2 evaluated one row above the image, which reaches input row 0 through conv2's last row of taps (jy = 2).
        # Same for column 0; the (ky, kx) = (0, 0) tap is taken back twice and returned once.
        self.w_top = torch.neg(_compose_taps(wd[:, 0:1].contiguous(), wc[:, 2:3].contiguous()))            # 1 x 5
        self.w_left = torch.neg(_compose_taps(wd[:, :, 0:1].contiguous(), wc[:, :, 2:3].contiguous()))     # 5 x 1
        self.w_corner = _compose_taps(wd[:, 0:1, 0:1].contiguous(), wc[:, 2:3, 2:3].contiguous())          # 1 x 1
        self.wsc = None
        fields = [cc.bias]
        if hasattr(res, "conv_shortcut"):
            ws = res.conv_shortcut.packed().view(self.mid, 1, 1, -1)
            self.wsc = _compose_taps(wd, ws)                                                               # 3 x 3, stride 2, pad 1
            fields.append(res.conv_shortcut.bias)
        # down(bias fields) + down's own bias: `down` applied to constant images (zero padded like any other input): exact at the borders
        m = None
        for i, f in enumerate(fields):
            const = f.view(1, 1, 1, self.mid).expand(1, 2 * oh, 2 * ow, self.mid).contiguous()
            m = ops.conv2d_nhwc(const, cd.packed(), cd.bias if i == 0 else None, 2 * oh, 2 * ow, self.mid, self.co, 3, 3, 2, 1, res=m)
        self.bias_map = m                                                                                  # [1, oh, ow, co]

    def apply(self, n, h, w, base, out=None):
        """[down o conv2](n) + base on the (h/2, w/2) grid; n bf16 [B, h, w, ci]; base bf16 [B or 1, h/2, w/2, co] (everything else that
        `down` sees: bias map, shortcut / identity path)."""
        return _composed_apply(self.w5, self.w_top, self.w_left, self.w_corner, self.ci, self.co, n, h, w, base, out=out)


def _composed_apply(w5, w_top, w_left, w_corner, ci, co, n, h, w, base, out=None, w_group=0):
    """_Composed.apply on explicit weights; with w_group > 0 they carry a leading group dimension (ControlNeXtBank: [nets, co, K]) and batch item b
    of `n` uses group b // w_group."""
    B = n.shape[0]
    oh, ow = h // 2, w // 2
    bs = 0 if base.shape[0] == 1 else oh * ow * co
    g = dict(w_group=w_group)
    d = ops.conv2d_nhwc(n, w5, None, h, w, ci, co, 5, 5, 2, 2, res=base, res_batch_stride=bs, out=out, **g)
    # first output row: a 1 x 5 stride-2 convolution over input row 0, added in place (negated weights)
    ops.conv2d_nhwc(n, w_top, None, 1, w, ci, co, 1, 5, 2, 0, pad_w=2, B=B, a_batch_stride=h * w * ci, out=d, c_batch_stride=oh * ow * co,
                    res=d, res_batch_stride=oh * ow * co, **g)
    # first output column: a 5 x 1 stride-2 convolution over input column 0 (gathered: the conv reads contiguous NHWC), written with
    # the output grid's row pitch so that result oy lands on pixel (oy, 0)
    col = n[:, :, 0:1, :].contiguous()
    ops.conv2d_nhwc(col, w_left, None, h, 1, ci, co, 5, 1, 2, 2, pad_w=0, out=d, c_batch_stride=oh * ow * co, ldc=ow * co, res=d,
                    res_batch_stride=oh * ow * co, ldr=ow * co, **g)
    # first pixel: the (0, 0) tap was taken back by both corrections
    ops.gemm(n, w_corner, None, out=d, M=1, batch=B, a_batch_stride=h * w * ci, lda=ci, c_batch_stride=oh * ow * co, ldc=co, res=d,
             res_batch_stride=oh * ow * co, ldr=co, w_batch_stride=co * ci if w_group else 0, w_group=w_group)
    return d


class _Affine(nn.Module):
    def __init__(self, c, device):
        super().__init__()
        self.weight = _param(c, device=device)
        self.bias = _param(c, device=device)


class _Lin(nn.Module):
    def __init__(self, i, o, device):
        super().__init__()
        self.weight = _param(o, i, device=device)
        self.bias = _param(o, device=device)


class _Sparse(nn.Module):
    def __init__(self, items):
        super().__init__()
        for k, v in items.items():
            self.add_module(str(k), v)

    def __getitem__(self, i):
        return self._modules[str(i)]


class _Res(nn.Module):
    def __init__(self, cin, cout, device):
        super().__init__()
        self.norm1 = _Affine(cin, device)
        self.conv1 = _Conv(cin, cout, 3, device)
        self.time_emb_proj = _Lin(256, cout, device)
        self.norm2 = _Affine(cout, device)
        self.conv2 = _Conv(cout, cout, 3, device)
        if cin != cout:
            self.conv_shortcut = _Conv(cin, cout, 1, device)


class _Down(nn.Module):
    def __init__(self, c, device):
        super().__init__()
        self.conv = _Conv(c, c, 3, device)


class ControlNeXtModel(nn.Module):
    """lightcontrol_flux.py:575-749.  forward(sample [B,3,H,W], timestep) -> {"out": [B,out_ch,H/16,W/16], "scale": 1.0}.
    `out_channels` generalises the hard-coded 3072 (:661-668) so reduced-width models can be tested."""

    def __init__(self, in_channels=(128, 128), out_channels=(128, 256), groups=(4, 8), time_embed_dim=256,
                 final_out_channels=320, device="cuda", control_out_channels=3072):
        super().__init__()
        assert tuple(in_channels) == (128, 128) and tuple(out_channels) == (128, 256) and time_embed_dim == 256
        self.groups = tuple(groups)
        self.scale = 1.0
        te = nn.Module()
        te.linear_1 = _Lin(128, 256, device)
        te.linear_2 = _Lin(256, 256, device)
        self.time_embedding = te
        self.embedding = _Sparse({0: _Conv(3, 64, 3, device), 1: _Affine(64, device), 3: _Conv(64, 64, 3, device),
                                  4: _Affine(64, device), 6: _Conv(64, 128, 3, device), 7: _Affine(128, device)})
        self.down_res = _Sparse({0: _Res(128, 128, device), 1: _Res(128, 256, device)})
        self.down_sample = _Sparse({0: _Down(128, device), 1: _Down(256, device)})
        mid0 = _Sparse({0: _Conv(256, 256, 3, device), 2: _Affine(256, device), 3: _Conv(256, 256, 3, device),
                        4: _Affine(256, device)})
        self.mid_convs = _Sparse({0: mid0, 1: _Conv(256, control_out_channels, 2, device)})
        self._hint_cache = None
        self.compose = True        # conv2 -> Downsample2D chains as one convolution each (module docstring); False: the chained form (A/B)
        self._composed_cache = None

    def _apply(self, fn, recurse=True):
        r = super()._apply(fn, recurse)
        for m in self.modules():
            if isinstance(m, _Conv):
                m._packed = None
        self._hint_cache = None
        self._composed_cache = None
        return r

    def _composed(self, h, w):
        """The two composed chains for a (h, w) = (H/2, W/2) grid; rebuilt when a weight changes (load_state_dict, .to())."""
        convs = [self.down_res[0].conv2, self.down_sample[0].conv, self.down_res[1].conv2, self.down_res[1].conv_shortcut, self.down_sample[1].conv]
        key = (h, w) + tuple((c.weight._version, c.weight.data_ptr(), c.bias._version, c.bias.data_ptr()) for c in convs)
        if self._composed_cache is None or self._composed_cache[0] != key:
            self._composed_cache = (key, _Composed(self.down_res[0], self.down_sample[0], h // 2, w // 2),
                                    _Composed(self.down_res[1], self.down_sample[1], h // 4, w // 4))
        return self._composed_cache[1], self._composed_cache[2]

    # ---- timestep-independent prefix (cached per hint tensor)
    @torch.no_grad()
    def prepare_hint(self, sample):
        """embedding (3 convs + GN + ReLU, :593-603,740) and ResnetBlock 0's norm1+SiLU+conv1 (before the time term)."""
        B, _, H, W = sample.shape
        e = self.embedding
        x = sample.to(torch.bfloat16).permute(0, 2, 3, 1).contiguous()  # NHWC
        w0 = e[0].weight.float().permute(0, 2, 3, 1).contiguous()  # [64, ky, kx, 3]
        x = ops.conv_stem(x, w0, e[0].bias.float(), 64)
        x = ops.groupnorm_nhwc(x, e[1].weight, e[1].bias, 2, 1e-5, act=ACT_RELU)
        h, w = H // 2, W // 2
        x = ops.conv2d_nhwc(x, e[3].packed(), e[3].bias, h, w, 64, 64, 3, 3, 1, 1)
        x = ops.groupnorm_nhwc(x, e[4].weight, e[4].bias, 2, 1e-5, act=ACT_RELU)
        x = ops.conv2d_nhwc(x, e[6].packed(), e[6].bias, h, w, 64, 128, 3, 3, 1, 1)
        x0 = ops.groupnorm_nhwc(x, e[7].weight, e[7].bias, 2, 1e-5, act=ACT_RELU)
        r = self.down_res[0]
        n = ops.groupnorm_nhwc(x0, r.norm1.weight, r.norm1.bias, self.groups[0], 1e-6, act=ACT_SILU)
        h1 = ops.conv2d_nhwc(n, r.conv1.packed(), r.conv1.bias, h, w, 128, 128, 3, 3, 1, 1)
        # norm2 of this block normalises h1 + time_emb_proj(...): h1 does not depend on the timestep, so its per-channel moments are taken
        # here, once per hint, and the denoising loop never runs a statistics pass over the [B, H/2, W/2, 128] tensor again
        prep = dict(x0=x0, h1=h1, h1_moments=ops.groupnorm_moments(h1), h=h, w=w, B=B, round_bf16=sample.dtype == torch.bfloat16)
        if self.compose:
            # down_sample[0] of (the block's identity shortcut + the bias fields): the timestep-independent part of what it will see
            ca, _ = self._composed(h, w)
            d = self.down_sample[0].conv
            prep["d0"] = ops.conv2d_nhwc(x0, d.packed(), None, h, w, 128, 128, 3, 3, 2, 1, res=ca.bias_map, res_batch_stride=0)
        return prep

    def _resblock_tail(self, r, x_in, h1, temb_act, G, h, w, cin, cout, h1_moments=None):
        """ResnetBlock2D after conv1: h = h1 + time_emb_proj(silu(temb)); h = conv2(silu(GN(h))); out = shortcut(x) + h."""
        tproj = ops.skinny_linear(temb_act, r.time_emb_proj.weight, r.time_emb_proj.bias, act_in=ACT_SILU)  # [B, cout] f32
        if h1_moments is not None:
            n = ops.groupnorm_nhwc_from_moments(h1, h1_moments, r.norm2.weight, r.norm2.bias, G, 1e-6, act=ACT_SILU, pre_add=tproj)
        else:
            n = ops.groupnorm_nhwc(h1, r.norm2.weight, r.norm2.bias, G, 1e-6, act=ACT_SILU, pre_add=tproj)
        if hasattr(r, "conv_shortcut"):
            sc = ops.conv2d_nhwc(x_in, r.conv_shortcut.packed(), r.conv_shortcut.bias, h, w, cin, cout, 1, 1, 1, 0)
        else:
            sc = x_in
        return ops.conv2d_nhwc(n, r.conv2.packed(), r.conv2.bias, h, w, cout, cout, 3, 3, 1, 1, res=sc)

    @staticmethod
    @torch.no_grad()
    def timestep_features(prep, timestep):
        """Timesteps(128)(timestep) for the B samples of a prepared hint: f32 [B, 128] (lightcontrol_flux.py:730-733).  The same for every
        net that was prepared with the same hint dtype, so the 19 nets of a step share ONE evaluation (make_control_fn)."""
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([float(t)], device=prep["x0"].device)
        t = t.reshape(-1).to(device=prep["x0"].device, dtype=torch.float32).expand(prep["B"]).contiguous()
        return ops.timestep_sinusoid(t, 128, round_bf16=prep["round_bf16"])  # Timesteps(128).to(sample.dtype)

    @torch.no_grad()
    def forward_nhwc(self, prep, timestep, add_into=None, add_offset=0, add_batch_stride=None, add_ld=None, tp=None):
        """Timestep-dependent part.  Returns NHWC [B, H/16, W/16, out_ch]; with `add_into` (the transformer's joint
        residual buffer) the final conv's epilogue adds its result straight into the image-token rows instead
        (hidden_states + control['out'] * 1.0, lightcontrol_flux.py:506-507).  `tp`: timestep_features(prep, timestep) when the caller
        has them already."""
        B, h, w = prep["B"], prep["h"], prep["w"]
        if tp is None:
            tp = self.timestep_features(prep, timestep)
        if self.compose and "d0" in prep:
            x, h, w = _trunk(_trunk_tensors([self], h, w), prep, tp, self.groups, 0)
            return self._final(x, h, w, add_into, add_offset, add_batch_stride, add_ld)
        te = self.time_embedding
        e1 = ops.skinny_linear(tp, te.linear_1.weight, te.linear_1.bias, act_out=ACT_SILU)
        emb = ops.skinny_linear(e1, te.linear_2.weight, te.linear_2.bias)  # [B,256] f32
        x = self._resblock_tail(self.down_res[0], prep["x0"], prep["h1"], emb, self.groups[0], h, w, 128, 128, h1_moments=prep.get("h1_moments"))
        d = self.down_sample[0].conv
        x = ops.conv2d_nhwc(x, d.packed(), d.bias, h, w, 128, 128, 3, 3, 2, 1)
        h, w = h // 2, w // 2
        r = self.down_res[1]
        n = ops.groupnorm_nhwc(x, r.norm1.weight, r.norm1.bias, self.groups[1], 1e-6, act=ACT_SILU)
        h1 = ops.conv2d_nhwc(n, r.conv1.packed(), r.conv1.bias, h, w, 128, 256, 3, 3, 1, 1)
        x = self._resblock_tail(r, x, h1, emb, self.groups[1], h, w, 128, 256)
        d = self.down_sample[1].conv
        x = ops.conv2d_nhwc(x, d.packed(), d.bias, h, w, 256, 256, 3, 3, 2, 1)
        h, w = h // 2, w // 2
        m = self.mid_convs[0]
        y = ops.conv2d_nhwc(x, m[0].packed(), m[0].bias, h, w, 256, 256, 3, 3, 1, 1, act=ACT_RELU)
        y = ops.groupnorm_nhwc(y, m[2].weight, m[2].bias, 8, 1e-5)
        y = ops.conv2d_nhwc(y, m[3].packed(), m[3].bias, h, w, 256, 256, 3, 3, 1, 1)
        x = ops.groupnorm_nhwc(y, m[4].weight, m[4].bias, 8, 1e-5, post_add=x)  # mid_convs[0](x) + x (:744)
        return self._final(x, h, w, add_into, add_offset, add_batch_stride, add_ld)

    def _final(self, x, h, w, add_into=None, add_offset=0, add_batch_stride=None, add_ld=None):
        """mid_convs[1] (the 2x2 stride-2 projection to the transformer width, :661-668,745) on the trunk's output x [B, h, w, 256]."""
        f = self.mid_convs[1]
        cout = f.weight.shape[0]
        if add_into is not None:
            ops.conv2d_nhwc(x, f.packed(), f.bias, h, w, 256, cout, 2, 2, 2, 0, out=add_into, c_offset=add_offset,
                            c_batch_stride=add_batch_stride, ldc=add_ld, res=add_into, res_offset=add_offset,
                            res_batch_stride=add_batch_stride, ldr=add_ld)
            return None
        return ops.conv2d_nhwc(x, f.packed(), f.bias, h, w, 256, cout, 2, 2, 2, 0)

    @torch.no_grad()
    def forward(self, sample, timestep):
        prep = self.prepare_hint(sample)
        out = self.forward_nhwc(prep, timestep)
        return {"out": out.permute(0, 3, 1, 2), "scale": self.scale}  # NCHW view like the reference


def _trunk_tensors(nets, h, w):
    """The parameters the timestep-dependent trunk reads (composed form), as a dict: of ONE net as they are, of several nets stacked along a new
    leading dimension ([nets, ...], contiguous) for the grouped launches of a bank."""
    per = []
    for net in nets:
        ca, cb = net._composed(h, w)
        te, r0, r1, m = net.time_embedding, net.down_res[0], net.down_res[1], net.mid_convs[0]
        per.append(dict(
            l1w=te.linear_1.weight, l1b=te.linear_1.bias, l2w=te.linear_2.weight, l2b=te.linear_2.bias,
            t0w=r0.time_emb_proj.weight, t0b=r0.time_emb_proj.bias, n02w=r0.norm2.weight, n02b=r0.norm2.bias,
            a5=ca.w5, atop=ca.w_top, aleft=ca.w_left, acorner=ca.w_corner,
            n11w=r1.norm1.weight, n11b=r1.norm1.bias, c1w=r1.conv1.packed(), c1b=r1.conv1.bias,
            t1w=r1.time_emb_proj.weight, t1b=r1.time_emb_proj.bias, n12w=r1.norm2.weight, n12b=r1.norm2.bias,
            bsc=cb.wsc, bmap=cb.bias_map, b5=cb.w5, btop=cb.w_top, bleft=cb.w_left, bcorner=cb.w_corner,
            m0w=m[0].packed(), m0b=m[0].bias, g2w=m[2].weight, g2b=m[2].bias, m3w=m[3].packed(), m3b=m[3].bias, g4w=m[4].weight, g4b=m[4].bias))
    if len(nets) == 1:
        return per[0]
    return {k: torch.stack([d[k] for d in per]).contiguous() for k in per[0]}


def _trunk(T, prep, tp, groups, wg):
    """Timestep-dependent trunk of ControlNeXtModel.forward up to (not including) mid_convs[1], composed form (module docstring).  T:
    _trunk_tensors of one net (wg = 0) or of a bank's nets (wg = samples per net: the batch of `prep` is (net, sample)-major and every launch
    takes the nets' parameters as grouped weights).  prep: prepare_hint's dict; tp: timestep_features.  Returns (x [B, h, w, 256], h, w)."""
    B, h, w = prep["B"], prep["h"], prep["w"]
    g = dict(w_group=wg)

    def lin(x, wt, b, **kw):
        return ops.skinny_linear_grouped(x, wt, b, rows=B, **kw) if wg else ops.skinny_linear(x, wt, b, **kw)

    e1 = lin(tp, T["l1w"], T["l1b"], act_out=ACT_SILU)
    emb = lin(e1, T["l2w"], T["l2b"])                                   # [nets * B, 256] f32
    tproj = lin(emb, T["t0w"], T["t0b"], act_in=ACT_SILU)
    n = ops.groupnorm_nhwc_from_moments(prep["h1"], prep["h1_moments"], T["n02w"], T["n02b"], groups[0], 1e-6, act=ACT_SILU, pre_add=tproj, **g)
    x = _composed_apply(T["a5"], T["atop"], T["aleft"], T["acorner"], 128, 128, n, h, w, prep["d0"], w_group=wg)   # = down_sample[0](conv2(n) + x0)
    h, w = h // 2, w // 2
    n = ops.groupnorm_nhwc(x, T["n11w"], T["n11b"], groups[1], 1e-6, act=ACT_SILU, **g)
    h1 = ops.conv2d_nhwc(n, T["c1w"], T["c1b"], h, w, 128, 256, 3, 3, 1, 1, **g)
    tproj = lin(emb, T["t1w"], T["t1b"], act_in=ACT_SILU)
    n = ops.groupnorm_nhwc(h1, T["n12w"], T["n12b"], groups[1], 1e-6, act=ACT_SILU, pre_add=tproj, **g)
    bmap = prep["bmap"] if wg else T["bmap"]                              # bank: the nets' bias maps, one copy per sample
    t1 = ops.conv2d_nhwc(x, T["bsc"], None, h, w, 128, 256, 3, 3, 2, 1, res=bmap, res_batch_stride=(h // 2) * (w // 2) * 256 if wg else 0, **g)
    x = _composed_apply(T["b5"], T["btop"], T["bleft"], T["bcorner"], 256, 256, n, h, w, t1, w_group=wg)       # = down_sample[1](conv2(n) + conv_shortcut(x))
    h, w = h // 2, w // 2
    y = ops.conv2d_nhwc(x, T["m0w"], T["m0b"], h, w, 256, 256, 3, 3, 1, 1, act=ACT_RELU, **g)
    y = ops.groupnorm_nhwc(y, T["g2w"], T["g2b"], 8, 1e-5, **g)
    y = ops.conv2d_nhwc(y, T["m3w"], T["m3b"], h, w, 256, 256, 3, 3, 1, 1, **g)
    x = ops.groupnorm_nhwc(y, T["g4w"], T["g4b"], 8, 1e-5, post_add=x, **g)   # mid_convs[0](x) + x (:744)
    return x, h, w


class ControlNeXtBank:
    """The control nets of a LightControl step (lightcontrol_flux.py:504-507: control_nets[i] behind double block i, every one fed the same
    guided_hint and timestep) evaluated as ONE batch: their trunks do not depend on the transformer's state, only their last convolution adds
    into it.  The batch is (net, sample)-major and each launch takes the nets' parameters as grouped weights (include/x2i.h:
    x2i_gemm_args.w_group, x2i_groupnorm_*_grouped_bf16, x2i_skinny_linear_grouped): 19 x fewer launches, and the persistent convolution
    kernels get tile lists of many rounds instead of exactly one.  A sample's results are bit-identical to ControlNeXtModel.forward_nhwc's
    (the same kernels on the same items; tests/test_lightcontrol_gpu.py)."""

    @staticmethod
    def eligible(nets):
        import os
        return (len(nets) > 1 and os.environ.get("X2I_CONTROL_BANK", "1") != "0" and all(isinstance(n, ControlNeXtModel) and n.compose for n in nets)
                and len({n.groups for n in nets}) == 1 and len({n.time_embedding.linear_1.weight.device for n in nets}) == 1)

    @torch.no_grad()
    def __init__(self, nets, guided_hint):
        self.nets = list(nets)
        preps = [n.prepare_hint(guided_hint) for n in self.nets]
        p0 = preps[0]
        self.B, h, w = p0["B"], p0["h"], p0["w"]
        self.T = _trunk_tensors(self.nets, h, w)
        self.prep = dict(h=h, w=w, B=self.B, round_bf16=p0["round_bf16"], x0=p0["x0"][:0],
                         h1=torch.cat([p["h1"] for p in preps]), h1_moments=torch.cat([p["h1_moments"] for p in preps]),
                         d0=torch.cat([p["d0"] for p in preps]), bmap=self.T["bmap"].repeat_interleave(self.B, dim=0).flatten(0, 1).contiguous())
        self.groups = self.nets[0].groups

    @torch.no_grad()
    def trunk(self, tp):
        """x bf16 [nets * B, H/16 * 2, W/16 * 2, 256]: net i's trunk output for sample b at row i * B + b."""
        return _trunk(self.T, self.prep, tp, self.groups, self.B)


def make_control_fn(control_nets, guided_hint):
    """Callable(i, timestep_x1000, X, St, S, D) used by FluxTransformer2DModel.denoise: adds control net i's output into the
    image rows of the joint residual buffer.  The t-independent prefix of every net is computed once per hint; the timestep-dependent trunks of
    all nets run as one batch at the first call of a step (ControlNeXtBank; X2I_CONTROL_BANK=0: net by net, A/B)."""
    nets = list(control_nets)
    shared = {"t": None, "tp": None, "x": None}   # the sinusoidal timestep features are the same for all nets of a step: one evaluation per step

    if ControlNeXtBank.eligible(nets):
        bank = ControlNeXtBank(nets, guided_hint)
        B = bank.B

        def fn(i, t1000, X, St, S, D):
            if i >= len(nets):
                return False
            if shared["t"] is not t1000:
                shared["t"] = t1000
                shared["x"] = bank.trunk(ControlNeXtModel.timestep_features(bank.prep, t1000))
            x, h, w = shared["x"]
            nets[i]._final(x[i * B:(i + 1) * B], h, w, add_into=X, add_offset=St * D, add_batch_stride=S * D, add_ld=D)
            if i == len(nets) - 1:
                shared["x"] = shared["t"] = None    # (the step's trunk outputs are not kept alive beyond their last reader)
            return True

        return fn

    preps = [n.prepare_hint(guided_hint) for n in nets]

    def fn(i, t1000, X, St, S, D):
        if i >= len(nets):
            return False
        if shared["t"] is not t1000:
            shared["t"], shared["tp"] = t1000, ControlNeXtModel.timestep_features(preps[0], t1000)
        nets[i].forward_nhwc(preps[i], t1000, add_into=X, add_offset=St * D, add_batch_stride=S * D, add_ld=D, tp=shared["tp"])
        return True

    return fn


class FluxTransformer2DModel(_BaseFlux):
    """The reference's modified transformer (lightcontrol_flux.py:208-553): extra `guided_hint`, `control_nets`
    arguments; return_dict=False returns the bare tensor."""

    def forward(self, hidden_states, encoder_hidden_states=None, pooled_projections=None, timestep=None, img_ids=None,
                txt_ids=None, guidance=None, joint_attention_kwargs=None, guided_hint=None, control_nets=None,
                return_dict: bool = True):
        if control_nets is None:
            raise TypeError("object of type 'NoneType' has no len()  (pass control_nets=[]; lightcontrol_flux.py:504)")
        state = self.prepare_conditioning(encoder_hidden_states, pooled_projections, txt_ids, img_ids, guidance)
        control = make_control_fn(control_nets, guided_hint.to(self.device)) if len(control_nets) else None
        out = self.denoise(state, hidden_states, timestep, control=control)
        if not return_dict:
            return out
        from .flux import Transformer2DModelOutput
        return Transformer2DModelOutput(sample=out)


class LightControlSampler:
    """Row L: N-step Euler sampling with the instruction-edit branch (FLUX.1-dev schedule by default: dynamic shift,
    guidance embedding), i.e. the inference counterpart of train_lightcontrol.py:732-743."""

    def __init__(self, transformer, control_nets, scheduler=None):
        from .pipeline import FluxPipeline, FlowMatchEulerDiscreteScheduler
        self.pipeline = FluxPipeline(transformer, scheduler or FlowMatchEulerDiscreteScheduler(shift=3.0, use_dynamic_shifting=True),
                                     control_nets=control_nets)

    def __call__(self, prompt_embeds, pooled_prompt_embeds, guided_hint, num_inference_steps=20, guidance_scale=3.5,
                 height=1024, width=1024, latents=None, generator=None, use_graph=False):
        return self.pipeline(prompt_embeds=prompt_embeds, pooled_prompt_embeds=pooled_prompt_embeds,
                             num_inference_steps=num_inference_steps, guidance_scale=guidance_scale, height=height,
                             width=width, output_type="latent", latents=latents, generator=generator,
                             guided_hint=guided_hint, use_graph=use_graph).images
