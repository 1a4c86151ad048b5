"""LightControl / ControlNeXt on the HIP path: conv + GroupNorm kernels against torch fp32 primitives, the hint encoder
against the reference-generated golden, the transformer with control injection against the reference's own
lightcontrol_flux.py output (Row L of SURVEY.md section 8), and the N-step sampler against the oracle sampler."""
import pytest
import torch
import torch.nn.functional as F

from oracle import flux as OF
from oracle import sampler as OS
from tests.util import golden, rel_l2, seeded

pytestmark = pytest.mark.gpu
DEV = "cuda"


def bf(x):
    return x.to(torch.bfloat16)


def rb(x):
    return x.to(torch.bfloat16).float()


@pytest.mark.parametrize("Cin,Cout,k,s,p,H,W", [(64, 64, 3, 1, 1, 20, 28), (64, 128, 3, 1, 1, 16, 16), (128, 128, 3, 2, 1, 32, 48),
                                                (128, 256, 1, 1, 0, 18, 10), (256, 320, 2, 2, 0, 16, 24), (256, 256, 3, 1, 1, 8, 12)])
def test_conv2d_implicit_gemm(Cin, Cout, k, s, p, H, W):
    from x2i_amd import ops
    B = 2
    x = bf(seeded((B, Cin, H, W), 1))
    w = bf(seeded((Cout, Cin, k, k), 2) / (Cin * k * k) ** 0.5)
    b = bf(seeded((Cout,), 3))
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(DEV)
    out = ops.conv2d_nhwc(xn, wp, b.to(DEV), H, W, Cin, Cout, k, k, s, p)
    ref = F.conv2d(x.float(), w.float(), b.float(), stride=s, padding=p)
    assert out.shape == (B, ref.shape[2], ref.shape[3], Cout)
    assert rel_l2(out.permute(0, 3, 1, 2), ref) < 1e-2
    # epilogue: per-sample bias2 + ReLU, and residual add
    b2 = seeded((B, Cout), 4)
    out = ops.conv2d_nhwc(xn, wp, b.to(DEV), H, W, Cin, Cout, k, k, s, p, act=ops.ACT_RELU, bias2=b2.to(DEV))
    assert rel_l2(out.permute(0, 3, 1, 2), F.relu(ref + b2[:, :, None, None])) < 1e-2
    res = bf(seeded(tuple(out.shape), 5))
    out = ops.conv2d_nhwc(xn, wp, b.to(DEV), H, W, Cin, Cout, k, k, s, p, res=res.to(DEV))
    assert rel_l2(out.permute(0, 3, 1, 2), ref + res.float().permute(0, 3, 1, 2)) < 1e-2


def test_conv_stem():
    from x2i_amd import ops
    x = bf(torch.rand((2, 3, 36, 52), generator=torch.Generator().manual_seed(1)) * 2 - 1)
    w, b = seeded((64, 3, 3, 3), 2) / 27 ** 0.5, seeded((64,), 3)
    out = ops.conv_stem(x.permute(0, 2, 3, 1).contiguous().to(DEV), w.permute(0, 2, 3, 1).contiguous().to(DEV), b.to(DEV), 64)
    ref = F.conv2d(x.float(), w, b, stride=2, padding=1)
    assert rel_l2(out.permute(0, 3, 1, 2), ref) < 5e-3


@pytest.mark.parametrize("C,G,act", [(64, 2, 4), (128, 4, 3), (256, 8, 0), (128, 2, 4)])
def test_groupnorm_nhwc(C, G, act):
    from x2i_amd import ops
    B, H, W = 2, 14, 18
    x = bf(seeded((B, C, H, W), 1, 2.0) + 0.3)
    w, b = bf(1 + 0.1 * seeded((C,), 2)), bf(0.1 * seeded((C,), 3))
    pre, post = seeded((B, C), 4), bf(seeded((B, C, H, W), 5))
    fn = {0: lambda t: t, 3: F.silu, 4: F.relu}[act]
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    out = ops.groupnorm_nhwc(xn, w.to(DEV), b.to(DEV), G, 1e-6, act=act)
    assert rel_l2(out.permute(0, 3, 1, 2), fn(F.group_norm(x.float(), G, w.float(), b.float(), 1e-6))) < 5e-3
    out = ops.groupnorm_nhwc(xn, w.to(DEV), b.to(DEV), G, 1e-5, act=act, pre_add=pre.to(DEV),
                             post_add=post.permute(0, 2, 3, 1).contiguous().to(DEV))
    ref = fn(F.group_norm(x.float() + pre[:, :, None, None], G, w.float(), b.float(), 1e-5)) + post.float()
    assert rel_l2(out.permute(0, 3, 1, 2), ref) < 5e-3


@pytest.mark.parametrize("C,G,H,W", [(128, 4, 38, 50), (256, 8, 16, 16), (128, 32, 20, 12), (512, 32, 9, 7)])
def test_groupnorm_from_cached_channel_moments(C, G, H, W):
    """ResnetBlock2D norm2 on conv1(...) + time-embedding term (diffusers ResnetBlock2D; lightcontrol_flux.py:620-640): with conv1's output
    fixed per hint, its per-channel moments (x2i_groupnorm_moments_f32) give the group statistics of x + v for every v without a pass
    over x.  Against torch's group_norm on x + v, for several v on ONE moment set, and against the two-pass kernel."""
    from x2i_amd import ops
    B = 3
    x = bf(seeded((B, C, H, W), 11, 1.5) + 0.4)
    w, b = bf(1 + 0.1 * seeded((C,), 12)), bf(0.1 * seeded((C,), 13))
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    mom = ops.groupnorm_moments(xn)
    ref_m = torch.stack([x.float().sum((2, 3)), (x.float() ** 2).sum((2, 3))], -1)
    assert torch.allclose(mom.cpu(), ref_m, rtol=2e-5, atol=1e-3)
    for k in range(3):
        pre = seeded((B, C), 20 + k, 0.7 * k)        # k = 0: no shift at all
        out = ops.groupnorm_nhwc_from_moments(xn, mom, w.to(DEV), b.to(DEV), G, 1e-6, act=ops.ACT_SILU, pre_add=pre.to(DEV))
        ref = F.silu(F.group_norm(x.float() + pre[:, :, None, None], G, w.float(), b.float(), 1e-6))
        assert rel_l2(out.permute(0, 3, 1, 2), ref) < 5e-3
        two = ops.groupnorm_nhwc(xn, w.to(DEV), b.to(DEV), G, 1e-6, act=ops.ACT_SILU, pre_add=pre.to(DEV))
        assert rel_l2(out, two) < 2e-3


def _load_cnext(seed, out_channels=3072):
    from x2i_amd.lightcontrol import ControlNeXtModel
    sd = OF.random_controlnext_state_dict(seed=seed, out_channels=out_channels)
    m = ControlNeXtModel(device=DEV, control_out_channels=out_channels)
    m.load_state_dict({k: bf(v) for k, v in sd.items()}, strict=True)
    return m, sd


def test_controlnext_vs_reference_golden():
    t, meta = golden("controlnext_full")
    m, sd = _load_cnext(meta["weight_seed"])
    o = m(t["hint"].to(DEV), t["timestep"].to(DEV))
    assert o["scale"] == meta["scale"] and o["out"].shape == t["out"].shape
    assert rel_l2(o["out"], t["out"]) < 3e-2  # reference ran fp32 weights
    ref = OF.controlnext_forward({k: rb(v) for k, v in sd.items()}, "", rb(t["hint"]), t["timestep"])
    assert rel_l2(o["out"], ref["out"]) < 2e-2


@pytest.mark.parametrize("H,W", [(64, 96), (128, 64)])
def test_composed_conv_chains_equal_the_chained_form_incl_borders(H, W):
    """Round 4: ResnetBlock2D.conv2 (+ shortcut) -> Downsample2D.conv evaluated as ONE 5x5 stride-2 convolution each (lightcontrol.py
    `_Composed`).  Zero padding of the INTERMEDIATE is reproduced exactly by the first-row / first-column / first-pixel corrections, so
    the composed form must agree with the chained form everywhere -- border outputs included -- up to bf16 rounding, and with the fp32
    oracle at the same tolerance as the chained form.  Non-square hint, batch 2, two timesteps on one prepared hint."""
    m, sd = _load_cnext(31)
    g = torch.Generator().manual_seed(5)
    hint = (torch.rand((2, 3, H, W), generator=g) * 2 - 1)
    outs = {}
    for compose in (False, True):
        m.compose = compose
        prep = m.prepare_hint(hint.to(DEV))
        assert ("d0" in prep) == compose
        outs[compose] = [m.forward_nhwc(prep, torch.tensor([t])).float().cpu() for t in (0.25, 0.9)]
    for k, tval in enumerate((0.25, 0.9)):
        a, b = outs[True][k], outs[False][k]
        e = rel_l2(a, b)
        # the control output grid is (H/16, W/16): its first row / column / pixel descend from the corrected border outputs of both chains
        eb = max(rel_l2(a[:, 0], b[:, 0]), rel_l2(a[:, :, 0], b[:, :, 0]), rel_l2(a[:, 0, 0], b[:, 0, 0]))
        ref = OF.controlnext_forward({kk: rb(v) for kk, v in sd.items()}, "", rb(hint), torch.tensor([tval]).expand(2))["out"].permute(0, 2, 3, 1)
        print(f"H={H} W={W} t={tval}: composed vs chained rel-L2 {e:.2e} (first row / column / pixel {eb:.2e}); vs fp32 oracle: composed "
              f"{rel_l2(a, ref):.2e}, chained {rel_l2(b, ref):.2e}")
        assert e < 1.5e-2 and eb < 2e-2
        assert rel_l2(a, ref) < 2e-2 and rel_l2(a, ref) < 1.5 * rel_l2(b, ref) + 2e-3
    m.compose = True


def test_composed_tap_weights_are_the_convolution_of_the_kernels():
    """_compose_taps against torch: composing a 3x3 stride-2 conv with a 3x3 conv (interior pixels, where padding plays no part)."""
    from x2i_amd.lightcontrol import _compose_taps
    g = torch.Generator().manual_seed(9)
    wd = bf(torch.randn((128, 3, 3, 128), generator=g) * 0.05)
    wc = bf(torch.randn((128, 3, 3, 128), generator=g) * 0.05)
    w5 = _compose_taps(wd.to(DEV), wc.to(DEV)).float().cpu().view(128, 5, 5, 128)
    x = torch.randn((1, 128, 13, 13), generator=g)
    y2 = F.conv2d(F.conv2d(x, wc.float().permute(0, 3, 1, 2)), wd.float().permute(0, 3, 1, 2), stride=2)
    y1 = F.conv2d(x, w5.permute(0, 3, 1, 2), stride=2)
    assert y1.shape == y2.shape and rel_l2(y1, y2) < 4e-3          # one bf16 rounding of the composed taps


def test_transformer_with_control_vs_reference_golden():
    """The reference's own FluxTransformer2DModel.forward(guided_hint=, control_nets=) output (tests/golden)."""
    from x2i_amd.lightcontrol import FluxTransformer2DModel
    t, meta = golden("flux_tiny_dev_control")
    cfg = meta["cfg"]
    sd = OF.random_flux_state_dict(cfg, seed=meta["weight_seed"], std=meta["weight_std"])
    m = FluxTransformer2DModel(**cfg, device=DEV)
    m.load_state_dict({k: bf(v) for k, v in sd.items()}, strict=True)
    nets, csds = [], []
    for s in meta["control_seeds"]:
        n, csd = _load_cnext(s, meta["control_out_channels"])
        nets.append(n)
        csds.append({k: rb(v) for k, v in csd.items()})
    kw = dict(hidden_states=t["hidden"].to(DEV), encoder_hidden_states=t["enc"].to(DEV), pooled_projections=t["pooled"].to(DEV),
              timestep=t["timestep"].to(DEV), img_ids=t["img_ids"].to(DEV), txt_ids=t["txt_ids"].to(DEV),
              guidance=t["guidance"].to(DEV), guided_hint=t["hint"].to(DEV))
    out = m(**kw, control_nets=nets, return_dict=False)
    assert torch.is_tensor(out) and out.shape == t["out"].shape  # bare tensor, as lightcontrol_flux.py:549-550
    assert rel_l2(out, t["out"]) < 3e-2
    ref = OF.flux_forward({k: rb(v) for k, v in sd.items()}, cfg, rb(t["hidden"]), rb(t["enc"]), rb(t["pooled"]), t["timestep"],
                          t["img_ids"], t["txt_ids"], guidance=t["guidance"], guided_hint=rb(t["hint"]), control_sds=csds)
    assert rel_l2(out, ref) < 2e-2
    out0 = m(**kw, control_nets=[], return_dict=False)
    assert rel_l2(out0, out) > 1e-3  # the control branch is really injected
    with pytest.raises(TypeError):
        m(**kw, control_nets=None)


def test_lightcontrol_sampler_vs_oracle():
    from x2i_amd.lightcontrol import FluxTransformer2DModel, LightControlSampler
    t, meta = golden("flux_tiny_dev_control")
    cfg = meta["cfg"]
    sd = OF.random_flux_state_dict(cfg, seed=meta["weight_seed"], std=meta["weight_std"])
    m = FluxTransformer2DModel(**cfg, device=DEV)
    m.load_state_dict({k: bf(v) for k, v in sd.items()}, strict=True)
    nets, csds = [], []
    for s in meta["control_seeds"]:
        n, csd = _load_cnext(s, meta["control_out_channels"])
        nets.append(n)
        csds.append({k: rb(v) for k, v in csd.items()})
    sampler = LightControlSampler(m, nets)
    pe, pooled, hint = bf(t["enc"]), bf(t["pooled"]), bf(t["hint"])
    noise = bf(OS.pack_latents(torch.randn((2, 16, 16, 24), generator=torch.Generator().manual_seed(3))))
    got = sampler(pe.to(DEV), pooled.to(DEV), hint.to(DEV), num_inference_steps=3, guidance_scale=3.5, height=128, width=192,
                  latents=noise.to(DEV))
    # oracle, dev schedule (dynamic shift), bf16 timestep arithmetic reproduced with torch bf16 ops
    sdr = {k: rb(v) for k, v in sd.items()}
    lat = noise.clone()
    ts, sig = OS.flow_match_sigmas(3, OS.SCHEDULER_DEV, lat.shape[1])
    gd = torch.full([2], 3.5)
    for i, tt in enumerate(ts):
        t1000 = ((tt.expand(2).to(torch.bfloat16) / 1000) * 1000).float()
        eps = OF.flux_forward(sdr, cfg, lat.float(), pe.float(), pooled.float(), t1000 / 1000, t["img_ids"], t["txt_ids"],
                              guidance=(gd.bfloat16() * 1000).float() / 1000, guided_hint=hint.float(), control_sds=csds)
        lat = OS.euler_step(lat, eps.bfloat16(), sig[i], sig[i + 1])
    assert rel_l2(got, lat) < 5e-2
    got2 = sampler(pe.to(DEV), pooled.to(DEV), hint.to(DEV), num_inference_steps=3, height=128, width=192, latents=noise.to(DEV),
                   use_graph=True)
    assert torch.equal(got2, got)


# ---------------------------------------------------------------------------------------------------- grouped weights / ControlNeXtBank
def _opt(**kw):
    import contextlib
    from x2i_amd import _lib

    @contextlib.contextmanager
    def cm():
        old = {k: _lib.get_option(k) for k in kw}
        for k, v in kw.items():
            _lib.set_option(k, v)
        try:
            yield
        finally:
            for k, v in old.items():
                _lib.set_option(k, v)
    return cm()


@pytest.mark.parametrize("Cin,Cout,k,s,p,H,W,res,min256", [
    (128, 128, 3, 1, 1, 64, 64, True, 1),      # 512 x 128 persistent kernel (8 tiles per item)
    (128, 256, 3, 1, 1, 64, 48, False, 1),     # 256^2 persistent kernel
    (256, 256, 5, 2, 2, 64, 64, True, 1),      # ... stride 2, residual
    (128, 128, 3, 2, 1, 32, 48, True, 256),    # 128^2 kernel
    (128, 128, 1, 1, 0, 8, 8, False, 256),     # a handful of pixels per item (the corrections' shape class)
])
def test_grouped_conv_equals_the_groups_launched_one_by_one(Cin, Cout, k, s, p, H, W, res, min256):
    """x2i_gemm_args.w_group: batch item b takes W[b // w_group], bias[b // w_group]; bit-identical to one launch per group."""
    from x2i_amd import _lib, ops
    G, B = 3, 2
    torch.manual_seed(11)
    with _opt(gemm_min256=min256):
        x = bf(torch.randn(G * B, H, W, Cin)).to(DEV)
        wt = bf(torch.randn(G, Cout, k * k * Cin) * 0.05).to(DEV)
        b = bf(torch.randn(G, Cout)).to(DEV)
        OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        r = bf(torch.randn(G * B, OH, OW, Cout)).to(DEV) if res else None
        y = ops.conv2d_nhwc(x, wt, b, H, W, Cin, Cout, k, k, s, p, res=r, w_group=B)
        tile = _lib.get_option("last_gemm_tile")
        for g in range(G):
            yg = ops.conv2d_nhwc(x[g * B:(g + 1) * B], wt[g], b[g], H, W, Cin, Cout, k, k, s, p, res=None if r is None else r[g * B:(g + 1) * B])
            assert _lib.get_option("last_gemm_tile") == tile
            assert torch.equal(y[g * B:(g + 1) * B], yg), (g, tile)
    print("tile", tile)


def test_grouped_conv_in_chunks_of_whole_groups_and_of_group_parts():
    """Batches beyond the persistent kernels' one 2 GB input descriptor are launched in chunks: of whole groups, or of parts of one group."""
    from x2i_amd import _lib, ops
    H = W = 512
    Cin, Cout = 256, 128           # 128 MiB per item
    for G, B in ((6, 4), (2, 16)):   # 15 items fit one descriptor: (6, 4) -> chunks of 12 = 3 whole groups; (2, 16) -> chunks of 8 = half a group
        torch.manual_seed(5)
        if True:
            x = bf(torch.randn(G * B, H, W, Cin)).to(DEV)
            wt = bf(torch.randn(G, Cout, 9 * Cin) * 0.03).to(DEV)
            b = bf(torch.randn(G, Cout)).to(DEV)
            y = ops.conv2d_nhwc(x, wt, b, H, W, Cin, Cout, 3, 3, 1, 1, w_group=B)
            assert _lib.get_option("last_gemm_tile") == 5512
            for g in (0, G - 1):
                yg = ops.conv2d_nhwc(x[g * B:(g + 1) * B], wt[g], b[g], H, W, Cin, Cout, 3, 3, 1, 1)
                assert torch.equal(y[g * B:(g + 1) * B], yg), (G, B, g)
    del x, y
    torch.cuda.empty_cache()


def test_grouped_gemm_groupnorm_and_skinny_linear():
    from x2i_amd import _lib, ops
    G, B = 3, 2
    torch.manual_seed(3)
    if True:
        a = bf(torch.randn(G * B, 40, 128)).to(DEV)
        wt = bf(torch.randn(G, 192, 128) * 0.1).to(DEV)
        b = bf(torch.randn(G, 192)).to(DEV)
        y = ops.gemm(a, wt, b, M=40, batch=G * B, a_batch_stride=40 * 128, lda=128, w_batch_stride=192 * 128, w_group=B)
        for g in range(G):
            yg = ops.gemm(a[g * B:(g + 1) * B], wt[g], b[g], M=40, batch=B, a_batch_stride=40 * 128, lda=128)
            assert torch.equal(y[g * B:(g + 1) * B], yg)
        x = bf(torch.randn(G * B, 20, 12, 128)).to(DEV)
        gw, gb = bf(torch.randn(G, 128)).to(DEV), bf(torch.randn(G, 128)).to(DEV)
        pre = torch.randn(G * B, 128).to(DEV)
        post = bf(torch.randn(G * B, 20, 12, 128)).to(DEV)
        mom = ops.groupnorm_moments(x)
        y = ops.groupnorm_nhwc(x, gw, gb, 4, 1e-6, act=ops.ACT_SILU, pre_add=pre, post_add=post, w_group=B)
        ym = ops.groupnorm_nhwc_from_moments(x, mom, gw, gb, 4, 1e-6, act=ops.ACT_SILU, pre_add=pre, w_group=B)
        for g in range(G):
            sl = slice(g * B, (g + 1) * B)
            assert torch.equal(y[sl], ops.groupnorm_nhwc(x[sl], gw[g], gb[g], 4, 1e-6, act=ops.ACT_SILU, pre_add=pre[sl], post_add=post[sl]))
            assert torch.equal(ym[sl], ops.groupnorm_nhwc_from_moments(x[sl], mom[sl], gw[g], gb[g], 4, 1e-6, act=ops.ACT_SILU, pre_add=pre[sl]))
        xs = torch.randn(G * B, 256).to(DEV)
        ws, bs = bf(torch.randn(G, 320, 256) * 0.1).to(DEV), bf(torch.randn(G, 320)).to(DEV)
        y = ops.skinny_linear_grouped(xs, ws, bs, rows=B, act_in=ops.ACT_SILU)
        y1 = ops.skinny_linear_grouped(xs[:B].contiguous(), ws, bs, rows=B, act_out=ops.ACT_SILU)     # one input for every group
        for g in range(G):
            sl = slice(g * B, (g + 1) * B)
            assert torch.equal(y[sl], ops.skinny_linear(xs[sl], ws[g], bs[g], act_in=ops.ACT_SILU))
            assert torch.equal(y1[sl], ops.skinny_linear(xs[:B], ws[g], bs[g], act_out=ops.ACT_SILU))
    with pytest.raises(_lib.X2IError):
        ops.gemm(a, wt, b, M=40, batch=G * B, a_batch_stride=40 * 128, lda=128, w_group=B)      # w_group without w_batch_stride


@pytest.mark.parametrize("H,W,min256", [(64, 96, 256), (256, 256, 1)])
def test_controlnext_bank_is_bit_identical_to_the_nets_one_by_one(H, W, min256, monkeypatch):
    """ControlNeXtBank (all nets' trunks as one (net, sample)-major batch with grouped weights) against forward_nhwc net by net: the same kernels
    on the same items -- bitwise.  (256, 256, gemm_min256 = 1: the persistent convolution kernels serve the trunk, as at 1024^2.)"""
    from x2i_amd.lightcontrol import ControlNeXtBank, make_control_fn
    nets = [_load_cnext(40 + i, 256)[0] for i in range(3)]
    g = torch.Generator().manual_seed(8)
    hint = bf(torch.rand((2, 3, H, W), generator=g) * 2 - 1).to(DEV)
    assert ControlNeXtBank.eligible(nets)
    S, St, D = (H // 16) * (W // 16) + 8, 8, 256
    outs = {}
    with _opt(gemm_min256=min256):
        for bank in ("1", "0"):
            monkeypatch.setenv("X2I_CONTROL_BANK", bank)
            fn = make_control_fn(nets, hint)
            res = []
            for tv in (0.3, 0.8):
                X = torch.zeros((2, S, D), device=DEV, dtype=torch.bfloat16)
                t1000 = torch.tensor([tv * 1000, tv * 1000], device=DEV)
                for i in range(len(nets)):
                    assert fn(i, t1000, X, St, S, D)
                assert not fn(len(nets), t1000, X, St, S, D)
                res.append(X.clone())
            outs[bank] = res
    for a, b in zip(outs["1"], outs["0"]):
        assert float(a.float().abs().max()) > 0 and torch.equal(a, b)
    assert torch.equal(outs["1"][0][:, :St], torch.zeros_like(outs["1"][0][:, :St]))   # text rows untouched
