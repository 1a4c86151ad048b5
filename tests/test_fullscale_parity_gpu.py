"""Parity at the sizes the benchmarked configurations actually run ("next round" item 1):

 (a) the 256^2 full-line kernel's implicit-GEMM convolution form (gemm256.hip, CONV = true) -- forced, value-checked against
     F.conv2d in fp32 with every epilogue it serves, and compared bit for bit with the 128^2 path;
 (b) one ControlNeXt hint encoder on a 1024x1024 hint (BASELINE configs[4]) and the VAE decoder on a 128x128 latent
     (1024^2 image) against the CPU oracle;
 (c) the full-depth 19 + 38 block transformer at 512^2 (BASELINE configs[0] shape) against the fp32 CPU oracle with a stated
     drift bound, plus the LightControl step (19 control nets) at full width;
 (d) the GEMM race screen: the 128^2 and 256^2 kernels accumulate in the same order, so repeated launches with unrelated
     traffic in flight must agree bit for bit.
"""
import pytest
import torch
import torch.nn.functional as F

from oracle import flux as OF
from oracle import sampler as OS
from oracle import vae as OV
from tests.util import rel_l2, seeded

pytestmark = pytest.mark.gpu
DEV = "cuda"


def bf(x):
    return x.to(torch.bfloat16)


def rb(x):
    return x.to(torch.bfloat16).float()


@pytest.fixture
def opt():
    from x2i_amd import _lib
    _lib.load()
    saved = {}

    def set_(name, value):
        old = _lib.set_option(name, value)
        saved.setdefault(name, old)
    yield set_
    for k, v in saved.items():
        _lib.set_option(k, v)


class _LazyF32(dict):
    """bf16 state dict read as fp32 one tensor at a time (the full model is 23.8 GB in bf16; never hold an fp32 copy)."""

    def __getitem__(self, k):
        return dict.__getitem__(self, k).float()

    def get(self, k, default=None):
        return self[k] if k in self else default


# ---------------------------------------------------------------------------------------------------- (a) conv on 256^2 tiles
@pytest.mark.parametrize("Cin,Cout,k,s,p,H,W,up", [
    (256, 256, 3, 1, 1, 128, 128, False),   # ControlNeXt mid convs / VAE-like 3x3, 64 tile rows x 1 column x B
    (256, 256, 3, 2, 1, 256, 256, False),   # Downsample2D (stride 2, ragged bottom/right taps)
    (256, 3072, 2, 2, 0, 128, 128, False),  # ControlNeXt mid_convs.1 (k2 s2, 12 tile columns)
    (512, 512, 3, 1, 1, 64, 64, True),      # VAE up-block conv with the x2 nearest upsample fused into the gather
    (256, 320, 3, 1, 1, 96, 100, False),    # ragged N (320 = 256 + 64) and ragged M (9600 = 37.5 tiles)
])
def test_conv256_kernel_vs_fp32_conv(opt, Cin, Cout, k, s, p, H, W, up):
    from x2i_amd import _lib, ops
    B = 2
    x = bf(seeded((B, Cin, H, W), 1))
    w = bf(seeded((Cout, Cin, k, k), 2) / (Cin * k * k) ** 0.5)
    b = bf(seeded((Cout,), 3))
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(DEV)
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest") if up else x.float()
    ref = F.conv2d(xin, w.float(), b.float(), stride=s, padding=p)
    OH, OW = ref.shape[2], ref.shape[3]
    assert OH * OW >= 4096

    def run(**kw):
        return ops.conv2d_nhwc(xn, wp, b.to(DEV), H, W, Cin, Cout, k, k, s, p, up=up, **kw)

    out = run()
    # (5256: the persistent four-wave kernel, csrc/gemm256c.hip -- every gather that is affine in the tap; 256: the eight-wave kernel, which keeps
    # the x2 upsampling fused into the gather)
    assert _lib.get_option("last_gemm_tile") == (256 if up else 5256), "the >= 256-channel convolution must take a 256^2 kernel"
    assert out.shape == (B, OH, OW, Cout)
    assert rel_l2(out.permute(0, 3, 1, 2), ref) < 1e-2
    # epilogues the ControlNeXt / VAE graphs use on this kernel: per-sample bias2 (time embedding) + ReLU; residual add
    b2 = seeded((B, Cout), 4)
    out_b2 = run(act=ops.ACT_RELU, bias2=b2.to(DEV))
    assert rel_l2(out_b2.permute(0, 3, 1, 2), F.relu(ref + b2[:, :, None, None])) < 1e-2
    res = bf(seeded((B, OH, OW, Cout), 5))
    out_res = run(res=res.to(DEV))
    assert rel_l2(out_res.permute(0, 3, 1, 2), ref + res.float().permute(0, 3, 1, 2)) < 1e-2
    # the 128^2 implicit-GEMM kernel accumulates in the weight layout's K order; so does the 256^2 kernel with conv_korder = 0: bit-identical results
    # (its product order -- a filter row's taps back to back -- is the same sum in another order: within 2e-3)
    if not up:
        opt("conv_korder", 0)
        out_k0, out_b2_k0, out_res_k0 = run(), run(act=ops.ACT_RELU, bias2=b2.to(DEV)), run(res=res.to(DEV))
        assert _lib.get_option("last_gemm_tile") == 5256
        for a_, b_ in ((out, out_k0), (out_b2, out_b2_k0), (out_res, out_res_k0)):
            assert rel_l2(a_, b_) < 2e-3
        out, out_b2, out_res = out_k0, out_b2_k0, out_res_k0
    opt("conv256", 0)
    out128 = run()
    assert _lib.get_option("last_gemm_tile") == 128
    assert torch.equal(out128, out)
    assert torch.equal(run(act=ops.ACT_RELU, bias2=b2.to(DEV)), out_b2)
    assert torch.equal(run(res=res.to(DEV)), out_res)


# ---------------------------------------------------------------------------------------------------- (b) 1024^2 hint / image
def _load_cnext(seed):
    from x2i_amd.lightcontrol import ControlNeXtModel
    sd = OF.random_controlnext_state_dict(seed=seed)
    m = ControlNeXtModel(device=DEV)
    m.load_state_dict({k: bf(v) for k, v in sd.items()}, strict=True)
    return m, {k: rb(v) for k, v in sd.items()}


def test_controlnext_at_1024_hint_vs_oracle():
    """BASELINE configs[4]: guided_hint [B,3,1024,1024] in [-1,1] -> [B,3072,64,64] (lightcontrol_flux.py:708-749); this is the
    size at which the hint encoder's 256-wide convolutions dispatch to the 256^2 kernel."""
    from x2i_amd import _lib
    m, sdr = _load_cnext(17)
    hint = bf(torch.rand((2, 3, 1024, 1024), generator=torch.Generator().manual_seed(5)) * 2 - 1)
    t = torch.tensor([752.0])
    o = m(hint.to(DEV), t.to(DEV))
    assert _lib.get_option("last_gemm_tile") == 5256  # mid_convs.1: 256 -> 3072, k2 s2 on the 128x128 map, on the persistent four-wave conv kernel
    ref = OF.controlnext_forward(sdr, "", hint.float(), t)
    assert o["out"].shape == ref["out"].shape == (2, 3072, 64, 64) and o["scale"] == ref["scale"] == 1.0
    assert rel_l2(o["out"], ref["out"]) < 2e-2


def test_vae_decode_1024_image_vs_oracle():
    """The FLUX decoder at its real widths on a 128x128 latent -> 1024x1024 image (infer/inference_qwenvl.py:213-214)."""
    from x2i_amd.vae import AutoencoderKL
    sd = OV.random_vae_decoder_state_dict(seed=7)
    vae = AutoencoderKL(device=DEV)
    vae.load_state_dict({k: bf(v) for k, v in sd.items()}, strict=True)
    z = seeded((1, 16, 128, 128), 8)
    img = vae.decode(z.to(DEV), return_dict=False)[0]
    ref = OV.vae_decode({k: rb(v) for k, v in sd.items()}, rb(z))
    assert img.shape == ref.shape == (1, 3, 1024, 1024)
    assert rel_l2(img, ref) < 3e-2


# ---------------------------------------------------------------------------------------------------- (c) full depth
@pytest.fixture(scope="module")
def full_dev_model():
    from x2i_amd.lightcontrol import FluxTransformer2DModel
    return FluxTransformer2DModel(guidance_embeds=True, device=DEV).init_random_(seed=9, std=0.02)


# e4m3 operands (3 mantissa bits) on 72 % / 97 % of the GEMM FLOPs: measured 1.04e-1 (mlp) / 1.11e-1 (all) after 57 random-weight
# blocks, against 1.47e-2 for bf16 -- the stated drift of the opt-in fp8 lines
FP8_FULL_DEPTH_TOL = 0.15


def test_full_depth_19_38_forward_at_512_vs_oracle(full_dev_model):
    """All 57 blocks (11.9 B parameters) at BASELINE configs[0]'s shape -- 512x512, S = 512 + 1024, B = 1 -- against the fp32
    CPU oracle evaluated on the same bf16-rounded weights.  Stated drift bound: the bf16 residual stream + bf16 GEMM inputs
    accumulate rounding over 57 blocks; rel-L2 of the final noise prediction <= 3e-2 (measured 1.47e-2 on MI355X), and no worse than
    3x the single-block error budget used elsewhere (2e-2 for 1+1 blocks)."""
    m = full_dev_model
    sd = _LazyF32({k: v.detach().to("cpu") for k, v in m.state_dict().items()})
    cfg = dict(OF.DEFAULT_CFG, guidance_embeds=True)
    hidden, enc, pooled = bf(seeded((1, 1024, 64), 1)), bf(seeded((1, 512, 4096), 2)), bf(seeded((1, 768), 3))
    ts, gd = torch.tensor([0.5]), torch.tensor([3.5])
    img_ids, txt_ids = OS.prepare_latent_image_ids(32, 32), torch.zeros(512, 3)
    out = m(hidden_states=hidden.to(DEV), encoder_hidden_states=enc.to(DEV), pooled_projections=pooled.to(DEV),
            timestep=ts.to(DEV), img_ids=img_ids.to(DEV), txt_ids=txt_ids.to(DEV), guidance=gd.to(DEV), control_nets=[],
            guided_hint=None, return_dict=False)
    # the reference multiplies timestep / guidance by 1000 in the caller's dtype (lightcontrol_flux.py:447,449): in its bf16 run
    # 3.5 * 1000 rounds to 3504, so the fp32 oracle is handed the value the bf16 run really embeds (0.5 * 1000 = 500 is exact)
    gd_eff = (gd.bfloat16() * 1000).float() / 1000
    ref = OF.flux_forward(sd, cfg, hidden.float(), enc.float(), pooled.float(), ts, img_ids, txt_ids, guidance=gd_eff)
    err = rel_l2(out, ref)
    print(f"full-depth 19+38 @512^2 rel-L2 vs fp32 oracle: {err:.3e}")
    assert out.shape == (1, 1024, 64) and torch.isfinite(out.float()).all()
    assert err < 3e-2
    # the opt-in e4m3 configurations at full depth, against the same oracle output (stated drift bound of the fp8 lines in bench.py)
    kw = dict(hidden_states=hidden.to(DEV), encoder_hidden_states=enc.to(DEV), pooled_projections=pooled.to(DEV), timestep=ts.to(DEV),
              img_ids=img_ids.to(DEV), txt_ids=txt_ids.to(DEV), guidance=gd.to(DEV), control_nets=[], guided_hint=None, return_dict=False)
    try:
        for mode in ("mlp", "all"):
            m.enable_fp8(mode)
            e8 = rel_l2(m(**kw), ref)
            print(f"full-depth 19+38 @512^2 fp8[{mode}] rel-L2 vs fp32 oracle: {e8:.3e}")
            assert e8 < FP8_FULL_DEPTH_TOL
    finally:
        m.enable_fp8(None)
    assert torch.equal(m(**kw), out)  # back on the bf16 path, bit for bit


def test_lightcontrol_step_full_width_19_nets_1024_vs_oracle_prefix(full_dev_model):
    """BASELINE configs[4] at its real size on the GPU: 19 ControlNeXt nets on a 1024^2 hint injected after the 19 double
    blocks.  The CPU oracle cannot afford the whole 1024^2 step, so the check is split: (1) every net's injected tensor equals
    the oracle's ControlNeXt output at 1024^2 (three nets sampled; the others share the code path and differ only in weights);
    (2) the full step with the 19 nets equals the same step computed with the control tensors added by a separate residual
    pass -- i.e. the fused add-into-X epilogue of the last conv is value-checked at full size; (3) batch independence."""
    from x2i_amd.lightcontrol import make_control_fn
    m = full_dev_model
    B = 2
    nets, sds = zip(*[_load_cnext(100 + i) for i in range(19)])
    g = torch.Generator(device=DEV).manual_seed(4)
    hint = (torch.rand((B, 3, 1024, 1024), device=DEV, generator=g) * 2 - 1).bfloat16()
    x = dict(hidden=torch.randn((B, 4096, 64), device=DEV, generator=g).bfloat16(),
             enc=torch.randn((B, 512, 4096), device=DEV, generator=g).bfloat16(),
             pooled=torch.randn((B, 768), device=DEV, generator=g).bfloat16(),
             t=torch.tensor([0.75, 0.25], device=DEV).bfloat16(), gd=torch.full((B,), 3.5, device=DEV),
             img_ids=OS.prepare_latent_image_ids(64, 64).to(DEV), txt_ids=torch.zeros(512, 3, device=DEV))

    def step(sl=slice(None), control_nets=nets):
        return m(hidden_states=x["hidden"][sl], encoder_hidden_states=x["enc"][sl], pooled_projections=x["pooled"][sl],
                 timestep=x["t"][sl], img_ids=x["img_ids"], txt_ids=x["txt_ids"], guidance=x["gd"][sl], guided_hint=hint[sl],
                 control_nets=list(control_nets), return_dict=False)

    full = step()
    assert full.shape == (B, 4096, 64) and torch.isfinite(full.float()).all()
    # (1) the tensors that get injected, against the oracle at the real hint size
    t1000 = (x["t"] * 1000).float()
    for i in (0, 9, 18):
        got = nets[i](hint[:1], t1000[:1])["out"]
        ref = OF.controlnext_forward(sds[i], "", hint[:1].float().cpu(), t1000[:1].cpu())["out"]
        assert rel_l2(got, ref) < 2e-2, i
    # (2) fused add-into-X epilogue == unfused add of the stand-alone control outputs (one bf16 rounding apart per block)
    D, St, Si = 3072, 512, 4096
    state = m.prepare_conditioning(x["enc"], x["pooled"], x["txt_ids"], x["img_ids"], x["gd"])

    def unfused(i, t1000_, X, St_, S_, D_):
        out = nets[i](hint, t1000_)["out"]  # [B, 3072, 64, 64]
        X[:, St_:] += out.flatten(2).transpose(1, 2).to(X.dtype)

    ref_unfused = m.denoise(state, x["hidden"], x["t"], control=unfused)
    fused = m.denoise(state, x["hidden"], x["t"], control=make_control_fn(list(nets), hint))
    assert torch.equal(fused, full)
    assert rel_l2(fused, ref_unfused) < 1.5e-2
    assert rel_l2(step(control_nets=[]), full) > 1e-3  # the control branch really changes the result
    # (3) batch independence, bit for bit
    assert torch.equal(step(slice(1, 2))[0], full[1])


# ---------------------------------------------------------------------------------------------------- (d) race screen
def test_gemm_kernel_forms_agree_bit_for_bit_under_repeated_launches(opt):
    """Race screen (was tools/stress_gemm.py): the 128^2 kernel and the pipelined 256^2 full-line kernel accumulate every output
    in the same order, so any disagreement -- on any launch, with unrelated HBM traffic between launches -- is a pipeline
    race (an LDS buffer read before its DMA landed or restaged before its last read)."""
    from x2i_amd import _lib, ops
    torch.manual_seed(0)
    noise = torch.randn(32 << 20, device=DEV)
    shapes = [(4096, 3072, 3072), (4608, 3072, 15360), (2304, 9216, 3072), (1280, 768, 64), (777, 520, 320), (8192, 12288, 3072)]
    bad = []
    for it in range(3):
        for (M, N, K) in shapes:
            A = torch.randn(M, K, device=DEV).bfloat16()
            W = (torch.randn(N, K, device=DEV) * 0.03).bfloat16()
            b = torch.randn(N, device=DEV).bfloat16()
            res = torch.randn(M, N, device=DEV).bfloat16()
            gate = torch.randn(1, N, device=DEV)
            opt("gemm_tile", 128)
            ref = ops.gemm(A, W, b, act=1)
            r128 = res.clone()
            ops.gemm(A, W, b, out=r128, res=r128, gate=gate)
            assert _lib.get_option("last_gemm_tile") == 128
            opt("gemm_tile", 256)
            for rep in range(3):
                noise.mul_(1.0001)  # unrelated traffic between launches
                out = ops.gemm(A, W, b, act=1)
                assert _lib.get_option("last_gemm_tile") == 256
                if not torch.equal(out, ref):
                    bad.append((it, M, N, K, rep, float((out.float() - ref.float()).abs().max())))
            r256 = res.clone()
            ops.gemm(A, W, b, out=r256, res=r256, gate=gate)
            if not torch.equal(r128, r256):
                bad.append((it, M, N, K, "gated"))
            opt("gemm_tile", 0)
            auto = ops.gemm(A, W, b, act=1)  # automatic choice, incl. the peeled-tail double launch
            if not torch.equal(auto, ref):
                bad.append((it, M, N, K, "auto"))
    assert not bad, bad


def test_full_depth_four_step_latents_and_decoded_psnr_bf16_and_fp8(full_dev_model):
    """the whole sampling chain at full depth and width -- 19 + 38 blocks, 4 Euler steps, 512 x 512, B = 1 -- on the
    bf16 path and on both opt-in e4m3 configurations against the fp32 CPU oracle (same bf16-rounded weights): rel-L2 of the final packed
    latents and PSNR of the images decoded from them by the real-width VAE (random weights; the oracle's latents decoded by the same
    HIP decoder, so that the figure isolates the transformer's arithmetic).  Stated bounds: bf16 <= 5e-2;
    e4m3 operands carry 3 mantissa bits -- about 3.7e-2 per GEMM whatever the scaling granularity (tools/fp8_scale_study.py) -- and
    random-weight blocks do not damp it: the bounds below are the measured drift + margin, and are why fp8 is opt-in."""
    import math
    from x2i_amd.pipeline import FluxPipeline, FlowMatchEulerDiscreteScheduler
    from x2i_amd.vae import AutoencoderKL
    m = full_dev_model
    H = W = 512
    pipe = FluxPipeline(m, FlowMatchEulerDiscreteScheduler(**OS.SCHEDULER_SCHNELL))
    pe, pooled = bf(seeded((1, 512, 4096), 21)), bf(seeded((1, 768), 22))
    noise = bf(OS.pack_latents(torch.randn((1, 16, H // 8, W // 8), generator=torch.Generator().manual_seed(3))))

    def sample():
        return pipe(prompt_embeds=pe.to(DEV), pooled_prompt_embeds=pooled.to(DEV), num_inference_steps=4, guidance_scale=3.5, height=H,
                    width=W, output_type="latent", latents=noise.to(DEV)).images
    got = {"bf16": sample()}
    try:
        for mode in ("mlp", "all"):
            m.enable_fp8(mode)
            got["fp8_" + mode] = sample()
    finally:
        m.enable_fp8(None)
    # fp32 oracle, 4 steps (about two minutes of host time)
    sd = _LazyF32({k: v.detach().to("cpu") for k, v in m.state_dict().items()})
    cfg = dict(OF.DEFAULT_CFG, guidance_embeds=True)
    lat = noise.clone()
    ts, sig = OS.flow_match_sigmas(4, OS.SCHEDULER_SCHNELL, lat.shape[1])
    img_ids, txt_ids = OS.prepare_latent_image_ids(H // 16, W // 16), torch.zeros(512, 3)
    gd = (torch.tensor([3.5]).bfloat16() * 1000).float() / 1000
    for i, tt in enumerate(ts):
        t1000 = ((tt.expand(1).to(torch.bfloat16) / 1000) * 1000).float()
        eps = OF.flux_forward(sd, cfg, lat.float(), pe.float(), pooled.float(), t1000 / 1000, img_ids, txt_ids, guidance=gd)
        lat = OS.euler_step(lat, eps.bfloat16(), sig[i], sig[i + 1])
    vae = AutoencoderKL(device=DEV)
    vae.load_state_dict({k: bf(v) for k, v in OV.random_vae_decoder_state_dict(seed=7).items()}, strict=True)

    def decode(l):
        z = FluxPipeline._unpack_latents(l.to(DEV), H, W, 16) / vae.config.scaling_factor + vae.config.shift_factor
        return vae.decode(z, return_dict=False)[0].float().cpu()
    img_ref = decode(lat)
    peak = float(img_ref.max() - img_ref.min())
    # measured on MI355X (profiles/r03k_fp8_evidence.log): bf16 7.0e-3 / 56.3 dB, fp8 mlp 4.9e-2 / 44.1 dB, fp8 all 5.4e-2 / 43.1 dB
    bounds = {"bf16": (2e-2, 45.0), "fp8_mlp": (8e-2, 38.0), "fp8_all": (8e-2, 38.0)}
    for k, l in got.items():
        e = rel_l2(l, lat)
        mse = float(((decode(l) - img_ref) ** 2).mean())
        psnr = 10 * math.log10(peak * peak / max(mse, 1e-30))
        print(f"full-depth 4-step @512^2 {k}: final-latent rel-L2 {e:.3e}, decoded PSNR {psnr:.1f} dB")
        assert torch.isfinite(l.float()).all()
        assert e < bounds[k][0] and psnr > bounds[k][1], (k, e, psnr)


def test_fp8_final_latent_drift_over_seeds(full_dev_model):
    """the e4m3 tolerance as a TEST over several seeds instead of one figure in prose.  Full depth and width (19 + 38 blocks),
    4 Euler steps, 512 x 512, B = 1, five (prompt, noise) seeds: rel-L2 of the final packed latents of each opt-in e4m3 configuration against the
    bf16 HIP path on the same inputs (whose own distance to the fp32 oracle is 7.0e-3, asserted <= 2e-2 above).  Measured on MI355X
    (profiles/r06g_test_fp8_drift_over_seeds.log): fp8_mlp 5.3e-2 .. 6.0e-2, fp8_all 5.7e-2 .. 6.4e-2 -- NEITHER configuration stays within the
    5e-2 that SURVEY.md section 8(d) proposed for the final latents (the single-seed 4.9e-2 of the test above was a 2 % margin, not a bound;
    BASELINE.md says so since round 6).  Asserted: the stated drift of these opt-in speed configurations, fp8_mlp <= 6.5e-2, fp8_all <= 7e-2."""
    from x2i_amd.pipeline import FluxPipeline, FlowMatchEulerDiscreteScheduler
    m = full_dev_model
    H = W = 512
    pipe = FluxPipeline(m, FlowMatchEulerDiscreteScheduler(**OS.SCHEDULER_SCHNELL))
    worst = {"fp8_mlp": 0.0, "fp8_all": 0.0}
    try:
        for seed in (3, 11, 29, 47, 101):
            pe, pooled = bf(seeded((1, 512, 4096), seed)).to(DEV), bf(seeded((1, 768), seed + 1)).to(DEV)
            noise = bf(OS.pack_latents(torch.randn((1, 16, H // 8, W // 8), generator=torch.Generator().manual_seed(seed)))).to(DEV)

            def sample():
                return pipe(prompt_embeds=pe, pooled_prompt_embeds=pooled, num_inference_steps=4, guidance_scale=3.5, height=H, width=W,
                            output_type="latent", latents=noise).images.float().cpu()
            m.enable_fp8(None)
            ref = sample()
            for mode in ("mlp", "all"):
                m.enable_fp8(mode)
                e = rel_l2(sample(), ref)
                worst["fp8_" + mode] = max(worst["fp8_" + mode], e)
                print(f"seed {seed}: fp8_{mode} final-latent rel-L2 vs the bf16 HIP path {e:.3e}")
    finally:
        m.enable_fp8(None)
    print("worst over 5 seeds:", worst)
    assert worst["fp8_mlp"] < 6.5e-2, worst
    assert worst["fp8_all"] < 7e-2, worst
