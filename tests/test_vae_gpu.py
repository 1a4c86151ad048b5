"""N1: FLUX VAE decoder on the HIP path vs the oracle restatement (diffusers is absent: parity unpinned, architecture from
the published vae/config.json), plus the new kernels it needs (fused upsample conv, GroupNorm with 4 channels per group,
row softmax, per-batch-W GEMM)."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import vae as OV
from tests.util import rel_l2, seeded

pytestmark = pytest.mark.gpu
DEV = "cuda"


def bf(x):
    return x.to(torch.bfloat16)


def rb(x):
    return x.to(torch.bfloat16).float()


def test_upsample_fused_conv():
    from x2i_amd import ops
    B, C, Co, H, W = 2, 64, 128, 10, 14
    x, w, b = bf(seeded((B, C, H, W), 1)), bf(seeded((Co, C, 3, 3), 2) / 24), bf(seeded((Co,), 3))
    out = ops.conv2d_nhwc(x.permute(0, 2, 3, 1).contiguous().to(DEV), w.permute(0, 2, 3, 1).reshape(Co, -1).contiguous().to(DEV),
                          b.to(DEV), H, W, C, Co, 3, 3, 1, 1, up=True)
    ref = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w.float(), b.float(), padding=1)
    assert out.shape == (B, 2 * H, 2 * W, Co) and rel_l2(out.permute(0, 3, 1, 2), ref) < 1e-2


@pytest.mark.parametrize("C,Co,H,W", [(64, 128, 10, 14), (128, 256, 16, 24), (256, 512, 8, 40)])
def test_upsample_conv_column_phase_form(C, Co, H, W):
    """x2i_conv_desc.up = 2 (rows doubled in the gather, columns not) with out_w (asymmetric padding) and ldc = 2 Cout: Upsample2D's conv as
    two 3 x 2 column-phase convolutions (vae._Conv.packed_up_phases) -- (1) the gather alone, arbitrary 3 x 2 weights, against F.conv2d on the
    row-doubled input with the phase's one-sided padding; (2) the composed pair against the fp32 conv on the doubled image and against the
    fused 3 x 3 form (x2i_conv_desc.up = 1), on both tile kernels (Cout = 128 / >= 256)."""
    from x2i_amd import ops
    from x2i_amd.vae import _Conv
    B = 2
    x = bf(seeded((B, C, H, W), 11))
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    rows2 = x.float().repeat_interleave(2, 2)                                   # [B, C, 2H, W]
    for px in (0, 1):
        w = bf(seeded((Co, C, 3, 2), 12 + px) / 20)
        b = bf(seeded((Co,), 14))
        y = torch.zeros((B, 2 * H, 2 * W, Co), device=DEV, dtype=torch.bfloat16)
        ops.conv2d_nhwc(xn, w.permute(0, 2, 3, 1).reshape(Co, -1).contiguous().to(DEV), b.to(DEV), H, W, C, Co, 3, 2, 1, 1, up=2,
                        pad_w=1 - px, out_w=W, out=y, ldc=2 * Co, c_offset=px * Co, c_batch_stride=4 * H * W * Co)
        ref = F.conv2d(F.pad(rows2, (1 - px, px, 1, 1)), w.float(), b.float())    # left / right / top / bottom
        assert ref.shape == (B, Co, 2 * H, W)
        assert rel_l2(y[:, :, px::2].permute(0, 3, 1, 2), ref) < 1e-2
        assert float(y[:, :, 1 - px::2].float().abs().max()) == 0.0             # the other phase's columns are not touched
    conv = _Conv(C, Co, 3, DEV)
    with torch.no_grad():
        conv.weight.copy_(bf(seeded((Co, C, 3, 3), 15) / 24))
        conv.bias.copy_(bf(seeded((Co,), 16)))
    w0, w1, bb = conv.packed_up_phases()
    y = torch.empty((B, 2 * H, 2 * W, Co), device=DEV, dtype=torch.bfloat16)
    for px, wp in ((0, w0), (1, w1)):
        ops.conv2d_nhwc(xn, wp, bb, H, W, C, Co, 3, 2, 1, 1, up=2, pad_w=1 - px, out_w=W, out=y, ldc=2 * Co, c_offset=px * Co,
                        c_batch_stride=4 * H * W * Co)
    wf, bf_ = conv.packed()
    fused = ops.conv2d_nhwc(xn, wf, bf_, H, W, C, Co, 3, 3, 1, 1, up=True)
    ref = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), conv.weight.float().cpu(), conv.bias.float().cpu(), padding=1)
    e_phase, e_fused = rel_l2(y.permute(0, 3, 1, 2), ref), rel_l2(fused.permute(0, 3, 1, 2), ref)
    assert e_phase < 1e-2 and e_phase < 2.0 * e_fused + 1e-3, (e_phase, e_fused)
    assert rel_l2(y, fused) < 1e-2
    # all four (row, column) phases: 2 x 2 taps on the un-doubled image, interleaved through ldc = 2 Cout and x2i_conv_desc.out_row_pitch,
    # top / left padding 1 - py / 1 - px, bottom / right implied by out_h / out_w
    wp, bb = conv.packed_up_phases(rows=True)
    y4 = torch.zeros((B, 2 * H, 2 * W, Co), device=DEV, dtype=torch.bfloat16)
    for py in (0, 1):
        for px in (0, 1):
            ops.conv2d_nhwc(xn, wp[py][px], bb, H, W, C, Co, 2, 2, 1, 1 - py, pad_w=1 - px, out_h=H, out_w=W, out=y4, ldc=2 * Co,
                            out_row_pitch=4 * W * Co, c_offset=(py * 2 * W + px) * Co, c_batch_stride=4 * H * W * Co)
            if (py, px) == (0, 0):   # only this phase's pixels are written
                assert float(y4[:, 1::2].float().abs().max()) == 0.0 and float(y4[:, :, 1::2].float().abs().max()) == 0.0
    e4 = rel_l2(y4.permute(0, 3, 1, 2), ref)
    assert e4 < 1e-2 and e4 < 2.0 * e_fused + 1e-3, (e4, e_fused)
    assert rel_l2(y4, fused) < 1e-2


@pytest.mark.parametrize("C,Co,H,W,res", [(64, 128, 20, 24, False), (128, 256, 40, 48, False), (128, 256, 33, 40, True), (64, 128, 17, 23, True),
                                          (256, 512, 32, 32, False)])
def test_conv_epilogue_channel_moments(C, Co, H, W, res):
    """x2i_conv_desc.moments: the conv epilogue's (sum, sum of squares) of the bf16 outputs per channel quad -- both tile kernels (Cout = 128: 128^2,
    Cout >= 256 with >= 1024 pixels: 256^2), ragged last tiles, with and without the residual epilogue, accumulation over two launches -- against the
    sums of the stored tensor, bit-reproducible, and feeding groupnorm_nhwc_from_moments = groupnorm_nhwc of the same tensor."""
    from x2i_amd import ops
    B = 2
    x = bf(seeded((B, H, W, C), 21)).to(DEV)
    w = bf(seeded((Co, 3, 3, C), 22) / 24).reshape(Co, -1).contiguous().to(DEV)
    b = bf(seeded((Co,), 23)).to(DEV)
    r = bf(seeded((B, H, W, Co), 24)).to(DEV) if res else None
    mom = torch.full((B, Co, 2), 7.0, device=DEV)
    y = ops.conv2d_nhwc(x, w, b, H, W, C, Co, 3, 3, 1, 1, res=r, moments=mom)
    y0 = ops.conv2d_nhwc(x, w, b, H, W, C, Co, 3, 3, 1, 1, res=r)
    assert torch.equal(y, y0)                                             # the outputs do not change
    yf = y.float().reshape(B, H * W, Co)
    want = torch.stack([yf.sum(1), (yf * yf).sum(1)], -1)                 # per channel ...
    quad = want.reshape(B, Co // 4, 4, 2).sum(2)
    want = torch.zeros_like(want)
    want[:, 0::4] = quad                                                  # ... stored per channel quad (entry c % 4 == 0), zeros elsewhere
    assert float((mom - want).abs().max() / want.abs().max()) < 2e-5
    mom2 = torch.zeros_like(mom)
    ops.conv2d_nhwc(x, w, b, H, W, C, Co, 3, 3, 1, 1, res=r, moments=mom2)
    assert torch.equal(mom, mom2)                                         # fixed summation order
    ops.conv2d_nhwc(x, w, b, H, W, C, Co, 3, 3, 1, 1, res=r, moments=mom2, moments_accumulate=True)
    assert float((mom2 - 2 * want).abs().max() / want.abs().max()) < 4e-5
    gw, gb = bf(1 + 0.1 * seeded((Co,), 25)).to(DEV), bf(0.1 * seeded((Co,), 26)).to(DEV)
    a1 = ops.groupnorm_nhwc_from_moments(y, mom, gw, gb, 32, 1e-6, act=3)
    a0 = ops.groupnorm_nhwc(y, gw, gb, 32, 1e-6, act=3)
    assert rel_l2(a1, a0) < 2e-3


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 9, 21, 128, 3), (1, 64, 80, 128, 3), (1, 37, 16, 64, 4), (3, 5, 130, 32, 1), (1, 130, 70, 96, 2)])
def test_conv3x3_narrow_output(B, H, W, Cin, Cout):
    """x2i_conv3x3_narrow_bf16 (the VAE's conv_out shape: up to four output channels in the rows of the 16 x 16 x 32 MFMA, a wave walking down a
    16-pixel strip with three rotating row accumulators): ragged widths / heights (partial strips, row blocks of any length, one-row images
    handled by the same loop), all supported Cin, against F.conv2d in fp32 on the bf16 inputs and against the implicit-GEMM conv."""
    from x2i_amd import ops
    x = bf(seeded((B, Cin, H, W), 31))
    w = bf(seeded((Cout, Cin, 3, 3), 32) / 20)
    b = bf(seeded((Cout,), 33))
    wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(DEV)
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    y = torch.full((B, H, W, 8), 5.0, device=DEV, dtype=torch.bfloat16)
    ops.conv3x3_narrow(xn, wp, b.to(DEV), Cout, out=y, ldy=8)
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=1)
    assert rel_l2(y[..., :Cout].permute(0, 3, 1, 2), ref) < 6e-3
    assert float(y[..., Cout:4].float().abs().max() if Cout < 4 else 0.0) == 0.0 and bool((y[..., 4:] == 5.0).all())   # zeros behind Cout, the rest untouched
    w8 = torch.zeros((8, 9 * Cin), dtype=torch.bfloat16, device=DEV)
    w8[:Cout] = wp
    b8 = torch.zeros((8,), dtype=torch.bfloat16, device=DEV)
    b8[:Cout] = b.to(DEV)
    if Cin % 64 == 0:
        g = ops.conv2d_nhwc(xn, w8, b8, H, W, Cin, 8, 3, 3, 1, 1)
        assert rel_l2(y[..., :Cout], g[..., :Cout]) < 6e-3
    y2 = ops.conv3x3_narrow(xn, wp, None, Cout)
    assert y2.shape == (B, H, W, 4) and rel_l2(y2[..., :Cout].permute(0, 3, 1, 2), ref - b.float().view(1, -1, 1, 1)) < 6e-3


def test_groupnorm_four_channels_per_group():
    from x2i_amd import ops
    x, w, b = bf(seeded((2, 128, 9, 11), 4, 2.0) + 0.2), bf(1 + 0.1 * seeded((128,), 5)), bf(0.1 * seeded((128,), 6))
    out = ops.groupnorm_nhwc(x.permute(0, 2, 3, 1).contiguous().to(DEV), w.to(DEV), b.to(DEV), 32, 1e-6, act=3)
    assert rel_l2(out.permute(0, 3, 1, 2), F.silu(F.group_norm(x.float(), 32, w.float(), b.float(), 1e-6))) < 5e-3


def test_softmax_rows_and_batched_weight_gemm():
    from x2i_amd import ops
    s = bf(seeded((3, 40, 256), 7, 4.0))
    got = ops.softmax_rows_(s.to(DEV).clone(), 0.25)
    assert rel_l2(got, torch.softmax(s.float() * 0.25, -1)) < 5e-3
    B, T, C = 2, 200, 64
    q, k = bf(seeded((B, T, C), 8)), bf(seeded((B, T, C), 9))
    out = torch.empty((B, T, T), device=DEV, dtype=torch.bfloat16)
    ops.gemm(q.to(DEV), k.to(DEV), None, out=out, M=T, N=T, K=C, batch=B, a_batch_stride=T * C, lda=C, c_batch_stride=T * T, ldc=T,
             w_batch_stride=T * C)
    assert rel_l2(out, q.float() @ k.float().transpose(1, 2)) < 1e-2


def test_vae_decode_vs_oracle_reduced_width():
    from x2i_amd.vae import AutoencoderKL
    cfg = dict(OV.FLUX_VAE_CFG, block_out_channels=(128, 128, 256, 256))
    sd = OV.random_vae_decoder_state_dict(cfg, seed=1)
    vae = AutoencoderKL(block_out_channels=cfg["block_out_channels"], device=DEV)
    full = dict(sd)
    full["encoder.conv_in.weight"] = torch.zeros(1)  # encoder keys of a full checkpoint are ignored
    vae.load_state_dict({k: bf(v) for k, v in full.items()}, strict=True)
    z = seeded((2, 16, 8, 12), 2)  # 96 mid-block tokens (the row softmax needs a multiple of 8)
    img = vae.decode(z.to(DEV), return_dict=False)[0]
    ref = OV.vae_decode({k: rb(v) for k, v in sd.items()}, rb(z), cfg)
    assert img.shape == ref.shape == (2, 3, 64, 96)
    assert rel_l2(img, ref) < 3e-2
    for form in (0, 1):       # Upsample2D's conv as one fused-gather 3 x 3 conv / as two column phases (A/B forms): same image within the same bound
        vae.up_phases = form
        img1 = vae.decode(z.to(DEV), return_dict=False)[0]
        assert rel_l2(img1, ref) < 3e-2 and rel_l2(img, img1) < 2e-2
    vae.up_phases, vae.narrow_conv_out = 2, False   # conv_out as an implicit GEMM with 3 -> 8 padded channels (A/B form): only the last conv's summation order differs
    img3 = vae.decode(z.to(DEV), return_dict=False)[0]
    assert rel_l2(img3, ref) < 3e-2 and rel_l2(img, img3) < 3e-3
    vae.narrow_conv_out = True
    vae.up_phases, vae.epilogue_moments = 2, False     # every GroupNorm with a statistics pass of its own (A/B form)
    img2 = vae.decode(z.to(DEV), return_dict=False)[0]
    assert rel_l2(img2, ref) < 3e-2 and rel_l2(img, img2) < 2e-2


def test_vae_decode_flux_config_small_latent():
    """The real FLUX decoder widths (512/512/256/128, GroupNorm(32), 512-wide attention) on a 16x16 latent -> 128x128 image."""
    from x2i_amd.vae import AutoencoderKL
    sd = OV.random_vae_decoder_state_dict(seed=3)
    vae = AutoencoderKL(device=DEV)
    assert set(vae.state_dict()) == set(sd)
    vae.load_state_dict({k: bf(v) for k, v in sd.items()}, strict=True)
    z = seeded((1, 16, 16, 16), 4)
    img = vae.decode(z.to(DEV), return_dict=False)[0]
    ref = OV.vae_decode({k: rb(v) for k, v in sd.items()}, rb(z))
    assert img.shape == (1, 3, 128, 128) and rel_l2(img, ref) < 3e-2
    assert vae.config.scaling_factor == 0.3611 and 2 ** len(vae.config.block_out_channels) == 16
