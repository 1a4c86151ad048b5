"""Implicit-GEMM convolutions on the persistent four-wave core (csrc/gemm256c.hip, round 6): BIT-IDENTICAL to the eight-wave one-tile form
(option conv_w4 = 0: same MFMA, same k order, same epilogue arithmetic) in outputs and in the epilogue's channel moments, over the gather's
cases -- 3 x 3 / 5 x 5 / 2 x 2 filters, strides 1 and 2, one-sided padding with out_w / out_h (the phases of Upsample2D's conv), an output row
pitch, residual, ReLU + per-batch bias, ragged M (a last tile that is partly out of range), N that is no multiple of 256, several batch
items in one tile list -- and against fp32 F.conv2d.  Reference: the convolutions of lightcontrol/lightcontrol_flux.py:593-668,708-749 and of
the VAE decoder behind infer/inference_qwenvl.py:209-217."""
import pytest
import torch
import torch.nn.functional as F

from tests.util import rel_l2, seeded

pytestmark = pytest.mark.gpu
DEV = "cuda"


def bf(x):
    return x.to(torch.bfloat16)


def _flat(o):
    return [t.float().flatten() for t in (o if isinstance(o, tuple) else (o,))]


def _both(fn, tile_new=5256):
    """fn() under conv_w4 = 1 and 0 -> (new, old, tile read-backs).  The new kernels run in the OTHER kernels' K order here (conv_korder = 0: the
    bit-identity claim); their product order (conv_korder = 1: a filter row's taps back to back, so their shifted re-reads hit L2) sums the same
    products in another order and is checked against it with a tolerance: every output tensor of fn() within 2e-3 (rel-L2)."""
    from x2i_amd import _lib
    out = []
    tiles = []
    for v in (1, 0):
        _lib.set_option("conv_w4", v)
        _lib.set_option("conv_korder", 0)
        _lib.set_option("gemm_min256", 1)    # (the 256^2 convolution kernels also for the few tiles of a test-sized image)
        try:
            out.append(fn())
            tiles.append(_lib.get_option("last_gemm_tile"))
        finally:
            _lib.set_option("conv_w4", 1)
            _lib.set_option("conv_korder", 1)
            _lib.set_option("gemm_min256", 128)
    _lib.set_option("gemm_min256", 1)
    try:
        prod = fn()                           # the product configuration: conv_w4 = 1, conv_korder = 1
        assert _lib.get_option("last_gemm_tile") == tile_new
    finally:
        _lib.set_option("gemm_min256", 128)
    for a, b in zip(_flat(prod), _flat(out[0])):
        assert float((a - b).norm() / b.norm().clamp_min(1e-30)) < 2e-3
    return out[0], out[1], tiles


CASES = [
    # Cin, Cout, KH, KW, stride, pad, H, W, B
    (256, 256, 3, 3, 1, 1, 64, 64, 2),     # VAE / ControlNeXt mid block
    (128, 256, 3, 3, 1, 1, 48, 80, 3),     # ragged: M = 3840 = 15 tiles; rows of 80 pixels straddle the tiles
    (512, 512, 3, 3, 1, 1, 32, 40, 2),     # M = 1280: five tiles, two tile columns
    (256, 320, 3, 3, 1, 1, 40, 40, 2),     # N no multiple of 256 (second tile column partly out of range), M = 1600 (ragged last tile)
    (128, 256, 3, 3, 2, 1, 96, 96, 2),     # Downsample2D: stride 2
    (128, 256, 5, 5, 2, 2, 96, 64, 2),     # the composed conv2 -> Downsample2D chain: 25 taps, stride 2
    (256, 3072, 2, 2, 2, 0, 64, 64, 1),    # ControlNeXt's last conv (2 x 2, stride 2, no padding)
    (64, 256, 3, 3, 1, 1, 40, 56, 2),      # Cin = 64: one channel slice per tap (the tap advances every K-tile)
]


@pytest.mark.parametrize("Cin,Cout,KH,KW,s,p,H,W,B", CASES)
def test_conv_w4_bit_identical_to_eight_wave_form(Cin, Cout, KH, KW, s, p, H, W, B):
    from x2i_amd import ops
    x = bf(seeded((B, Cin, H, W), 1))
    w = bf(seeded((Cout, Cin, KH, KW), 2) / (KH * KW * Cin) ** 0.5)
    b = bf(seeded((Cout,), 3))
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(DEV)
    new, old, tiles = _both(lambda: ops.conv2d_nhwc(xn, wp, b.to(DEV), H, W, Cin, Cout, KH, KW, s, p))
    assert tiles == [5256, 256], tiles           # the persistent four-wave kernel / the eight-wave kernel were really taken
    assert torch.equal(new, old)
    ref = F.conv2d(x.float(), w.float(), b.float(), stride=s, padding=p)
    assert rel_l2(new.permute(0, 3, 1, 2), ref) < 1e-2
    # ReLU + a per-batch f32 bias (ControlNeXt mid block; the time-embedding term)
    b2 = seeded((B, Cout), 4)
    new, old, tiles = _both(lambda: ops.conv2d_nhwc(xn, wp, b.to(DEV), H, W, Cin, Cout, KH, KW, s, p, act=ops.ACT_RELU, bias2=b2.to(DEV)))
    assert tiles == [5256, 256] and torch.equal(new, old)
    assert rel_l2(new.permute(0, 3, 1, 2), torch.relu(ref + b2[:, :, None, None])) < 1e-2
    # residual add (ResnetBlock2D's conv2 + shortcut), also in place
    res = bf(seeded(tuple(ref.shape), 5)).permute(0, 2, 3, 1).contiguous().to(DEV)
    new, old, tiles = _both(lambda: ops.conv2d_nhwc(xn, wp, b.to(DEV), H, W, Cin, Cout, KH, KW, s, p, res=res))
    assert tiles == [5256, 256] and torch.equal(new, old)
    assert rel_l2(new.permute(0, 3, 1, 2), ref + res.float().permute(0, 3, 1, 2).cpu()) < 1e-2


@pytest.mark.parametrize("Cin,Cout,H,W,B", [(256, 256, 64, 64, 2), (512, 512, 32, 40, 3), (128, 256, 72, 56, 1)])
def test_conv_w4_epilogue_moments_bit_identical(Cin, Cout, H, W, B):
    """x2i_conv_desc.moments from the chunked epilogue: the same per-row-block sums in the same order as the one-tile kernels."""
    from x2i_amd import ops
    x = bf(seeded((B, Cin, H, W), 11)).permute(0, 2, 3, 1).contiguous().to(DEV)
    wp = bf(seeded((Cout, 9 * Cin), 12) / (9 * Cin) ** 0.5).to(DEV)
    b = bf(seeded((Cout,), 13)).to(DEV)
    res = bf(seeded((B, H, W, Cout), 14)).to(DEV)
    for kw in (dict(), dict(res=res)):
        def run():
            mom = torch.full((B, Cout, 2), 7.0, device=DEV)
            y = ops.conv2d_nhwc(x, wp, b, H, W, Cin, Cout, 3, 3, 1, 1, moments=mom, **kw)
            return y, mom
        (yn, mn), (yo, mo), tiles = _both(run)
        assert tiles == [5256, 256]
        assert torch.equal(yn, yo) and torch.equal(mn, mo)
        q = yn.float().reshape(B, H * W, Cout // 4, 4)
        s1, s2 = q.sum((1, 3)), (q * q).sum((1, 3))
        assert rel_l2(mn[:, 0::4, 0], s1) < 1e-4 and rel_l2(mn[:, 0::4, 1], s2) < 1e-4
        assert float(mn[:, 1::4].abs().max()) == 0.0
    # accumulate: two launches into one set of moments
    def run2():
        mom = torch.zeros((B, Cout, 2), device=DEV)
        ops.conv2d_nhwc(x, wp, b, H, W, Cin, Cout, 3, 3, 1, 1, moments=mom)
        ops.conv2d_nhwc(x, wp, b, H, W, Cin, Cout, 3, 3, 1, 1, moments=mom, moments_accumulate=True)
        return mom
    mn, mo, _ = _both(run2)
    assert torch.equal(mn, mo)


@pytest.mark.parametrize("C,Co,H,W", [(256, 256, 32, 48), (512, 512, 32, 40)])
def test_conv_w4_four_phase_upsample_form(C, Co, H, W):
    """The four 2 x 2 phase convolutions of Upsample2D's conv (vae._Conv.packed_up_phases): one-sided padding through pad / pad_w / out_w / out_h,
    the phases interleaved through ldc = 2 Cout and an output ROW PITCH of two rows, moments accumulated over the four launches -- the new
    kernel's pitched store path.  Bit-identical to the eight-wave form; the composed result against fp32 on the doubled image."""
    from x2i_amd import ops
    from x2i_amd.vae import _Conv
    B = 2
    x = bf(seeded((B, C, H, W), 21))
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    conv = _Conv(C, Co, 3, DEV)
    with torch.no_grad():
        conv.weight.copy_(bf(seeded((Co, C, 3, 3), 22) / 24))
        conv.bias.copy_(bf(seeded((Co,), 23)))
    wp, pb = conv.packed_up_phases(rows=True)

    def run():
        y = torch.zeros((B, 2 * H, 2 * W, Co), device=DEV, dtype=torch.bfloat16)
        mom = torch.zeros((B, Co, 2), device=DEV)
        for py in (0, 1):
            for px in (0, 1):
                ops.conv2d_nhwc(xn, wp[py][px], pb, H, W, C, Co, 2, 2, 1, 1 - py, pad_w=1 - px, out_w=W, out_h=H, out=y, ldc=2 * Co,
                                c_offset=(py * 2 * W + px) * Co, c_batch_stride=4 * H * W * Co, out_row_pitch=4 * W * Co, moments=mom,
                                moments_accumulate=(py, px) != (0, 0))
        return y, mom
    (yn, mn), (yo, mo), tiles = _both(run)
    assert tiles == [5256, 256]
    assert torch.equal(yn, yo) and torch.equal(mn, mo)
    ref = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), conv.weight.float().cpu(), conv.bias.float().cpu(), padding=1)
    assert rel_l2(yn.permute(0, 3, 1, 2), ref) < 1.2e-2


def test_conv_w4_batch_independence():
    """A sample's output does not depend on the batch it rides in (one tile list over all batch items)."""
    from x2i_amd import ops
    Cin, Cout, H, W = 256, 256, 40, 48
    x = bf(seeded((3, H, W, Cin), 31)).to(DEV)
    wp = bf(seeded((Cout, 9 * Cin), 32) / 48).to(DEV)
    b = bf(seeded((Cout,), 33)).to(DEV)
    from x2i_amd import _lib
    _lib.set_option("gemm_min256", 1)
    try:
        y3 = ops.conv2d_nhwc(x, wp, b, H, W, Cin, Cout, 3, 3, 1, 1)
        y1 = ops.conv2d_nhwc(x[1:2].contiguous(), wp, b, H, W, Cin, Cout, 3, 3, 1, 1)
        assert _lib.get_option("last_gemm_tile") == 5256
    finally:
        _lib.set_option("gemm_min256", 128)
    assert torch.equal(y3[1:2], y1)


# ---- <= 128 output channels: the 512 x 128 tiles of csrc/gemm512c.hip (epilogue straight from registers) against the 128^2 kernel
CASES128 = [
    # Cin, Cout, KH, KW, stride, pad, H, W, B
    (128, 128, 3, 3, 1, 1, 64, 64, 2),     # the VAE's last up block / ControlNeXt ResnetBlock
    (256, 128, 3, 3, 1, 1, 48, 80, 2),     # M = 3840: 7.5 tiles of 512 (ragged last tile), rows of 80 pixels straddle the tiles
    (128, 128, 5, 5, 2, 2, 128, 96, 2),    # the composed conv2 -> Downsample2D chain (25 taps, stride 2)
    (128, 128, 3, 3, 2, 1, 128, 128, 1),   # Downsample2D
    (256, 128, 1, 1, 1, 0, 64, 64, 2),     # conv_shortcut (1 x 1: K = 256, four K-tiles)
    (64, 128, 3, 3, 1, 1, 72, 64, 2),      # ControlNeXt embedding conv (Cin = 64: the tap advances every K-tile)
    (128, 96, 3, 3, 1, 1, 64, 48, 2),      # fewer than 128 output channels: the tile's last 32 columns are out of range
]


def _both128(fn):
    return _both(fn, tile_new=5512)


@pytest.mark.parametrize("Cin,Cout,KH,KW,s,p,H,W,B", CASES128)
def test_conv512_bit_identical_to_the_128_tile_kernel(Cin, Cout, KH, KW, s, p, H, W, B):
    from x2i_amd import ops
    x = bf(seeded((B, Cin, H, W), 41))
    w = bf(seeded((Cout, Cin, KH, KW), 42) / (KH * KW * Cin) ** 0.5)
    b = bf(seeded((Cout,), 43))
    xn = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(DEV)
    ref = F.conv2d(x.float(), w.float(), b.float(), stride=s, padding=p)
    res = bf(seeded(tuple(ref.shape), 45)).permute(0, 2, 3, 1).contiguous().to(DEV)
    b2 = seeded((B, Cout), 44)
    for kw, want in ((dict(), ref), (dict(res=res), ref + res.float().permute(0, 3, 1, 2).cpu()),
                     (dict(act=ops.ACT_RELU, bias2=b2.to(DEV)), torch.relu(ref + b2[:, :, None, None]))):
        def run():
            mom = torch.zeros((B, Cout, 2), device=DEV)
            y = ops.conv2d_nhwc(xn, wp, b.to(DEV), H, W, Cin, Cout, KH, KW, s, p, moments=mom, **kw)
            return y, mom
        (yn, mn), (yo, mo), tiles = _both128(run)
        assert tiles == [5512, 128], tiles
        assert torch.equal(yn, yo)                               # same MFMA, same k order, same epilogue arithmetic
        assert rel_l2(yn.permute(0, 3, 1, 2), want) < 1e-2
        # moments: 128-row blocks here, 64-row blocks there -- the same sums up to the order of an f32 addition
        assert rel_l2(mn, mo) < 1e-5 and float(mn[:, 1::4].abs().max()) == 0.0
        q = yn.float().reshape(B, -1, Cout // 4, 4)
        assert rel_l2(mn[:, 0::4, 0], q.sum((1, 3))) < 1e-4 and rel_l2(mn[:, 0::4, 1], (q * q).sum((1, 3))) < 1e-4


def test_conv512_batch_independence_and_determinism():
    from x2i_amd import _lib, ops
    Cin, Cout, H, W = 128, 128, 64, 80
    x = bf(seeded((3, H, W, Cin), 51)).to(DEV)
    wp = bf(seeded((Cout, 9 * Cin), 52) / 34).to(DEV)
    b = bf(seeded((Cout,), 53)).to(DEV)
    _lib.set_option("gemm_min256", 1)
    try:
        mom = torch.zeros((3, Cout, 2), device=DEV)
        y3 = ops.conv2d_nhwc(x, wp, b, H, W, Cin, Cout, 3, 3, 1, 1, moments=mom)
        assert _lib.get_option("last_gemm_tile") == 5512
        mom1 = torch.zeros((1, Cout, 2), device=DEV)
        y1 = ops.conv2d_nhwc(x[1:2].contiguous(), wp, b, H, W, Cin, Cout, 3, 3, 1, 1, moments=mom1)
        mom_b = torch.zeros((3, Cout, 2), device=DEV)
        y3b = ops.conv2d_nhwc(x, wp, b, H, W, Cin, Cout, 3, 3, 1, 1, moments=mom_b)
    finally:
        _lib.set_option("gemm_min256", 128)
    assert torch.equal(y3[1:2], y1) and torch.equal(mom[1:2], mom1)      # a sample's outputs and moments do not depend on its batch
    assert torch.equal(y3, y3b) and torch.equal(mom, mom_b)              # run to run: bit-equal (fixed summation tree)
