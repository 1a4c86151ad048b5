"""fp8 (OCP e4m3fn) path, row F8: quantisation kernels bit-exact against torch's float8_e4m3fn conversion, the MX-scaled K = 128
MFMA GEMM against an fp32 matmul of the SAME dequantised operands (tight: only the accumulation order differs), its epilogues,
and the stated end-to-end tolerance of fp8 against the unquantised fp32 result (per GEMM rel-L2 <= 6e-2: two operands with a
3-bit mantissa each)."""
import pytest
import torch
import torch.nn.functional as F

from tests.util import rel_l2, seeded, span_permute

pytestmark = pytest.mark.gpu
DEV = "cuda"
FP8 = torch.float8_e4m3fn


def bf(x):
    return x.to(torch.bfloat16)


def deq(x8, scale=None):
    x = x8.float().cpu()
    return x if scale is None else x * scale.float().cpu()[:, None]


@pytest.mark.parametrize("rows,cols", [(5, 64), (300, 3072), (64, 12288), (7, 520)])
def test_quantize_rows_bit_exact_vs_torch_e4m3fn(rows, cols):
    from x2i_amd import ops
    x = bf(seeded((rows, cols), 1, 3.0))
    x[0, :8] = torch.tensor([0.0, -0.0, 1e-4, -2e-3, 448.0, -448.0, 0.0156, 0.3])
    if rows > 4:
        x[3] = 0  # an all-zero row: scale 1, zeros out
    y, s = ops.quantize_rows_fp8(x.to(DEV))
    amax = x.float().abs().amax(1)
    want_s = torch.where(amax > 0, amax * (1.0 / 448.0), torch.ones_like(amax))
    assert torch.equal(s.cpu(), want_s)
    want = (x.float() * (1.0 / want_s)[:, None]).clamp(-448, 448).to(FP8)
    assert torch.equal(y.cpu().view(torch.uint8), want.view(torch.uint8))
    # static form: saturating
    y2, s2 = ops.quantize_rows_fp8(x.to(DEV), static_inv_scale=200.0)
    assert s2 is None
    want2 = (x.float() * 200.0).clamp(-448, 448).to(FP8)
    assert torch.equal(y2.cpu().view(torch.uint8), want2.view(torch.uint8))


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 768, 256), (700, 520, 1152), (4096, 3072, 3072), (1024, 12288, 3072)])
def test_gemm_fp8_vs_fp32_matmul_of_the_same_operands(M, N, K):
    """Asymmetric operands + per-row / per-column scales: catches any k-permutation or row/column mix-up of the MX MFMA."""
    from x2i_amd import ops
    A, W = bf(seeded((M, K), 2)), bf(seeded((N, K), 3, 0.05))
    A[:, ::7] *= 3.0  # make columns distinguishable
    b = bf(seeded((N,), 4))
    A8, sa = ops.quantize_rows_fp8(A.to(DEV))
    W8, sw = ops.quantize_rows_fp8(W.to(DEV))
    out = ops.gemm_fp8(A8, W8, b.to(DEV), a_scale=sa, w_scale=sw)
    ref = deq(A8, sa) @ deq(W8, sw).T + b.float()
    assert out.shape == (M, N) and rel_l2(out, ref) < 4e-3  # bf16 output rounding dominates
    assert rel_l2(out, F.linear(A.float(), W.float(), b.float())) < 6e-2  # fp8 vs unquantised fp32 (stated tolerance)
    ident = ops.gemm_fp8(A8, W8, None, alpha=0.5)  # unit scales, alpha only
    assert rel_l2(ident, 0.5 * (deq(A8) @ deq(W8).T)) < 4e-3


def test_gemm_fp8_epilogues_gelu_e4m3_out_and_gated_residual():
    from x2i_amd import ops
    B, S, K, N = 2, 1536, 512, 1024
    X = bf(seeded((B, S, K), 5))
    W1, b1 = bf(seeded((N, K), 6, 0.05)), bf(seeded((N,), 7))
    W2, b2 = bf(seeded((K, N), 8, 0.03)), bf(seeded((K,), 9))
    X8, sx = ops.quantize_rows_fp8(X.to(DEV).view(B * S, K))
    W18, sw1 = ops.quantize_rows_fp8(W1.to(DEV))
    W28, sw2 = ops.quantize_rows_fp8(W2.to(DEV))
    # ff.net.0 + GELU(tanh) -> e4m3 hidden (static scale 1: GELU outputs live well inside +-448)
    H8 = ops.gemm_fp8(X8, W18, b1.to(DEV), a_scale=sx, w_scale=sw1, act=ops.ACT_GELU_TANH, out_fp8=True, out_inv_scale=1.0)
    assert H8.dtype == FP8
    h_ref = F.gelu(deq(X8, sx) @ deq(W18, sw1).T + b1.float(), approximate="tanh")
    want8 = h_ref.clamp(-448, 448).to(FP8)
    got = H8.float().cpu()
    # e4m3 rounding of a value that sits within fp32 accumulation noise of a rounding boundary may flip by one code
    mism = (H8.cpu().view(torch.uint8) != want8.view(torch.uint8)).float().mean()
    assert mism < 2e-3 and rel_l2(got, h_ref) < 4e-2
    # ff.net.2 with gate * v + residual, batched like the model launches it
    res = bf(seeded((B, S, K), 10))
    gate = seeded((B, K), 11)
    out = res.to(DEV).clone()
    ops.gemm_fp8(H8, W28, b2.to(DEV), out=out, M=S, batch=B, a_batch_stride=S * N, lda=N, w_scale=sw2, c_batch_stride=S * K, ldc=K,
                 res=out, res_batch_stride=S * K, ldr=K, gate=gate.to(DEV), gate_batch_stride=K)
    ref = res.float() + gate[:, None, :] * ((got.view(B, S, N) @ deq(W28, sw2).T) + b2.float())
    assert rel_l2(out, ref) < 4e-3
    # unsupported shapes are refused loudly (no silent slow path)
    from x2i_amd._lib import X2IError
    with pytest.raises(X2IError):
        ops.gemm_fp8(X8[:, :192].contiguous(), W18[:, :192].contiguous(), None)  # K % 128 != 0


def test_ln_modulate_fp8_matches_bf16_kernel_and_torch():
    from x2i_amd import ops
    B, S, D, S0 = 2, 300, 3072, 44
    X = bf(seeded((B, S, D), 12, 2.0) + 0.3)
    mod = seeded((B, 4 * D), 13, 0.3)
    Xd, md = X.to(DEV), mod.to(DEV)
    Y = torch.empty_like(Xd)
    ops.ln_modulate(Xd, Y, B, S, D, S0, md, md[:, D:], md[:, 2 * D:], md[:, 3 * D:], 4 * D)
    Y2 = torch.empty_like(Xd)
    Y8 = torch.empty((B, S, D), device=DEV, dtype=FP8)
    rs = torch.empty((B * S,), device=DEV, dtype=torch.float32)
    ops.ln_modulate_fp8(Xd, Y2, Y8, rs, B, S, D, S0, md, md[:, D:], md[:, 2 * D:], md[:, 3 * D:], 4 * D)
    assert torch.equal(Y2, Y)  # the bf16 output is the same arithmetic
    ln = F.layer_norm(X.float(), (D,), eps=1e-6)
    sh = torch.cat([mod[:, None, :D].expand(B, S0, D), mod[:, None, 2 * D:3 * D].expand(B, S - S0, D)], 1)
    sc = torch.cat([mod[:, None, D:2 * D].expand(B, S0, D), mod[:, None, 3 * D:].expand(B, S - S0, D)], 1)
    y = ln * (1 + sc) + sh
    want_s = y.abs().amax(-1).flatten() / 448
    assert torch.allclose(rs.cpu(), want_s, rtol=1e-4)
    assert rel_l2(Y8.float().cpu() * rs.cpu().view(B, S, 1), y) < 3e-2  # e4m3: 3 mantissa bits
    assert float(Y8.float().abs().max()) == 448.0
    ops.ln_modulate_fp8(Xd, None, Y8, rs, B, S, D, S0, md, md[:, D:], md[:, 2 * D:], md[:, 3 * D:], 4 * D)  # bf16 output optional


def test_ln_modulate_fp8_rows_form_equals_per_row_form():
    """D = 3072 with >= 4096 rows takes the four-rows-per-wave kernel (modulation vectors in registers, ln_fp8_rows_kernel): e4m3 rows,
    row scales and the optional bf16 output against the per-row kernel (option fp8 = 2) bit for bit; the text / image boundary S0 falls
    inside a wave's row group, and the last row group is ragged."""
    from x2i_amd import _lib, ops
    B, S, D, S0 = 2, 2307, 3072, 510
    g = torch.Generator(device=DEV).manual_seed(5)
    Xd = (torch.randn((B, S, D), device=DEV, generator=g) * 2 + 0.3).bfloat16()
    md = torch.randn((B, 4 * D), device=DEV, generator=g) * 0.3
    outs = []
    old = _lib.get_option("fp8")
    try:
        for form in (2, 0):
            _lib.set_option("fp8", form)
            Y = torch.zeros_like(Xd)
            Y8 = torch.zeros((B, S, D), device=DEV, dtype=torch.uint8)
            rs = torch.zeros((B * S,), device=DEV, dtype=torch.float32)
            ops.ln_modulate_fp8(Xd, Y, Y8.view(FP8), rs, B, S, D, S0, md, md[:, D:], md[:, 2 * D:], md[:, 3 * D:], 4 * D)
            outs.append((Y, Y8, rs))
    finally:
        _lib.set_option("fp8", old)
    for a, b_ in zip(outs[0], outs[1]):
        assert torch.equal(a, b_)
    Yref = torch.empty_like(Xd)
    ops.ln_modulate(Xd, Yref, B, S, D, S0, md, md[:, D:], md[:, 2 * D:], md[:, 3 * D:], 4 * D)
    assert torch.equal(outs[1][0], Yref)


@pytest.mark.parametrize("variant", [0, 9])
def test_attention_e4m3_output_equals_quantised_bf16_output(variant):
    """variant 9: the hand-scheduled kernel's e4m3 epilogue (16 consecutive bytes per lane), 0: the automatic choice at this size"""
    import math
    from x2i_amd import _lib, ops
    _lib.load().x2i_set_option(b"attn_variant", variant)
    B, H, S = 2, 4, 600
    Spad = ops.pad128(S)
    g = torch.Generator(device=DEV).manual_seed(3)
    Q = torch.randn((B, H, Spad, 128), device=DEV, generator=g).bfloat16()
    K = torch.randn((B, H, Spad, 128), device=DEV, generator=g).bfloat16()
    VT = (torch.randn((B, H, 128, Spad), device=DEV, generator=g) * 3).bfloat16()
    ld = H * 128 + 256  # strided destination, like the CAT buffer of the single blocks
    O = torch.zeros((B, S, ld), device=DEV, dtype=torch.bfloat16)
    O8 = torch.zeros((B, S, ld), device=DEV, dtype=FP8)
    ops.attention(Q, K, VT, O, B, H, S, Spad, ld, S * ld, 1 / math.sqrt(128))
    ops.attention_e4m3out(Q, K, VT, O8, B, H, S, Spad, ld, S * ld, 1 / math.sqrt(128))
    a, b8 = O[..., :H * 128].float(), O8[..., :H * 128].float()
    assert float(O8[..., H * 128:].float().abs().max()) == 0.0  # nothing written outside the head columns
    assert rel_l2(b8, a) < 3e-2  # one e4m3 rounding apart
    # e4m3(fp32 result) vs e4m3(bf16(result)): identical except where the bf16 rounding crosses an e4m3 boundary
    assert (b8 != a.to(FP8).float()).float().mean() < 0.03
    # a static output scale that saturates: sat(o * 2000) clamps at +-448
    ops.attention_e4m3out(Q, K, VT, O8, B, H, S, Spad, ld, S * ld, 1 / math.sqrt(128), out_inv_scale=2000.0)
    want = (a * 2000).clamp(-448, 448)
    assert float(O8.float().abs().max()) == 448.0 and rel_l2(O8[..., :H * 128].float(), want) < 3e-2
    _lib.load().x2i_set_option(b"attn_variant", 0)


@pytest.mark.parametrize("B,H,St,Si", [(2, 2, 256, 1024), (1, 4, 0, 1500), (2, 2, 200, 700)])
def test_gemm_qkv_fp8_equals_gemm_fp8_then_qkv_split(B, H, St, Si):
    """x2i_gemm_qkv_fp8 (dequantise, then the bf16 kernel's fused RMSNorm / RoPE / head-split epilogue) against the two-step form
    x2i_gemm_fp8 -> x2i_qkv_split_bf16 on the same e4m3 operands and scales: batched image rows behind a text offset, the
    flattened single-block geometry, and ragged (unaligned) token counts."""
    from x2i_amd import ops
    D, S, Kd = H * 128, St + Si, 256
    Spad = ops.pad128(S)
    W8, sw = ops.quantize_rows_fp8(bf(seeded((3 * D, Kd), 40, 0.08)).to(DEV))
    bias = bf(seeded((3 * D,), 41, 0.5)).to(DEV)
    X8, sx = ops.quantize_rows_fp8(bf(seeded((B * S, Kd), 44)).to(DEV))
    nq, nk = (bf(1 + 0.2 * seeded((128,), 45 + i)).to(DEV) for i in range(2))
    ang = seeded((S, 64), 50, 3.0)
    cos, sin = torch.cos(ang).repeat_interleave(2, 1).contiguous().to(DEV), torch.sin(ang).repeat_interleave(2, 1).contiguous().to(DEV)

    def bufs():
        z = lambda *sh: torch.zeros(sh, device=DEV, dtype=torch.bfloat16)
        return z(B, H, Spad, 128), z(B, H, Spad, 128), z(B, H, 128, Spad)
    Q0, K0, V0 = bufs()
    Q1, K1, V1 = bufs()
    if St > 0:
        # image rows of a [B, S, Kd] activation; row scales laid out [B, Si] as the LayerNorm kernel writes them
        sxi = sx.view(B, S)[:, St:].contiguous()
        QKV = torch.zeros((B * S, 3 * D), device=DEV, dtype=torch.bfloat16)
        off_img = B * St * 3 * D
        ops.gemm_fp8(X8, W8, bias, out=QKV, M=Si, batch=B, a_batch_stride=S * Kd, lda=Kd, a_offset=St * Kd, a_scale=sxi,
                     a_scale_batch_stride=Si, w_scale=sw, c_batch_stride=Si * 3 * D, ldc=3 * D, c_offset=off_img)
        ops.qkv_split(QKV, QKV.view(-1)[off_img:], 3 * D, 3 * D, B, S, St, H, nq, nk, nq, nk, cos, sin, Q0, K0, V0, Spad)
        ops.gemm_qkv_fp8(X8, W8, bias, Q1, K1, V1, nq, nk, cos, sin, M=Si, H=H, Spad=Spad, tok_off=St, rows_per_sample=Si, batch=B,
                         a_batch_stride=S * Kd, lda=Kd, a_offset=St * Kd, a_scale=sxi, a_scale_batch_stride=Si, w_scale=sw)
        sl = slice(St, S)
    else:
        QKV = torch.empty((B * S, 3 * D), device=DEV, dtype=torch.bfloat16)
        ops.gemm_fp8(X8, W8, bias, out=QKV, M=B * S, a_scale=sx, w_scale=sw)
        ops.qkv_split(None, QKV, 3 * D, 3 * D, B, S, 0, H, None, None, nq, nk, cos, sin, Q0, K0, V0, Spad)
        ops.gemm_qkv_fp8(X8, W8, bias, Q1, K1, V1, nq, nk, cos, sin, M=B * S, H=H, Spad=Spad, tok_off=0, rows_per_sample=S,
                         a_scale=sx, w_scale=sw)
        sl = slice(0, S)
    torch.cuda.synchronize()
    assert torch.equal(V1[..., sl], V0[..., sl]) and torch.equal(Q1[:, :, sl], Q0[:, :, sl]) and torch.equal(K1[:, :, sl], K0[:, :, sl])
    assert float(Q1[:, :, :St].float().abs().max() if St else 0.0) == 0.0  # rows outside the launch are untouched
    assert float(Q1[:, :, sl].float().abs().mean()) > 0.1
    # x2i_qkv_desc.vt_perm on the e4m3 form: the same V^T, span-permuted (the position map moves tokens within their own 32-token span,
    # so the permuted buffer holds the launched rows' values at their mapped places and zeros elsewhere)
    Q2, K2, V2 = bufs()
    kw = dict(M=Si, H=H, Spad=Spad, tok_off=St, rows_per_sample=Si, batch=B, a_batch_stride=S * Kd, lda=Kd, a_offset=St * Kd, a_scale=sxi,
              a_scale_batch_stride=Si, w_scale=sw) if St > 0 else dict(M=B * S, H=H, Spad=Spad, tok_off=0, rows_per_sample=S, a_scale=sx, w_scale=sw)
    ops.gemm_qkv_fp8(X8, W8, bias, Q2, K2, V2, nq, nk, cos, sin, vt_perm=True, **kw)
    assert torch.equal(V2, span_permute(V1)) and torch.equal(Q2, Q1) and torch.equal(K2, K1)
    # the pair-form RoPE table (x2i_qkv_desc.sin == NULL) on the e4m3 form: bit-identical
    Q3, K3, V3 = bufs()
    ops.gemm_qkv_fp8(X8, W8, bias, Q3, K3, V3, nq, nk, ops.rope_pairs(cos, sin), None, vt_perm=True, **kw)
    assert torch.equal(V3, V2) and torch.equal(Q3, Q2) and torch.equal(K3, K2)


def _tiny(fp8, mode="mlp"):
    from oracle import flux as OF
    from x2i_amd.flux import FluxTransformer2DModel
    cfg = dict(OF.DEFAULT_CFG)
    cfg.update(num_layers=2, num_single_layers=2, num_attention_heads=2, joint_attention_dim=64, pooled_projection_dim=32)
    sd = OF.random_flux_state_dict(cfg, seed=11, std=0.05)
    m = FluxTransformer2DModel(**cfg, device=DEV)
    m.load_state_dict({k: v.bfloat16() for k, v in sd.items()}, strict=True)
    if fp8:
        m.enable_fp8(mode)
    return m, sd, cfg


@pytest.mark.parametrize("mode", ["mlp", "all"])
def test_fp8_model_vs_bf16_model_and_fp32_oracle_and_graph_replay(mode):
    """Stated tolerance of the fp8 configuration: transformer output rel-L2 <= 6e-2 vs the fp32 oracle (bf16: <= 2e-2), 4-step latents
    <= 8e-2; per-sample results stay batch independent and hipGraph replay is bit-identical to eager."""
    from oracle import flux as OF
    from oracle import sampler as OS
    from x2i_amd.pipeline import FluxPipeline, FlowMatchEulerDiscreteScheduler
    m8, sd, cfg = _tiny(True, mode)
    m16, _, _ = _tiny(False)
    g = torch.Generator().manual_seed(0)
    B, St, h2, w2 = 3, 24, 6, 8
    hid = torch.randn((B, h2 * w2, 64), generator=g).bfloat16()
    enc, pooled = torch.randn((B, St, 64), generator=g).bfloat16(), torch.randn((B, 32), generator=g).bfloat16()
    ts = torch.tensor([0.5, 0.25, 1.0])
    ids, tids = OS.prepare_latent_image_ids(h2, w2), torch.zeros(St, 3)
    kw = dict(hidden_states=hid.to(DEV), encoder_hidden_states=enc.to(DEV), pooled_projections=pooled.to(DEV), timestep=ts.to(DEV),
              img_ids=ids.to(DEV), txt_ids=tids.to(DEV), return_dict=False)
    o8, o16 = m8(**kw)[0], m16(**kw)[0]
    ref = OF.flux_forward({k: v.bfloat16().float() for k, v in sd.items()}, cfg, hid.float(), enc.float(), pooled.float(), ts, ids, tids)
    e8, e16 = rel_l2(o8, ref), rel_l2(o16, ref)
    print(f"tiny 2+2 blocks [{mode}]: fp8 {e8:.3e}  bf16 {e16:.3e}  (rel-L2 vs fp32 oracle)")
    assert e16 < 2e-2 and e8 < 6e-2 and not torch.equal(o8, o16)
    kw1 = dict(kw, hidden_states=kw["hidden_states"][1:2], encoder_hidden_states=kw["encoder_hidden_states"][1:2],
               pooled_projections=kw["pooled_projections"][1:2], timestep=kw["timestep"][1:2])
    assert torch.equal(m8(**kw1)[0][0], o8[1])  # batch independence holds in fp8 too (per-row scales, no cross-sample statistic)
    pipe = FluxPipeline(m8, FlowMatchEulerDiscreteScheduler())
    pk = dict(prompt_embeds=enc.to(DEV), pooled_prompt_embeds=pooled.to(DEV), num_inference_steps=4, height=96, width=128,
              output_type="latent", latents=hid.to(DEV))
    a = pipe(**pk).images
    b = pipe(**pk, use_graph=True).images
    assert torch.equal(a, b)
    lat16 = FluxPipeline(m16, FlowMatchEulerDiscreteScheduler())(**pk).images
    assert rel_l2(a, lat16) < 8e-2
    m8.enable_fp8(None)
    assert torch.equal(m8(**kw)[0], o16)  # switching back restores the bf16 path bit for bit


@pytest.mark.parametrize("mode", ["mlp", "all"])
def test_fp8_full_width_blocks_vs_oracle(mode):
    """D = 3072, one double + one single block on 512 + 1024 tokens: the real GEMM shapes of the fp8 configuration."""
    from oracle import flux as OF
    from oracle import sampler as OS
    from x2i_amd.flux import FluxTransformer2DModel
    cfg = dict(OF.DEFAULT_CFG)
    cfg.update(num_layers=1, num_single_layers=1)
    sd = OF.random_flux_state_dict(cfg, seed=21, std=0.02)
    m = FluxTransformer2DModel(**cfg, device=DEV)
    m.load_state_dict({k: v.bfloat16() for k, v in sd.items()}, strict=True)
    hidden, enc, pooled = seeded((2, 1024, 64), 1), seeded((2, 512, 4096), 2), seeded((2, 768), 3)
    ts = torch.tensor([0.5, 0.75])
    img_ids, txt_ids = OS.prepare_latent_image_ids(32, 32), torch.zeros(512, 3)
    kw = dict(hidden_states=hidden.to(DEV), encoder_hidden_states=enc.to(DEV), pooled_projections=pooled.to(DEV), timestep=ts.to(DEV),
              img_ids=img_ids.to(DEV), txt_ids=txt_ids.to(DEV), return_dict=False)
    o16 = m(**kw)[0].clone()
    m.enable_fp8(mode)
    o8 = m(**kw)[0]
    ref = OF.flux_forward({k: v.bfloat16().float() for k, v in sd.items()}, cfg, hidden.bfloat16().float(), enc.bfloat16().float(),
                          pooled.bfloat16().float(), ts, img_ids, txt_ids)
    e8, e16 = rel_l2(o8, ref), rel_l2(o16, ref)
    print(f"full-width 1+1 blocks [{mode}]: fp8 {e8:.3e}  bf16 {e16:.3e}  (rel-L2 vs fp32 oracle)")
    assert e16 < 2e-2 and e8 < 6e-2


# ---------------------------------------------------------------------------------------------------------------------
# round 4: the persistent four-wave e4m3 kernel (gemm256p.hip, F8 = true; K-loop from gen_gemm256f8.py)
@pytest.fixture
def opt():
    from x2i_amd import _lib
    saved = {}

    def set_(name, value):
        old = _lib.set_option(name, value)
        saved.setdefault(name, old)
    yield set_
    for k, v in saved.items():
        _lib.set_option(k, v)


@pytest.mark.parametrize("kind", ["gelu_e4m3", "gated_residual", "plain", "gelu_bf16"])
@pytest.mark.parametrize("M,N,K", [(4 * 4608, 3072, 3072), (2 * 4608 + 40, 2048, 1536), (4608, 12288, 3072)])
def test_persistent_e4m3_gemm_equals_one_tile_kernel(opt, kind, M, N, K):
    """x2i_gemm_fp8 in the persistent four-wave form (one K = 128 MFMA per accumulator and K-tile, hand-scheduled; chained stream-K
    for the partly filled last round; pipelined dequantising epilogues) against the round-2 one-tile kernel, bit for bit: the
    accumulation order over K is the same and the epilogue arithmetic is spelled the same way.  Shapes: the single-block proj_out
    geometry with a stream-K remainder (864 tiles on 256 CUs), a ragged row count, and the wide GELU launch."""
    from x2i_amd import _lib, ops
    g = torch.Generator(device=DEV).manual_seed(7)
    A = (torch.randn((M, K), device=DEV, generator=g) * 1.5).bfloat16()
    W = (torch.randn((N, K), device=DEV, generator=g) * 0.03).bfloat16()
    b = torch.randn((N,), device=DEV, generator=g).bfloat16()
    A8, sa = ops.quantize_rows_fp8(A)
    W8, sw = ops.quantize_rows_fp8(W)
    res = torch.randn((M, N), device=DEV, generator=g).bfloat16()
    gate = torch.randn((1, N), device=DEV, generator=g)

    def run():
        if kind == "gelu_e4m3":
            return ops.gemm_fp8(A8, W8, b, a_scale=sa, w_scale=sw, act=ops.ACT_GELU_TANH, out_fp8=True, out_inv_scale=0.75).view(torch.uint8)
        if kind == "gated_residual":
            out = res.clone()
            ops.gemm_fp8(A8, W8, b, out=out, a_scale=sa, w_scale=sw, alpha=1.25, res=out, gate=gate)
            return out
        if kind == "gelu_bf16":
            return ops.gemm_fp8(A8, W8, b, a_scale=sa, w_scale=sw, act=ops.ACT_GELU_TANH)
        return ops.gemm_fp8(A8, W8, b, a_scale=sa, w_scale=sw)
    opt("gemm_fp8_persist", 0)
    want = run()
    assert _lib.get_option("last_gemm_tile") == 7256
    opt("gemm_fp8_persist", 1)
    for it in range(3):
        got = run()
        assert _lib.get_option("last_gemm_tile") in (8256, 9256)
        assert torch.equal(got, want), (kind, it)
    if (M, N, K) == (4 * 4608, 3072, 3072):
        assert _lib.get_option("last_gemm_tile") == 9256          # the 96-tile remainder went through the chained stream-K segments
        opt("gemm_streamk", 0)
        assert torch.equal(run(), want) and _lib.get_option("last_gemm_tile") == 8256
    ops.streamk_check(sync=True)
    if kind == "plain":
        ref = deq(A8[:64], sa[:64]) @ deq(W8, sw).T + b.float().cpu()
        assert rel_l2(got[:64], ref) < 4e-3


def test_persistent_e4m3_batched_launch_and_fused_qkv_at_full_width(opt):
    """The model's launches: (1) the batched image-stream geometry (batch items behind a text offset, per-item row scales) and (2)
    x2i_gemm_qkv_fp8 at H = 24, K = 3072 on the flattened single-block rows -- persistent form vs one-tile kernel, bit for bit."""
    from x2i_amd import _lib, ops
    g = torch.Generator(device=DEV).manual_seed(11)
    B, St, Si, Kd, N = 4, 512, 4096, 3072, 3072
    S = St + Si
    X = (torch.randn((B * S, Kd), device=DEV, generator=g)).bfloat16()
    X8, sx = ops.quantize_rows_fp8(X)
    W8, sw = ops.quantize_rows_fp8((torch.randn((N, Kd), device=DEV, generator=g) * 0.03).bfloat16())
    bias = torch.randn((N,), device=DEV, generator=g).bfloat16()
    sxi = sx.view(B, S)[:, St:].contiguous()
    gate = torch.randn((B, N), device=DEV, generator=g)
    res0 = torch.randn((B, S, N), device=DEV, generator=g).bfloat16()

    def batched():
        out = res0.clone()
        ops.gemm_fp8(X8, W8, bias, out=out, M=Si, batch=B, a_batch_stride=S * Kd, lda=Kd, a_offset=St * Kd, a_scale=sxi, a_scale_batch_stride=Si,
                     w_scale=sw, c_batch_stride=S * N, ldc=N, c_offset=St * N, res=out, res_batch_stride=S * N, ldr=N, res_offset=St * N,
                     gate=gate, gate_batch_stride=N)
        return out
    opt("gemm_fp8_persist", 0)
    want = batched()
    opt("gemm_fp8_persist", 1)
    got = batched()
    assert _lib.get_option("last_gemm_tile") in (8256, 9256) and torch.equal(got, want)
    assert torch.equal(got[:, :St], res0[:, :St])                      # text rows untouched
    # fused QKV, flattened rows (2592 tiles: 10 full rounds + a 32-tile stream-K remainder)
    H = 24
    D = H * 128
    Spad = ops.pad128(S)
    Wq8, swq = ops.quantize_rows_fp8((torch.randn((3 * D, Kd), device=DEV, generator=g) * 0.02).bfloat16())
    bq = (torch.randn((3 * D,), device=DEV, generator=g) * 0.5).bfloat16()
    nq, nk = ((1 + 0.2 * torch.randn((128,), device=DEV, generator=g)).bfloat16() for _ in range(2))
    ang = torch.randn((S, 64), device=DEV, generator=g) * 3
    cos, sin = torch.cos(ang).repeat_interleave(2, 1).contiguous(), torch.sin(ang).repeat_interleave(2, 1).contiguous()

    def qkv():
        z = lambda *sh: torch.zeros(sh, device=DEV, dtype=torch.bfloat16)
        Q, K_, VT = z(B, H, Spad, 128), z(B, H, Spad, 128), z(B, H, 128, Spad)
        ops.gemm_qkv_fp8(X8, Wq8, bq, Q, K_, VT, nq, nk, cos, sin, M=B * S, H=H, Spad=Spad, tok_off=0, rows_per_sample=S, a_scale=sx, w_scale=swq,
                         q_scale=0.1275)
        return Q, K_, VT
    opt("gemm_fp8_persist", 0)
    w3 = qkv()
    opt("gemm_fp8_persist", 1)
    g3 = qkv()
    assert _lib.get_option("last_gemm_tile") == 9256
    for a_, b_ in zip(g3, w3):
        assert torch.equal(a_, b_)
    ops.streamk_check(sync=True)
