"""Per-kernel parity: every C-ABI entry point (through x2i_amd.ops -> ctypes -> libx2i_hip.so) against the CPU
oracle's primitive on the same seeded inputs.  Tolerances (stated per test): bf16 storage with fp32 accumulate ->
rel-L2 <= 1e-2 vs the fp32 oracle evaluated on the SAME bf16-rounded inputs."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import primitives as P
from tests.util import rel_l2, seeded, span_permute

pytestmark = pytest.mark.gpu
DEV = "cuda"


def bf(x):
    return x.to(torch.bfloat16)


def g(x):
    return x.to(DEV)


@pytest.fixture(scope="module")
def ops():
    from x2i_amd import ops as o
    o._lib.load()
    return o


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 512), (200, 320, 192), (40, 64, 256), (513, 768, 1024),
                                   (96, 64, 3072)])
def test_gemm_plain_bias(ops, M, N, K):
    A, W, b = bf(seeded((M, K), 1)), bf(seeded((N, K), 2, 0.05)), bf(seeded((N,), 3))
    out = ops.gemm(g(A), g(W), g(b))
    ref = F.linear(A.float(), W.float(), b.float())
    assert rel_l2(out, ref) < 1e-2


@pytest.fixture
def opt(ops):
    """Set library A/B switches for one test (restored afterwards)."""
    from x2i_amd import _lib
    saved = {}

    def set_(name, value):
        old = _lib.set_option(name, value)
        saved.setdefault(name, old)
    yield set_
    for k, v in saved.items():
        _lib.set_option(k, v)


@pytest.mark.parametrize("tile", [128, 256])
@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (512, 768, 128), (700, 520, 320), (1024, 1024, 3072)])
def test_gemm_both_tile_kernels(ops, opt, tile, M, N, K):
    """Force the 128x128 kernel and the 256x256 full-line staging kernel on the same problems (ragged edges, 1..48 K-tiles)."""
    opt("gemm_tile", tile)
    A, W, b = bf(seeded((M, K), 40)), bf(seeded((N, K), 41, 0.05)), bf(seeded((N,), 42))
    res = bf(seeded((M, N), 43))
    gate = seeded((1, N), 44)
    out = ops.gemm(g(A), g(W), g(b), act=1)
    assert rel_l2(out, F.gelu(F.linear(A.float(), W.float(), b.float()), approximate="tanh")) < 1e-2
    out = g(res).clone()
    ops.gemm(g(A), g(W), g(b), out=out, res=out, gate=g(gate))
    assert rel_l2(out, res.float() + gate * F.linear(A.float(), W.float(), b.float())) < 1e-2


def test_gemm_detects_transposes_identity_times_asymmetric():
    from x2i_amd import ops
    K = N = 128
    A = bf(torch.eye(128))
    W = bf(torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251 / 64.0)
    out = ops.gemm(g(A), g(W))
    assert torch.equal(out.float().cpu(), W.float().t())


@pytest.mark.parametrize("act,fn", [(1, lambda x: F.gelu(x, approximate="tanh")), (2, F.gelu), (3, F.silu)])
def test_gemm_activation_epilogues(ops, act, fn):
    M, N, K = 192, 256, 320
    A, W, b = bf(seeded((M, K), 4)), bf(seeded((N, K), 5, 0.08)), bf(seeded((N,), 6))
    out = ops.gemm(g(A), g(W), g(b), act=act)
    assert rel_l2(out, fn(F.linear(A.float(), W.float(), b.float()))) < 1e-2


def test_gemm_gate_residual_inplace_batched_strided(ops):
    # the DiT pattern: X[b, S0:, :] = X[b, S0:, :] + gate[b] * (A[b, S0:, :] W^T + bias), joint buffers [B,S,D]
    B, S, S0, D, K = 2, 200, 72, 256, 512
    X = bf(seeded((B, S, D), 7))
    Aj = bf(seeded((B, S, K), 8))
    W, bias = bf(seeded((D, K), 9, 0.05)), bf(seeded((D,), 10))
    gate = seeded((B, 3 * D), 11)  # gate lives inside a wider modulation table
    Xg = g(X).clone()
    ops.gemm(g(Aj), g(W), g(bias), out=Xg, M=S - S0, batch=B, a_batch_stride=S * K, lda=K, a_offset=S0 * K,
             c_batch_stride=S * D, ldc=D, c_offset=S0 * D, res=Xg, res_batch_stride=S * D, ldr=D, res_offset=S0 * D,
             gate=g(gate)[:, D:], gate_batch_stride=3 * D)
    ref = X.float().clone()
    lin = F.linear(Aj.float()[:, S0:], W.float(), bias.float())
    ref[:, S0:] = ref[:, S0:] + gate[:, None, D:2 * D] * lin
    assert torch.equal(Xg[:, :S0].cpu(), X[:, :S0])  # untouched rows
    assert rel_l2(Xg[:, S0:], ref[:, S0:]) < 1e-2


def test_gemm_dual_output_and_f32(ops):
    M, N, K = 130, 192, 256
    A, W = bf(seeded((M, K), 12)), bf(seeded((N, K), 13, 0.06))
    out2 = torch.empty((M, N), device=DEV, dtype=torch.bfloat16)
    out = ops.gemm(g(A), g(W), out2=out2, act2=2)
    ref = F.linear(A.float(), W.float())
    assert rel_l2(out, ref) < 1e-2 and rel_l2(out2, F.gelu(ref)) < 1e-2
    o32 = ops.gemm(g(A), g(W), out_f32=True)
    assert o32.dtype == torch.float32 and rel_l2(o32, ref) < 1e-3


def test_gemm_generic_fallback_odd_k(ops):
    M, N, K = 37, 50, 27
    A, W, b = bf(seeded((M, K), 14)), bf(seeded((N, K), 15)), bf(seeded((N,), 16))
    out = ops.gemm(g(A), g(W), g(b), act=3)
    assert rel_l2(out, F.silu(F.linear(A.float(), W.float(), b.float()))) < 1e-2


def test_gemm_ldc_column_slab(ops):
    # single-block pattern: MLP-in GEMM writes GELU output into columns [D, 5D) of the [M, 5D] cat buffer
    M, D, K = 136, 128, 128
    A, W, b = bf(seeded((M, K), 17)), bf(seeded((4 * D, K), 18, 0.08)), bf(seeded((4 * D,), 19))
    cat = torch.zeros((M, 5 * D), device=DEV, dtype=torch.bfloat16)
    ops.gemm(g(A), g(W), g(b), out=cat, ldc=5 * D, c_offset=D, act=1)
    ref = F.gelu(F.linear(A.float(), W.float(), b.float()), approximate="tanh")
    assert torch.all(cat[:, :D] == 0) and rel_l2(cat[:, D:], ref) < 1e-2


# ------------------------------------------------------------------------------------------------ attention path
def _attention_case(ops, B, H, S, S0, seed, spike=False):
    D = H * 128
    ids = torch.cat([torch.zeros(S0, 3), torch.stack([torch.zeros(S - S0), torch.arange(S - S0) // 7, torch.arange(S - S0) % 7], 1).float()])
    cos, sin = P.flux_pos_embed(ids)
    qkv0 = bf(seeded((B, S0, 3 * D), seed)) if S0 else None
    qkv1 = bf(seeded((B, S - S0, 3 * D), seed + 1))
    if spike:  # force a late running-max jump (online-softmax rescale branch)
        qkv1[0, -5, :128] *= 6.0
        qkv1[0, -3, D:D + 128] *= 6.0
    nw = [bf(1 + 0.1 * seeded((128,), seed + 2 + i)) for i in range(4)]
    Spad = ops.pad128(S)
    Q = torch.zeros((B, H, Spad, 128), device=DEV, dtype=torch.bfloat16)
    K = torch.zeros_like(Q)
    VT = torch.zeros((B, H, 128, Spad), device=DEV, dtype=torch.bfloat16)
    ops.qkv_split(g(qkv0) if S0 else None, g(qkv1), 3 * D, 3 * D, B, S, S0, H, g(nw[0]), g(nw[1]), g(nw[2]), g(nw[3]),
                  g(cos), g(sin), Q, K, VT, Spad)
    # oracle: heads view -> rms_norm -> cat [txt, img] -> rope
    def heads(t):
        return t.view(B, -1, H, 128).transpose(1, 2)
    parts = []
    for src, (wq, wk) in ((qkv0, (nw[0], nw[1])), (qkv1, (nw[2], nw[3]))):
        if src is None:
            continue
        s = src.float()
        q, k, v = s[..., :D], s[..., D:2 * D], s[..., 2 * D:]
        parts.append((P.rms_norm(heads(q), wq.float()), P.rms_norm(heads(k), wk.float()), heads(v)))
    q = P.apply_rotary_emb(torch.cat([p[0] for p in parts], 2), (cos, sin))
    k = P.apply_rotary_emb(torch.cat([p[1] for p in parts], 2), (cos, sin))
    v = torch.cat([p[2] for p in parts], 2)
    assert rel_l2(Q[:, :, :S], q) < 6e-3 and rel_l2(K[:, :, :S], k) < 6e-3
    assert torch.equal(VT[:, :, :, :S].cpu().float(), v.transpose(2, 3))
    assert torch.all(Q[:, :, S:] == 0) and torch.all(VT[:, :, :, S:] == 0)
    out = torch.zeros((B, S, D), device=DEV, dtype=torch.bfloat16)
    ops.attention(Q, K, VT, out, B, H, S, Spad, D, S * D, 1.0 / math.sqrt(128))
    # reference attention on the bf16-rounded q,k,v the kernel saw
    qb, kb = Q[:, :, :S].float().cpu(), K[:, :, :S].float().cpu()
    ref = F.scaled_dot_product_attention(qb, kb, v).transpose(1, 2).reshape(B, S, D)
    return rel_l2(out, ref)


@pytest.mark.parametrize("B,H,S,S0", [(1, 1, 64, 0), (2, 2, 136, 40), (1, 2, 1152, 128), (1, 1, 200, 200)])
def test_qkv_split_and_attention(ops, B, H, S, S0):
    S0 = min(S0, S)
    if S0 == S:  # everything from source 0 is not a supported call shape; use source 1 only
        S0 = 0
    assert _attention_case(ops, B, H, S, S0, 100 + S) < 1e-2


def test_attention_forced_rescale_branch(ops):
    assert _attention_case(ops, 1, 1, 640, 64, 300, spike=True) < 1e-2


_FORM_CASES = [(1, 1, 64, 0), (2, 2, 136, 40), (1, 2, 1152, 128), (1, 3, 700, 100), (2, 1, 2000, 0)]


@pytest.mark.parametrize("variant", [4, 8, 9])
@pytest.mark.parametrize("B,H,S,S0", _FORM_CASES)
def test_attention_kernel_forms(ops, opt, variant, B, H, S, S0):
    """attn_variant 4 = the 4-wave kernel, 8 = the 8-wave ping-pong kernel (attention_pp.hip, the product schedule), 9 = the
    hand-scheduled one-wave-per-SIMD kernel (attention_w4.hip: 1, 2, 3, 11, 18 and 32 key tiles, ragged last tiles), on
    ragged sequence lengths (S % 64 != 0, S % 256 != 0, fewer rows than one 256-row workgroup) and with the forced-rescale spike."""
    opt("attn_variant", variant)
    assert _attention_case(ops, B, H, S, S0, 200 + S) < 1e-2
    assert _attention_case(ops, 1, 1, 640, 64, 300, spike=True) < 1e-2


@pytest.mark.ablation
@pytest.mark.parametrize("variant", [5, 6, 7, 10])
@pytest.mark.parametrize("B,H,S,S0", _FORM_CASES)
def test_attention_ab_forms_of_the_measurement_library(ops, opt, variant, B, H, S, S0):
    """The A/B forms that live only in libx2i_hip_ablate.so since round 6: 5 / 6 = the ping-pong kernel's schedule 0 with / without defer-max,
    7 = its schedule 1, 10 = the compiler-scheduled kernel on 16 x 16 x 32 MFMAs (attention16.hip)."""
    opt("attn_variant", variant)
    assert _attention_case(ops, B, H, S, S0, 200 + S) < 1e-2
    assert _attention_case(ops, 1, 1, 640, 64, 300, spike=True) < 1e-2


def test_attention_ping_pong_equals_four_wave_kernel_closely(ops, opt):
    """Same arithmetic per query row (tile order, defer-max rule, exp2 domain); only the wave -> row mapping differs, so the two
    kernels agree to the last bits of bf16 on the model's shape (B = 1, 24 heads, S = 4608)."""
    import math
    B, H, S = 1, 24, 4608
    Spad = ops.pad128(S)
    gen = torch.Generator(device=DEV).manual_seed(9)
    Q = torch.randn((B, H, Spad, 128), device=DEV, generator=gen).bfloat16()
    K = torch.randn((B, H, Spad, 128), device=DEV, generator=gen).bfloat16()
    VT = torch.randn((B, H, 128, Spad), device=DEV, generator=gen).bfloat16()
    outs = []
    for v in (4, 8):
        opt("attn_variant", v)
        O = torch.empty((B, S, H * 128), device=DEV, dtype=torch.bfloat16)
        ops.attention(Q, K, VT, O, B, H, S, Spad, H * 128, S * H * 128, 1 / math.sqrt(128))
        outs.append(O)
    assert torch.equal(outs[0], outs[1])


def test_attention_hand_scheduled_kernel_on_the_model_shape(ops, opt):
    """attention_w4.hip at the model's shape -- B = 2, 24 heads, S = 4608.  It is the automatic choice when Q already carries
    softmax_scale * log2(e) (x2i_qkv_desc.q_scale: the caller passes scale = ln 2): checked against fp32 softmax(ln 2 * Q~ K^T) V on
    sampled heads, where it must be as close as the 4-wave kernel is to ITS fp32 reference on the unscaled Q (both only round P);
    its log2-sum-exp rows against torch.logsumexp; twice in a row bit for bit (no state leaks between launches).  Forced on an
    unscaled Q (variant 9) it rescales its bf16 Q fragments itself: within bf16 rounding of P and of that second rounding of Q of
    the 4-wave kernel (which also takes the defer-max decision over different row groups: close, not identical)."""
    import math
    B, H, S = 2, 24, 4608
    Spad = ops.pad128(S)
    scale = 1 / math.sqrt(128)
    LOG2E = 1.4426950408889634
    gen = torch.Generator(device=DEV).manual_seed(9)
    Q32 = torch.randn((B, H, Spad, 128), device=DEV, generator=gen) * 1.5
    Q, Qs = Q32.bfloat16(), (Q32 * (scale * LOG2E)).bfloat16()
    K = (torch.randn((B, H, Spad, 128), device=DEV, generator=gen) * 1.5).bfloat16()
    VT = torch.randn((B, H, 128, Spad), device=DEV, generator=gen).bfloat16()

    def run(v, q, sc, with_lse=False):
        opt("attn_variant", v)
        O = torch.empty((B, S, H * 128), device=DEV, dtype=torch.bfloat16)
        if with_lse:
            lse = torch.zeros((B, H, Spad), device=DEV)
            ops.attention_lse(q, K, VT, O, lse, B, H, S, Spad, H * 128, S * H * 128, sc)
            return O, lse
        ops.attention(q, K, VT, O, B, H, S, Spad, H * 128, S * H * 128, sc)
        return O
    o9 = run(0, Qs, math.log(2))       # automatic choice = the hand-scheduled kernel at this size and scale
    o9b, lse = run(9, Qs, math.log(2), with_lse=True)
    o4 = run(4, Q, scale)
    o8 = run(0, Q, scale)              # any other scale: the ping-pong kernel, as before
    o9r = run(9, Q, scale)             # forced: in-kernel rescale of Q
    assert torch.equal(o9, o9b)
    assert rel_l2(o8, o4) < 4e-3 and rel_l2(o9r, o4) < 6e-3 and rel_l2(o9, o4) < 6e-3
    for (b, h) in ((0, 0), (1, 23), (1, 7)):
        k, v = K[b, h, :S].float(), VT[b, h, :, :S].float().t()
        sc4 = scale * Q[b, h, :S].float() @ k.t()
        sc9 = math.log(2) * Qs[b, h, :S].float() @ k.t()
        e9 = rel_l2(o9[b, :, h * 128:(h + 1) * 128], torch.softmax(sc9, -1) @ v)
        e4 = rel_l2(o4[b, :, h * 128:(h + 1) * 128], torch.softmax(sc4, -1) @ v)
        print(f"attention vs fp32 on its own inputs, head ({b},{h}): hand-scheduled {e9:.3e}, 4-wave {e4:.3e}")
        assert e9 < 4e-3 and e9 < 1.1 * e4 + 1e-4
        assert rel_l2(lse[b, h, :S], torch.logsumexp(sc9, -1) * LOG2E) < 1e-4


# ------------------------------------------------------------------------------------------------ norms / small linears
def test_ln_modulate_two_streams(ops):
    B, S, S0, D = 2, 50, 18, 3072
    X = bf(seeded((B, S, D), 20, 2.0) + 0.5)
    mod = seeded((B, 4 * D), 21, 0.5)
    Y = torch.empty((B, S, D), device=DEV, dtype=torch.bfloat16)
    m = g(mod)
    ops.ln_modulate(g(X), Y, B, S, D, S0, m[:, 0:], m[:, D:], m[:, 2 * D:], m[:, 3 * D:], 4 * D)
    ln = P.layer_norm_plain(X.float())
    ref = torch.empty_like(ln)
    ref[:, :S0] = ln[:, :S0] * (1 + mod[:, None, D:2 * D]) + mod[:, None, 0:D]
    ref[:, S0:] = ln[:, S0:] * (1 + mod[:, None, 3 * D:]) + mod[:, None, 2 * D:3 * D]
    assert rel_l2(Y, ref) < 5e-3


@pytest.mark.parametrize("D", [64, 896, 2048, 3584])
def test_ln_affine(ops, D):
    X, w, b = bf(seeded((3, 7, D), 22, 3.0)), bf(1 + 0.1 * seeded((D,), 23)), bf(0.1 * seeded((D,), 24))
    out = ops.ln_affine(g(X), g(w), g(b), 1e-6)
    assert rel_l2(out, F.layer_norm(X.float(), (D,), w.float(), b.float(), 1e-6)) < 5e-3


@pytest.mark.parametrize("B,N,K", [(1, 96, 256), (4, 3072, 768), (3, 1000, 3072), (9, 64, 128)])
def test_skinny_linear(ops, B, N, K):
    X, W, b = seeded((B, K), 25), bf(seeded((N, K), 26, 0.05)), bf(seeded((N,), 27))
    out = ops.skinny_linear(g(X), g(W), g(b), act_in=3, act_out=0)
    ref = F.linear(F.silu(X), W.float(), b.float())
    assert rel_l2(out, ref) < 1e-4
    out2 = ops.skinny_linear(g(bf(X)), g(W), None, act_in=0, act_out=3)
    assert rel_l2(out2, F.silu(F.linear(bf(X).float(), W.float()))) < 1e-4
    acc = ops.skinny_linear(g(X), g(W), g(b), out=out.clone(), act_in=3, accumulate=True)
    assert rel_l2(acc, 2 * ref) < 1e-4


def test_timestep_sinusoid(ops):
    t = torch.tensor([0.0, 250.0, 752.0, 1000.0])
    for dim in (256, 128):
        out = ops.timestep_sinusoid(g(t), dim)
        assert (out.cpu() - P.timesteps_proj(t, dim)).abs().max() < 2e-3  # fp32 cos/sin of arguments up to 1e3


def test_euler_step(ops):
    from oracle import sampler as OS
    x, e = bf(seeded((3, 4096, 64), 28)), bf(seeded((3, 4096, 64), 29))
    xg = g(x).clone()
    ops.euler_step_(xg, g(e), torch.tensor([-0.25], device=DEV))
    assert torch.equal(xg.cpu(), OS.euler_step(x, e, torch.tensor(0.75), torch.tensor(0.5)))
    x, e = bf(seeded((1, 3, 5), 30)), bf(seeded((1, 3, 5), 31))  # ragged tail path
    xg = g(x).clone()
    ops.euler_step_(xg, g(e), torch.tensor([0.5], device=DEV))
    assert torch.equal(xg.cpu(), (x.float() + 0.5 * e.float()).bfloat16())


# ------------------------------------------------------------------------------------------------ projector stage
@pytest.mark.parametrize("B,C,S,H", [(1, 3, 6, 64), (2, 29, 40, 896), (1, 37, 20, 2048)])
def test_proj_conv5x5(ops, B, C, S, H):
    x = bf(seeded((B, C, S, H), 32, 3.0))
    w, b = seeded((1, C, 5, 5), 33) / (C * 25) ** 0.5, seeded((1,), 34)
    out = ops.proj_conv5x5(g(x), g(w.reshape(C, 25).contiguous()), g(b))
    ref = F.conv2d(x.float(), w, b, padding=2).squeeze(1)
    assert rel_l2(out, ref) < 5e-3


@pytest.mark.parametrize("B,C,S,H", [(1, 3, 6, 64), (2, 29, 40, 896), (1, 37, 20, 2048), (2, 1, 12, 256), (1, 4, 13, 264), (1, 37, 512, 2048)])
def test_proj_conv5x5_matrix_core_form(ops, opt, B, C, S, H):
    """x2i_proj_conv5x5_pack + x2i_proj_conv5x5_packed_bf16 (banded-Toeplitz MFMA form, the one proj.py uses) against
    F.conv2d in fp32 on the same bf16 inputs and bf16-rounded taps (tight: only the summation order differs), against the
    unrounded taps (the VALU form's tolerance), and against the VALU form itself.  Shapes: single row block, ragged row blocks
    (S % 12 != 0), ragged column blocks (H % 256 != 0, H % 16 != 0), odd / even / single layer counts, the full Qwen2.5-VL-3B
    plane."""
    x = bf(seeded((B, C, S, H), 32, 3.0))
    w, b = seeded((1, C, 5, 5), 33) / (C * 25) ** 0.5, seeded((1,), 34)
    wg = g(w.reshape(C, 25).contiguous())
    table = ops.proj_conv5x5_pack(wg)
    assert table.shape == (C, 5, 64, 8)
    # Toeplitz fragments: lane (n, g), element j holds tap 8 g + j - n - 6 of the kernel row
    t = table.float().cpu().view(C, 5, 4, 16, 8)
    wb = bf(w).float().view(C, 5, 5)
    for gi, n, j in ((0, 0, 6), (1, 5, 3), (1, 0, 2), (2, 15, 7), (3, 15, 1), (0, 3, 0), (3, 0, 0)):
        dh = 8 * gi + j - n - 6
        want = wb[:, :, dh] if 0 <= dh <= 4 else torch.zeros(C, 5)
        assert torch.equal(t[:, :, gi, n, j], want)
    out = ops.proj_conv5x5_packed(g(x), table, g(b))
    ref_b = F.conv2d(x.float(), bf(w).float(), b, padding=2).squeeze(1)
    assert rel_l2(out, ref_b) < 3e-3  # bf16 output rounding
    assert rel_l2(out, F.conv2d(x.float(), w, b, padding=2).squeeze(1)) < 5e-3
    valu = ops.proj_conv5x5(g(x), wg, g(b))
    assert rel_l2(out, valu.float().cpu()) < 3e-3
    # staging variants (layers per stage x ring depth) are the same arithmetic in the same order
    for v in (1, 2, 3):
        opt("conv5_variant", v)
        assert torch.equal(ops.proj_conv5x5_packed(g(x), table, g(b)), out), v
    opt("conv5_variant", 0)
    # no bias pointer, and a table / layer-count mismatch is refused
    out0 = ops.proj_conv5x5_packed(g(x), table, None)
    assert rel_l2(out0, ref_b - b) < 3e-3
    if C > 1:
        with pytest.raises(ValueError):
            ops.proj_conv5x5_packed(g(x[:, :-1].contiguous()), table, g(b))


def test_proj_layer_mean_and_seq_mean(ops):
    x = bf(seeded((2, 25, 8, 896), 35, 3.0))
    sc = seeded((25,), 36)
    out = ops.proj_layer_mean(g(x), g(sc))
    assert rel_l2(out, (sc.view(1, 25, 1, 1) * x.float()).mean(1)) < 5e-3
    out = ops.proj_layer_mean(g(x), None)
    assert rel_l2(out, x.float().mean(1)) < 5e-3
    y = seeded((2, 12, 768), 37)
    assert rel_l2(ops.seq_mean(g(y)), y.mean(1)) < 1e-5


def test_errors_are_reported_not_fatal(ops):
    from x2i_amd._lib import X2IError
    with pytest.raises(X2IError):
        ops.attention(torch.zeros(1, device=DEV), torch.zeros(1, device=DEV), torch.zeros(1, device=DEV),
                      torch.zeros(8, device=DEV, dtype=torch.bfloat16), 1, 1, 100, 100, 128, 0, 1.0)  # Spad % 128 != 0
    with pytest.raises(X2IError):
        ops.gemm(torch.zeros((4, 8)), torch.zeros((4, 8)))  # CPU tensors: no fallback


def test_gemm_tile_quantisation_split_is_bit_identical(ops, opt):
    """M=4x1152 rows, N=3072 -> 4*5*12 = 240 + ... the launcher peels the trailing rows of each batch item into a second
    (128x128-tile) launch; both kernels accumulate in the same order, so the result must equal the unsplit launch."""
    B, S, N, K = 4, 4608, 3072, 256
    A = torch.randn((B, S, K), device=DEV).bfloat16()
    W, b = (torch.randn((N, K), device=DEV) * 0.05).bfloat16(), torch.randn(N, device=DEV).bfloat16()
    X = torch.randn((B, S, N), device=DEV).bfloat16()
    gate = torch.randn((B, N), device=DEV)

    def run():
        out = X.clone()
        ops.gemm(A, W, b, out=out, M=S, batch=B, a_batch_stride=S * K, lda=K, c_batch_stride=S * N, ldc=N, res=out,
                 res_batch_stride=S * N, ldr=N, gate=gate, gate_batch_stride=N)
        return out
    split = run()                      # 864 tiles -> 768 in the 256-kernel + trailing 512 rows per batch item in the 128-kernel
    opt("gemm_split_tail", 0)
    whole = run()
    assert torch.equal(split, whole)
    ref = X[1].float() + gate[1] * F.linear(A[1].float(), W.float(), b.float())
    assert rel_l2(split[1], ref.cpu()) < 1e-2


# ------------------------------------------------------------------------------------------------ fused QKV epilogue
@pytest.mark.parametrize("B,H,St,Si,tile", [(2, 2, 40, 216, "128"), (2, 2, 64, 448, "256"), (1, 4, 700, 324, "256"),
                                            (3, 2, 0, 600, "256"), (2, 24, 512, 1024, ""), (1, 24, 0, 5632, ""), (2, 24, 256, 2688, "")])
def test_gemm_qkv_fused_equals_gemm_then_qkv_split(ops, opt, B, H, St, Si, tile):
    """x2i_gemm_qkv_bf16 (norm + RoPE + head split + V transpose in the GEMM epilogue) against the two-step form
    x2i_gemm_bf16 -> x2i_qkv_split_bf16, on both tile kernels, with ragged text lengths (St = 700: unaligned token offsets
    take the element-wise V^T path), batched (double-block) and flattened (single-block, St = 0) row geometry; the last two
    cases have 792 / 756 tiles of 256^2, so the launcher peels the last tile rows into the 128^2 kernel (row offset path)."""
    if tile:
        opt("gemm_tile", int(tile))
    D, S, Kd = H * 128, St + Si, 256
    Spad = ops.pad128(S)
    W, bias = g(bf(seeded((3 * D, Kd), 40, 0.08))), g(bf(seeded((3 * D,), 41, 0.5)))
    Wc, biasc = g(bf(seeded((3 * D, Kd), 42, 0.08))), g(bf(seeded((3 * D,), 43, 0.5)))
    X = g(bf(seeded((B, S, Kd), 44)))
    nq, nk, nqa, nka = (g(bf(1 + 0.2 * seeded((128,), 45 + i))) for i in range(4))
    ang = seeded((S, 64), 50, 3.0)
    cos, sin = g(torch.cos(ang).repeat_interleave(2, 1).contiguous()), g(torch.sin(ang).repeat_interleave(2, 1).contiguous())

    def bufs():
        return (torch.zeros((B, H, Spad, 128), device=DEV, dtype=torch.bfloat16), torch.zeros((B, H, Spad, 128), device=DEV, dtype=torch.bfloat16),
                torch.zeros((B, H, 128, Spad), device=DEV, dtype=torch.bfloat16))
    Q0, K0, V0 = bufs()
    Q1, K1, V1 = bufs()
    if St > 0:  # double-block geometry: per-sample batched GEMMs over the image rows and the text rows
        QKV = torch.empty((B * S, 3 * D), device=DEV, dtype=torch.bfloat16)
        off_img = B * St * 3 * D
        ops.gemm(X, W, bias, out=QKV, M=Si, batch=B, a_batch_stride=S * Kd, lda=Kd, a_offset=St * Kd, c_batch_stride=Si * 3 * D,
                 ldc=3 * D, c_offset=off_img)
        ops.gemm(X, Wc, biasc, out=QKV, M=St, batch=B, a_batch_stride=S * Kd, lda=Kd, c_batch_stride=St * 3 * D, ldc=3 * D)
        ops.qkv_split(QKV, QKV.view(-1)[off_img:], 3 * D, 3 * D, B, S, St, H, nqa, nka, nq, nk, cos, sin, Q0, K0, V0, Spad)
        ops.gemm_qkv(X, W, bias, Q1, K1, V1, nq, nk, cos, sin, M=Si, H=H, Spad=Spad, tok_off=St, rows_per_sample=Si, batch=B,
                     a_batch_stride=S * Kd, lda=Kd, a_offset=St * Kd)
        ops.gemm_qkv(X, Wc, biasc, Q1, K1, V1, nqa, nka, cos, sin, M=St, H=H, Spad=Spad, tok_off=0, rows_per_sample=St, batch=B,
                     a_batch_stride=S * Kd, lda=Kd)
    else:  # single-block geometry: one flattened GEMM over B*S rows
        QKV = torch.empty((B * S, 3 * D), device=DEV, dtype=torch.bfloat16)
        ops.gemm(X, W, bias, out=QKV, M=B * S)
        ops.qkv_split(None, QKV, 3 * D, 3 * D, B, S, 0, H, None, None, nq, nk, cos, sin, Q0, K0, V0, Spad)
        ops.gemm_qkv(X, W, bias, Q1, K1, V1, nq, nk, cos, sin, M=B * S, H=H, Spad=Spad, tok_off=0, rows_per_sample=S)
    torch.cuda.synchronize()
    assert torch.equal(V1[..., :S], V0[..., :S])          # a pure move of the same bf16 values
    assert float(V1[..., S:].float().abs().max()) == 0.0 if Spad > S else True
    assert torch.equal(Q1, Q0) and torch.equal(K1, K0)    # same arithmetic, same order
    # x2i_qkv_desc.vt_perm: the same V^T with the keys of every 32-key span permuted (what attention_w16.hip reads), Q and K untouched --
    # both tile kernels, aligned runs (two 8-byte pieces) and the element-wise path, the grouped pair launch
    Q2, K2, V2 = bufs()
    if St > 0:
        g_img = dict(A=X, W=W, bias=bias, Q=Q2, K=K2, VT=V2, norm_q=nq, norm_k=nk, cos=cos, sin=sin, M=Si, H=H, Spad=Spad, tok_off=St,
                     rows_per_sample=Si, batch=B, a_batch_stride=S * Kd, lda=Kd, a_offset=St * Kd, vt_perm=True)
        g_txt = dict(A=X, W=Wc, bias=biasc, Q=Q2, K=K2, VT=V2, norm_q=nqa, norm_k=nka, cos=cos, sin=sin, M=St, H=H, Spad=Spad, tok_off=0,
                     rows_per_sample=St, batch=B, a_batch_stride=S * Kd, lda=Kd, vt_perm=True)
        ops.gemm_qkv_pair(g_img, g_txt)
        Q3, K3, V3 = bufs()
        ops.gemm_qkv(**dict(g_img, Q=Q3, K=K3, VT=V3))
        ops.gemm_qkv(**dict(g_txt, Q=Q3, K=K3, VT=V3))
        assert torch.equal(V3, V2)
    else:
        ops.gemm_qkv(X, W, bias, Q2, K2, V2, nq, nk, cos, sin, M=B * S, H=H, Spad=Spad, tok_off=0, rows_per_sample=S, vt_perm=True)
    assert torch.equal(V2, span_permute(V1)) and torch.equal(Q2, Q1) and torch.equal(K2, K1)
    # x2i_qkv_desc.sin == NULL: `cos` is the PAIR-form table f32 [S,64,2] (ops.rope_pairs) -- the same values in half the bytes, fetched two
    # half chunks ahead by the persistent kernel's q / k epilogue (round 6): bit-identical Q / K / V^T on every kernel form of this case
    pairs = ops.rope_pairs(cos, sin)
    assert pairs.shape == (S, 64, 2) and torch.equal(pairs[:, :, 0], cos[:, 0::2]) and torch.equal(pairs[:, :, 1], sin[:, 1::2])
    Q4, K4, V4 = bufs()
    if St > 0:
        ops.gemm_qkv_pair(dict(g_img, Q=Q4, K=K4, VT=V4, cos=pairs, sin=None), dict(g_txt, Q=Q4, K=K4, VT=V4, cos=pairs, sin=None))
    else:
        ops.gemm_qkv(X, W, bias, Q4, K4, V4, nq, nk, pairs, None, M=B * S, H=H, Spad=Spad, tok_off=0, rows_per_sample=S, vt_perm=True)
    assert torch.equal(Q4, Q2) and torch.equal(K4, K2) and torch.equal(V4, V2)
    with pytest.raises(ValueError):
        ops.rope_pairs(cos + torch.arange(128, device=DEV) * 1e-3, sin)   # not an interleaved-pair table: refused


def test_rope_table_matches_float64_reference(ops):
    """x2i_rope_table_f32 vs the float64 torch restatement of FluxPosEmbed (oracle/primitives.py)."""
    ids = torch.zeros((300, 3))
    ids[:, 1] = torch.arange(300) // 20
    ids[:, 2] = torch.arange(300) % 20
    ids[:40] = 0  # text rows
    cos, sin = ops.rope_table(g(ids), (16, 56, 56))
    rc, rs = P.flux_pos_embed(ids, (16, 56, 56))
    assert cos.shape == (300, 128) and torch.allclose(cos.cpu(), rc, atol=1e-6) and torch.allclose(sin.cpu(), rs, atol=1e-6)
    assert torch.equal(cos[:40].cpu(), torch.ones(40, 128)) and torch.equal(sin[:40].cpu(), torch.zeros(40, 128))


@pytest.mark.parametrize("B,H,S,pres", [(1, 24, 4608, 0), (2, 24, 4608, 0), (4, 24, 4608, 0), (3, 24, 4600, 0), (1, 40, 2000, 1), (5, 8, 3100, 0)])
def test_attention_streamk_bit_identical(ops, opt, B, H, S, pres):
    """x2i_attention_vp_ws_bf16: the items of a partly filled last round cut along the key axis (opening parts in front of the last whole round on
    some workgroups, one to three closing parts behind it on the others) and chained through the workspace; one to seven whole rounds, ragged last
    tiles, the kernel's own Q scaling -- bit for bit the undivided launch's output, flags back at zero, no give-up marker; twice in a row on the
    same workspace."""
    import math
    Spad = ops.pad128(S)
    D = H * 128
    gen = torch.Generator(device=DEV).manual_seed(2000 + S + B)
    Q = (torch.randn((B, H, Spad, 128), device=DEV, generator=gen) * 1.5).bfloat16()
    K = (torch.randn((B, H, Spad, 128), device=DEV, generator=gen) * 1.5).bfloat16()
    VTP = torch.randn((B, H, 128, Spad), device=DEV, generator=gen).bfloat16()
    if pres:
        scale = 1 / math.sqrt(128)
    else:
        Q = (Q.float() * (1.4426950408889634 / math.sqrt(128))).bfloat16()
        scale = math.log(2.0)
    opt("attn_streamk", 0)
    ref = torch.zeros((B, S, D), device=DEV, dtype=torch.bfloat16)
    ops.attention(Q, K, VTP, ref, B, H, S, Spad, D, S * D, scale, vt_perm=True)
    opt("attn_streamk", 1)
    for _ in range(2):
        out = torch.full((B, S, D), 7.0, device=DEV, dtype=torch.bfloat16)
        ops.attention(Q, K, VTP, out, B, H, S, Spad, D, S * D, scale, vt_perm=True)
        torch.cuda.synchronize()
        assert torch.isfinite(out.float()).all()
        bad = (out != ref)
        assert not bool(bad.any()), f"{int(bad.sum())} elements differ, first at {bad.nonzero()[0].tolist()}"
    ws = ops._sk_workspace()
    assert int(ws.buf[:4096].view(torch.int32).abs().sum()) == 0      # every flag returned to zero, no give-up marker
    ops.streamk_check(sync=True)
    # without a workspace: the persistent form with whole items only (each unit's exit requests the next item's Q block and first K tiles)
    with ops.streamk_scope(None):
        out = torch.full((B, S, D), 7.0, device=DEV, dtype=torch.bfloat16)
        ops.attention(Q, K, VTP, out, B, H, S, Spad, D, S * D, scale, vt_perm=True)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("B,H,S", [(1, 1, 64), (2, 2, 136), (1, 2, 1152), (1, 3, 700), (2, 1, 2000), (1, 24, 4608)])
def test_attention_hand_scheduled_16x16x32_kernel(ops, opt, B, H, S):
    """attention_w16.hip (attn_variant = 12, A/B): attention_w4.hip's program on v_mfma_f32_16x16x32_bf16 -- P^T from the lane's own registers
    through a key permutation within 32-key spans (V^T arrives permuted: the caller's job until the QKV epilogue writes that order), row sums
    on the matrix pipe.  Against fp32 softmax(Q K^T / sqrt(128)) V on the bf16 inputs (ragged lengths: 1 .. 72 key tiles, masked last tiles),
    its log2-sum-exp rows against torch.logsumexp, and twice in a row bit for bit."""
    import math
    Spad = ops.pad128(S)
    D = H * 128
    gen = torch.Generator(device=DEV).manual_seed(1000 + S)
    Q = (torch.randn((B, H, Spad, 128), device=DEV, generator=gen) * 1.5).bfloat16()
    K = (torch.randn((B, H, Spad, 128), device=DEV, generator=gen) * 1.5).bfloat16()
    VT = torch.randn((B, H, 128, Spad), device=DEV, generator=gen).bfloat16()
    perm = torch.tensor([16 * ((kk >> 2) & 1) + 4 * (kk >> 3) + (kk & 3) for kk in range(32)], device=DEV)
    VTP = VT.view(B, H, 128, Spad // 32, 32)[..., perm].reshape(B, H, 128, Spad).contiguous()
    scale = 1 / math.sqrt(128)
    opt("attn_variant", 12)
    outs = []
    for _ in range(2):
        O = torch.zeros((B, S, D), device=DEV, dtype=torch.bfloat16)
        lse = torch.zeros((B, H, Spad), device=DEV)
        ops.attention_lse(Q, K, VTP, O, lse, B, H, S, Spad, D, S * D, scale)
        outs.append((O, lse))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    O, lse = outs[0]
    for (b, h) in {(0, 0), (B - 1, H - 1)}:
        sc = scale * Q[b, h, :S].float() @ K[b, h, :S].float().t()
        ref = torch.softmax(sc, -1) @ VT[b, h, :, :S].float().t()
        assert rel_l2(O[b, :, h * 128:(h + 1) * 128], ref) < 8e-3, (b, h)
        ref_lse = torch.logsumexp(sc, -1) * 1.4426950408889634
        assert float((lse[b, h, :S] - ref_lse).abs().max()) < 2e-2
        assert bool((lse[b, h, S:] > 1e29).all())
    # the product entry point of that kernel (x2i_attention_vp_bf16: no LSE rows, the softmax scale already in Q as the sampling path has it)
    opt("attn_variant", 0)
    Qs = (Q.float() * (scale * 1.4426950408889634)).bfloat16()
    O2 = torch.zeros((B, S, D), device=DEV, dtype=torch.bfloat16)
    ops.attention(Qs, K, VTP, O2, B, H, S, Spad, D, S * D, math.log(2.0), vt_perm=True)
    opt("attn_variant", 12)
    O3 = torch.zeros((B, S, D), device=DEV, dtype=torch.bfloat16)
    ops.attention(Qs, K, VTP, O3, B, H, S, Spad, D, S * D, math.log(2.0))
    assert torch.equal(O2, O3)
    opt("attn_variant", 0)
    O4 = torch.zeros((B, S, D), device=DEV, dtype=torch.bfloat16)
    ops.attention(Qs, K, VT, O4, B, H, S, Spad, D, S * D, math.log(2.0))     # the 32 x 32 x 16 kernels on the natural layout
    assert rel_l2(O2, O4) < 6e-3
    opt("attn_w16", 0)
    assert not ops.attention_prefers_vt_perm(24, 4608, math.log(2.0))
    opt("attn_w16", 1)
    assert ops.attention_prefers_vt_perm(24, 4608, math.log(2.0)) and not ops.attention_prefers_vt_perm(24, 4608, scale)
    assert not ops.attention_prefers_vt_perm(2, 136, math.log(2.0))
    opt("attn_w16", 2)
    assert ops.attention_prefers_vt_perm(2, 136, math.log(2.0))
