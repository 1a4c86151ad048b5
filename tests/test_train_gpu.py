"""Row N4, backward half: the activation-gradient chain of the frozen transformer (x2i_amd/train.py, csrc/train.hip) against torch
autograd run on the fp32 CPU oracle -- kernel by kernel, then the whole chain (d loss / d encoder_hidden_states, d loss / d pooled)
for explicit tap gradients and for the reference's distillation loss (train/train_qwenvl.py:58-61,613-634)."""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.util import rel_l2, seeded

pytestmark = pytest.mark.gpu
DEV = "cuda"


def bf(x):
    return x.to(torch.bfloat16)


def g(x):
    return x.to(DEV)


@pytest.fixture(scope="module")
def ops():
    from x2i_amd import ops as o
    return o


def test_transpose_batched_and_strided(ops):
    x = bf(seeded((3, 72, 128), 1))
    assert torch.equal(ops.transpose(g(x)).cpu(), x.transpose(1, 2).contiguous())
    # head slices of a [S, D] matrix: batch stride 128, row stride D
    S, H = 40, 3
    m = bf(seeded((S, H * 128), 2))
    out = torch.zeros((H, 128, 64), device=DEV, dtype=torch.bfloat16)
    ops.transpose(g(m), out, batch=H, R=S, C=128, in_bs=128, ld_in=H * 128, out_bs=128 * 64, ld_out=64)
    ref = torch.zeros((H, 128, 64), dtype=torch.bfloat16)
    ref[:, :, :S] = m.view(S, H, 128).permute(1, 2, 0)
    assert torch.equal(out.cpu(), ref)


def test_softmax_pad_and_backward(ops):
    nz, Rt, Rv, Ct, Cv, scale = 2, 16, 13, 128, 100, 0.3
    x = bf(seeded((nz, Rt, Ct), 3, 4.0))
    dp = bf(seeded((nz, Rt, Ct), 4))
    xs = x[:, :Rv, :Cv].float().requires_grad_(True)
    p = F.softmax(scale * xs, dim=-1)
    P = g(x).clone()
    ops.softmax_pad_(P, nz, Rt, Rv, Ct, Cv, scale)
    assert rel_l2(P[:, :Rv, :Cv], p.detach()) < 5e-3
    assert float(P[:, Rv:].float().abs().max()) == 0.0 and float(P[:, :, Cv:].float().abs().max()) == 0.0
    # backward against autograd with the bf16-rounded P the kernel itself uses
    pb = P[:, :Rv, :Cv].float().cpu()
    ref = scale * pb * (dp[:, :Rv, :Cv].float() - (pb * dp[:, :Rv, :Cv].float()).sum(-1, keepdim=True))
    dS = g(dp).clone()
    ops.softmax_bwd_(P, dS, nz, Rt, Rv, Ct, Cv, scale)
    assert rel_l2(dS[:, :Rv, :Cv], ref) < 5e-3
    assert float(dS[:, Rv:].float().abs().max()) == 0.0 and float(dS[:, :, Cv:].float().abs().max()) == 0.0
    (p * dp[:, :Rv, :Cv].float()).sum().backward()
    assert rel_l2(dS[:, :Rv, :Cv], xs.grad) < 2e-2


@pytest.mark.parametrize("B,S,D,R", [(2, 48, 256, 16), (1, 40, 3072, 8), (3, 7, 512, 1)])
def test_ln_modulate_backward_vs_autograd(ops, B, S, D, R):
    x, dy = bf(seeded((B, S, D), 5, 2.0)), bf(seeded((B, S, D), 6))
    sc, dres = seeded((B, D), 7, 0.3), bf(seeded((B, S, D), 8))
    xr = x.float().requires_grad_(True)
    scr = sc.clone().requires_grad_(True)
    sh = torch.zeros((B, D), requires_grad=True)
    y = F.layer_norm(xr, (D,), eps=1e-6) * (1 + scr[:, None]) + sh[:, None]
    (y * dy.float()).sum().backward()
    nw = (S + R - 1) // R
    from x2i_amd import _lib
    old = _lib.get_option("train_rows_wg")
    try:
        for form in (1, 0):   # a workgroup per row group with a thread per eight columns (default) / a wave per row
            _lib.set_option("train_rows_wg", form)
            part = torch.full((B, nw, 2, D), float("nan"), device=DEV, dtype=torch.float32)
            dx = g(dres).clone()
            ops.ln_mod_bwd(g(x), g(dy), g(sc), dx, dx, part, B=B, S=S, D=D, R=R, mult_bs=D)
            assert rel_l2(dx, dres.float() + xr.grad) < 1e-2, form
            out = torch.zeros((B, 2 * D), device=DEV)
            ops.reduce_rows(part, out, np_=nw, len_=D, nz=B, in_zs=nw * 2 * D, in_ps=2 * D, out_zs=2 * D, accumulate=True)
            ops.reduce_rows(part, out, np_=nw, len_=D, nz=B, in_zs=nw * 2 * D, in_ps=2 * D, out_zs=2 * D, accumulate=True, in_offset=D, out_offset=D)
            assert rel_l2(out[:, :D], scr.grad) < 5e-3 and rel_l2(out[:, D:], sh.grad) < 5e-3, form
    finally:
        _lib.set_option("train_rows_wg", old)


def test_ln_modulate_backward_rows_with_a_large_mean(ops):
    """massive-activation rows (mean 200 x the spread): the workgroup-per-row-group form takes the variance as mean((x - mean)^2) like the
    forward and the wave-per-row form do; E[x^2] - mean^2 in f32 would lose it."""
    B, S, D, R = 1, 16, 3072, 8
    x = seeded((B, S, D), 15, 1.0)
    x[:, ::2] += 200.0
    x, dy, sc = bf(x), bf(seeded((B, S, D), 16)), seeded((B, D), 17, 0.3)
    xr = x.float().requires_grad_(True)
    y = F.layer_norm(xr, (D,), eps=1e-6) * (1 + sc[:, None])
    (y * dy.float()).sum().backward()
    from x2i_amd import _lib
    old = _lib.get_option("train_rows_wg")
    got = {}
    try:
        for form in (1, 0):
            _lib.set_option("train_rows_wg", form)
            part = torch.empty((B, S // R, 2, D), device=DEV, dtype=torch.float32)
            dx = torch.zeros((B, S, D), device=DEV, dtype=torch.bfloat16)
            ops.ln_mod_bwd(g(x), g(dy), g(sc), None, dx, part, B=B, S=S, D=D, R=R, mult_bs=D)
            got[form] = dx.float().cpu()
            assert rel_l2(dx, xr.grad) < 1e-2, form
    finally:
        _lib.set_option("train_rows_wg", old)
    assert rel_l2(got[1], got[0]) < 4e-3


def test_gate_backward_and_act_backward(ops):
    B, S, D, R = 2, 32, 256, 8
    dx, t, G = bf(seeded((B, S, D), 9)), bf(seeded((B, S, D), 10)), bf(seeded((B, S, D), 11))
    gate = seeded((B, D), 12)
    nw = S // R
    from x2i_amd import _lib
    old = _lib.get_option("train_rows_wg")
    try:
        for form in (1, 0):
            _lib.set_option("train_rows_wg", form)
            part = torch.full((B, nw, D), float("nan"), device=DEV, dtype=torch.float32)
            dT = torch.empty((B, S, D), device=DEV, dtype=torch.bfloat16)
            ops.gate_bwd(g(dx), g(t), g(gate), g(G), dT, part, B=B, S=S, D=D, R=R, gate_bs=D)
            assert rel_l2(dT, gate[:, None] * dx.float() + G.float()) < 5e-3, form
            dg = torch.zeros((B, D), device=DEV)
            ops.reduce_rows(part, dg, np_=nw, len_=D, nz=B, in_zs=nw * D, in_ps=D, out_zs=D)
            assert rel_l2(dg, (dx.float() * t.float()).sum(1)) < 5e-3, form
            # plain residual form (no gate: dT = dX + G, no partials), ragged last row group
            dT2 = torch.empty((B, S, D), device=DEV, dtype=torch.bfloat16)
            ops.gate_bwd(g(dx), None, None, g(G), dT2, None, B=B, S=S - 3, D=D, R=R, dx_bs=S * D, g_bs=S * D, dt_bs=S * D)
            assert rel_l2(dT2[:, :S - 3], (dx.float() + G.float())[:, :S - 3]) < 5e-3, form
    finally:
        _lib.set_option("train_rows_wg", old)
    from x2i_amd.ops import ACT_GELU_ERF, ACT_GELU_TANH, ACT_SILU
    pre = bf(seeded((64, 256), 13, 2.0))
    for act, fn in ((ACT_GELU_TANH, lambda v: F.gelu(v, approximate="tanh")), (ACT_GELU_ERF, F.gelu), (ACT_SILU, F.silu)):
        pr = pre.float().requires_grad_(True)
        fn(pr).sum().backward()
        d = torch.ones((64, 256), device=DEV, dtype=torch.bfloat16)
        ops.act_bwd_(d, g(pre), act)
        assert rel_l2(d, pr.grad) < 5e-3, act
    d32 = torch.ones((4, 256), device=DEV)
    p32 = seeded((4, 256), 14, 2.0)
    pr = p32.clone().requires_grad_(True)
    F.silu(pr).sum().backward()
    ops.act_bwd_(d32, g(p32), ACT_SILU)
    assert rel_l2(d32, pr.grad) < 1e-5


def test_qkv_split_backward_vs_autograd(ops):
    from oracle import primitives as P
    B, H, St, Si = 2, 2, 8, 24
    S, D = St + Si, H * 128
    Spad = ops.pad128(S)
    q0, q1 = bf(seeded((B * St, 3 * D), 15)), bf(seeded((B * Si, 3 * D), 16))
    nrm = [bf(1 + 0.2 * seeded((128,), 17 + i)) for i in range(4)]
    ang = seeded((S, 64), 21, 3.0)
    cos, sin = torch.cos(ang).repeat_interleave(2, 1).contiguous(), torch.sin(ang).repeat_interleave(2, 1).contiguous()
    dQ, dK, dV = (bf(seeded((B, H, Spad, 128), 22 + i)) for i in range(3))

    def fwd(a0, a1):
        """torch restatement of qkv_split: [B, S, 3, H, 128] -> rmsnorm(q, k) * w -> rope -> [B, H, S, 128]"""
        x = torch.cat((a0.view(B, St, 3, H, 128), a1.view(B, Si, 3, H, 128)), 1)
        outs = []
        for part in range(2):
            w = torch.stack([nrm[part].float()] * St + [nrm[2 + part].float()] * Si)[None, :, None, :]
            v = x[:, :, part]
            v = v * torch.rsqrt(v.pow(2).mean(-1, keepdim=True) + 1e-6) * w
            a, b_ = v[..., 0::2], v[..., 1::2]
            c, s_ = cos[None, :, None, 0::2], sin[None, :, None, 0::2]
            o = torch.stack((a * c - b_ * s_, b_ * c + a * s_), -1).flatten(-2)
            outs.append(o.permute(0, 2, 1, 3))
        outs.append(x[:, :, 2].permute(0, 2, 1, 3))
        return outs

    a0, a1 = q0.float().requires_grad_(True), q1.float().requires_grad_(True)
    Qr, Kr, Vr = fwd(a0, a1)
    ((Qr * dQ[:, :, :S].float()).sum() + (Kr * dK[:, :, :S].float()).sum() + (Vr * dV[:, :, :S].float()).sum()).backward()
    # the restatement is the forward kernel's arithmetic
    Qg, Kg = torch.zeros((B, H, Spad, 128), device=DEV, dtype=torch.bfloat16), torch.zeros((B, H, Spad, 128), device=DEV, dtype=torch.bfloat16)
    VT = torch.zeros((B, H, 128, Spad), device=DEV, dtype=torch.bfloat16)
    ops.qkv_split(g(q0), g(q1), 3 * D, 3 * D, B, S, St, H, g(nrm[0]), g(nrm[1]), g(nrm[2]), g(nrm[3]), g(cos), g(sin), Qg, Kg, VT, Spad)
    assert rel_l2(Qg[:, :, :S], Qr.detach()) < 5e-3 and rel_l2(Kg[:, :, :S], Kr.detach()) < 5e-3
    d0 = torch.empty((B * St, 3 * D), device=DEV, dtype=torch.bfloat16)
    d1 = torch.empty((B * Si, 3 * D), device=DEV, dtype=torch.bfloat16)
    ops.qkv_split_bwd(g(q0), g(q1), 3 * D, 3 * D, d0, d1, 3 * D, 3 * D, B, S, St, H, g(nrm[0]), g(nrm[1]), g(nrm[2]), g(nrm[3]), g(cos), g(sin),
                      g(dQ), g(dK), g(dV), Spad)
    assert rel_l2(d0, a0.grad) < 5e-3 and rel_l2(d1, a1.grad) < 5e-3


def test_skinny_backward_and_kd_loss_rows(ops):
    B, N, K = 3, 5000, 256
    dy, W = seeded((B, N), 30), bf(seeded((N, K), 31, 0.05))
    out = ops.skinny_linear_bwd(g(dy), g(W), chunk=512)
    assert rel_l2(out, dy @ W.float()) < 1e-5
    # distillation loss rows against the reference's formula under autograd
    from x2i_amd.distill import normalize
    rows, D, T = 37, 3072, 3.0
    t, s = bf(seeded((rows, D), 32, 0.7)), bf(seeded((rows, D), 33, 0.9) + 0.3 * seeded((rows, D), 32, 0.7))
    sr = s.float().requires_grad_(True)
    term = F.kl_div(F.softmax(normalize(t.float()) / T, dim=-1).log(), F.softmax(normalize(sr) / T, dim=-1), reduction="sum")
    (term / 4).backward()
    grad = torch.empty((rows, D), device=DEV, dtype=torch.bfloat16)
    rl = torch.empty((rows,), device=DEV)
    ops.kd_loss_rows(g(t), g(s), grad, rl, rows=rows, D=D, temperature=T, loss_scale=0.25)
    assert abs(float(rl.sum()) - float(term.detach())) < 1e-4 * abs(float(term.detach())) + 1e-6
    assert rel_l2(grad, sr.grad) < 1e-2
    flag = torch.tensor([float("nan")], device=DEV)
    ops.zero_if_nonfinite_(grad, flag)
    assert float(grad.float().abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------------------------------
def _tiny():
    from oracle import flux as OF
    from x2i_amd.flux import FluxTransformer2DModel
    cfg = dict(OF.DEFAULT_CFG)
    cfg.update(num_layers=2, num_single_layers=2, num_attention_heads=2, joint_attention_dim=64, pooled_projection_dim=32)
    sd = OF.random_flux_state_dict(cfg, seed=11, std=0.05)
    m = FluxTransformer2DModel(**cfg, device=DEV)
    m.load_state_dict({k: v.bfloat16() for k, v in sd.items()}, strict=True)
    return m, {k: v.bfloat16().float() for k, v in sd.items()}, cfg


def _inputs(B, St, h2, w2):
    from oracle import sampler as OS
    gen = torch.Generator().manual_seed(0)
    hid = torch.randn((B, h2 * w2, 64), generator=gen).bfloat16()
    enc, pooled = torch.randn((B, St, 64), generator=gen).bfloat16(), torch.randn((B, 32), generator=gen).bfloat16()
    ts = torch.tensor([0.5, 0.25, 1.0][:B])
    return hid, enc, pooled, ts, OS.prepare_latent_image_ids(h2, w2), torch.zeros(St, 3)


@pytest.mark.parametrize("fused", [True, False])
def test_activation_gradient_chain_vs_oracle_autograd_with_explicit_tap_gradients(fused):
    """d loss / d encoder_hidden_states and d loss / d pooled_projections for loss = sum over all taps of <tap, G> with random G: the HIP
    chain (bf16 activations, every block's backward) against torch autograd through the fp32 oracle on the same bf16-rounded weights."""
    from oracle import flux as OF
    from x2i_amd.train import DistillBackward
    m, sd, cfg = _tiny()
    B, St, h2, w2 = 2, 24, 6, 8
    Si, D = h2 * w2, 256
    hid, enc, pooled, ts, ids, tids = _inputs(B, St, h2, w2)
    gen = torch.Generator().manual_seed(5)
    G = [[(torch.randn((B, Si, D), generator=gen) * 0.05).bfloat16() for _ in range(2)],
         [(torch.randn((B, St, D), generator=gen) * 0.05).bfloat16() for _ in range(2)],
         [(torch.randn((B, St + Si, D), generator=gen) * 0.05).bfloat16() for _ in range(2)]]
    er, pr = enc.float().requires_grad_(True), pooled.float().requires_grad_(True)
    taps = [[], [], []]
    ref_out = OF.flux_forward(sd, cfg, hid.float(), er, pr, ts, ids, tids, taps=taps)
    loss = sum((t * gg.float()).sum() for k in range(3) for t, gg in zip(taps[k], G[k]))
    loss.backward()
    bw = DistillBackward(m, fused_attention_backward=fused)
    st = bw.prepare_conditioning(enc.to(DEV), pooled.to(DEV), tids.to(DEV), ids.to(DEV))
    out, _ = bw.forward_train(st, hid.to(DEV), ts.to(DEV), tap_grads=[[x.to(DEV) for x in k] for k in G])
    assert rel_l2(out, ref_out.detach()) < 2e-2
    # the saving forward is the sampling forward
    assert rel_l2(out, m.denoise(st, hid.to(DEV), ts.to(DEV))) < 1e-2
    d_enc, d_pooled = bw.backward()
    e1, e2 = rel_l2(d_enc, er.grad), rel_l2(d_pooled, pr.grad)
    print(f"activation-gradient chain (2+2 blocks, {'fused' if fused else 'explicit-matrix'} attention backward): d_enc rel-L2 {e1:.3e}, "
          f"d_pooled rel-L2 {e2:.3e}")
    assert e1 < 2.5e-2 and e2 < 2.5e-2  # measured 9.6e-3 / 5.4e-3


def test_distillation_loss_and_gradient_vs_reference_formula():
    """The reference's step (train/train_qwenvl.py:593-637) on the tiny model: teacher tensors as the batch holds them, loss value and the
    gradient reaching the projector's outputs against autograd through oracle + the reference's loss formula."""
    from oracle import flux as OF
    from x2i_amd.distill import kd_attention_loss
    from x2i_amd.train import DistillBackward
    m, sd, cfg = _tiny()
    B, St, h2, w2 = 2, 24, 6, 8
    Si, D = h2 * w2, 256
    hid, enc, pooled, ts, ids, tids = _inputs(B, St, h2, w2)
    # teacher = the same transformer on different (T5-like) conditioning
    gen = torch.Generator().manual_seed(9)
    enc_t, pooled_t = torch.randn((B, St, 64), generator=gen).bfloat16(), torch.randn((B, 32), generator=gen).bfloat16()
    tt = [[], [], []]
    with torch.no_grad():
        OF.flux_forward(sd, cfg, hid.float(), enc_t.float(), pooled_t.float(), ts, ids, tids, taps=tt)
    teacher = [torch.stack(k, dim=1).bfloat16() for k in tt]
    er, pr = enc.float().requires_grad_(True), pooled.float().requires_grad_(True)
    taps = [[], [], []]
    OF.flux_forward(sd, cfg, hid.float(), er, pr, ts, ids, tids, taps=taps)
    ref_loss = kd_attention_loss([t.float() for t in teacher], [torch.stack(k, dim=1) for k in taps])
    ref_loss.backward()
    bw = DistillBackward(m)
    st = bw.prepare_conditioning(enc.to(DEV), pooled.to(DEV), tids.to(DEV), ids.to(DEV))
    _, loss = bw.forward_train(st, hid.to(DEV), ts.to(DEV), teacher=[t.to(DEV) for t in teacher])
    d_enc, d_pooled = bw.backward()
    print(f"distillation loss: HIP {float(loss):.5f}  reference formula on the oracle {float(ref_loss.detach()):.5f}")
    assert abs(float(loss) - float(ref_loss.detach())) < 2e-3 * abs(float(ref_loss.detach()))  # measured 2.5e-5
    e1, e2 = rel_l2(d_enc, er.grad), rel_l2(d_pooled, pr.grad)
    print(f"distillation gradient: d_enc rel-L2 {e1:.3e}, d_pooled rel-L2 {e2:.3e}")
    assert e1 < 2.5e-2 and e2 < 2.5e-2  # measured 5.1e-3 / 2.0e-3


# ---------------------------------------------------------------------------------------------------------------------
def _tiny_proj(kind, seed=3):
    """Proj7Exp at reduced width with the reference's structure; returns (module on the GPU, fp32 state dict of its bf16 values)."""
    from x2i_amd.proj import Proj7Exp
    pr = Proj7Exp(in_channels=5, input_dim=128, output_dim0=32, output_dim1=64, use_t5=False, use_scale=kind == "scale",
                  use_cnn=kind == "conv", device=DEV).init_random_(seed)
    return pr, {k: v.detach().float().cpu() for k, v in pr.state_dict().items()}


@pytest.mark.parametrize("kind", ["conv", "scale", "mean"])
def test_projector_backward_vs_oracle_autograd(kind):
    """Weight gradients of every projector parameter (layer fusion, LayerNorm, the three linears) from given d prompt_embeds / d pooled,
    against torch autograd through oracle.projector.proj7exp on the same bf16-rounded weights and inputs."""
    from oracle import projector as OP
    from x2i_amd.train import ProjectorTrainer
    pr, sd = _tiny_proj(kind)
    B, C, S, H = 2, 5, 24, 128
    x = bf(seeded((B, C, S, H), 40, 2.0))
    d_enc, d_pooled = bf(seeded((B, S, 64), 41, 0.1)), seeded((B, 32), 42, 0.1)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x1, x2 = OP.proj7exp(sdr, x.float())
    ((x2 * d_enc.float()).sum() + (x1 * d_pooled).sum()).backward()
    tr = ProjectorTrainer(pr)
    pooled, prompt = tr.forward(g(x))
    assert rel_l2(prompt, x2.detach()) < 1e-2 and rel_l2(pooled, x1.detach()) < 1e-2
    tr.backward(g(d_enc), g(d_pooled))
    for n in tr.names:
        if n == "conv.bias":
            # a constant added in front of a LayerNorm has zero gradient: autograd leaves fp32 rounding noise, the bf16 chain its own --
            # both must be negligible against the conv taps' gradient
            assert float(tr.g(n).abs().max()) < 1e-2 * float(tr.g("conv.weight").norm()) and float(sdr[n].grad.abs().max()) < 1e-3
            continue
        e = rel_l2(tr.g(n), sdr[n].grad.reshape(-1))
        print(f"  {kind:5s} {n:28s} rel-L2 {e:.3e}")
        assert e < 2e-2, n
    # gradients accumulate across backward() calls until step()
    g0 = tr.grad.clone()
    tr.backward(g(d_enc), g(d_pooled))
    assert rel_l2(tr.grad, 2 * g0) < 1e-5


def test_clip_and_adamw_vs_fp32_restatement_and_torch():
    """clip_grad_norm_ + AdamW on bf16 parameters: bit-level agreement with an fp32 restatement that rounds the parameter to bf16 after
    every step (what the kernel does), and agreement with torch.optim.AdamW on bf16 parameters to bf16 rounding."""
    from x2i_amd import ops
    n, lr, b1, b2, eps, wd, max_norm = 5000, 1e-2, 0.9, 0.999, 1e-8, 1e-2, 0.5
    p0 = bf(seeded((n,), 50))
    grads = [seeded((n,), 51 + i, 0.05) for i in range(3)]
    p = g(p0).clone()
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pr = p0.float().clone()
    mr, vr = torch.zeros(n), torch.zeros(n)
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pt], lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd)
    for i, gr in enumerate(grads, 1):
        coef = ops.clip_coef(ops.sum_all(g(gr), squares=True), max_norm)
        nrm = float(gr.norm())
        assert abs(float(coef[1]) - nrm) < 1e-4 * nrm and abs(float(coef[0]) - min(1.0, max_norm / (nrm + 1e-6))) < 1e-5
        ops.adamw_(p, g(gr), m, v, lr=lr, beta1=b1, beta2=b2, eps=eps, weight_decay=wd, step=i, coef=coef)
        gc = gr * float(coef[0])
        mr = b1 * mr + (1 - b1) * gc
        vr = b2 * vr + (1 - b2) * gc * gc
        pr = (pr * (1 - lr * wd) - lr * (mr / (1 - b1 ** i)) / ((vr / (1 - b2 ** i)).sqrt() + eps)).bfloat16().float()
        pt.grad = gr.bfloat16()
        torch.nn.utils.clip_grad_norm_([pt], max_norm)
        opt.step()
    assert (p.float().cpu() != pr).float().mean() < 2e-3          # identical up to fp32 contraction order on a handful of elements
    assert rel_l2(p, pr) < 1e-4 and rel_l2(p, pt.detach().float()) < 2e-3


def test_distill_step_end_to_end_loss_decreases_and_matches_autograd():
    """The reference's whole step (projector -> student transformer with the distillation loss at every tap -> backward -> clip ->
    AdamW) on tiny modules: the projector gradients equal autograd through oracle projector + oracle transformer + the reference's loss
    formula, and repeating the step on the same batch lowers the loss."""
    from oracle import flux as OF
    from oracle import projector as OP
    from oracle import sampler as OS
    from x2i_amd.distill import kd_attention_loss
    from x2i_amd.proj import Proj7Exp
    from x2i_amd.train import DistillBackward, ProjectorTrainer, distill_step
    m, sd, cfg = _tiny()
    # projector emitting the tiny transformer's conditioning widths: prompt_embeds 64 (joint_attention_dim), pooled 32
    pr = Proj7Exp(in_channels=5, input_dim=128, output_dim0=32, output_dim1=64, use_t5=False, use_scale=False, use_cnn=True, device=DEV).init_random_(7)
    psd = {k: v.detach().float().cpu() for k, v in pr.state_dict().items()}
    B, St, h2, w2 = 2, 24, 6, 8
    hid, enc_t, pooled_t, ts, ids, tids = _inputs(B, St, h2, w2)
    x = bf(seeded((B, 5, St, 128), 60, 2.0))
    tt = [[], [], []]
    with torch.no_grad():
        OF.flux_forward(sd, cfg, hid.float(), enc_t.float(), pooled_t.float(), ts, ids, tids, taps=tt)
    teacher = [torch.stack(k, dim=1).bfloat16() for k in tt]
    # oracle gradient of the projector parameters
    psr = {k: v.clone().requires_grad_(True) for k, v in psd.items()}
    x1, x2 = OP.proj7exp(psr, x.float())
    taps = [[], [], []]
    # (the projector hands bf16 tensors to the transformer: round in the forward value, identity in the backward)
    OF.flux_forward(sd, cfg, hid.float(), x2 + (x2.bfloat16().float() - x2).detach(), x1 + (x1.bfloat16().float() - x1).detach(), ts, ids, tids,
                    taps=taps)
    ref_loss = kd_attention_loss([t.float() for t in teacher], [torch.stack(k, dim=1) for k in taps])
    ref_loss.backward()
    tr = ProjectorTrainer(pr, lr=2e-3, max_grad_norm=1.0)
    chain = DistillBackward(m)
    kw = dict(teacher=[t.to(DEV) for t in teacher], txt_ids=tids.to(DEV), img_ids=ids.to(DEV))
    loss0 = distill_step(tr, chain, g(x), hid.to(DEV), ts.to(DEV), optimizer_step=False, **kw)
    assert abs(float(loss0) - float(ref_loss.detach())) < 5e-3 * abs(float(ref_loss.detach()))
    for n in tr.names:
        if n == "conv.bias":
            continue  # zero gradient (see test_projector_backward_vs_oracle_autograd)
        e = rel_l2(tr.g(n), psr[n].grad.reshape(-1))
        print(f"  step gradient {n:28s} rel-L2 {e:.3e}")
        assert e < 2.5e-2, n  # measured <= 7e-3
    coef = tr.step()
    assert 0.0 < float(coef[0]) <= 1.0 and float(coef[1]) > 0.0
    losses = [float(loss0)]
    for _ in range(4):
        losses.append(float(distill_step(tr, chain, g(x), hid.to(DEV), ts.to(DEV), **kw)))
    print("  loss over repeated steps on one batch:", " ".join(f"{v:.4f}" for v in losses))
    assert losses[-1] < losses[0]


@pytest.mark.parametrize("B,H,S", [(1, 2, 200), (2, 3, 192), (1, 1, 64), (1, 2, 333 // 8 * 8), (1, 2, 640)])
def test_fused_attention_backward_vs_autograd(ops, B, H, S):
    """x2i_attention_bwd_bf16 (statistics pass, dQ pass, dK / dV pass) against torch autograd through fp32 softmax attention on the same
    bf16 operands; ragged sequence lengths (S % 64 != 0, S % 128 != 0) exercise the key mask and the row neutralisation."""
    Spad = ops.pad128(S)
    scale = 1.0 / math.sqrt(128.0)

    def padded(seed, sc=1.0):
        t = torch.zeros((B, H, Spad, 128))
        t[:, :, :S] = seeded((B, H, S, 128), seed, sc)
        return bf(t)
    Q, K, V = padded(70), padded(71), padded(72)
    dO = bf(seeded((B, S, H * 128), 73))
    q, k, v = (t[:, :, :S].float().requires_grad_(True) for t in (Q, K, V))
    o = torch.softmax(scale * q @ k.transpose(-1, -2), dim=-1) @ v          # [B,H,S,128]
    o_tok = o.permute(0, 2, 1, 3).reshape(B, S, H * 128)
    (o_tok * dO.float()).sum().backward()
    Qg, Kg, Vg, dOg, Og = g(Q), g(K), g(V), g(dO), g(bf(o_tok.detach()))
    QT, KT = ops.transpose(Qg.view(B * H, Spad, 128)), ops.transpose(Kg.view(B * H, Spad, 128))
    dOT = torch.zeros((B * H, 128, Spad), device=DEV, dtype=torch.bfloat16)
    for b in range(B):
        ops.transpose(dOg, dOT[b * H:], batch=H, R=S, C=128, in_bs=128, ld_in=H * 128, out_bs=128 * Spad, ld_out=Spad, in_offset=b * S * H * 128)
    dOh = ops.transpose(dOT)                                                    # [B*H, Spad, 128], zero beyond S
    Dv = torch.empty((B, H, Spad), device=DEV)
    ops.attention_bwd_prep(dOg, Og, Dv, B, H, S, Spad, do_bs=S * H * 128, lddo=H * 128, o_bs=S * H * 128, ldo=H * 128)
    assert rel_l2(Dv[:, :, :S], (o.detach() * dO.float().view(B, S, H, 128).permute(0, 2, 1, 3)).sum(-1)) < 1e-2
    lse = torch.empty((B, H, Spad), device=DEV)
    dQ, dK, dV = (torch.zeros((B, H, Spad, 128), device=DEV, dtype=torch.bfloat16) for _ in range(3))
    ops.attention_bwd(Qg, Kg, Vg, QT, KT, dOh, dOT, lse, Dv, dQ, dK, dV, B, H, S, Spad, scale)
    ref_lse = torch.logsumexp(scale * q.detach() @ k.detach().transpose(-1, -2), dim=-1) * 1.4426950408889634
    assert rel_l2(lse[:, :, :S], ref_lse) < 1e-4
    eq, ek, ev = rel_l2(dQ[:, :, :S], q.grad), rel_l2(dK[:, :, :S], k.grad), rel_l2(dV[:, :, :S], v.grad)
    print(f"fused attention backward B={B} H={H} S={S}: dQ {eq:.3e} dK {ek:.3e} dV {ev:.3e}")
    assert eq < 1.5e-2 and ek < 1.5e-2 and ev < 1.5e-2
    # the forward kernel can hand over the statistics itself (x2i_attention_lse_bf16): same output as x2i_attention_bf16, same lse, and the
    # backward without its statistics pass gives the same gradients
    VT = ops.transpose(Vg.view(B * H, Spad, 128)).view(B, H, 128, Spad)
    o1 = torch.empty((B, S, H * 128), device=DEV, dtype=torch.bfloat16)
    o2 = torch.empty_like(o1)
    lse_f = torch.zeros((B, H, Spad), device=DEV)
    ops.attention(Qg, Kg, VT, o1, B, H, S, Spad, H * 128, S * H * 128, scale)
    ops.attention_lse(Qg, Kg, VT, o2, lse_f, B, H, S, Spad, H * 128, S * H * 128, scale)
    assert torch.equal(o1, o2) and rel_l2(o2, o_tok.detach()) < 1e-2
    assert rel_l2(lse_f[:, :, :S], ref_lse) < 1e-4 and bool((lse_f[:, :, S:] > 1e29).all())
    dQ2, dK2, dV2 = (torch.zeros((B, H, Spad, 128), device=DEV, dtype=torch.bfloat16) for _ in range(3))
    ops.attention_bwd(Qg, Kg, Vg, QT, KT, dOh, dOT, lse_f, Dv, dQ2, dK2, dV2, B, H, S, Spad, scale, have_lse=True)
    for a, b_ in ((dQ2, dQ), (dK2, dK), (dV2, dV)):
        assert rel_l2(a[:, :, :S], b_[:, :, :S].float().cpu()) < 2e-3
    # the two passes as one launch (default) and one after the other: the same blocks, bit for bit
    from x2i_amd import _lib
    old = _lib.set_option("attn_bwd_overlap", 0)
    try:
        dQ3, dK3, dV3 = (torch.zeros((B, H, Spad, 128), device=DEV, dtype=torch.bfloat16) for _ in range(3))
        ops.attention_bwd(Qg, Kg, Vg, QT, KT, dOh, dOT, lse_f, Dv, dQ3, dK3, dV3, B, H, S, Spad, scale, have_lse=True)
    finally:
        _lib.set_option("attn_bwd_overlap", old)
    assert torch.equal(dQ3, dQ2) and torch.equal(dK3, dK2) and torch.equal(dV3, dV2)


@pytest.mark.ablation
@pytest.mark.parametrize("S", [4608, 4600, 200, 640])
def test_pipelined_attention_backward_bit_identical_at_model_length(ops, S):
    """The software-pipelined dQ and dK / dV passes (the product's; one fused launch) against the round-2 phase-after-phase kernels, which live in the
    measurement library (X2I_LIB_VARIANT=ablate: options attn_bwd_pipe / attn_bwd_dq64), at the model's sequence length (72 streamed tiles; S = 4600: a
    ragged last tile, the masked form of the dQ pass) and at short ragged ones -- bit for bit, on random operands; also the 32-row dQ form and the
    passes one after the other."""
    from x2i_amd import _lib
    B, H = 1, 2
    Spad = ops.pad128(S)
    scale = 1.0 / math.sqrt(128.0)
    gen = torch.Generator(device=DEV).manual_seed(90 + S)

    def padded(sc=1.0):
        t = torch.zeros((B, H, Spad, 128), device=DEV)
        t[:, :, :S] = torch.randn((B, H, S, 128), device=DEV, generator=gen) * sc
        return t.bfloat16()
    Q, K, V, dOh4 = padded(), padded(), padded(), padded(0.5)
    dOh = dOh4.view(B * H, Spad, 128)
    QT, KT, dOT = ops.transpose(Q.view(B * H, Spad, 128)), ops.transpose(K.view(B * H, Spad, 128)), ops.transpose(dOh)
    Dv = torch.zeros((B, H, Spad), device=DEV)
    Dv[:, :, :S] = torch.randn((B, H, S), device=DEV, generator=gen) * 0.1
    lse = torch.empty((B, H, Spad), device=DEV)
    outs = {}
    forms = {"product": {}, "serial": {"attn_bwd_overlap": 0}, "round 2": {"attn_bwd_pipe": 0}, "round 2, 32-row dQ": {"attn_bwd_pipe": 0, "attn_bwd_dq64": 0},
             "round 2, serial": {"attn_bwd_pipe": 0, "attn_bwd_overlap": 0}}
    for name, opts in forms.items():
        old = {k: _lib.set_option(k, v) for k, v in opts.items()}
        try:
            dQ, dK, dV = (torch.zeros((B, H, Spad, 128), device=DEV, dtype=torch.bfloat16) for _ in range(3))
            ops.attention_bwd(Q, K, V, QT, KT, dOh, dOT, lse, Dv, dQ, dK, dV, B, H, S, Spad, scale)   # (statistics pass included)
            torch.cuda.synchronize()
        finally:
            for k, v in old.items():
                _lib.set_option(k, v)
        outs[name] = (dQ, dK, dV)
    for name in forms:
        for a, b_ in zip(outs["product"], outs[name]):
            assert torch.isfinite(a.float()).all() and float(a.float().abs().max()) > 0
            assert torch.equal(a, b_), name


def test_full_width_distillation_gradient_vs_oracle_autograd():
    """D = 3072, 24 heads, one double + one single block on 512 + 1024 tokens, B = 2: the training chain at the real GEMM / attention
    shapes (256^2 MFMA kernels in the dgrad launches, the fused attention backward at S = 1536, the 43 008-row modulation table) with
    the reference's loss against teacher tensors -- loss value and d loss / d encoder_hidden_states, d loss / d pooled against torch
    autograd through the fp32 oracle."""
    from oracle import flux as OF
    from oracle import sampler as OS
    from x2i_amd.distill import kd_attention_loss
    from x2i_amd.flux import FluxTransformer2DModel
    from x2i_amd.train import DistillBackward
    cfg = dict(OF.DEFAULT_CFG)
    cfg.update(num_layers=1, num_single_layers=1)
    sd = OF.random_flux_state_dict(cfg, seed=21, std=0.02)
    m = FluxTransformer2DModel(**cfg, device=DEV)
    m.load_state_dict({k: v.bfloat16() for k, v in sd.items()}, strict=True)
    sdf = {k: v.bfloat16().float() for k, v in sd.items()}
    B = 2
    hidden, enc, pooled = bf(seeded((B, 1024, 64), 1)), bf(seeded((B, 512, 4096), 2)), bf(seeded((B, 768), 3))
    enc_t, pooled_t = bf(seeded((B, 512, 4096), 4)), bf(seeded((B, 768), 5))
    ts = torch.tensor([0.5, 0.25])  # t * 1000 is formed in the latents' dtype (lightcontrol_flux.py:447): values exact in bf16, as the fp32 oracle sees them
    img_ids, txt_ids = OS.prepare_latent_image_ids(32, 32), torch.zeros(512, 3)
    tt = [[], [], []]
    with torch.no_grad():
        OF.flux_forward(sdf, cfg, hidden.float(), enc_t.float(), pooled_t.float(), ts, img_ids, txt_ids, taps=tt)
    teacher = [torch.stack(k, dim=1).bfloat16() for k in tt]
    er, pr = enc.float().requires_grad_(True), pooled.float().requires_grad_(True)
    taps = [[], [], []]
    OF.flux_forward(sdf, cfg, hidden.float(), er, pr, ts, img_ids, txt_ids, taps=taps)
    ref_loss = kd_attention_loss([t.float() for t in teacher], [torch.stack(k, dim=1) for k in taps])
    ref_loss.backward()
    bw = DistillBackward(m)
    st = bw.prepare_conditioning(enc.to(DEV), pooled.to(DEV), txt_ids.to(DEV), img_ids.to(DEV))
    _, loss = bw.forward_train(st, hidden.to(DEV), ts.to(DEV), teacher=[t.to(DEV) for t in teacher])
    d_enc, d_pooled = bw.backward()
    e1, e2 = rel_l2(d_enc, er.grad), rel_l2(d_pooled, pr.grad)
    print(f"full-width 1+1 blocks, B=2: loss {float(loss):.5f} vs {float(ref_loss.detach()):.5f}; d_enc rel-L2 {e1:.3e}, d_pooled rel-L2 {e2:.3e}")
    assert abs(float(loss) - float(ref_loss.detach())) < 1e-3 * abs(float(ref_loss.detach()))   # measured 250.90887 vs 250.90886
    assert e1 < 1.5e-2 and e2 < 1.5e-2                                                          # measured 3.8e-3 / 4.0e-4


def test_training_harness_synthetic_tiny_writes_loadable_checkpoints(tmp_path):
    """x2i_amd.train_distill (counterpart of the reference's training loop): a few synthetic steps with gradient accumulation, warm-up
    schedule and checkpointing; the written .bin is the projector state dict the inference loaders accept."""
    from x2i_amd import train_distill as TD
    from x2i_amd.checkpoints import load_projector_checkpoint
    losses = TD.main(["--synthetic", "--tiny", "--batch_size", "2", "--max_train_steps", "4", "--gradient_accumulation_steps", "2",
                      "--checkpointing_steps", "2", "--lr_scheduler", "constant_with_warmup", "--lr_warmup_steps", "2", "--learning_rate", "1e-3",
                      "--output_dir", str(tmp_path), "--seed", "1"])
    assert len(losses) == 4 and all(math.isfinite(v) for v in losses)
    for step in (2, 4):
        f = tmp_path / str(step) / "diffusion_pytorch_model.bin"
        assert f.exists()
    pr = load_projector_checkpoint(str(tmp_path / "4" / "diffusion_pytorch_model.bin"), device=DEV, in_channels=5)
    assert pr.use_cnn and pr.conv.weight.shape == (1, 5, 5, 5)


def test_training_harness_resumes_from_the_newest_checkpoint(tmp_path, capsys):
    """re-launching into the same output_dir continues from the highest-numbered checkpoint (train/train_qwenvl.py:404-409,
    :535) instead of restarting at step 0 and overwriting <output_dir>/2, /4 ... with a fresh run."""
    from x2i_amd import train_distill as TD
    common = ["--synthetic", "--tiny", "--batch_size", "2", "--checkpointing_steps", "2", "--learning_rate", "1e-3", "--output_dir", str(tmp_path),
              "--seed", "1"]
    first = TD.main(common + ["--max_train_steps", "4"])
    assert len(first) == 4
    w4 = torch.load(tmp_path / "4" / "diffusion_pytorch_model.bin")
    again = TD.main(common + ["--max_train_steps", "6"])
    assert len(again) == 2                                  # steps 5 and 6 only
    assert "resuming from" in capsys.readouterr().out
    assert (tmp_path / "6" / "diffusion_pytorch_model.bin").exists()
    w4b = torch.load(tmp_path / "4" / "diffusion_pytorch_model.bin")
    assert all(torch.equal(w4[k], w4b[k]) for k in w4)     # the earlier checkpoint was not rewritten


def test_resumed_optimizer_restarts_moments_and_bias_correction_together(tmp_path):
    """on resume the AdamW moments restart at zero, so the optimizer's step count (the bias corrections) must
    restart with them -- as in the reference, which builds a fresh AdamW and a fresh lr_scheduler and continues only global_step
    (train/train_qwenvl.py:404-409, :447-459, :476-481, :535).  The first post-resume update must not depend on the step NUMBER of the
    checkpoint: resuming the same weights as step 1 and as step 1000 gives bit-identical parameters, and the warm-up restarts."""
    import shutil
    from x2i_amd import train_distill as TD
    base = ["--synthetic", "--tiny", "--batch_size", "2", "--checkpointing_steps", "1", "--learning_rate", "1e-3", "--seed", "1"]
    a, b, c = tmp_path / "a", tmp_path / "b", tmp_path / "c"
    TD.main(base + ["--max_train_steps", "1", "--output_dir", str(a)])
    for d, n in ((b, "1"), (c, "1000")):
        (d / n).mkdir(parents=True)
        shutil.copy(a / "1" / "diffusion_pytorch_model.bin", d / n / "diffusion_pytorch_model.bin")
    TD.main(base + ["--max_train_steps", "2", "--output_dir", str(b)])
    assert TD.run.last["trainer"].step_count == 1 and TD.run.last["resume_step"] == 1
    TD.main(base + ["--max_train_steps", "1001", "--output_dir", str(c)])
    assert TD.run.last["trainer"].step_count == 1 and TD.run.last["resume_step"] == 1000 and TD.run.last["global_step"] == 1001
    wb, wc = torch.load(b / "2" / "diffusion_pytorch_model.bin"), torch.load(c / "1001" / "diffusion_pytorch_model.bin")
    w1 = torch.load(a / "1" / "diffusion_pytorch_model.bin")
    assert all(torch.equal(wb[k], wc[k]) for k in wb)
    # ... and it is a first Adam step: |update| = lr * |m_hat / (sqrt(v_hat) + eps)| <= lr (+ weight decay, + one bf16 rounding)
    k = max((k for k in wb if wb[k].dim() == 2), key=lambda k: wb[k].numel())
    delta = (wb[k].float() - w1[k].float()).abs()
    ulp = w1[k].float().abs().max().item() * 2.0 ** -7
    assert delta.max().item() <= 1e-3 * 1.05 + 1e-2 * 1e-3 * w1[k].float().abs().max().item() + ulp and delta.max().item() > 0
    # the warm-up restarts too: with constant_with_warmup the first post-resume step runs at factor 0 -> parameters unchanged
    d = tmp_path / "d"
    (d / "7").mkdir(parents=True)
    shutil.copy(a / "1" / "diffusion_pytorch_model.bin", d / "7" / "diffusion_pytorch_model.bin")
    TD.main(base + ["--max_train_steps", "8", "--output_dir", str(d), "--lr_scheduler", "constant_with_warmup", "--lr_warmup_steps", "4"])
    wd = torch.load(d / "8" / "diffusion_pytorch_model.bin")
    assert all(torch.equal(wd[k], w1[k]) for k in wd)


def test_distillation_loss_kernel_vs_reference_golden(ops):
    """x2i_kd_loss_bf16 against the golden produced by EXECUTING the reference's statements (train/train_qwenvl.py normalize and the
    kl_div loops, tests/golden/make_golden.py gen_distill): the summed loss and the gradient with respect to every student tensor."""
    import os
    from safetensors.torch import load_file
    gold = load_file(os.path.join(os.path.dirname(__file__), "golden", "distill_loss.safetensors"))
    total = 0.0
    for k in range(3):
        t, s = bf(gold[f"teacher{k}"]), bf(gold[f"student{k}"])
        assert torch.equal(t.float(), gold[f"teacher{k}"]) and torch.equal(s.float(), gold[f"student{k}"])  # fixture values are bf16-exact
        B, n, S, D = t.shape
        grad = torch.empty((B, n, S, D), device=DEV, dtype=torch.bfloat16)
        rl = torch.empty((B * n * S,), device=DEV)
        ops.kd_loss_rows(g(t), g(s), grad, rl, rows=B * n * S, D=D, temperature=3.0, loss_scale=1.0 / B)   # 'batchmean' = / B per block
        total += float(rl.sum()) / B
        e = rel_l2(grad, gold[f"grad{k}"])
        print(f"  reference-statement golden, tensor {k}: gradient rel-L2 {e:.3e}")
        assert e < 5e-3
    assert abs(total - float(gold["loss"])) < 1e-4 * abs(float(gold["loss"]))


def test_graphed_step_equals_eager_step():
    """GraphedDistillStep (one hipGraph for everything but the optimizer) produces bit-identical losses, gradients and updated parameters
    to the eager distill_step over several steps with changing batches."""
    from oracle import sampler as OS
    from x2i_amd.proj import Proj7Exp
    from x2i_amd.train import DistillBackward, GraphedDistillStep, ProjectorTrainer, distill_step

    def setup():
        m, _, _ = _tiny()
        pr = Proj7Exp(in_channels=5, input_dim=128, output_dim0=32, output_dim1=64, use_t5=False, use_scale=False, use_cnn=True, device=DEV).init_random_(7)
        return ProjectorTrainer(pr, lr=2e-3), DistillBackward(m)
    B, St, h2, w2 = 2, 24, 6, 8
    ids, tids = OS.prepare_latent_image_ids(h2, w2).to(DEV), torch.zeros(St, 3, device=DEV)

    def batch(i):
        gen = torch.Generator().manual_seed(100 + i)
        rn = lambda *s: torch.randn(s, generator=gen)  # noqa: E731
        return (bf(rn(B, 5, St, 128) * 2).to(DEV), bf(rn(B, h2 * w2, 64)).to(DEV), torch.tensor([0.5, 0.25]).to(DEV),
                [bf(rn(B, 2, h2 * w2, 256)).to(DEV), bf(rn(B, 2, St, 256)).to(DEV), bf(rn(B, 2, St + h2 * w2, 256)).to(DEV)])
    tr_e, ch_e = setup()
    tr_g, ch_g = setup()
    gs = GraphedDistillStep(tr_g, ch_g, tids, ids)
    for i in range(5):
        x, lat, ts, teacher = batch(i)
        le = distill_step(tr_e, ch_e, x, lat, ts, teacher, tids, ids, optimizer_step=False)
        lg = gs(x, lat, ts, teacher, optimizer_step=False)
        assert torch.equal(le, lg), i
        assert torch.equal(tr_e.grad, tr_g.grad), i
        tr_e.step()
        tr_g.step()
        for pe, pg in zip(tr_e.params, tr_g.params):
            assert torch.equal(pe, pg), i
    assert gs.graph is not None


def test_graphed_step_gradient_accumulation_and_shape_change():
    """(1) the first calls run with optimizer_step=False and NO trainer.step() in between (gradient accumulation 2), so the
    conv-table cache filled by the eager warm-up is still there when the graph is captured -- the capture must record the pack
    anyway, or replays after the first optimizer step read a table packed from the old conv weights; (2) a batch-shape change re-warms
    eagerly before the next capture.  Losses and parameters must follow the eager step bit for bit throughout."""
    from oracle import sampler as OS
    from x2i_amd.proj import Proj7Exp
    from x2i_amd.train import DistillBackward, GraphedDistillStep, ProjectorTrainer, distill_step

    def setup():
        m, _, _ = _tiny()
        pr = Proj7Exp(in_channels=5, input_dim=128, output_dim0=32, output_dim1=64, use_t5=False, use_scale=False, use_cnn=True, device=DEV).init_random_(7)
        return ProjectorTrainer(pr, lr=5e-3), DistillBackward(m)
    St, h2, w2 = 24, 6, 8
    ids, tids = OS.prepare_latent_image_ids(h2, w2).to(DEV), torch.zeros(St, 3, device=DEV)

    def batch(i, B):
        gen = torch.Generator().manual_seed(300 + i)
        rn = lambda *s: torch.randn(s, generator=gen)  # noqa: E731
        return (bf(rn(B, 5, St, 128) * 2).to(DEV), bf(rn(B, h2 * w2, 64)).to(DEV), torch.linspace(0.2, 0.8, B).to(DEV),
                [bf(rn(B, 2, h2 * w2, 256)).to(DEV), bf(rn(B, 2, St, 256)).to(DEV), bf(rn(B, 2, St + h2 * w2, 256)).to(DEV)])
    tr_e, ch_e = setup()
    tr_g, ch_g = setup()
    gs = GraphedDistillStep(tr_g, ch_g, tids, ids)
    for i, B in enumerate([2, 2, 2, 2, 2, 2, 3, 3, 3, 3]):
        x, lat, ts, teacher = batch(i, B)
        le = distill_step(tr_e, ch_e, x, lat, ts, teacher, tids, ids, optimizer_step=False)
        lg = gs(x, lat, ts, teacher, optimizer_step=False)
        assert torch.equal(le, lg), i
        if i % 2 == 1:  # an optimizer step every second micro-step
            tr_e.step()
            tr_g.step()
            for pe, pg in zip(tr_e.params, tr_g.params):
                assert torch.equal(pe, pg), i
    assert gs.graph is not None and gs.static["shapes"][0][0] == 3


@pytest.mark.parametrize("B,H,S", [(2, 24, 1536), (1, 24, 2800)])
def test_ping_pong_attention_kernel_lse_output_vs_logsumexp(B, H, S):
    """x2i_attention_lse_bf16 at a size that takes the 8-wave ping-pong kernel (B * H * ceil(S / 256) >= 256 workgroups):
    its log2-sum-exp output against torch.logsumexp directly (S = 2800: a ragged last key tile, padded query rows get the +inf marker),
    and its attention output bit for bit against x2i_attention_bf16 and against the 4-wave kernel's statistics."""
    from x2i_amd import _lib, ops
    assert B * H * ((S + 255) // 256) >= 256
    Spad = ops.pad128(S)
    scale = 1.0 / 128 ** 0.5
    gq = torch.Generator(device=DEV).manual_seed(11)
    Q = torch.zeros((B, H, Spad, 128), device=DEV, dtype=torch.bfloat16)
    K = torch.zeros_like(Q)
    VT = torch.zeros((B, H, 128, Spad), device=DEV, dtype=torch.bfloat16)
    Q[:, :, :S] = (torch.randn((B, H, S, 128), device=DEV, generator=gq) * 1.5).bfloat16()
    K[:, :, :S] = (torch.randn((B, H, S, 128), device=DEV, generator=gq) * 1.5).bfloat16()
    VT[:, :, :, :S] = torch.randn((B, H, 128, S), device=DEV, generator=gq).bfloat16()
    o_plain = torch.empty((B, S, H * 128), device=DEV, dtype=torch.bfloat16)
    o_lse = torch.empty_like(o_plain)
    lse = torch.zeros((B, H, Spad), device=DEV)
    old = _lib.set_option("attn_variant", 0)
    try:
        ops.attention(Q, K, VT, o_plain, B, H, S, Spad, H * 128, S * H * 128, scale)
        ops.attention_lse(Q, K, VT, o_lse, lse, B, H, S, Spad, H * 128, S * H * 128, scale)
        _lib.set_option("attn_variant", 4)  # the 4-wave kernel on the same problem
        o4 = torch.empty_like(o_plain)
        lse4 = torch.zeros_like(lse)
        ops.attention_lse(Q, K, VT, o4, lse4, B, H, S, Spad, H * 128, S * H * 128, scale)
    finally:
        _lib.set_option("attn_variant", old)
    assert torch.equal(o_plain, o_lse)
    ref = torch.logsumexp(scale * Q[:, :, :S].float() @ K[:, :, :S].float().transpose(-1, -2), dim=-1) * 1.4426950408889634
    assert rel_l2(lse[:, :, :S], ref) < 1e-4
    assert float((lse[:, :, :S] - ref).abs().max()) < 2e-2
    assert bool((lse[:, :, S:] > 1e29).all())
    assert torch.equal(o4, o_lse) and float((lse4[:, :, :S] - lse[:, :, :S]).abs().max()) < 1e-4
