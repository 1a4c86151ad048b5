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
    part = torch.empty((B, nw, 2, D), device=DEV, dtype=torch.float32)
    dx = g(dres).clone()
    ops.ln_mod_bwd(g(x), g(dy), g(sc), dx, dx, part, B=B, S=S, D=D, R=R, mult_bs=D)
    assert rel_l2(dx, dres.float() + xr.grad) < 1e-2
    out = torch.zeros((B, 2 * D), device=DEV)
    ops.reduce_rows(part, out, np_=nw, len_=D, nz=B, in_zs=nw * 2 * D, in_ps=2 * D, out_zs=2 * D, accumulate=True)
    ops.reduce_rows(part, out, np_=nw, len_=D, nz=B, in_zs=nw * 2 * D, in_ps=2 * D, out_zs=2 * D, accumulate=True, in_offset=D, out_offset=D)
    assert rel_l2(out[:, :D], scr.grad) < 5e-3 and rel_l2(out[:, D:], sh.grad) < 5e-3


def test_gate_backward_and_act_backward(ops):
    B, S, D, R = 2, 32, 256, 8
    dx, t, G = bf(seeded((B, S, D), 9)), bf(seeded((B, S, D), 10)), bf(seeded((B, S, D), 11))
    gate = seeded((B, D), 12)
    nw = S // R
    part = torch.empty((B, nw, D), device=DEV, dtype=torch.float32)
    dT = torch.empty((B, S, D), device=DEV, dtype=torch.bfloat16)
    ops.gate_bwd(g(dx), g(t), g(gate), g(G), dT, part, B=B, S=S, D=D, R=R, gate_bs=D)
    assert rel_l2(dT, gate[:, None] * dx.float() + G.float()) < 5e-3
    dg = torch.zeros((B, D), device=DEV)
    ops.reduce_rows(part, dg, np_=nw, len_=D, nz=B, in_zs=nw * D, in_ps=D, out_zs=D)
    assert rel_l2(dg, (dx.float() * t.float()).sum(1)) < 5e-3
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


def test_activation_gradient_chain_vs_oracle_autograd_with_explicit_tap_gradients():
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
    bw = DistillBackward(m)
    st = bw.prepare_conditioning(enc.to(DEV), pooled.to(DEV), tids.to(DEV), ids.to(DEV))
    out, _ = bw.forward_train(st, hid.to(DEV), ts.to(DEV), tap_grads=[[x.to(DEV) for x in k] for k in G])
    assert rel_l2(out, ref_out.detach()) < 2e-2
    # the saving forward is the sampling forward
    assert rel_l2(out, m.denoise(st, hid.to(DEV), ts.to(DEV))) < 1e-2
    d_enc, d_pooled = bw.backward()
    e1, e2 = rel_l2(d_enc, er.grad), rel_l2(d_pooled, pr.grad)
    print(f"activation-gradient chain (2+2 blocks): d_enc rel-L2 {e1:.3e}, d_pooled rel-L2 {e2:.3e}")
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
    print(f"distillation loss: HIP {float(loss):.5f}  reference formula on the oracle {float(ref_loss):.5f}")
    assert abs(float(loss) - float(ref_loss)) < 2e-3 * abs(float(ref_loss))  # measured 2.5e-5
    e1, e2 = rel_l2(d_enc, er.grad), rel_l2(d_pooled, pr.grad)
    print(f"distillation gradient: d_enc rel-L2 {e1:.3e}, d_pooled rel-L2 {e2:.3e}")
    assert e1 < 2.5e-2 and e2 < 2.5e-2  # measured 5.1e-3 / 2.0e-3
