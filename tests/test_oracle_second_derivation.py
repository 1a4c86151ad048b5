"""Second, independent derivations of every diffusers primitive the oracle restates (rows A7-A12).

diffusers==0.31.0 is absent from the image, so oracle/primitives.py cannot be replayed against the library itself.  Each test
below recomputes the operator a DIFFERENT way -- complex multiplication for RoPE, an explicit float64 softmax(QK^T/sqrt d)V,
scalar Python loops for the sinusoid / schedule formulas, hand-chosen constants that make every AdaLN chunk distinguishable --
and never calls oracle/primitives.py to produce the expected value.  A primitive mistake (chunk order, pair layout, promotion
rule, text/image order) therefore has to be made twice, in two unrelated formulations, to go unnoticed.
"""
import math

import numpy as np
import torch

from oracle import primitives as P
from oracle import sampler as OS


def g(seed):
    return torch.Generator().manual_seed(seed)


# ------------------------------------------------------------------------------------------------- A9 RoPE
def _rope_angles_scalar(ids, axes_dim, theta=10000.0):
    """angle[s][pair] by scalar Python arithmetic (float64): axis a, pair k -> ids[s][a] * theta^(-2k/d_a)."""
    out = []
    for s in range(ids.shape[0]):
        row = []
        for a, d in enumerate(axes_dim):
            for k in range(d // 2):
                row.append(float(ids[s, a]) * (1.0 / math.pow(theta, (2 * k) / d)))
        out.append(row)
    return torch.tensor(out, dtype=torch.float64)


def test_rope_tables_and_rotation_via_complex_multiplication():
    """FluxPosEmbed + apply_rotary_emb == multiplying each ADJACENT pair, read as a complex number, by exp(i * angle)."""
    ids = torch.zeros(40, 3)
    ids[8:, 1] = torch.arange(32) // 8
    ids[8:, 2] = torch.arange(32) % 8
    ang = _rope_angles_scalar(ids, (16, 56, 56))  # [S, 64]
    cos, sin = P.flux_pos_embed(ids, (16, 56, 56))
    assert cos.shape == sin.shape == (40, 128) and cos.dtype == torch.float32
    assert torch.allclose(cos[:, 0::2].double(), ang.cos(), atol=1e-7) and torch.equal(cos[:, 0::2], cos[:, 1::2])
    assert torch.allclose(sin[:, 0::2].double(), ang.sin(), atol=1e-7) and torch.equal(sin[:, 0::2], sin[:, 1::2])
    x = torch.randn(2, 3, 40, 128, generator=g(1))
    xc = torch.view_as_complex(x.double().reshape(2, 3, 40, 64, 2).contiguous())
    want = torch.view_as_real(xc * torch.polar(torch.ones_like(ang), ang)[None, None]).reshape(2, 3, 40, 128)
    got = P.apply_rotary_emb(x, (cos, sin))
    assert got.dtype == x.dtype and torch.allclose(got.double(), want, atol=2e-6)
    # NOT the half-split (GPT-NeoX) pairing: that formulation must disagree
    half = torch.cat([x[..., :64] * cos[:, 0::2] - x[..., 64:] * sin[:, 0::2], x[..., 64:] * cos[:, 0::2] + x[..., :64] * sin[:, 0::2]], -1)
    assert not torch.allclose(half[:, :, 8:].double(), want[:, :, 8:], atol=1e-2)


# ------------------------------------------------------------------------------------------------- A7 attention processor
def _rms64(x, w, eps=1e-6):
    x = x.double()
    return x / torch.sqrt((x * x).mean(-1, keepdim=True) + eps) * w.double()


def test_joint_attention_explicit_float64_softmax_text_first():
    """Attention + FluxAttnProcessor2_0 (double-stream form) recomputed with per-head matmuls in float64: q/k RMSNorm per head,
    joint sequence = [text, image] (text FIRST), RoPE on the joint positions by complex multiplication, softmax(QK^T/sqrt(d))V,
    split back, to_out / to_add_out; returns (image, text) in that order."""
    torch.manual_seed(0)
    B, Si, St, H, hd = 2, 10, 6, 2, 128
    D = H * hd
    names = ["to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"]
    sd = {}
    for i, n in enumerate(names):
        sd[f"a.{n}.weight"] = torch.randn(D, D, generator=g(10 + i)) / D ** 0.5
        sd[f"a.{n}.bias"] = 0.1 * torch.randn(D, generator=g(30 + i))
    for i, n in enumerate(["norm_q", "norm_k", "norm_added_q", "norm_added_k"]):
        sd[f"a.{n}.weight"] = 1 + 0.2 * torch.randn(hd, generator=g(50 + i))
    hid, enc = torch.randn(B, Si, D, generator=g(2)), torch.randn(B, St, D, generator=g(3))
    ids = torch.zeros(St + Si, 3)
    ids[St:, 1] = torch.arange(Si) // 5
    ids[St:, 2] = torch.arange(Si) % 5
    rot = P.flux_pos_embed(ids)
    got_img, got_txt = P.flux_attention(sd, "a", hid, H, rot, encoder_hidden=enc)

    ang = _rope_angles_scalar(ids, (16, 56, 56))
    phase = torch.polar(torch.ones_like(ang), ang)

    def lin(n, x):
        return x.double() @ sd[f"a.{n}.weight"].double().T + sd[f"a.{n}.bias"].double()

    def rope(x):  # [S, hd] float64
        xc = torch.view_as_complex(x.reshape(-1, 64, 2).contiguous())
        return torch.view_as_real(xc * phase).reshape(-1, hd)

    want_img, want_txt = torch.zeros(B, Si, D, dtype=torch.float64), torch.zeros(B, St, D, dtype=torch.float64)
    for b in range(B):
        o = torch.zeros(St + Si, D, dtype=torch.float64)
        for h in range(H):
            sl = slice(h * hd, (h + 1) * hd)
            q = torch.cat([_rms64(lin("add_q_proj", enc[b])[:, sl], sd["a.norm_added_q.weight"]),
                           _rms64(lin("to_q", hid[b])[:, sl], sd["a.norm_q.weight"])])
            k = torch.cat([_rms64(lin("add_k_proj", enc[b])[:, sl], sd["a.norm_added_k.weight"]),
                           _rms64(lin("to_k", hid[b])[:, sl], sd["a.norm_k.weight"])])
            v = torch.cat([lin("add_v_proj", enc[b])[:, sl], lin("to_v", hid[b])[:, sl]])
            s = rope(q) @ rope(k).T / math.sqrt(hd)
            p = torch.exp(s - s.max(-1, keepdim=True).values)
            o[:, sl] = (p / p.sum(-1, keepdim=True)) @ v
        want_txt[b] = o[:St] @ sd["a.to_add_out.weight"].double().T + sd["a.to_add_out.bias"].double()
        want_img[b] = o[St:] @ sd["a.to_out.0.weight"].double().T + sd["a.to_out.0.bias"].double()
    assert got_img.shape == (B, Si, D) and got_txt.shape == (B, St, D)
    assert torch.allclose(got_img.double(), want_img, atol=2e-4) and torch.allclose(got_txt.double(), want_txt, atol=2e-4)
    # single-stream form: no added projections, no output projection, joint sequence returned as is
    joint = torch.cat([enc, hid], 1)
    got = P.flux_attention(sd, "a", joint, H, rot)
    want = torch.zeros(B, St + Si, D, dtype=torch.float64)
    for b in range(B):
        for h in range(H):
            sl = slice(h * hd, (h + 1) * hd)
            q, k = (_rms64(lin(n, joint[b])[:, sl], sd[f"a.norm_{n[-1]}.weight"]) for n in ("to_q", "to_k"))
            s = rope(q) @ rope(k).T / math.sqrt(hd)
            want[b][:, sl] = torch.softmax(s, -1) @ lin("to_v", joint[b])[:, sl]
    assert torch.allclose(got.double(), want, atol=2e-4)


def test_rmsnorm_float64_and_the_bf16_weight_cast_rule():
    """diffusers RMSNorm: variance in fp32, x * rsqrt(var + eps) in fp32, THEN (bf16 weight) cast to bf16 BEFORE the multiply."""
    x = torch.randn(3, 5, 128, generator=g(4))
    w = 1 + 0.3 * torch.randn(128, generator=g(5))
    want = x.double() / torch.sqrt((x.double() ** 2).mean(-1, keepdim=True) + 1e-6) * w.double()
    assert torch.allclose(P.rms_norm(x, w).double(), want, atol=1e-5)
    xb, wb = x.bfloat16(), w.bfloat16()
    got = P.rms_norm(xb, wb)
    assert got.dtype == torch.bfloat16
    n64 = xb.double() / torch.sqrt((xb.double() ** 2).mean(-1, keepdim=True) + 1e-6)
    want_b = (n64.float().bfloat16().double() * wb.double()).float().bfloat16()  # round, multiply, round
    assert (got.float() - want_b.float()).abs().max() <= 2 ** -7 * want_b.float().abs().max()  # <= 1 bf16 ulp (fp32 vs fp64 rsqrt)
    one_round = (n64 * wb.double()).float().bfloat16()  # the single-rounding formulation is a DIFFERENT function
    assert (want_b != one_round).any()
    frac_equal = (got == want_b).float().mean()
    assert frac_equal > 0.99


# ------------------------------------------------------------------------------------------------- A8 AdaLN family
def test_adaln_chunk_orders_with_distinguishable_constants():
    D = 16
    x = torch.randn(2, 7, D, generator=g(6))
    ln = (x.double() - x.double().mean(-1, keepdim=True)) / torch.sqrt(x.double().var(-1, unbiased=False, keepdim=True) + 1e-6)
    emb = torch.randn(2, D, generator=g(7))

    def table(n):  # zero weight, bias chunk c == constant (c + 1): the modulation is independent of emb
        return {"n.linear.weight": torch.zeros(n * D, D), "n.linear.bias": torch.arange(1, n + 1).float().repeat_interleave(D)}

    # AdaLayerNormZero: shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = chunk(6)
    y, gate_msa, shift_mlp, scale_mlp, gate_mlp = P.ada_layer_norm_zero(table(6), "n", x, emb)
    assert torch.allclose(y.double(), ln * (1 + 2.0) + 1.0, atol=1e-5)
    assert torch.all(gate_msa == 3) and torch.all(shift_mlp == 4) and torch.all(scale_mlp == 5) and torch.all(gate_mlp == 6)
    # AdaLayerNormZeroSingle: shift, scale, gate = chunk(3)
    y, gate = P.ada_layer_norm_zero_single(table(3), "n", x, emb)
    assert torch.allclose(y.double(), ln * (1 + 2.0) + 1.0, atol=1e-5) and torch.all(gate == 3)
    # AdaLayerNormContinuous: SCALE first, then shift
    y = P.ada_layer_norm_continuous(table(2), "n", x, emb)
    assert torch.allclose(y.double(), ln * (1 + 1.0) + 2.0, atol=1e-5)
    # the linear really sees SiLU(emb): identity weight on one chunk
    sd = table(3)
    sd["n.linear.weight"][:D] = torch.eye(D)
    sd["n.linear.bias"][:D] = 0
    y, _ = P.ada_layer_norm_zero_single(sd, "n", x, emb)
    silu = emb.double() / (1 + torch.exp(-emb.double()))
    assert torch.allclose(y.double(), ln * 3.0 + silu[:, None, :], atol=1e-5)


# ------------------------------------------------------------------------------------------------- A10 embeddings
def test_timesteps_sinusoid_scalar_formula_and_embedder_composition():
    for dim in (256, 128):
        t = [0.0, 1.0, 250.0, 752.0, 1000.0, 3500.0]
        got = P.timesteps_proj(torch.tensor(t), dim)
        half = dim // 2
        for i, tv in enumerate(t):
            for k in (0, 1, 17, half - 1):
                f = math.exp(-math.log(10000.0) * k / half)
                assert abs(float(got[i, k]) - math.cos(tv * f)) < 2e-4 * max(1.0, tv * f), (dim, tv, k)  # cos half FIRST
                assert abs(float(got[i, half + k]) - math.sin(tv * f)) < 2e-4 * max(1.0, tv * f)
    # CombinedTimestepGuidanceTextProjEmbeddings = MLP(Timesteps(t)) + MLP(Timesteps(g)) + MLP(pooled), SiLU between linears
    D, Pd = 32, 8
    sd = {}
    for i, (n, k) in enumerate((("timestep_embedder", 256), ("guidance_embedder", 256), ("text_embedder", Pd))):
        sd[f"e.{n}.linear_1.weight"] = torch.randn(D, k, generator=g(60 + i)) / k ** 0.5
        sd[f"e.{n}.linear_1.bias"] = 0.1 * torch.randn(D, generator=g(63 + i))
        sd[f"e.{n}.linear_2.weight"] = torch.randn(D, D, generator=g(66 + i)) / D ** 0.5
        sd[f"e.{n}.linear_2.bias"] = 0.1 * torch.randn(D, generator=g(69 + i))
    tt, gd, pooled = torch.tensor([500.0, 752.0]), torch.tensor([3500.0, 3500.0]), torch.randn(2, Pd, generator=g(8))

    def sinus(v):
        return torch.tensor([[math.cos(x * math.exp(-math.log(10000.0) * k / 128)) for k in range(128)] +
                             [math.sin(x * math.exp(-math.log(10000.0) * k / 128)) for k in range(128)] for x in v.tolist()],
                            dtype=torch.float64)

    def mlp(n, x):
        h = x @ sd[f"e.{n}.linear_1.weight"].double().T + sd[f"e.{n}.linear_1.bias"].double()
        h = h / (1 + torch.exp(-h))
        return h @ sd[f"e.{n}.linear_2.weight"].double().T + sd[f"e.{n}.linear_2.bias"].double()

    want = mlp("timestep_embedder", sinus(tt)) + mlp("guidance_embedder", sinus(gd)) + mlp("text_embedder", pooled.double())
    got = P.combined_time_text_embed(sd, "e", tt, pooled, gd)
    assert torch.allclose(got.double(), want, atol=5e-4)
    want_ng = mlp("timestep_embedder", sinus(tt)) + mlp("text_embedder", pooled.double())
    assert torch.allclose(P.combined_time_text_embed(sd, "e", tt, pooled, None).double(), want_ng, atol=5e-4)


def test_feed_forward_gelu_tanh_scalar_formula():
    D = 8
    sd = {"f.net.0.proj.weight": torch.randn(4 * D, D, generator=g(9)), "f.net.0.proj.bias": torch.randn(4 * D, generator=g(10)),
          "f.net.2.weight": torch.randn(D, 4 * D, generator=g(11)), "f.net.2.bias": torch.randn(D, generator=g(12))}
    x = torch.randn(3, D, generator=g(13))
    h = x.double() @ sd["f.net.0.proj.weight"].double().T + sd["f.net.0.proj.bias"].double()
    h = 0.5 * h * (1 + torch.tanh(math.sqrt(2 / math.pi) * (h + 0.044715 * h ** 3)))
    want = h @ sd["f.net.2.weight"].double().T + sd["f.net.2.bias"].double()
    assert torch.allclose(P.feed_forward(sd, "f", x).double(), want, atol=1e-4)


# ------------------------------------------------------------------------------------------------- A12 scheduler, A11 loop
def _schedule_scalar(n, seq_len, dynamic, shift=1.0, base_shift=0.5, max_shift=1.15, base_len=256, max_len=4096):
    """Closed form by scalar Python arithmetic: sigma_i = 1 - i (1 - 1/n)/(n-1), then the static or exponential shift."""
    raw = [1.0 - i * (1.0 - 1.0 / n) / (n - 1) if n > 1 else 1.0 for i in range(n)]
    if dynamic:
        mu = base_shift + (max_shift - base_shift) * (seq_len - base_len) / (max_len - base_len)
        return [math.exp(mu) / (math.exp(mu) + (1.0 / s - 1.0)) for s in raw]
    return [shift * s / (1 + (shift - 1) * s) for s in raw]


def test_schedule_closed_form_for_oracle_and_product_scheduler():
    from x2i_amd.pipeline import FlowMatchEulerDiscreteScheduler, calculate_shift
    for (n, seq, cfg) in ((4, 4096, OS.SCHEDULER_SCHNELL), (20, 4096, OS.SCHEDULER_DEV), (20, 1024, OS.SCHEDULER_DEV),
                          (8, 2304, dict(OS.SCHEDULER_SCHNELL, shift=3.0)), (1, 4096, OS.SCHEDULER_DEV)):
        want = _schedule_scalar(n, seq, cfg["use_dynamic_shifting"], cfg["shift"], cfg["base_shift"], cfg["max_shift"])
        ts, sig = OS.flow_match_sigmas(n, cfg, seq)
        assert np.allclose(sig[:-1].double().numpy(), want, atol=1e-6) and float(sig[-1]) == 0.0
        assert np.allclose(ts.double().numpy(), [1000 * s for s in want], atol=1e-3)
        sch = FlowMatchEulerDiscreteScheduler.from_config(cfg)
        mu = calculate_shift(seq, cfg["base_image_seq_len"], cfg["max_image_seq_len"], cfg["base_shift"], cfg["max_shift"])
        sch.set_timesteps(sigmas=np.linspace(1.0, 1 / n, n), mu=mu)
        assert torch.equal(sch.timesteps, ts) and torch.equal(sch.sigmas, sig)


def test_euler_loop_is_the_flow_matching_ode_integrator():
    """With a model that returns the constant velocity v, N Euler steps from sigma=1 to 0 give x - v exactly (sum of dt = -1)."""
    x = torch.randn(2, 6, 8, generator=g(14))
    v = torch.randn(2, 6, 8, generator=g(15))
    for cfg, n in ((OS.SCHEDULER_SCHNELL, 4), (OS.SCHEDULER_DEV, 20)):
        _, sig = OS.flow_match_sigmas(n, cfg, 4096)
        y = x.clone()
        for i in range(n):
            y = OS.euler_step(y, v, sig[i], sig[i + 1])
        assert torch.allclose(y, x - v, atol=1e-5)


def test_scheduler_protocol_fixture_from_the_reference_statements():
    """tests/golden/scheduler_protocol.safetensors: the reference's OWN retrieve_timesteps (train/train_qwenvl.py:248-281) and its
    sigma / mu / timestep construction statements (:754-764), plus get_sigmas and the noising statement of
    lightcontrol/train_lightcontrol.py (:412-421, :706), executed (make_golden.py, by `ast`) against this build's scheduler class.
    Replayed here three ways: the product scheduler driven by the product pipeline's own call sequence, the oracle's
    flow_match_sigmas, and the scalar closed form above."""
    from tests.util import golden
    from x2i_amd.pipeline import FlowMatchEulerDiscreteScheduler, calculate_shift
    t, meta = golden("scheduler_protocol")
    assert "retrieve_timesteps" in meta["ref"] and len(meta["cases"]) == 5
    for case in meta["cases"]:
        tag, cfgk, n, seq = case["tag"], case["config"], case["num_inference_steps"], case["image_seq_len"]
        cfg = dict(OS.SCHEDULER_SCHNELL, **cfgk)
        want = _schedule_scalar(n, seq, cfg["use_dynamic_shifting"], cfg["shift"], cfg["base_shift"], cfg["max_shift"])
        assert np.allclose(t[tag + ".sigmas"][:-1].double().numpy(), want, atol=1e-6) and float(t[tag + ".sigmas"][-1]) == 0
        assert np.allclose(t[tag + ".timesteps"].double().numpy(), [1000 * s for s in want], atol=1e-3)
        ts, sig = OS.flow_match_sigmas(n, cfg, seq)
        assert torch.equal(ts, t[tag + ".timesteps"]) and torch.equal(sig, t[tag + ".sigmas"])
        sch = FlowMatchEulerDiscreteScheduler.from_config(cfg)
        mu = calculate_shift(seq, sch.config.base_image_seq_len, sch.config.max_image_seq_len, sch.config.base_shift, sch.config.max_shift)
        assert abs(mu - float(t[tag + ".mu"])) < 1e-12
        sch.set_timesteps(sigmas=np.linspace(1.0, 1 / n, n), device="cpu", mu=mu)
        assert torch.equal(sch.timesteps, t[tag + ".timesteps"]) and torch.equal(sch.sigmas, t[tag + ".sigmas"])
    # training-side attributes the reference reads straight off the object: .timesteps / .sigmas of a fresh scheduler (1000 train
    # steps, no shift applied under dynamic shifting), and z_t = (1 - sigma) x + sigma z1
    sch = FlowMatchEulerDiscreteScheduler(shift=3.0, use_dynamic_shifting=True)
    idx = t["train.indices"]
    assert torch.equal(sch.timesteps[idx], t["train.timesteps"])
    want_sig = torch.tensor([(1000 - int(i)) / 1000 for i in idx], dtype=torch.float32)
    assert torch.allclose(t["train.sigmas"].flatten(), want_sig, atol=1e-7)
    s4 = want_sig.view(4, 1, 1, 1)
    assert torch.allclose(t["train.noisy"], (1 - s4) * t["train.model_input"] + s4 * t["train.noise"], atol=1e-6)
