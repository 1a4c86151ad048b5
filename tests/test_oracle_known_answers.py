"""Known-answer checks for the restated diffusers primitives (parity unpinned:
diffusers==0.31.0 is not available offline).  Each check is a property the
published algorithm must satisfy."""
import math

import torch

from oracle import flux as OF
from oracle import primitives as P
from oracle import sampler as OS


def test_schnell_four_step_schedule():
    ts, sig = OS.flow_match_sigmas(4, OS.SCHEDULER_SCHNELL, 4096)
    assert torch.allclose(sig, torch.tensor([1.0, 0.75, 0.5, 0.25, 0.0]))
    assert torch.allclose(ts, torch.tensor([1000.0, 750.0, 500.0, 250.0]))


def test_dev_dynamic_shift_schedule():
    mu = OS.calculate_shift(4096, 256, 4096, 0.5, 1.15)
    assert abs(mu - 1.15) < 1e-12 and abs(OS.calculate_shift(256) - 0.5) < 1e-12
    assert abs(OS.calculate_shift(4096) - 1.16) < 1e-12
    ts, sig = OS.flow_match_sigmas(20, OS.SCHEDULER_DEV, 4096)
    raw = torch.linspace(1.0, 1 / 20, 20, dtype=torch.float64)
    want = math.exp(mu) / (math.exp(mu) + (1 / raw - 1))
    assert torch.allclose(sig[:-1].double(), want, atol=1e-6) and sig[-1] == 0 and sig[0] == 1.0
    assert torch.all(sig[1:] < sig[:-1])


def test_rope_identity_at_position_zero_and_norm_preserving():
    ids = torch.zeros(5, 3)
    cos, sin = P.flux_pos_embed(ids)
    assert cos.shape == (5, 128) and torch.all(cos == 1) and torch.all(sin == 0)
    x = torch.randn(1, 2, 5, 128)
    assert torch.equal(P.apply_rotary_emb(x, (cos, sin)), x)
    ids = torch.tensor([[0.0, 3.0, 7.0], [0.0, 1.0, 2.0]])
    cos, sin = P.flux_pos_embed(ids)
    x = torch.randn(1, 1, 2, 128)
    y = P.apply_rotary_emb(x, (cos, sin))
    # adjacent pairs rotate rigidly
    assert torch.allclose(y.view(1, 1, 2, 64, 2).norm(dim=-1), x.view(1, 1, 2, 64, 2).norm(dim=-1), atol=1e-5)
    # axis 0 (first 16 dims) always has id 0 -> untouched
    assert torch.allclose(y[..., :16], x[..., :16])
    # pair (16,17) rotates by angle id_1 * theta^0 = 3 rad for token 0
    a = 3.0
    want0 = x[0, 0, 0, 16] * math.cos(a) - x[0, 0, 0, 17] * math.sin(a)
    assert abs(float(y[0, 0, 0, 16] - want0)) < 1e-5


def test_timestep_embedding_is_cos_then_sin():
    e = P.timesteps_proj(torch.tensor([0.0, 1000.0]), 256)
    assert e.shape == (2, 256)
    assert torch.all(e[0, :128] == 1) and torch.all(e[0, 128:] == 0)  # cos(0)=1 first, sin(0)=0 second
    assert abs(float(e[1, 0]) - math.cos(1000.0)) < 1e-4 and abs(float(e[1, 128]) - math.sin(1000.0)) < 1e-4


def test_zero_gates_make_blocks_identity():
    cfg = dict(OF.DEFAULT_CFG)
    cfg.update(num_layers=1, num_single_layers=1, num_attention_heads=2, joint_attention_dim=32, pooled_projection_dim=16)
    sd = OF.random_flux_state_dict(cfg, seed=1, std=0.05)
    for k in ("transformer_blocks.0.norm1.linear", "transformer_blocks.0.norm1_context.linear",
              "single_transformer_blocks.0.norm.linear"):
        sd[k + ".weight"].zero_()
        sd[k + ".bias"].zero_()
    D = 256
    hid, enc, temb = torch.randn(1, 6, D), torch.randn(1, 4, D), torch.randn(1, D)
    rot = P.flux_pos_embed(torch.zeros(10, 3))
    e, h = OF.double_block(sd, "transformer_blocks.0", hid, enc, temb, rot, 2)
    assert torch.equal(e, enc) and torch.equal(h, hid)
    j = torch.cat([enc, hid], 1)
    assert torch.equal(OF.single_block(sd, "single_transformer_blocks.0", j, temb, rot, 2), j)


def test_adaln_continuous_scale_first():
    D = 8
    sd = {"n.linear.weight": torch.zeros(2 * D, D), "n.linear.bias": torch.cat([torch.full((D,), 1.0), torch.full((D,), 5.0)])}
    x = torch.randn(1, 3, D)
    y = P.ada_layer_norm_continuous(sd, "n", x, torch.zeros(1, D))
    assert torch.allclose(y, P.layer_norm_plain(x) * 2.0 + 5.0, atol=1e-6)


def test_param_count_matches_published_flux():
    n = sum(math.prod(s) for s in OF.flux_param_shapes(OF.DEFAULT_CFG).values())
    assert abs(n / 1e9 - 11.891) < 0.002, n
    dev = dict(OF.DEFAULT_CFG, guidance_embeds=True)
    n2 = sum(math.prod(s) for s in OF.flux_param_shapes(dev).values())
    assert abs(n2 / 1e9 - 11.901) < 0.002
    nc = sum(math.prod(s) for s in OF.controlnext_param_shapes().values())
    assert nc == 6592960, nc  # strict-loaded into the reference ControlNeXtModel by make_golden.py


def test_euler_step_fp32_add_model_dtype_store():
    x = torch.randn(2, 4, 8).bfloat16()
    eps = torch.randn(2, 4, 8).bfloat16()
    y = OS.euler_step(x, eps, torch.tensor(0.75), torch.tensor(0.5))
    assert y.dtype == torch.bfloat16
    assert torch.equal(y, (x.float() - 0.25 * eps.float()).bfloat16())


def test_sampler_runs_cpu_plumbing_config():
    """BASELINE config 1 (CPU eager plumbing) at reduced width: 4 steps, batch 1, latents= given."""
    cfg = dict(OF.DEFAULT_CFG)
    cfg.update(num_layers=1, num_single_layers=1, num_attention_heads=2, joint_attention_dim=32, pooled_projection_dim=16)
    sd = OF.random_flux_state_dict(cfg, seed=2, std=0.05)
    pe, pooled = torch.randn(1, 8, 32), torch.randn(1, 16)
    lat = OS.sample_latents(sd, cfg, pe, pooled, 64, 64, 4, generator=torch.Generator().manual_seed(0))
    assert lat.shape == (1, 16, 64) and torch.isfinite(lat).all()
    lat2 = OS.sample_latents(sd, cfg, pe, pooled, 64, 64, 4, generator=torch.Generator().manual_seed(0))
    assert torch.equal(lat, lat2)
