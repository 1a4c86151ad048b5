"""CPU-side checks (no GPU): the C-ABI library loads and exports every declared symbol, the host modules mirror the
reference's parameter tree, the scheduler/pipeline helpers match the oracle and the reference goldens, and the product
path refuses to run without the HIP device (no silent fallback)."""
import math
import os
import re

import pytest
import torch

from oracle import flux as OF
from oracle import projector as OP
from oracle import sampler as OS
from tests.util import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    from x2i_amd import _lib
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "x2i.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(x2i_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/x2i.h but not exported by libx2i_hip.so"
    bound = set(_lib.SIGNATURES) | {"x2i_abi_version", "x2i_last_error", "x2i_is_ablation_build", "x2i_groupnorm_scratch_floats",
                                    "x2i_streamk_workspace_bytes", "x2i_groupnorm_moments_scratch_floats", "x2i_conv_moments_scratch_floats"}
    assert declared == bound, (declared ^ bound)
    ver = int(re.search(r"#define X2I_ABI_VERSION (\d+)", hdr).group(1))
    assert lib.x2i_abi_version() == ver == _lib.ABI_VERSION == 5   # header, library and binding move together
    assert lib.x2i_streamk_workspace_bytes() == 4096 + 512 * 256 * 1024   # the caller-owned workspace: flags + 512 slabs of 256 KiB (round 5: the K split double-buffers)


def test_options_are_resolved_once_and_product_library_has_no_ablation_kernels():
    """include/x2i.h: x2i_set_option / x2i_get_option; the "wrong results by design" kernels live only in the measurement build."""
    from x2i_amd import _lib
    from x2i_amd._lib import X2IError
    lib = _lib.load()
    assert os.path.basename(_lib.LIB_PATH) == "libx2i_hip.so" and lib.x2i_is_ablation_build() == 0
    assert _lib.get_option("gemm_tile") == 0 and _lib.get_option("gemm_min256") == 128 and _lib.get_option("conv256") == 1
    os.environ["X2I_GEMM_TILE"] = "128"  # the environment is read once (first use): changing it now has no effect
    try:
        assert _lib.get_option("gemm_tile") == 0
    finally:
        del os.environ["X2I_GEMM_TILE"]
    assert _lib.set_option("gemm_tile", 256) == 0 and _lib.get_option("gemm_tile") == 256
    _lib.set_option("gemm_tile", 0)
    for name in ("gemm_ablate", "attn_ablate", "gemm_lform", "no_such_option"):
        with pytest.raises(X2IError):
            _lib.set_option(name, 1)
    # no getenv left on any launch path: only the one-time option table in c_api.hip reads the environment
    csrc = os.path.join(ROOT, "x2i_amd", "csrc")
    for f in os.listdir(csrc):
        if f != "c_api.hip":
            assert "getenv" not in open(os.path.join(csrc, f)).read().replace("no getenv", ""), f
    # the measurement-only build exists beside it and says what it is
    import ctypes
    abl = os.path.join(ROOT, "x2i_amd", "libx2i_hip_ablate.so")
    if os.path.exists(abl):
        assert ctypes.CDLL(abl).x2i_is_ablation_build() == 1


def test_no_cpu_fallback():
    from x2i_amd import ops
    from x2i_amd._lib import X2IError
    with pytest.raises(X2IError):
        ops.gemm(torch.zeros((64, 64), dtype=torch.bfloat16), torch.zeros((64, 64), dtype=torch.bfloat16))
    with pytest.raises(X2IError):
        ops.ln_affine(torch.zeros((4, 64), dtype=torch.bfloat16), torch.ones(64, dtype=torch.bfloat16),
                      torch.zeros(64, dtype=torch.bfloat16), 1e-6)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "x2i_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"


@pytest.mark.parametrize("guidance", [False, True])
def test_transformer_parameter_tree_matches_reference(guidance):
    from x2i_amd.flux import FluxTransformer2DModel
    cfg = dict(OF.DEFAULT_CFG, guidance_embeds=guidance)
    m = FluxTransformer2DModel(**cfg, device="meta")
    want = OF.flux_param_shapes(cfg)  # strict-loaded into the reference module tree by make_golden.py
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert got == {k: tuple(v) for k, v in want.items()}
    n = sum(math.prod(s) for s in got.values())
    assert abs(n / 1e9 - (11.901 if guidance else 11.891)) < 0.002
    assert m.config.in_channels == 64 and m.config.guidance_embeds == guidance and m.dtype == torch.bfloat16


def test_fused_storage_is_filled_through_reference_names():
    from x2i_amd.flux import FluxTransformer2DModel
    cfg = dict(OF.DEFAULT_CFG, num_layers=1, num_single_layers=1, num_attention_heads=2, joint_attention_dim=64,
               pooled_projection_dim=32)
    sd = {k: v.bfloat16() for k, v in OF.random_flux_state_dict(cfg, seed=5).items()}
    m = FluxTransformer2DModel(**cfg, device="cpu")
    m.load_state_dict(sd, strict=True)
    D = 256
    assert torch.equal(m._fused["s0.in.w"][:D], sd["single_transformer_blocks.0.attn.to_q.weight"])
    assert torch.equal(m._fused["s0.in.w"][3 * D:], sd["single_transformer_blocks.0.proj_mlp.weight"])
    assert torch.equal(m._fused["d0.cqkv.w"][2 * D:], sd["transformer_blocks.0.attn.add_v_proj.weight"])
    assert torch.equal(m._fused["mod.w"][6 * D:12 * D], sd["transformer_blocks.0.norm1_context.linear.weight"])
    assert torch.equal(m._fused["mod.b"][-2 * D:], sd["norm_out.linear.bias"])
    with pytest.raises(ValueError):
        FluxTransformer2DModel(attention_head_dim=64, device="meta")


@pytest.mark.parametrize("kind", list(OP.FACTORIES))
def test_projector_parameter_tree_matches_reference(kind):
    import x2i_amd.proj as XP
    C = OP.FACTORIES[kind]["in_channels"]
    make = dict(qwen3b=XP.create_proj3_qwen3b, qwen7b=XP.create_proj3_qwen7b, internvl1b=XP.create_proj_internvl1b,
                internvl4b=XP.create_proj_internvl4b, minicpm=XP.create_proj_minicpm)[kind]
    use_scale = kind == "internvl1b"
    p = make(in_channels=C, use_t5=False, use_scale=use_scale, use_cnn=True, device="meta")
    want = OP.random_proj_state_dict(kind, seed=0)
    assert {k: tuple(v.shape) for k, v in p.state_dict().items()} == {k: tuple(v.shape) for k, v in want.items()}


def test_scheduler_and_helpers_match_oracle_and_reference():
    from x2i_amd.pipeline import FluxPipeline, FlowMatchEulerDiscreteScheduler, calculate_shift
    for sc, n, seq in ((OS.SCHEDULER_SCHNELL, 4, 4096), (OS.SCHEDULER_DEV, 20, 4096), (OS.SCHEDULER_DEV, 28, 1024)):
        s = FlowMatchEulerDiscreteScheduler(**sc)
        import numpy as np
        mu = calculate_shift(seq, sc["base_image_seq_len"], sc["max_image_seq_len"], sc["base_shift"], sc["max_shift"])
        s.set_timesteps(sigmas=np.linspace(1.0, 1 / n, n), mu=mu)
        ts, sig = OS.flow_match_sigmas(n, sc, seq)
        assert torch.allclose(s.sigmas, sig, atol=1e-7) and torch.allclose(s.timesteps, ts, atol=1e-4)
    t, meta = golden("helpers")
    assert torch.equal(FluxPipeline._pack_latents(t["lat"], 2, 16, 8, 12), t["packed"])
    assert torch.equal(FluxPipeline._unpack_latents(t["packed"], 64, 96, 16), t["unpacked"])
    assert torch.equal(FluxPipeline._prepare_latent_image_ids(2, 4, 6, "cpu", torch.float32), t["ids"])
    assert abs(calculate_shift(4096) - float(t["shifts"][2])) < 1e-12


def test_scheduler_step_fp32_path_on_cpu():
    from x2i_amd.pipeline import FlowMatchEulerDiscreteScheduler
    import numpy as np
    s = FlowMatchEulerDiscreteScheduler()
    s.set_timesteps(sigmas=np.linspace(1.0, 0.25, 4))
    x, e = torch.randn(2, 8, 4), torch.randn(2, 8, 4)
    y = s.step(e, s.timesteps[0], x)[0]
    assert torch.allclose(y, x - 0.25 * e)


def test_vae_decoder_parameter_tree_and_oracle_known_answers():
    """N1 (parity unpinned: diffusers absent).  Key/shape table == oracle table; FLUX decoder has 49.5 M parameters; a zero
    conv_out weight turns the decoder into its bias image; nearest upsampling makes a constant latent decode to a constant."""
    import math
    from oracle import vae as OV
    from x2i_amd.vae import AutoencoderKL
    vae = AutoencoderKL(device="meta")
    want = OV.vae_decoder_param_shapes()
    assert {k: tuple(v.shape) for k, v in vae.state_dict().items()} == {k: tuple(v) for k, v in want.items()}
    assert abs(sum(math.prod(s) for s in want.values()) / 1e6 - 49.545) < 0.01
    assert 2 ** len(vae.config.block_out_channels) == 16  # vae_scale_factor the reference derives (infer/inference_qwenvl.py:209)
    cfg = dict(OV.FLUX_VAE_CFG, block_out_channels=(32, 32, 64, 64), norm_num_groups=8)
    sd = OV.random_vae_decoder_state_dict(cfg, seed=0)
    sd["decoder.conv_out.weight"].zero_()
    y = OV.vae_decode(sd, torch.randn(1, 16, 4, 4), cfg)
    assert y.shape == (1, 3, 32, 32)
    assert torch.allclose(y, sd["decoder.conv_out.bias"].view(1, 3, 1, 1).expand_as(y))


def test_c_abi_argument_validation_returns_codes_without_a_gpu():
    """Every entry point validates its arguments before touching HIP: bad shapes / null pointers come back as negative
    status codes with a message in x2i_last_error() (include/x2i.h error contract) -- never a crash, never a launch."""
    import ctypes as C
    from x2i_amd import _lib
    lib = _lib.load()
    lib.x2i_last_error.restype = C.c_char_p
    fake = C.c_void_p(0x1000)  # never dereferenced: validation fails first

    a = _lib.GemmArgs()
    assert lib.x2i_gemm_bf16(C.byref(a), None) < 0 and b"null" in lib.x2i_last_error()
    a.A, a.W, a.C = 0x1000, 0x1000, 0x1000
    a.M, a.N, a.K, a.batch = 0, 128, 64, 1
    assert lib.x2i_gemm_bf16(C.byref(a), None) < 0 and b"shape" in lib.x2i_last_error()
    a.M = 128
    a.gate = 0x1000  # gate without residual
    assert lib.x2i_gemm_bf16(C.byref(a), None) < 0 and b"gate" in lib.x2i_last_error()

    a = _lib.GemmArgs()
    a.A, a.W, a.M, a.N, a.K, a.batch = 0x1000, 0x1000, 256, 3 * 2 * 128 + 8, 64, 1
    q = _lib.QkvDesc()
    q.norm_q = q.norm_k = q.cos = q.sin = q.Q = q.K = q.VT = 0x1000
    q.H, q.Spad, q.tok_off, q.rows_per_sample, q.eps = 2, 256, 0, 256, 1e-6
    assert lib.x2i_gemm_qkv_bf16(C.byref(a), C.byref(q), None) < 0 and b"3*H*128" in lib.x2i_last_error()
    a.N = 3 * 2 * 128
    q.Spad = 200
    assert lib.x2i_gemm_qkv_bf16(C.byref(a), C.byref(q), None) < 0 and b"geometry" in lib.x2i_last_error()
    q.Spad, a.act = 256, 1
    assert lib.x2i_gemm_qkv_bf16(C.byref(a), C.byref(q), None) < 0 and b"plain bias" in lib.x2i_last_error()
    assert lib.x2i_gemm_qkv_bf16(C.byref(a), None, None) < 0
    # grouped launches: same validation as the two calls they stand for (no GPU needed to be refused)
    g = _lib.GemmArgs()
    assert lib.x2i_gemm_pair_bf16(C.byref(g), None, None) < 0 and lib.x2i_gemm_pair_bf16(None, C.byref(g), None) < 0
    assert lib.x2i_gemm_pair_bf16(C.byref(g), C.byref(g), None) < 0 and b"null" in lib.x2i_last_error()   # falls back to x2i_gemm_bf16's checks
    assert lib.x2i_gemm_qkv_pair_bf16(C.byref(a), None, C.byref(a), C.byref(q), None) < 0
    a.act = 0
    q.Spad = 200
    assert lib.x2i_gemm_qkv_pair_bf16(C.byref(a), C.byref(q), C.byref(a), C.byref(q), None) < 0 and b"geometry" in lib.x2i_last_error()

    cd = _lib.ConvDesc(H=8, W=8, Cin=48, KH=3, KW=3, stride=1, pad=1, up=0)
    a = _lib.GemmArgs()
    a.A, a.W, a.C, a.M, a.N, a.K, a.batch = 0x1000, 0x1000, 0x1000, 64, 64, 9 * 48, 1
    assert lib.x2i_conv2d_nhwc_bf16(C.byref(a), C.byref(cd), None) < 0 and b"multiple of 64" in lib.x2i_last_error()
    # ABI 3: a descriptor whose ninth field was never set (0) pads W like H -- nn.Conv2d(padding=pad); pad_w_p1 = 1 means NO padding along W
    cd = _lib.ConvDesc(H=8, W=8, Cin=64, KH=3, KW=3, stride=1, pad=1, up=0)
    a.M, a.K = 48, 9 * 64
    assert lib.x2i_conv2d_nhwc_bf16(C.byref(a), C.byref(cd), None) < 0 and b"OH*OW=64" in lib.x2i_last_error()
    cd.pad_w_p1 = 1
    a.M = 64
    assert lib.x2i_conv2d_nhwc_bf16(C.byref(a), C.byref(cd), None) < 0 and b"OH*OW=48" in lib.x2i_last_error()

    # the round-5 fields of x2i_conv_desc: up in {0, 1, 2}; out_w / out_h (one-sided padding) must leave the last tap inside the padded input; an output
    # row pitch goes with the plain bf16 epilogue only; moments need their scratch, whole-line outputs and no gate anywhere near a convolution
    cd = _lib.ConvDesc(H=8, W=8, Cin=64, KH=3, KW=3, stride=1, pad=1, up=3)
    a.M, a.N, a.K, a.ldc = 64, 64, 9 * 64, 64
    assert lib.x2i_conv2d_nhwc_bf16(C.byref(a), C.byref(cd), None) < 0 and b"up must be" in lib.x2i_last_error()
    cd = _lib.ConvDesc(H=8, W=8, Cin=64, KH=2, KW=2, stride=1, pad=1, up=0, pad_w_p1=2, out_w=12, out_h=8)
    a.K = 4 * 64
    assert lib.x2i_conv2d_nhwc_bf16(C.byref(a), C.byref(cd), None) < 0 and b"out_w=12 lies outside" in lib.x2i_last_error()
    cd.out_w, cd.out_h = 8, 12
    assert lib.x2i_conv2d_nhwc_bf16(C.byref(a), C.byref(cd), None) < 0 and b"out_h=12 lies outside" in lib.x2i_last_error()
    cd.out_h, cd.out_row_pitch = 8, 100
    assert lib.x2i_conv2d_nhwc_bf16(C.byref(a), C.byref(cd), None) < 0 and b"out_row_pitch" in lib.x2i_last_error()
    cd.out_row_pitch, a.res = 1024, 0x1000
    assert lib.x2i_conv2d_nhwc_bf16(C.byref(a), C.byref(cd), None) < 0 and b"plain bf16 epilogue only" in lib.x2i_last_error()
    a.res, cd.out_row_pitch, cd.moments = None, 0, 0x1000
    assert lib.x2i_conv2d_nhwc_bf16(C.byref(a), C.byref(cd), None) < 0 and b"moments_scratch" in lib.x2i_last_error()
    cd.moments_scratch, a.ldc = 0x1000, 60
    assert lib.x2i_conv2d_nhwc_bf16(C.byref(a), C.byref(cd), None) < 0 and b"whole-line bf16 epilogue" in lib.x2i_last_error()
    a.ldc, a.res, a.gate = 64, 0x1000, 0x1000
    assert lib.x2i_conv2d_nhwc_bf16(C.byref(a), C.byref(cd), None) < 0 and b"gated residual" in lib.x2i_last_error()
    a.res, a.gate = None, None
    assert lib.x2i_conv_moments_scratch_floats(1024 * 1024, 128, 4) == 4 * (16384 + 64) * 64 and lib.x2i_conv_moments_scratch_floats(0, 128, 4) == 0
    # the span-permuted V^T: only the 16 x 16 x 32 kernel reads it, and that one wants 16-byte aligned output rows
    assert lib.x2i_attention_vp_bf16(fake, fake, fake, fake, 1, 1, 100, 100, 128, 12800, 0.1, None) < 0 and b"Spad" in lib.x2i_last_error()
    assert lib.x2i_attention_vp_bf16(fake, fake, None, fake, 1, 1, 128, 128, 128, 16384, 0.1, None) < 0
    # ... and its stream-K form: a workspace that is too small or misaligned is an argument error (never a silent whole-item launch)
    assert lib.x2i_attention_vp_ws_bf16(fake, fake, fake, fake, 1, 1, 128, 128, 128, 16384, 0.1, fake, 4096, None) < 0 and b"workspace" in lib.x2i_last_error()
    assert lib.x2i_attention_vp_ws_bf16(fake, fake, fake, fake, 1, 1, 100, 100, 128, 12800, 0.1, None, 0, None) < 0 and b"Spad" in lib.x2i_last_error()
    _lib.set_option("attn_w16", 0)
    assert lib.x2i_attention_prefers_vt_perm(24, 4608, 0.6931471805599453) == 0
    _lib.set_option("attn_w16", 1)
    assert lib.x2i_attention_prefers_vt_perm(24, 4608, 0.6931471805599453) == 1 and lib.x2i_attention_prefers_vt_perm(24, 4608, 0.0883883) == 0

    assert lib.x2i_attention_bf16(fake, fake, fake, fake, 1, 1, 100, 100, 128, 12800, 0.1, None) < 0  # Spad % 128
    assert lib.x2i_qkv_split_bf16(None, fake, 384, 384, 1, 64, 0, 1, None, None, fake, fake, fake, fake, fake, fake, fake, 100, 1e-6, None) < 0
    assert lib.x2i_groupnorm_nhwc_bf16(fake, fake, 1, 64, 60, 4, fake, fake, 1e-5, 0, None, None, fake, None) < 0  # C % 8
    assert lib.x2i_softmax_rows_bf16(fake, 4, 12, 1.0, None) < 0  # cols % 8
    assert lib.x2i_skinny_linear(fake, 0, fake, None, fake, 8, 1, 8, 12, 0, 0, 0, None) < 0  # K % 8
    assert lib.x2i_rope_table_f32(fake, 8, 16, 55, 56, 10000.0, fake, fake, None) < 0  # odd axis width
    assert lib.x2i_timestep_sinusoid(fake, fake, 1, 255, 0, None) < 0
    assert lib.x2i_conv_stem_bf16(None, None, None, None, 1, 8, 8, 64, None) < 0
    assert lib.x2i_abi_version() >= 1


def test_training_harness_arguments_and_schedules():
    """x2i_amd.train_distill: the reference's argument names / defaults (train/train_qwenvl.py:63-167) and the lr schedule factors."""
    from x2i_amd import train_distill as TD
    a = TD.parse_args([])
    assert (a.learning_rate, a.adam_beta1, a.adam_beta2, a.adam_weight_decay, a.adam_epsilon, a.max_grad_norm) == (1e-4, 0.9, 0.999, 1e-2, 1e-8, 1.0)
    assert (a.max_train_steps, a.checkpointing_steps, a.gradient_accumulation_steps, a.lr_scheduler, a.lr_warmup_steps) == (200000, 500, 1, "constant", 500)
    assert TD.lr_factor("constant", 7, 500, 1000) == 1.0
    assert TD.lr_factor("constant_with_warmup", 250, 500, 1000) == 0.5 and TD.lr_factor("constant_with_warmup", 900, 500, 1000) == 1.0
    assert abs(TD.lr_factor("linear", 750, 500, 1000) - 0.5) < 1e-12 and TD.lr_factor("linear", 1000, 500, 1000) == 0.0
    assert abs(TD.lr_factor("cosine", 750, 500, 1000) - 0.5) < 1e-12 and abs(TD.lr_factor("cosine", 500, 500, 1000) - 1.0) < 1e-12


def test_training_harness_finds_the_newest_checkpoint(tmp_path):
    from x2i_amd import train_distill as TD
    assert TD.latest_checkpoint(str(tmp_path)) == (0, None)
    for d in ("500", "1500", "1000", "notes"):
        (tmp_path / d).mkdir()
    (tmp_path / "500" / "diffusion_pytorch_model.bin").write_bytes(b"x")
    (tmp_path / "1000" / "diffusion_pytorch_model.bin").write_bytes(b"x")
    step, path = TD.latest_checkpoint(str(tmp_path))      # 1500 has no model file: not a checkpoint
    assert step == 1000 and path.endswith("1000/diffusion_pytorch_model.bin")


def test_generated_gemm_loop_is_current():
    """csrc/gemm256w_loop.inc (the hand-scheduled K-loop of the 4-wave GEMM) is what csrc/gen_gemm256w.py emits: the generator asserts
    the register / buffer hazards of its schedule table and derives the wait counts, so a stale or hand-edited .inc fails here."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gen = os.path.join(here, "x2i_amd", "csrc", "gen_gemm256w.py")
    assert subprocess.run([sys.executable, gen, "--check"]).returncode == 0
    # ... and the e4m3 loop of the same kernel (csrc/gemm256f8_loop.inc <- gen_gemm256f8.py: fragment lifetimes, buffer reuse, SCC and
    # the M0 / piece pairing are asserted by the generator, lgkmcnt counts derived from the simulated LDS queue)
    gen8 = os.path.join(here, "x2i_amd", "csrc", "gen_gemm256f8.py")
    assert subprocess.run([sys.executable, gen8, "--check"]).returncode == 0
    # ... and the K-loop of the "two residents" A/B kernel (csrc/gemm_r2_loop.inc <- gen_gemm_r2.py: wait counts from the simulated queues)
    gen_r2 = os.path.join(here, "x2i_amd", "csrc", "gen_gemm_r2.py")
    assert subprocess.run([sys.executable, gen_r2, "--check"]).returncode == 0


def test_generated_attention_statement_is_current():
    """csrc/attn_w4_loop.inc (the hand-scheduled attention kernel body) is what csrc/gen_attn_w4.py emits with its default switches
    (no measurement ablation, product fragment lead): the generator asserts its own hazards (a P group is not converted over before
    the last MFMA that reads it, the defer-max decision precedes the MFMAs that read the -max copies, ring occupancy)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gen = os.path.join(here, "x2i_amd", "csrc", "gen_attn_w4.py")
    env = {k: v for k, v in os.environ.items() if not k.startswith("X2I_ATTN_")}
    assert subprocess.run([sys.executable, gen, "--check"], env=env).returncode == 0
    # ... and the same program on the 16 x 16 x 32 MFMA shape (csrc/attn_w16_loop.inc <- gen_attn_w16.py; A/B kernel, attn_variant = 12)
    gen16 = os.path.join(here, "x2i_amd", "csrc", "gen_attn_w16.py")
    assert subprocess.run([sys.executable, gen16, "--check"], env=env).returncode == 0
