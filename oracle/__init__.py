"""CPU oracle for the X2I sampling hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (CPU, fp32-capable) restatement of the
reference algorithm for the path named in BASELINE.json `north_star`:

  * alignment projector            -> oracle/projector.py   (utils/proj.py, model_internvl/proj.py)
  * FLUX DiT blocks + model        -> oracle/flux.py        (lightcontrol/lightcontrol_flux.py)
  * diffusers 0.31.0 primitives    -> oracle/primitives.py  (third-party, NOT under /root/reference)
  * sampling loop + scheduler      -> oracle/sampler.py     (diffusers FluxPipeline / FlowMatchEuler...)
  * ControlNeXt hint encoder       -> oracle/flux.py `controlnext_forward` (lightcontrol/lightcontrol_flux.py:575-749)
  * VAE decode (next row N1)       -> oracle/vae.py         (diffusers AutoencoderKL.decode, third-party)

Who may import it: tests/, __graft_entry__.smoke(), and bench.py's
`cpu_baseline` leg -- always as the checker / reported baseline, never as the
thing that is shipped or measured as the product.  Nothing under x2i_amd/
imports this package; the product path raises if the HIP library is missing.

How it is pinned (see DESIGN.md "Oracle"):
  * projector: PINNED -- tests/golden/make_golden.py imports the reference's own
    utils/proj.py and model_internvl/proj.py and records (weights, input, output)
    fixtures; tests/test_oracle_golden.py replays them.
  * block / model / ControlNeXt COMPOSITION: PINNED -- the reference's
    lightcontrol/lightcontrol_flux.py is imported under a shim `diffusers`
    package built from oracle/primitives.py, so the reference's own forward
    code produced the fixtures.
  * diffusers==0.31.0 PRIMITIVES (Attention/FluxAttnProcessor2_0, RMSNorm,
    AdaLayerNorm*, FluxPosEmbed/apply_rotary_emb, Timesteps, FeedForward,
    ResnetBlock2D, FlowMatchEulerDiscreteScheduler, FluxPipeline): PARITY
    UNPINNED.  diffusers is an un-vendored third-party dependency
    (requirements.txt:3, `diffusers==0.31.0`), absent from /root/reference and
    from this image; primitives.py restates its published algorithm and is
    anchored by (1) known-answer tests (tests/test_oracle_known_answers.py), (2) an
    independent SECOND derivation of every primitive -- complex-multiplication RoPE,
    explicit float64 softmax(QK^T/sqrt d)V with text-first joint order, float64
    RMSNorm with the bf16 weight-cast rule, AdaLN chunk orders with distinguishable
    constants, scalar-loop Timesteps / schedule closed forms
    (tests/test_oracle_second_derivation.py; no expected value there comes from
    primitives.py), (3) the in-repo copies of pack/unpack/ids/calculate_shift
    (train/train_qwenvl.py:216-246, lightcontrol/train_lightcontrol.py:403-410) and
    (4) the reference's own retrieve_timesteps / sigma construction / get_sigmas /
    noising statements executed against the build's scheduler
    (tests/golden/scheduler_protocol.safetensors).
"""
