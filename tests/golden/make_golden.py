#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference); the fixtures it
writes are data (seeds, inputs, expected outputs) and are committed.  Nothing
from the reference's source text is written anywhere.

  python tests/golden/make_golden.py            # regenerate everything

What executes reference code:
  * utils/proj.py                       imported as-is               -> proj_*.safetensors
  * model_internvl/proj.py              imported with stub modules   -> legacy_*.safetensors
  * lightcontrol/lightcontrol_flux.py   imported under diffusers_shim -> flux_*.safetensors, controlnext_*.safetensors
  * train/train_qwenvl.py, lightcontrol/train_lightcontrol.py: the helper
    functions _pack_latents/_unpack_latents/_prepare_latent_image_ids/
    calculate_shift are extracted with `ast` and executed         -> helpers.safetensors
  * train/train_qwenvl.py: retrieve_timesteps (:248-281) and the sigma / mu / timestep construction statements of the
    teacher step (:753-771); lightcontrol/train_lightcontrol.py: get_sigmas (:412-421) and the flow-matching noising
    statement (:706) -- extracted with `ast` and executed against the build's scheduler class (diffusers' own class is
    not installed), which pins the scheduler's call protocol and attributes            -> scheduler_protocol.safetensors

  python tests/golden/make_golden.py scheduler   # regenerate only the named sections (manifest is merged)
Weights are NOT stored: they are regenerated from seeds by oracle.*.random_*_state_dict
and loaded into the reference modules with load_state_dict(strict=True) (which
also checks our key/shape tables against the reference module tree).
"""
import ast
import json
import os
import sys
import types

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import flux as OF  # noqa: E402
from oracle import projector as OP  # noqa: E402

MANIFEST = {}


def save(name, tensors, meta):
    tensors = {k: v.detach().contiguous().clone() for k, v in tensors.items()}
    save_file(tensors, os.path.join(HERE, name + ".safetensors"))
    MANIFEST[name] = meta
    print(f"  wrote {name}: " + ", ".join(f"{k}{tuple(v.shape)}" for k, v in tensors.items()))


def seeded(shape, seed, scale=1.0):
    return scale * torch.randn(shape, generator=torch.Generator().manual_seed(seed))


# --------------------------------------------------------------------------- projector (utils/proj.py)
def gen_projector():
    sys.path.insert(0, REF)
    import importlib

    proj = importlib.import_module("utils.proj")
    factories = {
        "qwen3b": lambda: proj.create_proj3_qwen3b(in_channels=37, use_t5=False, use_scale=False, use_cnn=True),
        "qwen7b": lambda: proj.create_proj3_qwen7b(in_channels=29, use_t5=False, use_scale=False, use_cnn=True),
        "internvl1b": lambda: proj.create_proj_internvl1b(in_channels=25, use_t5=False, use_scale=True),
        "internvl4b": lambda: proj.create_proj_internvl4b(in_channels=37, use_t5=False, use_scale=False),
        "minicpm": lambda: proj.create_proj_minicpm(in_channels=29, use_t5=False, use_scale=False, use_cnn=True),
    }
    for i, (kind, make) in enumerate(factories.items()):
        m = make().eval()
        sd = OP.random_proj_state_dict(kind, seed=100 + i)
        m.load_state_dict(sd, strict=True)
        f = OP.FACTORIES[kind]
        B, S = (2, 12) if kind == "qwen7b" else (1, 8)
        x = seeded((B, f["in_channels"], S, f["input_dim"]), 200 + i, 3.0)
        with torch.no_grad():
            x1, x2 = m(x)
        save(f"proj_{kind}", {"x1": x1, "x2": x2},
             dict(ref="utils/proj.py", kind=kind, weight_seed=100 + i, input_seed=200 + i, input_scale=3.0,
                  input_shape=list(x.shape)))
    # mean-over-layers branch (use_scale=False,use_cnn=False), utils/proj.py:70-71
    m = proj.create_proj_internvl1b(in_channels=25, use_t5=False, use_scale=False, use_cnn=False).eval()
    sd = OP.random_proj_state_dict("internvl1b", seed=110, use_scale=True)
    sd.pop("cha_scale")
    m.load_state_dict(sd, strict=True)
    x = seeded((1, 25, 8, 896), 210, 3.0)
    with torch.no_grad():
        x1, x2 = m(x)
    save("proj_internvl1b_mean", {"x1": x1, "x2": x2},
         dict(ref="utils/proj.py", kind="internvl1b", weight_seed=110, input_seed=210, input_scale=3.0,
              input_shape=list(x.shape), drop=["cha_scale"]))


# --------------------------------------------------------------------------- legacy (model_internvl/proj.py)
def gen_legacy():
    import importlib.machinery
    import transformers  # noqa: F401  (real one first)
    import transformers.models.t5.modeling_t5  # noqa: F401

    class _Any(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return type(k, (), {})

    for name in ["pytorch_lightning", "pytorch_lightning.callbacks", "deepspeed", "torchvision", "torchvision.utils",
                 "diffusers", "diffusers.image_processor", "diffusers.models", "diffusers.models.transformers",
                 "diffusers.models.autoencoders", "diffusers.schedulers", "diffusers.utils",
                 "diffusers.utils.torch_utils", "diffusers.models.attention", "diffusers.training_utils"]:
        if name not in sys.modules:
            m = _Any(name)
            m.__spec__ = importlib.machinery.ModuleSpec(name, None)
            m.__path__ = []
            sys.modules[name] = m
    sys.path.insert(0, os.path.join(REF, "model_internvl"))
    import importlib

    lp = importlib.import_module("proj")
    torch.manual_seed(7)
    for cls_name in ("MLP", "MLP2", "MLP_plus"):
        cls = getattr(lp, cls_name)
        kw = dict(in_dim=64, out_dim=128, hidden_dim=128, out_dim1=32)
        m = cls(**kw).eval()
        x = seeded((2, 6, 64), 300, 2.0)
        with torch.no_grad():
            x1, x2 = m(x)
        t = {"sd." + k: v for k, v in m.state_dict().items()}
        t.update(x=x, x1=x1, x2=x2)
        eps = 1e-5
        save(f"legacy_{cls_name}", t, dict(ref="model_internvl/proj.py", cls=cls_name, eps=eps))
    # full legacy Proj / Proj2 / Proj3 / Transformer_proj forwards (T5Stack / TransformerEncoder as installed: transformers 5.15)
    pkw = dict(in_channels=3, kernel_size=5, input_dim=64, output_dim0=32, output_dim1=128, num_layers=2, num_heads=2,
               layer_norm_eps=1e-6, head_dim=32)
    for i, cls_name in enumerate(("Proj", "Proj2", "Proj3")):
        torch.manual_seed(20 + i)
        m = getattr(lp, cls_name)(**pkw).eval()
        with torch.no_grad():
            for n, prm in m.named_parameters():  # liven up the default inits (LayerNorm = identity), then round to bf16 so that
                if prm.dim() == 1:               # the fixture stores exactly the weights the reference forward used, at half the size
                    prm.add_(0.1 * torch.randn(prm.shape))
                prm.copy_(prm.bfloat16().float())
            x = seeded((2, 3, 6, 64), 310 + i, 2.0)
            x1, x2 = m(x)
        # the token-embedding table (32128 x 64) is never read with inputs_embeds=: kept out of the fixture
        t = {"sd." + k: v.bfloat16() for k, v in m.state_dict().items() if k != "t5stack.embed_tokens.weight"}
        t.update(x=x, x1=x1, x2=x2)
        save(f"legacy_{cls_name}_full", t, dict(ref="model_internvl/proj.py:149-211", cls=cls_name, cfg=pkw,
                                                t5stack="transformers " + transformers.__version__))
    torch.manual_seed(30)
    m = lp.Transformer_proj(64, 2, 32, 48, num_layers=2).eval()
    x = seeded((2, 6, 64), 320, 1.0)
    with torch.no_grad():
        for prm in m.parameters():
            prm.copy_(prm.bfloat16().float())
        x1, x2 = m(x)
    t = {"sd." + k: v.bfloat16() for k, v in m.state_dict().items()}
    t.update(x=x, x1=x1, x2=x2)
    save("legacy_Transformer_proj", t, dict(ref="model_internvl/proj.py:133-147", d_model=64, n_heads=2, out_dim1=32, out_dim2=48,
                                             num_layers=2))
    # Proj front stage (norm0 -> conv -> norm1), captured with a forward hook on the reference module
    try:
        m = lp.Proj(in_channels=3, kernel_size=5, input_dim=64, output_dim0=32, output_dim1=128, num_layers=1,
                    num_heads=2, layer_norm_eps=1e-6, head_dim=32).eval()
        cap = {}
        m.norm1.register_forward_hook(lambda mod, i, o: cap.__setitem__("pre", o))
        x = seeded((2, 3, 6, 64), 301, 2.0)
        with torch.no_grad():
            try:
                m(x)
            except Exception as e:  # T5Stack of transformers 5.x may reject; the hook already fired
                print("   (Proj T5Stack stage skipped:", type(e).__name__, ")")
        t = {"sd." + k: v for k, v in m.state_dict().items() if k.split(".")[0] in ("norm0", "conv", "norm1")}
        t.update(x=x, pre=cap["pre"])
        save("legacy_Proj_pre", t, dict(ref="model_internvl/proj.py:163-166", eps=1e-6))
    except Exception as e:
        print("   legacy Proj pre-stage not generated:", repr(e))


# --------------------------------------------------------------------------- helpers extracted by ast
def _extract(path, names, extra_ns=None):
    src = open(path).read()
    tree = ast.parse(src)
    ns = {"torch": torch}
    ns.update(extra_ns or {})
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            code = compile(ast.Module(body=[node], type_ignores=[]), path, "exec")
            exec(code, ns)
    return ns


def gen_helpers():
    a = _extract(os.path.join(REF, "train/train_qwenvl.py"), {"_pack_latents", "_prepare_latent_image_ids", "calculate_shift"})
    b = _extract(os.path.join(REF, "lightcontrol/train_lightcontrol.py"), {"_unpack_latents"})
    lat = seeded((2, 16, 8, 12), 400)
    packed = a["_pack_latents"](lat, 2, 16, 8, 12)
    ids = a["_prepare_latent_image_ids"](2, 8, 12, "cpu", torch.float32)
    unpacked = b["_unpack_latents"](packed, 64, 96, 16)
    shifts = torch.tensor([a["calculate_shift"](n) for n in (256, 1024, 4096)], dtype=torch.float64)
    shifts115 = torch.tensor([a["calculate_shift"](n, 256, 4096, 0.5, 1.15) for n in (256, 1024, 4096)], dtype=torch.float64)
    save("helpers", dict(lat=lat, packed=packed, ids=ids, unpacked=unpacked, shifts=shifts, shifts115=shifts115),
         dict(ref="train/train_qwenvl.py:216-246; lightcontrol/train_lightcontrol.py:403-410", seq_lens=[256, 1024, 4096]))


# --------------------------------------------------------------------------- scheduler protocol (reference statements, our class)
def _extract_statements(path, func_pred, targets):
    """The Assign statements with the given target names, in source order, from the first function accepted by func_pred."""
    tree = ast.parse(open(path).read())

    def names(t):
        return tuple(e.id for e in t.elts) if isinstance(t, ast.Tuple) else (getattr(t, "id", None),)

    for fn in ast.walk(tree):
        if isinstance(fn, ast.FunctionDef) and func_pred(fn):
            body = [n for n in ast.walk(fn) if isinstance(n, ast.Assign) and len(n.targets) == 1 and names(n.targets[0]) in targets]
            body.sort(key=lambda n: n.lineno)
            return body
    raise RuntimeError("statements not found in " + path)


def gen_scheduler():
    import inspect
    from typing import List, Optional, Union

    import numpy as np

    from x2i_amd.pipeline import FlowMatchEulerDiscreteScheduler  # host-side class, no GPU needed

    qw = os.path.join(REF, "train/train_qwenvl.py")
    lc = os.path.join(REF, "lightcontrol/train_lightcontrol.py")
    ns = _extract(qw, {"retrieve_timesteps", "calculate_shift"},
                  dict(inspect=inspect, List=List, Optional=Optional, Union=Union, np=np))  # annotations are evaluated at def time
    calls_rt = lambda fn: any(isinstance(n, ast.Call) and getattr(n.func, "id", "") == "retrieve_timesteps" for n in ast.walk(fn))  # noqa: E731
    stmts = _extract_statements(qw, calls_rt, {("sigmas",), ("image_seq_len",), ("mu",), ("timesteps", "num_inference_steps")})
    lines = [n.lineno for n in stmts]
    code = compile(ast.Module(body=stmts, type_ignores=[]), qw, "exec")
    out, meta_cases = {}, []
    cases = [("schnell", dict(shift=1.0, use_dynamic_shifting=False), 4, 4096), ("dev", dict(shift=3.0, use_dynamic_shifting=True), 20, 4096),
             ("dev", dict(shift=3.0, use_dynamic_shifting=True), 1, 4096), ("dev", dict(shift=3.0, use_dynamic_shifting=True), 20, 1024),
             ("shift3", dict(shift=3.0, use_dynamic_shifting=False), 8, 2304)]
    for kind, kw, n, seq in cases:
        sch = FlowMatchEulerDiscreteScheduler(**kw)
        env = dict(ns, scheduler=sch, infer_device="cpu", latents=torch.zeros(1, seq, 64))
        # the reference sets num_inference_steps = 1 on the line before (:752); we vary it, the statements are unchanged
        src_n = [m for m in stmts if m.targets[0].__class__ is ast.Name and m.targets[0].id == "num_inference_steps"]
        assert not src_n
        env["num_inference_steps"] = n
        exec(code, env)
        tag = f"{kind}_n{n}_s{seq}"
        out[tag + ".timesteps"] = env["timesteps"].to(torch.float32)
        out[tag + ".sigmas"] = sch.sigmas.to(torch.float32)
        out[tag + ".mu"] = torch.tensor([env["mu"]], dtype=torch.float64)
        assert env["num_inference_steps"] == n
        meta_cases.append(dict(tag=tag, config=kw, num_inference_steps=n, image_seq_len=seq))
    # training-side use of the same class: get_sigmas + the noising statement (z_t = (1 - sigma) x + sigma z1)
    ns2 = _extract(lc, {"get_sigmas"})
    noising = _extract_statements(lc, lambda fn: any(isinstance(n, ast.Call) and getattr(n.func, "id", "") == "get_sigmas" for n in ast.walk(fn)),
                                  {("noisy_model_input",)})
    noising = [n for n in noising if isinstance(n.value, ast.BinOp)][:1]
    sch = FlowMatchEulerDiscreteScheduler(shift=3.0, use_dynamic_shifting=True)  # FLUX.1-dev training scheduler (:495-499)
    idx = torch.tensor([0, 17, 500, 999])
    ts = sch.timesteps[idx]
    x, z = seeded((4, 16, 8, 8), 410), seeded((4, 16, 8, 8), 411)
    sig = ns2["get_sigmas"](ts, sch, "cpu", n_dim=4, dtype=torch.float32)
    env = dict(sigmas=sig, model_input=x, noise=z)
    exec(compile(ast.Module(body=noising, type_ignores=[]), lc, "exec"), env)
    out.update({"train.indices": idx, "train.timesteps": ts, "train.sigmas": sig, "train.model_input": x, "train.noise": z,
                "train.noisy": env["noisy_model_input"]})
    save("scheduler_protocol", out,
         dict(ref="train/train_qwenvl.py:248-281 (retrieve_timesteps), statements at lines %s; lightcontrol/train_lightcontrol.py:412-421 "
                  "(get_sigmas), noising statement at line %d" % (lines, noising[0].lineno),
              executed_against="x2i_amd.pipeline.FlowMatchEulerDiscreteScheduler (diffusers is not installed)", cases=meta_cases))


# --------------------------------------------------------------------------- composition (lightcontrol_flux.py under shim)
def gen_flux():
    import diffusers_shim

    diffusers_shim.install()
    sys.path.insert(0, os.path.join(REF, "lightcontrol"))
    import importlib

    lf = importlib.import_module("lightcontrol_flux")

    def tiny_cfg(guidance):
        return dict(patch_size=1, in_channels=64, num_layers=2, num_single_layers=2, attention_head_dim=128,
                    num_attention_heads=2, joint_attention_dim=128, pooled_projection_dim=64,
                    guidance_embeds=guidance, axes_dims_rope=(16, 56, 56))

    def inputs(B, St, h2, w2, cfg, seed):
        from oracle import sampler as OS
        return dict(
            hidden=seeded((B, h2 * w2, cfg["in_channels"]), seed),
            enc=seeded((B, St, cfg["joint_attention_dim"]), seed + 1),
            pooled=seeded((B, cfg["pooled_projection_dim"]), seed + 2),
            timestep=torch.tensor([0.75, 0.25][:B] if B <= 2 else [0.5] * B),
            img_ids=OS.prepare_latent_image_ids(h2, w2),
            txt_ids=torch.zeros(St, 3),
        )

    # D1: schnell-like tiny model, ragged sequence (40 text + 96 image tokens)
    cfg = tiny_cfg(False)
    model = lf.FluxTransformer2DModel(**cfg).eval()
    model.load_state_dict(OF.random_flux_state_dict(cfg, seed=500, std=0.05), strict=True)
    inp = inputs(2, 40, 8, 12, cfg, 510)
    with torch.no_grad():
        out = model(hidden_states=inp["hidden"], encoder_hidden_states=inp["enc"], pooled_projections=inp["pooled"],
                    timestep=inp["timestep"], img_ids=inp["img_ids"], txt_ids=inp["txt_ids"], guidance=None,
                    control_nets=[], return_dict=False)
    save("flux_tiny_schnell", dict(out=out, **inp),
         dict(ref="lightcontrol/lightcontrol_flux.py:390-553", cfg=cfg, weight_seed=500, weight_std=0.05))

    # D2: dev-like (guidance) tiny model + 2 ControlNeXt on the 2 double blocks (Row L)
    cfg = tiny_cfg(True)
    D = 256
    model = lf.FluxTransformer2DModel(**cfg).eval()
    model.load_state_dict(OF.random_flux_state_dict(cfg, seed=501, std=0.05), strict=True)
    nets = []
    for j in range(2):
        net = lf.ControlNeXtModel().eval()
        # the reference hard-codes 3072 output channels (lightcontrol_flux.py:661-668); at reduced width the
        # final conv is swapped for one of the model's width -- forward code path is unchanged.
        net.mid_convs[1] = torch.nn.Conv2d(256, D, kernel_size=2, stride=2)
        net.load_state_dict(OF.random_controlnext_state_dict(seed=520 + j, out_channels=D), strict=True)
        nets.append(net)
    inp = inputs(2, 40, 8, 12, cfg, 530)
    hint = torch.rand((2, 3, 128, 192), generator=torch.Generator().manual_seed(540)) * 2 - 1
    guidance = torch.full([2], 3.5)
    with torch.no_grad():
        out = model(hidden_states=inp["hidden"], encoder_hidden_states=inp["enc"], pooled_projections=inp["pooled"],
                    timestep=inp["timestep"], img_ids=inp["img_ids"], txt_ids=inp["txt_ids"], guidance=guidance,
                    guided_hint=hint, control_nets=torch.nn.ModuleList(nets), return_dict=False)
    save("flux_tiny_dev_control", dict(out=out, hint=hint, guidance=guidance, **inp),
         dict(ref="lightcontrol/lightcontrol_flux.py:390-553,708-749", cfg=cfg, weight_seed=501, weight_std=0.05,
              control_seeds=[520, 521], control_out_channels=D))

    # D3: ONE full-width (D=3072, 24 heads) double block and single block, tiny sequence
    from oracle import primitives as P
    full = dict(OF.DEFAULT_CFG)
    full.update(num_layers=1, num_single_layers=1)
    sd = OF.random_flux_state_dict(full, seed=502, std=0.02)
    dbl = lf.FluxTransformerBlock(3072, 24, 128).eval()
    dbl.load_state_dict({k[len("transformer_blocks.0."):]: v for k, v in sd.items() if k.startswith("transformer_blocks.0.")}, strict=True)
    sgl = lf.FluxSingleTransformerBlock(3072, 24, 128).eval()
    sgl.load_state_dict({k[len("single_transformer_blocks.0."):]: v for k, v in sd.items() if k.startswith("single_transformer_blocks.0.")}, strict=True)
    St, h2, w2 = 16, 4, 4
    from oracle import sampler as OS
    ids = torch.cat([torch.zeros(St, 3), OS.prepare_latent_image_ids(h2, w2)], 0)
    rotary = P.flux_pos_embed(ids)
    hidden = seeded((1, h2 * w2, 3072), 550)
    enc = seeded((1, St, 3072), 551)
    temb = seeded((1, 3072), 552)
    with torch.no_grad():
        enc_o, hid_o = dbl(hidden_states=hidden, encoder_hidden_states=enc, temb=temb, image_rotary_emb=rotary)
        joint = torch.cat([enc, hidden], 1)
        sgl_o = sgl(hidden_states=joint, temb=temb, image_rotary_emb=rotary)
    save("flux_full_width_blocks", dict(hidden=hidden, enc=enc, temb=temb, enc_out=enc_o, hidden_out=hid_o, single_out=sgl_o),
         dict(ref="lightcontrol/lightcontrol_flux.py:82-104,159-204", weight_seed=502, weight_std=0.02, St=St, h2=h2, w2=w2))

    # D4: unmodified ControlNeXtModel
    net = lf.ControlNeXtModel().eval()
    net.load_state_dict(OF.random_controlnext_state_dict(seed=560), strict=True)
    hint = torch.rand((1, 3, 64, 64), generator=torch.Generator().manual_seed(561)) * 2 - 1
    t = torch.tensor([500.0])
    with torch.no_grad():
        o = net(hint, t)
    save("controlnext_full", dict(hint=hint, timestep=t, out=o["out"]),
         dict(ref="lightcontrol/lightcontrol_flux.py:575-749", weight_seed=560, scale=float(o["scale"])))


# --------------------------------------------------------------------------- distillation loss (reference statements executed as they stand)
def gen_distill():
    """train/train_qwenvl.py:58-61 (`normalize`) and :611-634 (`loss = 0`, `temperature0 = 3`, the two `for i in range(19 / 38)` loops with
    their F.kl_div terms and non-finite guards) extracted by `ast` and executed on seeded tensors; the loss value and, through autograd, its
    gradient with respect to the three student tensors are the golden outputs."""
    import torch.nn.functional as F
    qw = os.path.join(REF, "train/train_qwenvl.py")
    ns = _extract(qw, {"normalize"})
    tree = ast.parse(open(qw).read())

    def has_kl(node):
        return any(isinstance(n, ast.Attribute) and n.attr == "kl_div" for n in ast.walk(node))

    stmts = None
    for fn in ast.walk(tree):
        if isinstance(fn, ast.FunctionDef) and has_kl(fn):
            body = []
            for n in ast.walk(fn):
                if isinstance(n, ast.Assign) and len(n.targets) == 1 and getattr(n.targets[0], "id", None) in ("loss", "temperature0") \
                        and isinstance(n.value, ast.Constant):
                    body.append(n)
                if isinstance(n, ast.For) and has_kl(n) and isinstance(n.iter, ast.Call) and getattr(n.iter.func, "id", "") == "range" \
                        and getattr(n.target, "id", "") == "i":   # the innermost per-block loops, not the epoch / batch loops around them
                    body.append(n)
            body.sort(key=lambda n: n.lineno)
            stmts = body
            break
    assert stmts and sum(isinstance(n, ast.For) for n in stmts) == 2, "reference loss statements not found"
    B, S0, S1, S2, D = 2, 6, 4, 10, 64
    # (values rounded to bf16 so that the bf16 HIP kernel sees exactly the tensors the reference statements saw)
    t = [seeded(sh, 700 + i, 0.8).bfloat16().float() for i, sh in enumerate(((B, 19, S0, D), (B, 19, S1, D), (B, 38, S2, D)))]
    s_ = [(t[i] * 0.5 + seeded(t[i].shape, 710 + i, 0.9)).bfloat16().float() for i in range(3)]
    with torch.enable_grad():
        st = [x.clone().requires_grad_(True) for x in s_]
        ns.update(F=F, torch=torch, KD_teacher_tensor0=t[0], KD_teacher_tensor1=t[1], KD_teacher_tensor2=t[2], KD_student_tensor0=st[0],
                  KD_student_tensor1=st[1], KD_student_tensor2=st[2], print=lambda *a, **k: None)
        exec(compile(ast.Module(body=stmts, type_ignores=[]), qw, "exec"), ns)
        loss = ns["loss"]
        loss.backward()
    save("distill_loss", dict(teacher0=t[0], teacher1=t[1], teacher2=t[2], student0=s_[0], student1=s_[1], student2=s_[2],
                              loss=loss.detach().reshape(1), grad0=st[0].grad, grad1=st[1].grad, grad2=st[2].grad),
         dict(ref="train/train_qwenvl.py:58-61 and the statements at the listed lines", temperature=float(ns["temperature0"]),
              lines=[int(n.lineno) for n in stmts]))


if __name__ == "__main__":
    torch.set_grad_enabled(False)
    sections = [("legacy", gen_legacy),  # before gen_flux: its permissive stub modules must not shadow the shim
                ("projector", gen_projector), ("helpers", gen_helpers), ("scheduler", gen_scheduler), ("flux", gen_flux), ("distill", gen_distill)]
    only = set(sys.argv[1:])
    if only:  # partial regeneration: keep the other sections' manifest entries
        MANIFEST.update(json.load(open(os.path.join(HERE, "manifest.json"))))
    for name, fn in sections:
        if not only or name in only:
            print(name + ":")
            fn()
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(MANIFEST, f, indent=1, sort_keys=True)
    print("manifest written")
