"""Oracle restatement of the MLLM -> T5/CLIP-slot alignment projectors.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PINNED against the reference's
own utils/proj.py and model_internvl/proj.py (tests/golden/make_golden.py).

Functional over the reference's state-dict key names:
  conv.weight [1,C,5,5], conv.bias [1]  |  cha_scale [1,C,1,1]
  mlp.layernorm.{weight,bias} [H], mlp.projector.{0,2}.weight, mlp.fc.1.{weight,bias}
"""
import torch
import torch.nn.functional as F

# (input_dim H, in_channels C used by the inference scripts) per factory, utils/proj.py:74-96
FACTORIES = {
    "qwen3b": dict(input_dim=2048, in_channels=37),      # infer/inference_qwenvl.py:80
    "qwen7b": dict(input_dim=3584, in_channels=29),      # infer/inference_qwenvl.py:82
    "internvl1b": dict(input_dim=896, in_channels=25),   # infer/inference_internvl.py:76
    "internvl4b": dict(input_dim=2048, in_channels=37),  # infer/inference_internvl.py:78
    "minicpm": dict(input_dim=3584, in_channels=29),     # infer/inference_minicpm.py:78
}


def mlp3(sd, prefix, x, eps=1e-6):
    """MLP3.forward -- utils/proj.py:28-33 (ctor :15-26).  Returns (x1 pooled, x2 prompt_embeds)."""
    H = x.shape[-1]
    x = F.layer_norm(x, (H,), sd[prefix + "layernorm.weight"], sd[prefix + "layernorm.bias"], eps)  # :29
    h = F.gelu(F.linear(x, sd[prefix + "projector.0.weight"]))  # :18-19 (erf GELU)
    x2 = F.linear(h, sd[prefix + "projector.2.weight"])  # :20, :30
    x1 = F.linear(F.gelu(x2), sd[prefix + "fc.1.weight"], sd[prefix + "fc.1.bias"])  # :22-25, :31
    x1 = torch.mean(x1, 1)  # :32
    return x1, x2


def proj7exp(sd, x, eps=1e-6):
    """Proj7Exp.forward with use_t5=False -- utils/proj.py:62-72.

    Layer fusion is chosen by which weights the checkpoint holds, as the
    constructor does (:47-50): cha_scale -> scaled mean; conv -> 5x5 conv over
    the (S,H) plane; neither -> plain mean over layers.
    """
    B, C, S, H = x.shape
    if "cha_scale" in sd:
        x = (sd["cha_scale"] * x.view(B, C, S, H)).mean(dim=1)  # :66-67
    elif "conv.weight" in sd:
        k = sd["conv.weight"].shape[-1]
        x = F.conv2d(x.view(B, C, S, H), sd["conv.weight"], sd["conv.bias"], padding=(k - 1) // 2).squeeze(1)  # :68-69
    else:
        x = x.view(B, C, S, H).mean(dim=1)  # :70-71
    return mlp3(sd, "mlp.", x, eps)  # :72


def legacy_mlp(sd, x, eps=1e-5, variant="MLP"):
    """MLP / MLP2 / MLP_plus -- model_internvl/proj.py:53-73, :76-102, :104-130.

    projector = Linear,GELU,... (no bias); x2 = GELU(projector(LN(x))); x1 = mean_S(fc(x2)).
    """
    H = x.shape[-1]
    x = F.layer_norm(x, (H,), sd["layernorm.weight"], sd["layernorm.bias"], eps)
    idx = sorted(int(k.split(".")[1]) for k in sd if k.startswith("projector.") and k.endswith(".weight"))
    for n, i in enumerate(idx):
        x = F.linear(x, sd[f"projector.{i}.weight"])
        if n != len(idx) - 1:
            x = F.gelu(x)
    x2 = F.gelu(x)
    if "fc.weight" in sd:  # MLP / MLP_plus: single biased Linear
        x1 = F.linear(x2, sd["fc.weight"], sd["fc.bias"])
    else:  # MLP2: Linear,GELU,Linear,GELU,Linear (no bias)
        x1 = x2
        fidx = sorted(int(k.split(".")[1]) for k in sd if k.startswith("fc.") and k.endswith(".weight"))
        for n, i in enumerate(fidx):
            x1 = F.linear(x1, sd[f"fc.{i}.weight"])
            if n != len(fidx) - 1:
                x1 = F.gelu(x1)
    return torch.mean(x1, 1), x2


def legacy_proj_pre(sd, x, eps=1e-6):
    """The non-T5 front stage of Proj/Proj2 -- model_internvl/proj.py:163-166: LN -> conv5x5 -> LN."""
    H = x.shape[-1]
    x = F.layer_norm(x, (H,), sd["norm0.weight"], sd["norm0.bias"], eps)
    k = sd["conv.weight"].shape[-1]
    x = F.conv2d(x, sd["conv.weight"], sd["conv.bias"], padding=(k - 1) // 2).squeeze(1)
    return F.layer_norm(x, (H,), sd["norm1.weight"], sd["norm1.bias"], eps)


def random_proj_state_dict(kind, seed=0, use_scale=None, dtype=torch.float32, device="cpu"):
    """Seeded weights with the reference key names (for tests / bench)."""
    f = FACTORIES[kind]
    H, C = f["input_dim"], f["in_channels"]
    if use_scale is None:
        use_scale = kind == "internvl1b"
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd = {}
    if use_scale:
        sd["cha_scale"] = torch.randn(1, C, 1, 1, generator=g) * (2.0 / (C + 1)) ** 0.5
    else:
        sd["conv.weight"] = torch.randn(1, C, 5, 5, generator=g) / (C * 25) ** 0.5
        sd["conv.bias"] = torch.randn(1, generator=g) * 0.01
    sd["mlp.layernorm.weight"] = 1.0 + 0.1 * torch.randn(H, generator=g)
    sd["mlp.layernorm.bias"] = 0.02 * torch.randn(H, generator=g)
    sd["mlp.projector.0.weight"] = torch.randn(4096, H, generator=g) / H ** 0.5
    sd["mlp.projector.2.weight"] = torch.randn(4096, 4096, generator=g) / 64.0
    sd["mlp.fc.1.weight"] = torch.randn(768, 4096, generator=g) / 64.0
    sd["mlp.fc.1.bias"] = 0.02 * torch.randn(768, generator=g)
    return {k: v.to(device=device, dtype=dtype) for k, v in sd.items()}


def _t5stack(cfg, sd, device="cpu"):
    """transformers' T5Stack with the reference's configuration (model_internvl/proj.py:152-153) -- third-party code on both sides."""
    from transformers import T5Config
    from transformers.models.t5.modeling_t5 import T5Stack
    config = T5Config(num_heads=cfg["num_heads"], num_layers=cfg["num_layers"], num_decoder_layers=0, layer_norm_epsilon=cfg["layer_norm_eps"],
                      is_encoder_decoder=False, is_decoder=False, d_ff=cfg["input_dim"] * 4, d_kv=cfg["head_dim"], d_model=cfg["input_dim"],
                      dense_act_fn="gelu_new", feed_forward_proj="gated-gelu", use_cache=False)
    m = T5Stack(config).eval()
    m.load_state_dict({k[len("t5stack."):]: v for k, v in sd.items() if k.startswith("t5stack.")}, strict=True)
    return m


def legacy_proj(sd, x, cfg, t5_first=False):
    """Proj / Proj2 (t5_first=False, model_internvl/proj.py:162-167,182-187) and Proj3 (t5_first=True, :203-211):
    front stage (LN -> conv5x5 -> LN) and MLP / MLP2 head restated here, T5Stack from `transformers` (as in the reference)."""
    t5 = _t5stack(cfg, sd)
    mlp = {k[4:]: v for k, v in sd.items() if k.startswith("mlp.")}
    with torch.no_grad():
        if t5_first:
            B, C, S, H = x.shape
            x = t5(inputs_embeds=x.contiguous().view(B * C, S, H)).last_hidden_state
            x = legacy_proj_pre(sd, x.view(B, C, S, H), cfg["layer_norm_eps"])
        else:
            x = t5(inputs_embeds=legacy_proj_pre(sd, x, cfg["layer_norm_eps"])).last_hidden_state
        return legacy_mlp(mlp, x, eps=cfg["layer_norm_eps"])


def transformer_proj(sd, x, d_model, n_heads, num_layers):
    """Transformer_proj (model_internvl/proj.py:133-147); nn.TransformerEncoder from torch on both sides."""
    layer = torch.nn.TransformerEncoderLayer(d_model=d_model, nhead=n_heads, dim_feedforward=2048, batch_first=True)
    enc = torch.nn.TransformerEncoder(layer, num_layers=num_layers).eval()
    enc.load_state_dict({k[len("transformer_encoder."):]: v for k, v in sd.items() if k.startswith("transformer_encoder.")}, strict=True)
    with torch.no_grad():
        x = enc(x)
        x1 = torch.mean(F.linear(x, sd["linear1.weight"], sd["linear1.bias"]), 1)
        return x1, F.linear(x, sd["linear2.weight"], sd["linear2.bias"])
