"""MLLM -> T5-slot / CLIP-slot alignment projector on the HIP path.

Same surface as the reference's utils/proj.py (class names, factory names and arguments, state-dict keys, return
order): `pooled, prompt_embeds = proj(text_embeddings)` with text_embeddings [B, C, S, H]
(infer/inference_qwenvl.py:77-94,179).  `use_t5=True` is dead code in the reference (T5Stack is never imported,
utils/proj.py:42,46) and is rejected here.

Kernel chain (all in libx2i_hip.so):
  layer fusion   proj_conv5x5 | proj_layer_mean           HBM-bound, reads the 78-106 MB/sample input once
  LayerNorm(H)   ln_affine                                 wave-per-row
  Linear+GELU    gemm (GELU-erf epilogue)                  MFMA
  Linear         gemm, dual output: x2 and GELU(x2)        MFMA (second output feeds fc without an extra pass)
  fc Linear      gemm -> f32 [B,S,768], then seq_mean      MFMA + tiny reduction
"""
import torch
import torch.nn as nn

from . import ops
from .ops import ACT_GELU_ERF, ACT_NONE


def _param(*shape, device, dtype):
    return nn.Parameter(torch.empty(shape, device=device, dtype=dtype), requires_grad=False)


def _conv5x5(mod, x):
    """Conv2d(C->1, 5, pad 2) of `mod.conv` over the (S,H) plane on the matrix cores.  The banded-Toeplitz fragment table is packed
    once per weight value (cache keyed on the parameter's version counter / storage / device, like the packed VAE weights)."""
    w, b = mod.conv.weight, mod.conv.bias
    key = (w._version, w.data_ptr(), b._version, b.data_ptr(), str(w.device))
    cache = mod.__dict__.get("_conv5x5_cache")
    if cache is None or cache[0] != key:
        C = w.shape[1]
        cache = (key, ops.proj_conv5x5_pack(w.detach().float().reshape(C, 25).contiguous()), b.detach().float().contiguous())
        mod.__dict__["_conv5x5_cache"] = cache
    return ops.proj_conv5x5_packed(x, cache[1], cache[2])


class _LN(nn.Module):
    def __init__(self, dim, device, dtype):
        super().__init__()
        self.weight = _param(dim, device=device, dtype=dtype)
        self.bias = _param(dim, device=device, dtype=dtype)


class _Lin(nn.Module):
    def __init__(self, i, o, bias, device, dtype):
        super().__init__()
        self.weight = _param(o, i, device=device, dtype=dtype)
        if bias:
            self.bias = _param(o, device=device, dtype=dtype)
        else:
            self.bias = None


class _Sparse(nn.Module):
    def __init__(self, items):
        super().__init__()
        for k, v in items.items():
            self.add_module(str(k), v)

    def __getitem__(self, i):
        return self._modules[str(i)]


class MLP3(nn.Module):
    """utils/proj.py:14-33.  Keys: layernorm.{weight,bias}, projector.{0,2}.weight, fc.1.{weight,bias}."""

    def __init__(self, in_dim=4096, out_dim=4096, hidden_dim=4096, out_dim1=768, layer_norm_eps=1e-5, use_residual=True,
                 device="cuda", dtype=torch.bfloat16):
        super().__init__()
        self.eps = layer_norm_eps
        self.layernorm = _LN(in_dim, device, dtype)
        self.projector = _Sparse({0: _Lin(in_dim, hidden_dim, False, device, dtype), 2: _Lin(hidden_dim, hidden_dim, False, device, dtype)})
        self.fc = _Sparse({1: _Lin(out_dim, out_dim1, True, device, dtype)})

    @torch.no_grad()
    def forward(self, x):
        B, S, H = x.shape
        x = x.to(torch.bfloat16).contiguous()
        xn = ops.ln_affine(x, self.layernorm.weight, self.layernorm.bias, self.eps)  # :29
        h = ops.gemm(xn, self.projector[0].weight, act=ACT_GELU_ERF, M=B * S)  # :18-19
        x2 = torch.empty((B, S, self.projector[2].weight.shape[0]), device=x.device, dtype=torch.bfloat16)
        g2 = torch.empty_like(x2)
        ops.gemm(h, self.projector[2].weight, out=x2, out2=g2, act2=ACT_GELU_ERF, M=B * S)  # :20 and the GELU of :23
        x1_tok = ops.gemm(g2, self.fc[1].weight, self.fc[1].bias, M=B * S, out_f32=True)  # :24
        x1 = ops.seq_mean(x1_tok.view(B, S, -1))  # :32
        return x1.to(torch.bfloat16), x2


class Proj7Exp(nn.Module):
    """utils/proj.py:35-72 with use_t5=False.  Keys: conv.{weight,bias} | cha_scale, mlp.*"""

    def __init__(self, in_channels=25, kernel_size=5, input_dim=896, output_dim0=768, output_dim1=4096, num_layers=2,
                 num_heads=12, norm_eps=1e-6, head_dim=64, use_t5=True, use_scale=True, use_cnn=True, device="cuda",
                 dtype=torch.bfloat16):
        super().__init__()
        if use_t5:
            raise NotImplementedError(
                "use_t5=True is dead code in the reference (utils/proj.py:42 raises NameError: T5Config); "
                "every inference script passes use_t5=False")
        if kernel_size != 5:
            raise ValueError("the HIP layer-fusion kernel implements the reference's 5x5 convolution")
        self.use_t5, self.use_scale, self.use_cnn = use_t5, use_scale, use_cnn
        if use_scale:
            self.cha_scale = _param(1, in_channels, 1, 1, device=device, dtype=dtype)
        elif use_cnn:
            conv = nn.Module()
            conv.weight = _param(1, in_channels, kernel_size, kernel_size, device=device, dtype=dtype)
            conv.bias = _param(1, device=device, dtype=dtype)
            self.conv = conv
        self.mlp = MLP3(input_dim, output_dim1, output_dim1, output_dim0, norm_eps, device=device, dtype=dtype)

    @torch.no_grad()
    def forward(self, x):
        B, C, S, H = x.shape
        x = x.to(torch.bfloat16).contiguous()
        if self.use_scale:
            x = ops.proj_layer_mean(x, self.cha_scale.float().reshape(-1).contiguous())  # :66-67
        elif self.use_cnn:
            x = _conv5x5(self, x)  # :68-69
        else:
            x = ops.proj_layer_mean(x, None)  # :70-71
        return self.mlp(x)  # :72

    @torch.no_grad()
    def init_random_(self, seed=0):
        g = torch.Generator(device="cpu").manual_seed(seed)
        for n, p in self.named_parameters():
            if n.endswith("layernorm.weight"):
                v = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
            elif p.dim() >= 2:
                fan_in = p[0].numel()
                v = torch.randn(p.shape, generator=g) / fan_in ** 0.5
            else:
                v = 0.02 * torch.randn(p.shape, generator=g)
            p.copy_(v)
        return self


# ---- factories: names, arguments and per-model dimensions as utils/proj.py:74-96 (prints dropped)
def create_proj3_qwen3b(in_channels, use_t5=True, use_scale=True, use_cnn=False, **kw):
    use_cnn = False if use_scale else use_cnn
    return Proj7Exp(in_channels=in_channels, kernel_size=5, input_dim=2048, output_dim0=768, output_dim1=4096, num_layers=2,
                    num_heads=28, norm_eps=1e-6, head_dim=128, use_t5=use_t5, use_scale=use_scale, use_cnn=use_cnn, **kw)


def create_proj3_qwen7b(in_channels, use_t5=True, use_scale=True, use_cnn=False, **kw):
    use_cnn = False if use_scale else use_cnn
    return Proj7Exp(in_channels=in_channels, kernel_size=5, input_dim=3584, output_dim0=768, output_dim1=4096, num_layers=2,
                    num_heads=28, norm_eps=1e-6, head_dim=128, use_t5=use_t5, use_scale=use_scale, use_cnn=use_cnn, **kw)


def create_proj_internvl1b(in_channels, use_t5=True, use_scale=True, use_cnn=True, **kw):
    return Proj7Exp(in_channels=in_channels, kernel_size=5, input_dim=896, output_dim0=768, output_dim1=4096, num_layers=2,
                    num_heads=12, norm_eps=1e-6, head_dim=64, use_t5=use_t5, use_scale=use_scale, use_cnn=use_cnn, **kw)


def create_proj_internvl4b(in_channels, use_t5=True, use_scale=False, use_cnn=True, **kw):
    return Proj7Exp(in_channels=in_channels, kernel_size=5, input_dim=2048, output_dim0=768, output_dim1=4096, num_layers=2,
                    num_heads=16, norm_eps=1e-6, head_dim=128, use_t5=use_t5, use_scale=use_scale, use_cnn=use_cnn, **kw)


def create_proj_minicpm(in_channels, use_t5=True, use_scale=True, use_cnn=False, **kw):
    use_cnn = False if use_scale else use_cnn
    return Proj7Exp(in_channels=in_channels, kernel_size=5, input_dim=3584, output_dim0=768, output_dim1=4096, num_layers=2,
                    num_heads=28, norm_eps=1e-6, head_dim=128, use_t5=use_t5, use_scale=use_scale, use_cnn=use_cnn, **kw)


# ------------------------------------------------------------------------------------------------------------------
# Legacy projector heads (model_internvl/proj.py) -- imported by no reference script, kept for checkpoint compatibility.
# ------------------------------------------------------------------------------------------------------------------
class _LegacyMLP(nn.Module):
    """Common body of MLP / MLP2 / MLP_plus (model_internvl/proj.py:53-130): LayerNorm -> [Linear,GELU]*n -> Linear
    (no bias) ; x2 = GELU(.) ; x1 = mean_S(fc(x2))."""

    n_proj = 3
    fc_chain = False

    def __init__(self, in_dim=4096, out_dim=4096, hidden_dim=4096, out_dim1=768, layer_norm_eps=1e-5, use_residual=True,
                 device="cuda", dtype=torch.bfloat16):
        super().__init__()
        self.eps = layer_norm_eps
        self.layernorm = _LN(in_dim, device, dtype)
        dims = [in_dim] + [hidden_dim] * (self.n_proj - 1) + [out_dim]
        self.projector = _Sparse({2 * i: _Lin(dims[i], dims[i + 1], False, device, dtype) for i in range(self.n_proj)})
        if self.fc_chain:  # MLP2: Linear,GELU,Linear,GELU,Linear without bias (:91-97)
            self.fc = _Sparse({0: _Lin(out_dim, out_dim1, False, device, dtype), 2: _Lin(out_dim1, out_dim1, False, device, dtype),
                               4: _Lin(out_dim1, out_dim1, False, device, dtype)})
        else:  # MLP / MLP_plus: one biased Linear (:64, :123)
            self.fc = _Lin(out_dim, out_dim1, True, device, dtype)

    @torch.no_grad()
    def forward(self, x):
        B, S, H = x.shape
        h = ops.ln_affine(x.to(torch.bfloat16).contiguous(), self.layernorm.weight, self.layernorm.bias, self.eps)
        for i in range(self.n_proj - 1):
            h = ops.gemm(h, self.projector[2 * i].weight, act=ACT_GELU_ERF, M=B * S)
        x2 = ops.gemm(h, self.projector[2 * (self.n_proj - 1)].weight, act=ACT_GELU_ERF, M=B * S)  # x2 = GELU(projector(x))
        if self.fc_chain:
            t = ops.gemm(x2, self.fc[0].weight, act=ACT_GELU_ERF, M=B * S)
            t = ops.gemm(t, self.fc[2].weight, act=ACT_GELU_ERF, M=B * S)
            x1_tok = ops.gemm(t, self.fc[4].weight, M=B * S, out_f32=True)
        else:
            x1_tok = ops.gemm(x2, self.fc.weight, self.fc.bias, M=B * S, out_f32=True)
        x1 = ops.seq_mean(x1_tok.view(B, S, -1))
        return x1.to(torch.bfloat16), x2.view(B, S, -1)


class MLP(_LegacyMLP):
    """model_internvl/proj.py:53-73"""


class MLP2(_LegacyMLP):
    """model_internvl/proj.py:76-102"""
    fc_chain = True


class MLP_plus(_LegacyMLP):
    """model_internvl/proj.py:104-130 (six projector linears, LayerNorm default eps)"""
    n_proj = 6

    def __init__(self, in_dim=4096, out_dim=4096, hidden_dim=4096, out_dim1=768, use_residual=True, device="cuda",
                 dtype=torch.bfloat16):
        super().__init__(in_dim, out_dim, hidden_dim, out_dim1, 1e-5, use_residual, device, dtype)


class ProjFrontStage(nn.Module):
    """The kernelised front stage of legacy Proj / Proj2 / Proj3 (model_internvl/proj.py:163-165): norm0 -> Conv2d(C->1,5x5) -> norm1,
    on its own (the full classes, with the T5Stack on `transformers`, are Proj / Proj2 / Proj3 below)."""

    def __init__(self, in_channels=2, input_dim=896, layer_norm_eps=1e-6, device="cuda", dtype=torch.bfloat16):
        super().__init__()
        self.eps = layer_norm_eps
        self.norm0 = _LN(input_dim, device, dtype)
        conv = nn.Module()
        conv.weight = _param(1, in_channels, 5, 5, device=device, dtype=dtype)
        conv.bias = _param(1, device=device, dtype=dtype)
        self.conv = conv
        self.norm1 = _LN(input_dim, device, dtype)

    @torch.no_grad()
    def forward(self, x):
        B, C, S, H = x.shape
        x = ops.ln_affine(x.to(torch.bfloat16).contiguous(), self.norm0.weight, self.norm0.bias, self.eps)
        x = _conv5x5(self, x)
        return ops.ln_affine(x, self.norm1.weight, self.norm1.bias, self.eps)


class _ProjT5(nn.Module):
    """Legacy Proj / Proj2 / Proj3 (model_internvl/proj.py:149-211): LayerNorm -> Conv2d(C->1, 5x5) -> LayerNorm front stage and an
    MLP / MLP2 head on the HIP path around a `transformers` T5Stack encoder (gated-gelu, relative attention bias), which stays on
    PyTorch-ROCm as SURVEY.md section 8 row A3 allows.  Same constructor arguments and state-dict keys (norm0.*, conv.*, norm1.*,
    t5stack.*, mlp.*) as the reference classes; no reference script imports them (checkpoint compatibility only)."""

    mlp_cls = None      # MLP for Proj, MLP2 for Proj2 / Proj3
    t5_first = False    # Proj3 runs the T5Stack on every layer's sequence BEFORE the layer-fusion stage (:203-210)

    def __init__(self, in_channels=2, kernel_size=5, input_dim=896, output_dim0=768, output_dim1=4096, num_layers=4, num_heads=12,
                 layer_norm_eps=1e-6, head_dim=64, device="cuda", dtype=torch.bfloat16):
        super().__init__()
        from transformers import T5Config
        from transformers.models.t5.modeling_t5 import T5Stack
        if kernel_size != 5:
            raise ValueError("the HIP layer-fusion kernel implements the reference's 5x5 convolution")
        config = T5Config(num_heads=num_heads, num_layers=num_layers, num_decoder_layers=0, layer_norm_epsilon=layer_norm_eps,
                          is_encoder_decoder=False, is_decoder=False, d_ff=input_dim * 4, d_kv=head_dim, d_model=input_dim,
                          dense_act_fn="gelu_new", feed_forward_proj="gated-gelu", use_cache=False)
        self.eps = layer_norm_eps
        self.norm0 = _LN(input_dim, device, dtype)
        conv = nn.Module()
        conv.weight = _param(1, in_channels, kernel_size, kernel_size, device=device, dtype=dtype)
        conv.bias = _param(1, device=device, dtype=dtype)
        self.conv = conv
        self.norm1 = _LN(input_dim, device, dtype)
        self.t5stack = T5Stack(config).to(device=device, dtype=dtype).eval().requires_grad_(False)
        self.mlp = self.mlp_cls(input_dim, output_dim1, output_dim1, output_dim0, layer_norm_eps, device=device, dtype=dtype)

    def _front(self, x):
        B, C, S, H = x.shape
        x = ops.ln_affine(x.to(torch.bfloat16).contiguous(), self.norm0.weight, self.norm0.bias, self.eps)          # :163 / :207
        x = _conv5x5(self, x)       # :164 / :208
        return ops.ln_affine(x, self.norm1.weight, self.norm1.bias, self.eps)                                       # :165 / :209

    @torch.no_grad()
    def forward(self, x):
        B, C, S, H = x.shape
        if self.t5_first:
            x = self.t5stack(inputs_embeds=x.to(torch.bfloat16).contiguous().view(B * C, S, H)).last_hidden_state   # :206
            x = self._front(x.view(B, C, S, H))
        else:
            x = self.t5stack(inputs_embeds=self._front(x)).last_hidden_state                                        # :166
        return self.mlp(x.contiguous())


class Proj(_ProjT5):
    """model_internvl/proj.py:149-167"""
    mlp_cls = MLP


class Proj2(_ProjT5):
    """model_internvl/proj.py:169-187"""
    mlp_cls = MLP2


class Proj3(_ProjT5):
    """model_internvl/proj.py:190-211"""
    mlp_cls = MLP2
    t5_first = True


class Transformer_proj(nn.Module):
    """model_internvl/proj.py:133-147: nn.TransformerEncoder (stays on PyTorch-ROCm, like the T5Stack above) followed by two
    biased linears on the HIP path; x1 = mean_S(linear1(x)), x2 = linear2(x).  Keys: transformer_encoder.*, linear1.*, linear2.*"""

    def __init__(self, d_model, n_heads, out_dim1, out_dim2, num_layers=3, device="cuda", dtype=torch.bfloat16):
        super().__init__()
        layer = nn.TransformerEncoderLayer(d_model=d_model, nhead=n_heads, dim_feedforward=2048, batch_first=True)
        self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=num_layers).to(device=device, dtype=dtype).eval()
        self.transformer_encoder.requires_grad_(False)
        self.linear1 = _Lin(d_model, out_dim1, True, device, dtype)
        self.linear2 = _Lin(d_model, out_dim2, True, device, dtype)

    @torch.no_grad()
    def forward(self, x):
        B, S, H = x.shape
        x = self.transformer_encoder(x.to(torch.bfloat16)).contiguous()
        x1_tok = ops.gemm(x, self.linear1.weight, self.linear1.bias, M=B * S, out_f32=True)
        x1 = ops.seq_mean(x1_tok.view(B, S, -1))
        x2 = ops.gemm(x, self.linear2.weight, self.linear2.bias, M=B * S)
        return x1.to(torch.bfloat16), x2.view(B, S, -1)
