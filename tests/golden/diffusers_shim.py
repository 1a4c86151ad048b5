"""A stand-in `diffusers` package, just big enough for the reference's
lightcontrol/lightcontrol_flux.py to import (its imports: lines 22-39).

Runs ONLY in the build container, from tests/golden/make_golden.py, to let the
reference's OWN block / model / ControlNeXt forward code produce composition
goldens.  Every primitive is a thin nn.Module whose parameters carry the
diffusers key names and whose forward calls oracle/primitives.py -- so the
fixtures pin composition (residual/gate order, cat order, control injection,
timestep scaling, slicing), not the primitives themselves (those stay
"parity unpinned", oracle/__init__.py).
"""
import sys
import types

import torch
import torch.nn as nn

from oracle import primitives as P


def _sd(mod, prefix="m"):
    return {prefix + "." + k: v for k, v in mod.state_dict().items()}


class _RMSNormW(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, added_kv_proj_dim=None, dim_head=64, heads=8,
                 out_dim=None, context_pre_only=None, bias=True, processor=None, qk_norm=None, eps=1e-5,
                 pre_only=False):
        super().__init__()
        inner = out_dim if out_dim is not None else dim_head * heads
        self.heads = heads
        self.eps = eps
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(query_dim, inner, bias=bias)
        self.to_v = nn.Linear(query_dim, inner, bias=bias)
        assert qk_norm == "rms_norm"
        self.norm_q = _RMSNormW(dim_head)
        self.norm_k = _RMSNormW(dim_head)
        if added_kv_proj_dim is not None:
            self.add_q_proj = nn.Linear(added_kv_proj_dim, inner, bias=bias)
            self.add_k_proj = nn.Linear(added_kv_proj_dim, inner, bias=bias)
            self.add_v_proj = nn.Linear(added_kv_proj_dim, inner, bias=bias)
            self.norm_added_q = _RMSNormW(dim_head)
            self.norm_added_k = _RMSNormW(dim_head)
            self.to_add_out = nn.Linear(inner, query_dim, bias=bias)
        if not pre_only:
            self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=bias), nn.Dropout(0.0)])
        self.processor = processor

    def get_processor(self):
        return self.processor

    def forward(self, hidden_states, encoder_hidden_states=None, image_rotary_emb=None, **kw):
        return P.flux_attention(_sd(self), "m", hidden_states, self.heads, image_rotary_emb,
                                encoder_hidden=encoder_hidden_states, eps=self.eps)


class FluxAttnProcessor2_0:
    pass


class FusedFluxAttnProcessor2_0:
    pass


AttentionProcessor = object


class _GELUProj(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.proj = nn.Linear(i, o)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, activation_fn="geglu", **kw):
        super().__init__()
        assert activation_fn == "gelu-approximate"
        self.net = nn.ModuleList([_GELUProj(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim_out or dim)])

    def forward(self, x):
        return P.feed_forward(_sd(self), "m", x)


class AdaLayerNormZero(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.linear = nn.Linear(dim, 6 * dim)

    def forward(self, x, emb=None):
        return P.ada_layer_norm_zero(_sd(self), "m", x, emb)


class AdaLayerNormZeroSingle(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.linear = nn.Linear(dim, 3 * dim)

    def forward(self, x, emb=None):
        return P.ada_layer_norm_zero_single(_sd(self), "m", x, emb)


class AdaLayerNormContinuous(nn.Module):
    def __init__(self, dim, cond_dim, elementwise_affine=True, eps=1e-5):
        super().__init__()
        assert not elementwise_affine and eps == 1e-6
        self.linear = nn.Linear(cond_dim, 2 * dim)

    def forward(self, x, cond):
        return P.ada_layer_norm_continuous(_sd(self), "m", x, cond)


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.args = (num_channels, flip_sin_to_cos, downscale_freq_shift)

    def forward(self, t):
        return P.timesteps_proj(t, *self.args)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, x):
        return P.timestep_embedding(_sd(self), "m", x)


class _TextProj(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.linear_1 = nn.Linear(i, o)
        self.linear_2 = nn.Linear(o, o)


class CombinedTimestepTextProjEmbeddings(nn.Module):
    guidance = False

    def __init__(self, embedding_dim, pooled_projection_dim):
        super().__init__()
        self.timestep_embedder = TimestepEmbedding(256, embedding_dim)
        if self.guidance:
            self.guidance_embedder = TimestepEmbedding(256, embedding_dim)
        self.text_embedder = _TextProj(pooled_projection_dim, embedding_dim)

    def forward(self, timestep, *rest):
        if self.guidance:
            guidance, pooled = rest
        else:
            (pooled,) = rest
            guidance = None
        return P.combined_time_text_embed(_sd(self), "m", timestep, pooled, guidance)


class CombinedTimestepGuidanceTextProjEmbeddings(CombinedTimestepTextProjEmbeddings):
    guidance = True


class FluxPosEmbed(nn.Module):
    def __init__(self, theta, axes_dim):
        super().__init__()
        self.theta, self.axes_dim = theta, tuple(axes_dim)

    def forward(self, ids):
        return P.flux_pos_embed(ids, self.axes_dim, self.theta)


class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, groups):
        super().__init__()
        self.groups = groups
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=1e-6)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=1e-6)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        if in_channels != out_channels:
            self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1)

    def forward(self, x, temb):
        return P.resnet_block2d(_sd(self), "m", x, temb, self.groups)


class Downsample2D(nn.Module):
    def __init__(self, channels, use_conv=False, out_channels=None, padding=1, name="conv"):
        super().__init__()
        assert use_conv and padding == 1
        self.conv = nn.Conv2d(channels, out_channels, 3, stride=2, padding=1)

    def forward(self, x, *args):
        return P.downsample2d(_sd(self), "m", x)


class _Config(dict):
    __getattr__ = dict.__getitem__


def register_to_config(init):
    import functools
    import inspect

    @functools.wraps(init)
    def wrapper(self, *args, **kwargs):
        sig = inspect.signature(init)
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self"}
        object.__setattr__(self, "_cfg", _Config(cfg))
        init(self, *args, **kwargs)

    return wrapper


class ConfigMixin:
    @property
    def config(self):
        return self._cfg


class ModelMixin(nn.Module):
    pass


class FromOriginalModelMixin:
    pass


class PeftAdapterMixin:
    pass


class BaseOutput(dict):
    pass


class Transformer2DModelOutput(BaseOutput):
    def __init__(self, sample=None):
        super().__init__(sample=sample)
        self.sample = sample


class _Logger:
    def warning(self, *a, **k):
        pass

    info = debug = warning


class _Logging:
    @staticmethod
    def get_logger(name):
        return _Logger()


def install():
    """Register the shim under the exact module names lightcontrol_flux.py:22-39 imports."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("diffusers")
    mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
    mod("diffusers.loaders", FromOriginalModelMixin=FromOriginalModelMixin, PeftAdapterMixin=PeftAdapterMixin)
    mod("diffusers.models")
    mod("diffusers.models.attention", FeedForward=FeedForward)
    mod("diffusers.models.attention_processor", Attention=Attention, AttentionProcessor=AttentionProcessor,
        FluxAttnProcessor2_0=FluxAttnProcessor2_0, FusedFluxAttnProcessor2_0=FusedFluxAttnProcessor2_0)
    mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    mod("diffusers.models.normalization", AdaLayerNormContinuous=AdaLayerNormContinuous,
        AdaLayerNormZero=AdaLayerNormZero, AdaLayerNormZeroSingle=AdaLayerNormZeroSingle)
    mod("diffusers.utils", USE_PEFT_BACKEND=False, is_torch_version=lambda *a: True, logging=_Logging,
        scale_lora_layers=lambda *a, **k: None, unscale_lora_layers=lambda *a, **k: None, BaseOutput=BaseOutput)
    mod("diffusers.utils.torch_utils", maybe_allow_in_graph=lambda c: c)
    mod("diffusers.models.embeddings",
        CombinedTimestepGuidanceTextProjEmbeddings=CombinedTimestepGuidanceTextProjEmbeddings,
        CombinedTimestepTextProjEmbeddings=CombinedTimestepTextProjEmbeddings, FluxPosEmbed=FluxPosEmbed,
        TimestepEmbedding=TimestepEmbedding, Timesteps=Timesteps)
    mod("diffusers.models.modeling_outputs", Transformer2DModelOutput=Transformer2DModelOutput)
    mod("diffusers.models.resnet", Downsample2D=Downsample2D, ResnetBlock2D=ResnetBlock2D)
