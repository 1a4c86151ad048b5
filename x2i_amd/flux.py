"""FLUX / shuttle-3 diffusion transformer on the HIP path.

Mirrors the reference interface (diffusers 0.31 `FluxTransformer2DModel` as called at
infer/inference_qwenvl.py:72-73,188-197 and train/train_qwenvl.py:578-587; in-repo composition
lightcontrol/lightcontrol_flux.py:208-553): same constructor arguments, same state-dict key names
(SURVEY.md Appendix B), same forward signature.  Everything numeric runs in libx2i_hip.so through x2i_amd.ops; this
module only owns device memory and sequences kernel launches on the current HIP stream.

Data layout in HBM (all bf16 unless noted)
  X    [B, S, D]        joint residual stream, text tokens FIRST (S = S_txt + S_img) -- the reference's
                        torch.cat([encoder_hidden_states, hidden_states], dim=1) (lightcontrol_flux.py:510) is free
  NRM  [B, S, D]        LayerNorm+modulate output (A operand of the next GEMM)
  QKV  [B*S, 3D]        fused q|k|v rows (double blocks: text rows then image rows)
  Q,K  [B, H, Spad, 128], VT [B, H, 128, Spad]   attention operands (Spad = ceil128(S), zero padded)
  CAT  [B*S, 5D]        single blocks: cols [0,D) attention out, [D,5D) GELU(proj_mlp) -- torch.cat(..., dim=2)
                        (lightcontrol_flux.py:97) is never materialised separately
  MOD  [B, Ntot] f32    every AdaLayerNorm* linear of the step in ONE skinny GEMM (temb is block-independent)
Weights are stored fused (q|k|v, q|k|v|proj_mlp, all AdaLN linears) and exposed under the reference's parameter
names as views, so load_state_dict() of a diffusers checkpoint fills the fused storage directly.
"""
import math
import os
from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .ops import ACT_GELU_TANH, ACT_NONE, ACT_SILU


class _P(nn.Module):
    """A bag of parameters with reference names (weight / bias); values are views into fused storage."""

    def __init__(self, **tensors):
        super().__init__()
        for k, v in tensors.items():
            self.register_parameter(k, nn.Parameter(v, requires_grad=False))


class _AttnTap(nn.Module):
    """The `attn` sub-module of a block: holds the reference's parameter names (to_q / to_k / ... are views into fused storage) and
    is CALLED with the block's attention outputs when forward hooks are registered on it -- the reference's distillation code hooks
    `block.attn` (train/train_qwenvl.py:206-214: double blocks return (image, text) projections, single blocks the joint sequence).
    Identity otherwise; the sampling path never calls it (the projections' outputs are never materialised there)."""

    def forward(self, *outs):
        return outs if len(outs) > 1 else outs[0]


class _Seq(nn.Module):
    """ModuleList-like container whose children are named by integers but may be sparse (e.g. net.0 / net.2)."""

    def __init__(self, items):
        super().__init__()
        for k, v in items.items():
            self.add_module(str(k), v)

    def __getitem__(self, i):
        return self._modules[str(i)]


class FluxTransformer2DModel(nn.Module):
    """Drop-in for diffusers' FluxTransformer2DModel (constructor: lightcontrol_flux.py:230-242)."""

    def __init__(self, patch_size: int = 1, in_channels: int = 64, num_layers: int = 19, num_single_layers: int = 38,
                 attention_head_dim: int = 128, num_attention_heads: int = 24, joint_attention_dim: int = 4096,
                 pooled_projection_dim: int = 768, guidance_embeds: bool = False, axes_dims_rope=(16, 56, 56),
                 device="cuda", dtype=torch.bfloat16):
        super().__init__()
        if attention_head_dim != 128:
            raise ValueError("x2i_amd: the HIP attention kernel is built for attention_head_dim == 128 (FLUX)")
        if dtype != torch.bfloat16:
            raise ValueError("x2i_amd: the HIP path computes in bf16 (fp32 statistics/accumulation)")
        if sum(axes_dims_rope) != attention_head_dim:
            raise ValueError("sum(axes_dims_rope) must equal attention_head_dim")
        self.config = _Config(patch_size=patch_size, in_channels=in_channels, num_layers=num_layers,
                              num_single_layers=num_single_layers, attention_head_dim=attention_head_dim,
                              num_attention_heads=num_attention_heads, joint_attention_dim=joint_attention_dim,
                              pooled_projection_dim=pooled_projection_dim, guidance_embeds=guidance_embeds,
                              axes_dims_rope=tuple(axes_dims_rope))
        self.out_channels = in_channels
        D = self.inner_dim = num_attention_heads * attention_head_dim
        H = num_attention_heads
        dev = torch.device(device)
        self._fused = {}
        self._views = []  # (param, fused name, index expression)
        self.rope_pair_tables = os.environ.get("X2I_ROPE_PAIRS", "1") != "0"   # fused QKV epilogues read the pair-form RoPE table (denoise)

        def store(name, *shape):
            t = torch.empty(shape, device=dev, dtype=dtype)
            self._fused[name] = t
            return t

        def view(name, sl):
            t = self._fused[name][sl]
            p = nn.Parameter(t, requires_grad=False)
            self._views.append((p, name, sl))
            return p

        def lin_from(wname, bname, r0, r1):
            m = nn.Module()
            m.register_parameter("weight", view(wname, slice(r0, r1)))
            m.register_parameter("bias", view(bname, slice(r0, r1)))
            return m

        def lin(prefix, out_f, in_f):
            store(prefix + ".w", out_f, in_f)
            store(prefix + ".b", out_f)
            return lin_from(prefix + ".w", prefix + ".b", 0, out_f)

        def norm_w(prefix):
            store(prefix, 128)
            m = nn.Module()
            m.register_parameter("weight", view(prefix, slice(0, 128)))
            return m

        # ---- embedders (lightcontrol_flux.py:247-257)
        self.x_embedder = lin("x_embedder", D, in_channels)
        self.context_embedder = lin("context_embedder", D, joint_attention_dim)
        tte = nn.Module()
        for nm, in_f in (("timestep_embedder", 256), ("guidance_embedder", 256), ("text_embedder", pooled_projection_dim)):
            if nm == "guidance_embedder" and not guidance_embeds:
                continue
            e = nn.Module()
            e.add_module("linear_1", lin(f"tte.{nm}.1", D, in_f))
            e.add_module("linear_2", lin(f"tte.{nm}.2", D, D))
            tte.add_module(nm, e)
        self.time_text_embed = tte

        # ---- one modulation table for every AdaLayerNorm linear (norm1, norm1_context, norm, norm_out)
        self._mod_rows = num_layers * 12 * D + num_single_layers * 3 * D + 2 * D
        store("mod.w", self._mod_rows, D)
        store("mod.b", self._mod_rows)

        def mod_lin(r0, n):
            m = nn.Module()
            m.add_module("linear", lin_from("mod.w", "mod.b", r0, r0 + n))
            return m

        # ---- double-stream blocks (lightcontrol_flux.py:108-157)
        blocks = []
        for i in range(num_layers):
            p = f"d{i}"
            blk = nn.Module()
            off = i * 12 * D
            blk.add_module("norm1", mod_lin(off, 6 * D))
            blk.add_module("norm1_context", mod_lin(off + 6 * D, 6 * D))
            store(p + ".qkv.w", 3 * D, D), store(p + ".qkv.b", 3 * D)
            store(p + ".cqkv.w", 3 * D, D), store(p + ".cqkv.b", 3 * D)
            attn = _AttnTap()
            for j, nm in enumerate(("to_q", "to_k", "to_v")):
                attn.add_module(nm, lin_from(p + ".qkv.w", p + ".qkv.b", j * D, (j + 1) * D))
            for j, nm in enumerate(("add_q_proj", "add_k_proj", "add_v_proj")):
                attn.add_module(nm, lin_from(p + ".cqkv.w", p + ".cqkv.b", j * D, (j + 1) * D))
            for nm in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
                attn.add_module(nm, norm_w(p + "." + nm))
            attn.add_module("to_out", _Seq({0: lin(p + ".to_out", D, D)}))
            attn.add_module("to_add_out", lin(p + ".to_add_out", D, D))
            blk.add_module("attn", attn)
            for ffn in ("ff", "ff_context"):
                g0 = nn.Module()
                g0.add_module("proj", lin(f"{p}.{ffn}.0", 4 * D, D))
                ff = nn.Module()
                ff.add_module("net", _Seq({0: g0, 2: lin(f"{p}.{ffn}.2", D, 4 * D)}))
                blk.add_module(ffn, ff)
            blocks.append(blk)
        self.transformer_blocks = nn.ModuleList(blocks)

        # ---- single-stream blocks (lightcontrol_flux.py:45-80): q|k|v|proj_mlp fused along N
        singles = []
        base = num_layers * 12 * D
        for i in range(num_single_layers):
            p = f"s{i}"
            blk = nn.Module()
            blk.add_module("norm", mod_lin(base + i * 3 * D, 3 * D))
            store(p + ".in.w", 7 * D, D), store(p + ".in.b", 7 * D)
            attn = _AttnTap()
            for j, nm in enumerate(("to_q", "to_k", "to_v")):
                attn.add_module(nm, lin_from(p + ".in.w", p + ".in.b", j * D, (j + 1) * D))
            for nm in ("norm_q", "norm_k"):
                attn.add_module(nm, norm_w(p + "." + nm))
            blk.add_module("attn", attn)
            blk.add_module("proj_mlp", lin_from(p + ".in.w", p + ".in.b", 3 * D, 7 * D))
            blk.add_module("proj_out", lin(p + ".proj_out", D, 5 * D))
            singles.append(blk)
        self.single_transformer_blocks = nn.ModuleList(singles)

        # ---- output head (lightcontrol_flux.py:281-282)
        self.norm_out = mod_lin(base + num_single_layers * 3 * D, 2 * D)
        self.proj_out = lin("proj_out", patch_size * patch_size * self.out_channels, D)
        self._ws = {}
        self._H = H
        # X2I_QKV_FUSE=0 (read once, here) keeps the two-step form GEMM -> x2i_qkv_split for A/B measurements; bit-identical results
        self.fuse_qkv = os.environ.get("X2I_QKV_FUSE", "1") != "0"
        self._fp8 = None  # see enable_fp8()
        self._fp8_mode = None

    # ------------------------------------------------------------------ fp8 (e4m3) configuration, opt-in
    @torch.no_grad()
    def enable_fp8(self, mode="mlp"):
        """Switch the large image-/joint-stream linears to the e4m3 MFMA path (BASELINE north_star "MFMA bf16/fp8 for the QKV/out-proj
        and MLP GEMMs"; bf16 stays the default).
          mode "mlp": double blocks' ff.net.0 / ff.net.2 on the image rows and the single blocks' proj_mlp / proj_out (72 % of the
                      GEMM FLOPs); QKV / attention-output projections stay bf16.
          mode "all": additionally the image-stream to_q|k|v (fused RMSNorm/RoPE epilogue, x2i_gemm_qkv_fp8), to_out / to_add_out
                      and the single blocks' to_q|k|v (97 % of the GEMM FLOPs).  Text-stream QKV / feed-forward and all small
                      linears stay bf16 in both modes.
        Weights are quantised ONCE here (per-output-channel scales, x2i_quantize_rows_fp8) from the bf16 parameters currently
        loaded -- call again after load_state_dict().  Activations: the preceding LayerNorm+modulate emits e4m3 rows with
        per-row scales; GELU outputs and attention outputs are written as e4m3 with a static scale of 1 (saturating at 448).
        mode None / "off" returns to bf16."""
        if mode in (None, "off", False):
            self._fp8 = None
            self._fp8_mode = None
            self._ws = {}
            return self
        if mode not in ("mlp", "all"):
            raise ValueError("enable_fp8: mode must be 'mlp', 'all' or None")
        if self.inner_dim % 128:
            raise ValueError("enable_fp8: inner_dim must be a multiple of 128 (K-tile of the e4m3 MFMA kernel)")
        if mode == "all" and (not self.fuse_qkv or self._H % 2):
            raise ValueError("enable_fp8('all'): needs the fused QKV epilogue and an even head count (256-column tiles)")
        D, f, q = self.inner_dim, self._fused, {}
        for i in range(self.config.num_layers):
            for nm in (f"d{i}.ff.0", f"d{i}.ff.2") + ((f"d{i}.qkv", f"d{i}.to_out", f"d{i}.to_add_out") if mode == "all" else ()):
                q[nm] = ops.quantize_rows_fp8(f[nm + ".w"])
        for i in range(self.config.num_single_layers):
            if mode == "all":
                q[f"s{i}.qkv"] = ops.quantize_rows_fp8(f[f"s{i}.in.w"][:3 * D])
            q[f"s{i}.mlp"] = ops.quantize_rows_fp8(f[f"s{i}.in.w"][3 * D:])
            q[f"s{i}.proj_out"] = ops.quantize_rows_fp8(f[f"s{i}.proj_out.w"])
        self._fp8 = q
        self._fp8_mode = mode
        self._ws = {}
        return self

    # ------------------------------------------------------------------ nn.Module plumbing
    @property
    def dtype(self):
        return torch.bfloat16

    @property
    def device(self):
        return self._fused["proj_out.w"].device

    def _apply(self, fn, recurse=True):
        # parameters are views of fused storage: move the storage, then re-point the views
        for k in list(self._fused):
            new = fn(self._fused[k])
            if new.dtype != torch.bfloat16:
                raise ValueError("x2i_amd FluxTransformer2DModel is bf16-only")
            self._fused[k] = new
        for p, name, sl in self._views:
            p.data = self._fused[name][sl]
        self._ws = {}
        if getattr(self, "_fp8", None) is not None:
            self.enable_fp8(self._fp8_mode)  # re-quantise on the new device
        return self

    @torch.no_grad()
    def init_random_(self, seed=0, std=0.02):
        """Random-init weights in place on the device (bench / smoke; there are no checkpoints offline)."""
        gen = torch.Generator(device=self.device).manual_seed(seed)
        for k, t in self._fused.items():
            if k.endswith("norm_q") or k.endswith("norm_k") or k.endswith("norm_added_q") or k.endswith("norm_added_k"):
                t.copy_(1.0 + 0.1 * torch.randn(t.shape, device=t.device, generator=gen))
            else:
                # fill in chunks to bound fp32 temporaries
                flat = t.view(-1)
                step = 1 << 26
                for s in range(0, flat.numel(), step):
                    n = min(step, flat.numel() - s)
                    flat[s:s + n].copy_(std * torch.randn(n, device=t.device, generator=gen))
        return self

    # ------------------------------------------------------------------ workspace
    def _workspace(self, B, St, Si):
        key = (B, St, Si)
        ws = self._ws.get(key)
        if ws is not None:
            return ws
        D, H = self.inner_dim, self._H
        S = St + Si
        Spad = ops.pad128(S)
        dev = self.device
        bf = dict(device=dev, dtype=torch.bfloat16)
        ws = dict(
            S=S, Spad=Spad,
            X=torch.empty((B, S, D), **bf), NRM=torch.empty((B, S, D), **bf), QKV=torch.empty((B * S, 3 * D), **bf),
            Q=torch.zeros((B, H, Spad, 128), **bf), K=torch.zeros((B, H, Spad, 128), **bf),
            VT=torch.zeros((B, H, 128, Spad), **bf), ATT=torch.empty((B, S, D), **bf),
            CAT=torch.empty((B * S, 5 * D), **bf), NRMF=torch.empty((B * Si, D), **bf),
            MOD=torch.empty((B, self._mod_rows), device=dev, dtype=torch.float32),
            TEMB=torch.empty((B, D), device=dev, dtype=torch.float32),
            H1=torch.empty((B, D), device=dev, dtype=torch.float32),
        )
        if self._fp8 is not None:
            f8 = dict(device=dev, dtype=ops.FP8)
            ws.update(NRM8=torch.empty((B, S, D), **f8), RS=torch.empty((B * S,), device=dev, dtype=torch.float32),
                      H8=torch.empty((B * Si, 4 * D), **f8), CAT8=torch.empty((B * S, 5 * D), **f8))
            if self._fp8_mode == "all":
                ws.update(ATT8=torch.empty((B, S, D), **f8))
        self._ws = {key: ws}  # keep one shape resident
        return ws

    # ------------------------------------------------------------------ forward pieces
    @torch.no_grad()
    def prepare_conditioning(self, encoder_hidden_states, pooled_projections, txt_ids, img_ids, guidance=None):
        """Step-invariant work hoisted out of the denoising loop: context_embedder, RoPE tables, and the
        text/guidance halves of the conditioning embedding ("step-invariant -> hoist")."""
        cfg = self.config
        D = self.inner_dim
        enc = encoder_hidden_states
        B, St, Kj = enc.shape
        if txt_ids.ndim == 3:
            txt_ids = txt_ids[0]
        if img_ids.ndim == 3:
            img_ids = img_ids[0]
        Si = img_ids.shape[0]
        ws = self._workspace(B, St, Si)
        S = ws["S"]
        f = self._fused
        enc = enc.to(device=self.device, dtype=torch.bfloat16).contiguous()
        # context_embedder -> text rows of X (kept in CTX so that X can be rebuilt every step)
        ctx = torch.empty((B, St, D), device=self.device, dtype=torch.bfloat16)
        ops.gemm(enc, f["context_embedder.w"], f["context_embedder.b"], out=ctx, M=B * St)
        # RoPE tables: FluxPosEmbed semantics (float64 frequencies), once per prompt
        ids = torch.cat((txt_ids.to(self.device), img_ids.to(self.device)), dim=0)
        cos, sin = ops.rope_table(ids, cfg.axes_dims_rope)  # FluxPosEmbed, once per prompt (x2i_rope_table_f32)
        pairs = ops.rope_pairs(cos, sin, check=False)                    # the same values as (cos, sin) per dim pair: what the fused QKV epilogues read
        # conditioning: text_embedder(pooled) [+ guidance_embedder(guidance*1000)]
        # per-state buffer (NOT the shared workspace): two prepared states of the same shape -- positive / negative prompts,
        # prepare A, prepare B, denoise A -- must not see each other's pooled / guidance conditioning
        cond = torch.empty((B, D), device=self.device, dtype=torch.float32)
        pooled = pooled_projections.to(device=self.device, dtype=torch.bfloat16).contiguous()
        h1 = ops.skinny_linear(pooled, f["tte.text_embedder.1.w"], f["tte.text_embedder.1.b"], out=ws["H1"], act_out=ACT_SILU)
        ops.skinny_linear(h1, f["tte.text_embedder.2.w"], f["tte.text_embedder.2.b"], out=cond)
        if cfg.guidance_embeds:
            if guidance is None:
                raise ValueError("guidance is required when config.guidance_embeds is True")
            # `guidance.to(hidden_states.dtype) * 1000` (lightcontrol_flux.py:449): rounding follows the caller's dtype
            g = (guidance.to(device=self.device, dtype=pooled_projections.dtype) * 1000).float().contiguous()
            gp = ops.timestep_sinusoid(g, 256, round_bf16=pooled_projections.dtype == torch.bfloat16)
            h1 = ops.skinny_linear(gp, f["tte.guidance_embedder.1.w"], f["tte.guidance_embedder.1.b"], out=ws["H1"], act_out=ACT_SILU)
            ops.skinny_linear(h1, f["tte.guidance_embedder.2.w"], f["tte.guidance_embedder.2.b"], out=cond, accumulate=True)
        return dict(B=B, St=St, Si=Si, ctx=ctx, cos=cos, sin=sin, rope_pairs=pairs, cond=cond, ws=ws,
                    round_bf16=pooled_projections.dtype == torch.bfloat16)

    MOD_GROUP = 4   # (t, sample) pairs per pass of the modulation table: x2i_skinny_linear stages its activations in LDS, and beyond four
                    # samples of K = 3072 only one workgroup fits a CU -- measured: eight pairs per pass run no faster than two passes of four

    @torch.no_grad()
    def prepare_modulation(self, state, timesteps, dtype=torch.bfloat16):
        """The AdaLN modulation tables of SEVERAL denoise calls: temb = timestep_embedder(Timesteps(t * 1000)) + cond and
        `mod = Linear(SiLU(temb))` for every t of `timesteps` (a list of [B] tensors, each what `denoise` would be given; `dtype` = the
        dtype of its hidden_states).  The table is a pass over 27 % of the model's weights (6.4 GB, 1.34 ms at any batch: 2 % of a
        batch-1 step) that depends on (t, pooled text, guidance) only: a sampler that knows its schedule evaluates MOD_GROUP // B steps
        per pass.  Returns one [B, mod_rows] f32 table per entry of `timesteps` for `denoise(mod=...)`, or None where a pass would hold
        a single step anyway (B > MOD_GROUP / 2: `denoise` then computes its own, into the workspace).  Per-sample results are those of
        the per-step evaluation (x2i_skinny_linear treats every sample alike; tests/test_fullsize_gpu.py)."""
        f = self._fused
        B, n = state["B"], len(timesteps)
        g = self.MOD_GROUP // B
        if g < 2:
            return [None] * n
        out = []
        for i0 in range(0, n, g):
            ts = timesteps[i0:i0 + g]
            t1000 = torch.cat([(t.to(device=self.device, dtype=dtype) * 1000).float().reshape(-1).expand(B) for t in ts]).contiguous()
            tp = ops.timestep_sinusoid(t1000, 256, round_bf16=state["round_bf16"])
            h1 = ops.skinny_linear(tp, f["tte.timestep_embedder.1.w"], f["tte.timestep_embedder.1.b"], act_out=ACT_SILU)
            temb = state["cond"].repeat(len(ts), 1)
            ops.skinny_linear(h1, f["tte.timestep_embedder.2.w"], f["tte.timestep_embedder.2.b"], out=temb, accumulate=True)
            table = ops.skinny_linear(temb, f["mod.w"], f["mod.b"], act_in=ACT_SILU)
            out += [table[k * B:(k + 1) * B] for k in range(len(ts))]
        return out

    @torch.no_grad()
    def denoise(self, state, hidden_states, timestep, control=None, mod=None):
        """One transformer evaluation given prepared conditioning.  `control`: optional callable
        (i, timestep_x1000, X, St, S, D) that adds control net i's output into the image rows of X after double block i.
        `mod`: this call's modulation table from prepare_modulation (None: computed here)."""
        cfg = self.config
        f = self._fused
        ws = state["ws"]
        B, St, Si = state["B"], state["St"], state["Si"]
        S, Spad = ws["S"], ws["Spad"]
        D, H = self.inner_dim, self._H
        X, NRM, QKV, Q, K, VT, ATT, CAT, MOD = (ws[k] for k in ("X", "NRM", "QKV", "Q", "K", "VT", "ATT", "CAT", "MOD"))
        Ntot = self._mod_rows
        hs = hidden_states.to(device=self.device, dtype=torch.bfloat16).contiguous()
        # ---- embed: text rows <- cached context embedding, image rows <- x_embedder(latents)
        X[:, :St].copy_(state["ctx"])
        ops.gemm(hs, f["x_embedder.w"], f["x_embedder.b"], out=X, M=Si, batch=B, a_batch_stride=Si * cfg.in_channels,
                 lda=cfg.in_channels, c_batch_stride=S * D, ldc=D, c_offset=St * D)
        # ---- temb = timestep_embedder(Timesteps(t*1000)) + cond   (lightcontrol_flux.py:447,452-456)
        # `timestep.to(hidden_states.dtype) * 1000` -- in the reference's bf16 run this multiply rounds to bf16
        # (750 -> 752); we follow the dtype the caller hands us, exactly as the reference module does.
        t1000 = None
        if mod is None or control is not None:
            t1000 = (timestep.to(device=self.device, dtype=hidden_states.dtype) * 1000).float().contiguous()
        if mod is not None:
            MOD = mod   # (prepare_modulation: the same values, computed with the other steps' tables)
        else:
            tp = ops.timestep_sinusoid(t1000, 256, round_bf16=state["round_bf16"])
            h1 = ops.skinny_linear(tp, f["tte.timestep_embedder.1.w"], f["tte.timestep_embedder.1.b"], out=ws["H1"], act_out=ACT_SILU)
            temb = ws["TEMB"]
            temb.copy_(state["cond"])
            ops.skinny_linear(h1, f["tte.timestep_embedder.2.w"], f["tte.timestep_embedder.2.b"], out=temb, accumulate=True)
            # ---- every AdaLN modulation vector of this step in one HBM-bound pass over 27% of the weights
            ops.skinny_linear(temb, f["mod.w"], f["mod.b"], out=MOD, act_in=ACT_SILU)
        cos, sin = state["cos"], state["sin"]
        # what the FUSED QKV epilogues read: the pair-form table (half the bytes per token, fetched two half chunks ahead in the persistent
        # kernel; X2I_ROPE_PAIRS=0: the separate tables, A/B -- same values, bit-identical results); x2i_qkv_split_bf16 keeps cos / sin
        rc, rs = (state["rope_pairs"], None) if self.rope_pair_tables else (cos, sin)
        scale = 1.0 / math.sqrt(128.0)

        def mod(off):
            return MOD[:, off:]

        fuse_qkv = self.fuse_qkv and D % 64 == 0
        fp8 = self._fp8
        fp8_all = fp8 is not None and self._fp8_mode == "all"
        # with the fused QKV epilogues, softmax_scale * log2(e) rides in Q (one f32 multiply in front of the epilogue's bf16 rounding,
        # x2i_qkv_desc.q_scale) and the attention kernels get scale = ln 2: their inner loops lose the score multiply
        qs = scale * 1.4426950408889634 if (fuse_qkv or fp8_all) else 1.0
        scale = scale / qs if qs != 1.0 else scale
        # the 16 x 16 x 32 attention kernel reads V^T with the keys of every 32-key span permuted (x2i_qkv_desc.vt_perm): the fused QKV
        # epilogues write it that way where that kernel serves the call (decided by H, S and the scale only -- never by the batch)
        vp_d = fuse_qkv and not fp8_all and ops.attention_prefers_vt_perm(H, S, scale)   # double blocks: bf16 attention unless fp8 "all"
        vp_s = fuse_qkv and fp8 is None and ops.attention_prefers_vt_perm(H, S, scale)   # single blocks: bf16 attention without fp8
        # forward hooks on block.attn (attention-distillation capture): materialise the attention outputs of every block
        taps = any(len(b.attn._forward_hooks) for b in self.transformer_blocks) or \
            any(len(b.attn._forward_hooks) for b in self.single_transformer_blocks)
        if taps and fp8 is not None:
            raise RuntimeError("attention taps (forward hooks on block.attn) are served by the bf16 configuration only")
        qkv_txt = QKV  # rows [0, B*St)
        qkv_img_off = B * St * 3 * D
        # ---- double-stream blocks (lightcontrol_flux.py:159-204)
        for i in range(cfg.num_layers):
            p = f"d{i}"
            oi = i * 12 * D
            oc = oi + 6 * D
            if fp8_all:
                # text rows: bf16 norm for the bf16 add_q|k|v projection; image rows: e4m3 rows + per-row scales
                if St > 0:
                    ops.ln_modulate(X, NRM, B, St, D, St, mod(oc), mod(oc + D), mod(oc), mod(oc + D), Ntot, x_bs=S * D, y_bs=S * D)
                ops.ln_modulate_fp8(X, None, ws["NRM8"], ws["RS"], B, Si, D, 0, None, None, mod(oi), mod(oi + D), Ntot,
                                    x_bs=S * D, x_offset=St * D, y8_bs=S * D, y8_offset=St * D)
                wq, sq = fp8[p + ".qkv"]
                ops.gemm_qkv_fp8(ws["NRM8"], wq, f[p + ".qkv.b"], Q, K, VT, f[p + ".norm_q"], f[p + ".norm_k"], rc, rs, M=Si, H=H,
                                 Spad=Spad, tok_off=St, rows_per_sample=Si, batch=B, a_batch_stride=S * D, lda=D, a_offset=St * D,
                                 a_scale=ws["RS"], a_scale_batch_stride=Si, w_scale=sq, q_scale=qs)
                ops.gemm_qkv(NRM, f[p + ".cqkv.w"], f[p + ".cqkv.b"], Q, K, VT, f[p + ".norm_added_q"], f[p + ".norm_added_k"],
                             rc, rs, M=St, H=H, Spad=Spad, tok_off=0, rows_per_sample=St, batch=B, a_batch_stride=S * D, lda=D, q_scale=qs)
            else:
                ops.ln_modulate(X, NRM, B, S, D, St, mod(oc), mod(oc + D), mod(oi), mod(oi + D), Ntot)
            if fp8_all:
                pass
            elif fuse_qkv:
                # q/k RMSNorm + RoPE + head split + V transpose ride in the QKV GEMM's epilogue (no [B*S, 3D] round trip)
                # (image rows and text rows: two problems, ONE grouped launch -- the text tiles ride in the image launch's rounds)
                g_img = dict(A=NRM, W=f[p + ".qkv.w"], bias=f[p + ".qkv.b"], Q=Q, K=K, VT=VT, norm_q=f[p + ".norm_q"], norm_k=f[p + ".norm_k"],
                             cos=rc, sin=rs, M=Si, H=H, Spad=Spad, tok_off=St, rows_per_sample=Si, batch=B, a_batch_stride=S * D, lda=D,
                             a_offset=St * D, q_scale=qs, vt_perm=vp_d)
                g_txt = dict(A=NRM, W=f[p + ".cqkv.w"], bias=f[p + ".cqkv.b"], Q=Q, K=K, VT=VT, norm_q=f[p + ".norm_added_q"],
                             norm_k=f[p + ".norm_added_k"], cos=rc, sin=rs, M=St, H=H, Spad=Spad, tok_off=0, rows_per_sample=St, batch=B,
                             a_batch_stride=S * D, lda=D, q_scale=qs, vt_perm=vp_d)
                if St > 0:
                    ops.gemm_qkv_pair(g_img, g_txt)
                else:
                    ops.gemm_qkv(**g_img)
            else:
                ops.gemm(NRM, f[p + ".qkv.w"], f[p + ".qkv.b"], out=QKV, M=Si, batch=B, a_batch_stride=S * D, lda=D,
                         a_offset=St * D, c_batch_stride=Si * 3 * D, ldc=3 * D, c_offset=qkv_img_off)
                ops.gemm(NRM, f[p + ".cqkv.w"], f[p + ".cqkv.b"], out=QKV, M=St, batch=B, a_batch_stride=S * D, lda=D,
                         c_batch_stride=St * 3 * D, ldc=3 * D)
                ops.qkv_split(qkv_txt, QKV.view(-1)[qkv_img_off:], 3 * D, 3 * D, B, S, St, H, f[p + ".norm_added_q"],
                              f[p + ".norm_added_k"], f[p + ".norm_q"], f[p + ".norm_k"], cos, sin, Q, K, VT, Spad)
            if fp8_all:
                ops.attention_e4m3out(Q, K, VT, ws["ATT8"], B, H, S, Spad, D, S * D, scale)
                (wo, so), (wa, sa) = fp8[p + ".to_out"], fp8[p + ".to_add_out"]
                ops.gemm_fp8(ws["ATT8"], wo, f[p + ".to_out.b"], out=X, M=Si, batch=B, a_batch_stride=S * D, lda=D, a_offset=St * D,
                             w_scale=so, c_batch_stride=S * D, ldc=D, c_offset=St * D, res=X, res_batch_stride=S * D, ldr=D,
                             res_offset=St * D, gate=mod(oi + 2 * D), gate_batch_stride=Ntot)
                ops.gemm_fp8(ws["ATT8"], wa, f[p + ".to_add_out.b"], out=X, M=St, batch=B, a_batch_stride=S * D, lda=D, w_scale=sa,
                             c_batch_stride=S * D, ldc=D, res=X, res_batch_stride=S * D, ldr=D, gate=mod(oc + 2 * D),
                             gate_batch_stride=Ntot)
            else:
                ops.attention(Q, K, VT, ATT, B, H, S, Spad, D, S * D, scale, vt_perm=vp_d)
            # hidden += gate_msa * to_out(attn_img) ; enc += c_gate_msa * to_add_out(attn_txt)
            if fp8_all:
                pass
            elif not taps:
                g_img = dict(A=ATT, W=f[p + ".to_out.w"], bias=f[p + ".to_out.b"], out=X, M=Si, batch=B, a_batch_stride=S * D, lda=D,
                             a_offset=St * D, c_batch_stride=S * D, ldc=D, c_offset=St * D, res=X, res_batch_stride=S * D, ldr=D,
                             res_offset=St * D, gate=mod(oi + 2 * D), gate_batch_stride=Ntot)
                g_txt = dict(A=ATT, W=f[p + ".to_add_out.w"], bias=f[p + ".to_add_out.b"], out=X, M=St, batch=B, a_batch_stride=S * D, lda=D,
                             c_batch_stride=S * D, ldc=D, res=X, res_batch_stride=S * D, ldr=D, gate=mod(oc + 2 * D),
                             gate_batch_stride=Ntot)
                if St > 0:
                    ops.gemm_pair(g_img, g_txt)
                else:
                    ops.gemm(**g_img)
            else:
                # the module's outputs as the reference's Attention returns them (image, text), fresh tensors per block for the hooks
                t_img = torch.empty((B, Si, D), device=self.device, dtype=torch.bfloat16)
                t_txt = torch.empty((B, St, D), device=self.device, dtype=torch.bfloat16)
                ops.gemm(ATT, f[p + ".to_out.w"], f[p + ".to_out.b"], out=t_img, M=Si, batch=B, a_batch_stride=S * D, lda=D,
                         a_offset=St * D, c_batch_stride=Si * D, ldc=D)
                ops.gemm(ATT, f[p + ".to_add_out.w"], f[p + ".to_add_out.b"], out=t_txt, M=St, batch=B, a_batch_stride=S * D, lda=D,
                         c_batch_stride=St * D, ldc=D)
                self.transformer_blocks[i].attn(t_img, t_txt)
                ops.gated_residual_(X, t_img, mod(oi + 2 * D), B, Si, D, S * D, D, Si * D, D, Ntot, x_offset=St * D)
                ops.gated_residual_(X, t_txt, mod(oc + 2 * D), B, St, D, S * D, D, St * D, D, Ntot)
            # feed-forward of both streams
            ffh_img_off = B * St * 4 * D
            if fp8 is None:
                ops.ln_modulate(X, NRM, B, S, D, St, mod(oc + 3 * D), mod(oc + 4 * D), mod(oi + 3 * D), mod(oi + 4 * D), Ntot)
                # ff and ff_context: hidden layers of both streams, then both output layers, each pair one grouped launch
                h_img = dict(A=NRM, W=f[p + ".ff.0.w"], bias=f[p + ".ff.0.b"], out=CAT, M=Si, batch=B, a_batch_stride=S * D, lda=D,
                             a_offset=St * D, c_batch_stride=Si * 4 * D, ldc=4 * D, c_offset=ffh_img_off, act=ACT_GELU_TANH)
                o_img = dict(A=CAT, W=f[p + ".ff.2.w"], bias=f[p + ".ff.2.b"], out=X, M=Si, batch=B, a_batch_stride=Si * 4 * D, lda=4 * D,
                             a_offset=ffh_img_off, c_batch_stride=S * D, ldc=D, c_offset=St * D, res=X, res_batch_stride=S * D,
                             ldr=D, res_offset=St * D, gate=mod(oi + 5 * D), gate_batch_stride=Ntot)
                if St > 0:
                    ops.gemm_pair(h_img, dict(A=NRM, W=f[p + ".ff_context.0.w"], bias=f[p + ".ff_context.0.b"], out=CAT, M=St, batch=B,
                                              a_batch_stride=S * D, lda=D, c_batch_stride=St * 4 * D, ldc=4 * D, act=ACT_GELU_TANH))
                    ops.gemm_pair(o_img, dict(A=CAT, W=f[p + ".ff_context.2.w"], bias=f[p + ".ff_context.2.b"], out=X, M=St, batch=B,
                                              a_batch_stride=St * 4 * D, lda=4 * D, c_batch_stride=S * D, ldc=D, res=X, res_batch_stride=S * D,
                                              ldr=D, gate=mod(oc + 5 * D), gate_batch_stride=Ntot))
                else:
                    ops.gemm(**h_img)
                    ops.gemm(**o_img)
            else:
                # text rows: bf16 norm for the bf16 ff_context; image rows: the norm emits the e4m3 operand + per-row scales
                if St > 0:
                    ops.ln_modulate(X, NRM, B, St, D, St, mod(oc + 3 * D), mod(oc + 4 * D), mod(oc + 3 * D), mod(oc + 4 * D), Ntot,
                                    x_bs=S * D, y_bs=S * D)
                ops.ln_modulate_fp8(X, None, ws["NRM8"], ws["RS"], B, Si, D, 0, None, None, mod(oi + 3 * D), mod(oi + 4 * D), Ntot,
                                    x_bs=S * D, x_offset=St * D, y8_bs=S * D, y8_offset=St * D)
                (w0, s0), (w2, s2) = fp8[p + ".ff.0"], fp8[p + ".ff.2"]
                ops.gemm_fp8(ws["NRM8"], w0, f[p + ".ff.0.b"], out=ws["H8"], M=Si, batch=B, a_batch_stride=S * D, lda=D, a_offset=St * D,
                             a_scale=ws["RS"], a_scale_batch_stride=Si, w_scale=s0, c_batch_stride=Si * 4 * D, ldc=4 * D,
                             act=ACT_GELU_TANH, out_fp8=True)
                ops.gemm_fp8(ws["H8"], w2, f[p + ".ff.2.b"], out=X, M=Si, batch=B, a_batch_stride=Si * 4 * D, lda=4 * D, w_scale=s2,
                             c_batch_stride=S * D, ldc=D, c_offset=St * D, res=X, res_batch_stride=S * D, ldr=D, res_offset=St * D,
                             gate=mod(oi + 5 * D), gate_batch_stride=Ntot)
            if fp8 is not None and St > 0:
                ops.gemm(NRM, f[p + ".ff_context.0.w"], f[p + ".ff_context.0.b"], out=CAT, M=St, batch=B, a_batch_stride=S * D,
                         lda=D, c_batch_stride=St * 4 * D, ldc=4 * D, act=ACT_GELU_TANH)
                ops.gemm(CAT, f[p + ".ff_context.2.w"], f[p + ".ff_context.2.b"], out=X, M=St, batch=B,
                         a_batch_stride=St * 4 * D, lda=4 * D, c_batch_stride=S * D, ldc=D, res=X, res_batch_stride=S * D, ldr=D,
                         gate=mod(oc + 5 * D), gate_batch_stride=Ntot)
            if control is not None:
                # hidden_states += control_nets[i](guided_hint, timestep)['out'] * 1.0 (lightcontrol_flux.py:504-507),
                # fused into the last ControlNeXt conv's epilogue (residual add into the image rows of X)
                control(i, t1000, X, St, S, D)
        # ---- single-stream blocks on the joint sequence (lightcontrol_flux.py:82-104)
        base = cfg.num_layers * 12 * D
        for i in range(cfg.num_single_layers):
            p = f"s{i}"
            o = base + i * 3 * D
            if fp8 is None:
                ops.ln_modulate(X, NRM, B, S, D, 0, None, None, mod(o), mod(o + D), Ntot)
            else:
                ops.ln_modulate_fp8(X, None if fp8_all else NRM, ws["NRM8"], ws["RS"], B, S, D, 0, None, None, mod(o), mod(o + D), Ntot)
            w, bias = f[p + ".in.w"], f[p + ".in.b"]
            if fp8_all:
                wq, sq = fp8[p + ".qkv"]
                ops.gemm_qkv_fp8(ws["NRM8"], wq, bias[:3 * D], Q, K, VT, f[p + ".norm_q"], f[p + ".norm_k"], rc, rs, M=B * S, H=H,
                                 Spad=Spad, tok_off=0, rows_per_sample=S, a_scale=ws["RS"], w_scale=sq, q_scale=qs)
            elif fuse_qkv:
                ops.gemm_qkv(NRM, w[:3 * D], bias[:3 * D], Q, K, VT, f[p + ".norm_q"], f[p + ".norm_k"], rc, rs, M=B * S, H=H,
                             Spad=Spad, tok_off=0, rows_per_sample=S, q_scale=qs, vt_perm=vp_s)
            else:
                ops.gemm(NRM, w, bias, out=QKV, M=B * S, N=3 * D)
            if fp8 is None:
                ops.gemm(NRM, w[3 * D:], bias[3 * D:], out=CAT, M=B * S, N=4 * D, ldc=5 * D, c_offset=D, act=ACT_GELU_TANH)
            else:
                wm, sm = fp8[p + ".mlp"]
                ops.gemm_fp8(ws["NRM8"], wm, bias[3 * D:], out=ws["CAT8"], M=B * S, a_scale=ws["RS"], w_scale=sm, ldc=5 * D, c_offset=D,
                             act=ACT_GELU_TANH, out_fp8=True)
            if not fuse_qkv:
                ops.qkv_split(None, QKV, 3 * D, 3 * D, B, S, 0, H, None, None, f[p + ".norm_q"], f[p + ".norm_k"], cos, sin,
                              Q, K, VT, Spad)
            if fp8 is None:
                ops.attention(Q, K, VT, CAT, B, H, S, Spad, 5 * D, S * 5 * D, scale, vt_perm=vp_s)
                if taps:  # single blocks' Attention (pre_only) returns the un-projected joint sequence [B, S, D]
                    self.single_transformer_blocks[i].attn(CAT.view(B, S, 5 * D)[:, :, :D].clone())
                ops.gemm(CAT, f[p + ".proj_out.w"], f[p + ".proj_out.b"], out=X, M=S, batch=B, a_batch_stride=S * 5 * D,
                         lda=5 * D, c_batch_stride=S * D, ldc=D, res=X, res_batch_stride=S * D, ldr=D, gate=mod(o + 2 * D),
                         gate_batch_stride=Ntot)
            else:
                ops.attention_e4m3out(Q, K, VT, ws["CAT8"], B, H, S, Spad, 5 * D, S * 5 * D, scale)
                wo, so = fp8[p + ".proj_out"]
                ops.gemm_fp8(ws["CAT8"], wo, f[p + ".proj_out.b"], out=X, M=S, batch=B, a_batch_stride=S * 5 * D, lda=5 * D, w_scale=so,
                             c_batch_stride=S * D, ldc=D, res=X, res_batch_stride=S * D, ldr=D, gate=mod(o + 2 * D),
                             gate_batch_stride=Ntot)
        # ---- norm_out (AdaLayerNormContinuous: scale first, then shift) + proj_out on the image tokens (:540-543)
        o = base + cfg.num_single_layers * 3 * D
        NRMF = ws["NRMF"]
        ops.ln_modulate(X, NRMF, B, Si, D, 0, None, None, mod(o + D), mod(o), Ntot, x_bs=S * D, ldx=D, y_bs=Si * D, ldy=D,
                        x_offset=St * D)
        out = torch.empty((B, Si, self.out_channels * cfg.patch_size ** 2), device=self.device, dtype=torch.bfloat16)
        ops.gemm(NRMF, f["proj_out.w"], f["proj_out.b"], out=out, M=B * Si)
        return out

    def forward(self, hidden_states, encoder_hidden_states=None, pooled_projections=None, timestep=None, img_ids=None,
                txt_ids=None, guidance=None, joint_attention_kwargs=None, return_dict: bool = True):
        """diffusers signature; returns (sample,) when return_dict=False (train/train_qwenvl.py:578-587 takes [0])."""
        state = self.prepare_conditioning(encoder_hidden_states, pooled_projections, txt_ids, img_ids, guidance)
        out = self.denoise(state, hidden_states, timestep)
        if not return_dict:
            return (out,)
        return Transformer2DModelOutput(sample=out)

    # ------------------------------------------------------------------ loading
    @classmethod
    def from_config(cls, config, **kw):
        keys = ("patch_size", "in_channels", "num_layers", "num_single_layers", "attention_head_dim",
                "num_attention_heads", "joint_attention_dim", "pooled_projection_dim", "guidance_embeds", "axes_dims_rope")
        return cls(**{k: config[k] for k in keys if k in config}, **kw)

    @classmethod
    def from_pretrained(cls, path, subfolder: Optional[str] = None, torch_dtype=torch.bfloat16, device=None, **kw):
        """Load a diffusers-format FLUX transformer directory (config.json + *.safetensors shards)."""
        import glob
        import json
        import os

        from safetensors import safe_open

        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        d = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(d, "config.json")) as fh:
            cfg = json.load(fh)
        model = cls.from_config(cfg, device=device, dtype=torch_dtype)
        own = dict(model.named_parameters())
        seen = set()
        for shard in sorted(glob.glob(os.path.join(d, "*.safetensors"))):
            with safe_open(shard, framework="pt", device="cpu") as sf:
                for k in sf.keys():
                    if k not in own:
                        raise KeyError(f"unexpected key {k} in {shard}")
                    own[k].copy_(sf.get_tensor(k))  # (not .data.copy_: keeps the version counter honest)
                    seen.add(k)
        missing = set(own) - seen
        if missing:
            raise KeyError(f"missing keys in checkpoint: {sorted(missing)[:8]} ...")
        return model


class _Config(dict):
    """config object with attribute access (`transformer.config.in_channels`, `.guidance_embeds`)."""

    __getattr__ = dict.__getitem__


class Transformer2DModelOutput:
    def __init__(self, sample):
        self.sample = sample
