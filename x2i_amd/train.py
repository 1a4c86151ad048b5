"""Attention-distillation training step on the HIP path (reference train/train_qwenvl.py:556-654).

The reference trains ONLY the projector: `loss.backward()` runs back through the frozen student transformer into
`prompt_embeds_zh` / `add_text_embeds` (train/train_qwenvl.py:577-590,626).  So the transformer's backward is an activation-gradient
chain -- no weight gradients -- and that is what `DistillBackward` implements:

  forward_train()   the transformer forward with every activation the chain needs kept (block inputs, pre-norm q|k|v rows, attention
                    projections, pre-GELU rows, feed-forward outputs); at every `block.attn` tap the distillation loss of that tap
                    (x2i_kd_loss_bf16) is evaluated against the teacher's tensor and its gradient stored -- the taps themselves are
                    not kept;
  backward()        57 blocks in reverse: gated-residual, GELU, LayerNorm+modulate, q/k RMSNorm + RoPE and attention backward, every
                    matrix product a launch of the forward MFMA GEMM on a transposed operand.  Attention backward recomputes
                    P = softmax(QK^T) per sample with explicit [H, S, S] matrices (5 GEMMs + 2 row kernels + 3 transposes) -- correct and
                    deterministic, HBM-bound rather than flash-style (the fused form is the next optimisation);
                    returns d loss / d encoder_hidden_states and d loss / d pooled_projections.

`ProjectorTrainer` (below) is the other half: the projector's forward with saves, its backward (weight gradients), gradient
clipping and AdamW, and `distill_step` strings the reference's step together.  Everything numeric runs in the C-ABI library; torch
is storage and launch order.
"""
import math

import torch

from . import ops
from .ops import ACT_GELU_ERF, ACT_GELU_TANH, ACT_NONE, ACT_SILU


def _gcd_rows(*ns):
    """rows per wave of the two-stage column-sum kernels: a divisor of every segment length, small enough that B * S / R waves fill the chip"""
    g = 8
    for n in ns:
        if n:
            g = math.gcd(g, n)
    return max(g, 1)


class DistillBackward:
    """Activation-gradient chain of a frozen x2i_amd FluxTransformer2DModel (see the module docstring)."""

    def __init__(self, model, fused_attention_backward=True):
        self.m = model
        self.WT = {}
        self._pool = {}
        self.saved = None
        # True: x2i_attention_bwd_bf16 (flash-style, no [S, S] matrices); False: the explicit-matrix form built from GEMM launches and
        # row kernels (kept as the A/B reference: both are tested against autograd)
        self.fused_attention_backward = fused_attention_backward

    # ------------------------------------------------------------------ buffers: allocated once per shape, reused by every step
    def _buf(self, name, shape, dtype=torch.bfloat16, zero=False):
        """Named device buffer (saved activation, gradient, scratch).  Reusing them across steps keeps the caching allocator out of
        the step: at B = 4 the kept activations are ~140 GiB and re-allocating them every step cost more than the kernels."""
        key = (name, tuple(shape), dtype)
        t = self._pool.get(key)
        if t is None:
            t = torch.empty(tuple(shape), device=self.m.device, dtype=dtype)
            self._pool[key] = t
        if zero:
            t.zero_()
        return t

    def _keep(self, name, src):
        t = self._buf(name, src.shape, src.dtype)
        t.copy_(src)
        return t

    # ------------------------------------------------------------------ frozen weights, transposed once
    @torch.no_grad()
    def _wt(self, key):
        """W^T of fused weight `key`: the B operand of dX = dY W as the GEMM wants it ([K_in, N_out]); made once (the weights are frozen)."""
        t = self.WT.get(key)
        if t is None:
            t = ops.transpose(self.m._fused[key].contiguous())
            self.WT[key] = t
        return t

    # ------------------------------------------------------------------ forward with saves
    @torch.no_grad()
    def forward_train(self, state, hidden_states, timestep, teacher=None, tap_grads=None, temperature=3.0):
        """One transformer evaluation (prepared conditioning `state` as in FluxTransformer2DModel.denoise) that keeps what backward()
        needs.  Exactly one of
          teacher   = three stacked tensors / lists [(B, 19, Si, D), (B, 19, St, D), (B, 38, S, D)] as the reference's batch holds them
                      (KD_teacher_tensor0/1/2, train/train_qwenvl.py:570-572): the loss is evaluated tap by tap, or
          tap_grads = three lists of explicit gradients with the taps' shapes (tests, other losses)
        may be given.  Returns (noise_pred, loss) with loss a device f32 scalar (0 when tap_grads is used)."""
        m = self.m
        cfg, f, ws = m.config, m._fused, state["ws"]
        B, St, Si = state["B"], state["St"], state["Si"]
        S, Spad = ws["S"], ws["Spad"]
        D, H = m.inner_dim, m._H
        if S % 8:
            raise ValueError("forward_train: the joint sequence length must be a multiple of 8")
        dev = m.device
        X, NRM, QKV, Q, K, VT, ATT, CAT, MOD = (ws[k] for k in ("X", "NRM", "QKV", "Q", "K", "VT", "ATT", "CAT", "MOD"))
        Ntot = m._mod_rows
        bf = dict(device=dev, dtype=torch.bfloat16)
        hs = hidden_states.to(device=dev, dtype=torch.bfloat16).contiguous()
        X[:, :St].copy_(state["ctx"])
        ops.gemm(hs, f["x_embedder.w"], f["x_embedder.b"], out=X, M=Si, batch=B, a_batch_stride=Si * cfg.in_channels, lda=cfg.in_channels,
                 c_batch_stride=S * D, ldc=D, c_offset=St * D)
        t1000 = (timestep.to(device=dev, dtype=hidden_states.dtype) * 1000).float().contiguous()
        tp = ops.timestep_sinusoid(t1000, 256, round_bf16=state["round_bf16"])
        h1 = ops.skinny_linear(tp, f["tte.timestep_embedder.1.w"], f["tte.timestep_embedder.1.b"], out=ws["H1"], act_out=ACT_SILU)
        temb = self._buf("temb", (B, D), torch.float32)
        temb.copy_(state["cond"])
        ops.skinny_linear(h1, f["tte.timestep_embedder.2.w"], f["tte.timestep_embedder.2.b"], out=temb, accumulate=True)
        ops.skinny_linear(temb, f["mod.w"], f["mod.b"], out=MOD, act_in=ACT_SILU)
        cos, sin = state["cos"], state["sin"]
        scale = 1.0 / math.sqrt(128.0)

        def mod(off):
            return MOD[:, off:]

        loss_terms = []
        row_loss = self._buf("row_loss", (B * S,), torch.float32)

        def tap(kind, i, tensor, rows, ld, offset_rows=0):
            """gradient injected at a tap: explicit, or d(loss term) / d(tap) from the teacher's tensor; `tensor` rows = [B][rows][D] with
            row stride ld inside a [B, S or rows, *] buffer"""
            if tap_grads is not None:
                return tap_grads[kind][i].to(**bf).contiguous()
            if teacher is None:
                return None
            t_all = teacher[kind]
            t = self._keep(f"teacher{kind}", (t_all[i] if isinstance(t_all, (list, tuple)) else t_all[:, i]).to(**bf))  # [B, rows, D]
            g = self._buf(f"G{kind}.{i}", (B, rows, D))
            term = self._buf(f"term{kind}.{i}", (1,), torch.float32)
            # F.kl_div(..., reduction='batchmean') divides the summed rows by B (:616); one launch per sample keeps the strides simple
            for b in range(B):
                ops.kd_loss_rows(t[b], tensor[b, offset_rows:offset_rows + rows], g[b], row_loss[b * rows:], rows=rows, D=D,
                                 temperature=temperature, loss_scale=1.0 / B, ldt=D, lds=ld, ldg=D)
            ops.reduce_rows(row_loss, term, np_=B * rows, len_=1, in_ps=1, alpha=1.0 / B)
            ops.zero_if_nonfinite_(g, term)
            loss_terms.append(term)
            return g

        saved = dict(state=state, B=B, St=St, Si=Si, S=S, Spad=Spad, temb=temb, MOD=self._keep("MOD", MOD), double=[], single=[])
        qkv_img_off = B * St * 3 * D
        for i in range(cfg.num_layers):
            p = f"d{i}"
            oi = i * 12 * D
            oc = oi + 6 * D
            sv = dict(Xin=self._keep(p + ".Xin", X))
            ops.ln_modulate(X, NRM, B, S, D, St, mod(oc), mod(oc + D), mod(oi), mod(oi + D), Ntot)
            qkv = self._buf(p + ".QKV", (B * S, 3 * D))
            ops.gemm_pair(dict(A=NRM, W=f[p + ".qkv.w"], bias=f[p + ".qkv.b"], out=qkv, M=Si, batch=B, a_batch_stride=S * D, lda=D,
                               a_offset=St * D, c_batch_stride=Si * 3 * D, ldc=3 * D, c_offset=qkv_img_off),
                          dict(A=NRM, W=f[p + ".cqkv.w"], bias=f[p + ".cqkv.b"], out=qkv, M=St, batch=B, a_batch_stride=S * D, lda=D,
                               c_batch_stride=St * 3 * D, ldc=3 * D))
            ops.qkv_split(qkv, qkv.view(-1)[qkv_img_off:], 3 * D, 3 * D, B, S, St, H, f[p + ".norm_added_q"], f[p + ".norm_added_k"],
                          f[p + ".norm_q"], f[p + ".norm_k"], cos, sin, Q, K, VT, Spad)
            sv["L"] = self._buf(p + ".L", (B, H, Spad), torch.float32)
            ops.attention_lse(Q, K, VT, ATT, sv["L"], B, H, S, Spad, D, S * D, scale)
            OP = self._buf(p + ".OP", (B, S, D))
            ops.gemm_pair(dict(A=ATT, W=f[p + ".to_out.w"], bias=f[p + ".to_out.b"], out=OP, M=Si, batch=B, a_batch_stride=S * D, lda=D,
                               a_offset=St * D, c_batch_stride=S * D, ldc=D, c_offset=St * D),
                          dict(A=ATT, W=f[p + ".to_add_out.w"], bias=f[p + ".to_add_out.b"], out=OP, M=St, batch=B, a_batch_stride=S * D, lda=D,
                               c_batch_stride=S * D, ldc=D))
            sv["O"] = self._keep(p + ".O", ATT)                             # attention output (rowsum(dO * O) of the fused attention backward)
            sv["Gimg"] = tap(0, i, OP, Si, D, offset_rows=St)   # reference lists[0]: image-stream attention output
            sv["Gtxt"] = tap(1, i, OP, St, D, offset_rows=0)    # lists[1]: text-stream attention output
            ops.gated_residual_(X, OP, mod(oi + 2 * D), B, Si, D, S * D, D, S * D, D, Ntot, x_offset=St * D, t_offset=St * D)
            ops.gated_residual_(X, OP, mod(oc + 2 * D), B, St, D, S * D, D, S * D, D, Ntot)
            sv.update(QKV=qkv, OP=OP, Xmid=self._keep(p + ".Xmid", X))
            ops.ln_modulate(X, NRM, B, S, D, St, mod(oc + 3 * D), mod(oc + 4 * D), mod(oi + 3 * D), mod(oi + 4 * D), Ntot)
            PRE = self._buf(p + ".PRE", (B * S, 4 * D))   # text rows [B*St] first, image rows behind (as CAT in denoise)
            Hh = self._buf("Hh", (B * S, 4 * D))
            ff_img = B * St * 4 * D
            ops.gemm_pair(dict(A=NRM, W=f[p + ".ff.0.w"], bias=f[p + ".ff.0.b"], out=PRE, out2=Hh, act2=ACT_GELU_TANH, M=Si, batch=B,
                               a_batch_stride=S * D,
                               lda=D, a_offset=St * D, c_batch_stride=Si * 4 * D, ldc=4 * D, c_offset=ff_img),
                          dict(A=NRM, W=f[p + ".ff_context.0.w"], bias=f[p + ".ff_context.0.b"], out=PRE, out2=Hh, act2=ACT_GELU_TANH, M=St, batch=B,
                               a_batch_stride=S * D, lda=D, c_batch_stride=St * 4 * D, ldc=4 * D))
            FF = self._buf(p + ".FF", (B, S, D))
            ops.gemm_pair(dict(A=Hh, W=f[p + ".ff.2.w"], bias=f[p + ".ff.2.b"], out=FF, M=Si, batch=B, a_batch_stride=Si * 4 * D, lda=4 * D,
                               a_offset=ff_img, c_batch_stride=S * D, ldc=D, c_offset=St * D),
                          dict(A=Hh, W=f[p + ".ff_context.2.w"], bias=f[p + ".ff_context.2.b"], out=FF, M=St, batch=B, a_batch_stride=St * 4 * D,
                               lda=4 * D, c_batch_stride=S * D, ldc=D))
            ops.gated_residual_(X, FF, mod(oi + 5 * D), B, Si, D, S * D, D, S * D, D, Ntot, x_offset=St * D, t_offset=St * D)
            ops.gated_residual_(X, FF, mod(oc + 5 * D), B, St, D, S * D, D, S * D, D, Ntot)
            sv.update(PRE=PRE, FF=FF)
            saved["double"].append(sv)
        base = cfg.num_layers * 12 * D
        for i in range(cfg.num_single_layers):
            p = f"s{i}"
            o = base + i * 3 * D
            sv = dict(Xin=self._keep(p + ".Xin", X))
            ops.ln_modulate(X, NRM, B, S, D, 0, None, None, mod(o), mod(o + D), Ntot)
            w, bias = f[p + ".in.w"], f[p + ".in.b"]
            IN = self._buf(p + ".IN", (B * S, 7 * D))   # [q|k|v pre-norm | proj_mlp pre-GELU]
            ops.gemm(NRM, w, bias, out=IN, M=B * S)
            ops.qkv_split(None, IN, 7 * D, 7 * D, B, S, 0, H, None, None, f[p + ".norm_q"], f[p + ".norm_k"], cos, sin, Q, K, VT, Spad)
            sv["L"] = self._buf(p + ".L", (B, H, Spad), torch.float32)
            ops.attention_lse(Q, K, VT, CAT, sv["L"], B, H, S, Spad, 5 * D, S * 5 * D, scale)
            sv["O"] = self._keep(p + ".O", CAT.view(B, S, 5 * D)[:, :, :D])
            sv["G"] = tap(2, i, CAT.view(B, S, 5 * D), S, 5 * D)   # lists[2]: the un-projected joint attention output
            # GELU(proj_mlp) into CAT[:, D:]: one elementwise pass through the GEMM epilogue is not available here, so the kernel that
            # owns the activation (x2i_gemm_bf16 with act) recomputes that slice from NRM -- the saved pre-activation stays exact
            ops.gemm(NRM, w[3 * D:], bias[3 * D:], out=CAT, M=B * S, N=4 * D, ldc=5 * D, c_offset=D, act=ACT_GELU_TANH)
            PO = self._buf(p + ".PO", (B, S, D))
            ops.gemm(CAT, f[p + ".proj_out.w"], f[p + ".proj_out.b"], out=PO, M=S, batch=B, a_batch_stride=S * 5 * D, lda=5 * D,
                     c_batch_stride=S * D, ldc=D)
            ops.gated_residual_(X, PO, mod(o + 2 * D), B, S, D, S * D, D, S * D, D, Ntot)
            sv.update(IN=IN, PO=PO)
            saved["single"].append(sv)
        o = base + cfg.num_single_layers * 3 * D
        NRMF = ws["NRMF"]
        ops.ln_modulate(X, NRMF, B, Si, D, 0, None, None, mod(o + D), mod(o), Ntot, x_bs=S * D, ldx=D, y_bs=Si * D, ldy=D, x_offset=St * D)
        out = torch.empty((B, Si, m.out_channels * cfg.patch_size ** 2), **bf)
        ops.gemm(NRMF, f["proj_out.w"], f["proj_out.b"], out=out, M=B * Si)
        self.saved = saved
        loss = torch.zeros((1,), device=dev, dtype=torch.float32)
        for t in loss_terms:   # non-finite terms are skipped (:617-620): their gradients were zeroed on the device, here the value
            loss += torch.where(torch.isfinite(t), t, torch.zeros_like(t))
        return out, loss

    # ------------------------------------------------------------------ backward pieces
    @torch.no_grad()
    def _attention_bwd(self, qkv0, qkv1, ld, S0, norms, dATT, ld_datt, dQKV0, dQKV1, O=None, L=None):
        """d(q|k|v rows) from d(attention output) [B, S, *] (row stride ld_datt): recompute Q / K / V^T, then per sample the explicit
        P = softmax(scale Q K^T), dP = dO V^T, dS, dQ = dS K, dK = dS^T Q, dV = P^T dO -- all x2i_gemm_bf16 launches."""
        sv = self.saved
        m = self.m
        B, S, Spad = sv["B"], sv["S"], sv["Spad"]
        H = m._H
        ws = sv["state"]["ws"]
        Q, K, VT = ws["Q"], ws["K"], ws["VT"]
        cos, sin = sv["state"]["cos"], sv["state"]["sin"]
        dev = m.device
        bf = dict(device=dev, dtype=torch.bfloat16)
        nq0, nk0, nq1, nk1 = norms
        ops.qkv_split(qkv0, qkv1, ld, ld, B, S, S0, H, nq0, nk0, nq1, nk1, cos, sin, Q, K, VT, Spad)
        scale = 1.0 / math.sqrt(128.0)
        if self.fused_attention_backward and O is not None:
            w2 = self.__dict__.get("_attn_ws2")
            if w2 is None or w2["key"] != (B, H, Spad):
                hm = lambda: torch.empty((B * H, Spad, 128), **bf)  # noqa: E731
                tm = lambda: torch.empty((B * H, 128, Spad), **bf)  # noqa: E731
                w2 = dict(key=(B, H, Spad), V=hm(), QT=tm(), KT=tm(), dOT=torch.zeros((B * H, 128, Spad), **bf), dOh=hm(), dQ=hm(), dK=hm(), dV=hm(),
                          D=torch.empty((B, H, Spad), device=dev, dtype=torch.float32), L=torch.empty((B, H, Spad), device=dev, dtype=torch.float32))
                self._attn_ws2 = w2
            BH = B * H
            ops.transpose(VT, w2["V"], batch=BH, R=128, C=Spad, in_bs=128 * Spad, ld_in=Spad, out_bs=Spad * 128, ld_out=128)
            ops.transpose(Q, w2["QT"], batch=BH, R=Spad, C=128, in_bs=Spad * 128, ld_in=128, out_bs=128 * Spad, ld_out=Spad)
            ops.transpose(K, w2["KT"], batch=BH, R=Spad, C=128, in_bs=Spad * 128, ld_in=128, out_bs=128 * Spad, ld_out=Spad)
            for b in range(B):  # token-major d(attention output) -> per-head transposed and row-major copies (zero beyond S)
                ops.transpose(dATT, w2["dOT"][b * H:], batch=H, R=S, C=128, in_bs=128, ld_in=ld_datt, out_bs=128 * Spad, ld_out=Spad,
                              in_offset=b * S * ld_datt)
            ops.transpose(w2["dOT"], w2["dOh"], batch=BH, R=128, C=Spad, in_bs=128 * Spad, ld_in=Spad, out_bs=Spad * 128, ld_out=128)
            ops.attention_bwd_prep(dATT, O, w2["D"], B, H, S, Spad, do_bs=S * ld_datt, lddo=ld_datt, o_bs=S * O.shape[-1], ldo=O.shape[-1])
            ops.attention_bwd(Q, K, w2["V"], w2["QT"], w2["KT"], w2["dOh"], w2["dOT"], L if L is not None else w2["L"], w2["D"], w2["dQ"], w2["dK"],
                              w2["dV"], B, H, S, Spad, scale, have_lse=L is not None)
            ops.qkv_split_bwd(qkv0, qkv1, ld, ld, dQKV0, dQKV1, ld, ld, B, S, S0, H, nq0, nk0, nq1, nk1, cos, sin, w2["dQ"], w2["dK"], w2["dV"], Spad)
            return
        big = self.__dict__.get("_attn_ws")
        if big is None or big["key"] != (H, Spad):
            big = dict(key=(H, Spad), P=torch.empty((H, Spad, Spad), **bf), dP=torch.empty((H, Spad, Spad), **bf),
                       T=torch.empty((H, Spad, Spad), **bf), V=torch.empty((H, Spad, 128), **bf), KT=torch.empty((H, 128, Spad), **bf),
                       QT=torch.empty((H, 128, Spad), **bf), dOT=torch.zeros((H, 128, Spad), **bf))
            self._attn_ws = big
        P, dP, T, V, KT, QT, dOT = (big[k] for k in ("P", "dP", "T", "V", "KT", "QT", "dOT"))
        dQ, dK, dV = (self._buf(n, (B, H, Spad, 128)) for n in ("xdQ", "xdK", "xdV"))
        L2 = Spad * Spad
        for b in range(B):
            ops.gemm(Q[b], K[b], out=P, M=Spad, N=Spad, K=128, batch=H, a_batch_stride=Spad * 128, lda=128, w_batch_stride=Spad * 128,
                     c_batch_stride=L2, ldc=Spad)
            ops.softmax_pad_(P, H, Spad, S, Spad, S, scale)
            ops.transpose(VT[b], V, batch=H, R=128, C=Spad, in_bs=128 * Spad, ld_in=Spad, out_bs=Spad * 128, ld_out=128)
            ops.gemm(dATT, V, out=dP, M=S, N=Spad, K=128, batch=H, a_batch_stride=128, lda=ld_datt, a_offset=b * S * ld_datt,
                     w_batch_stride=Spad * 128, c_batch_stride=L2, ldc=Spad)
            ops.softmax_bwd_(P, dP, H, Spad, S, Spad, S, scale)   # dP now holds dS (scale folded in)
            ops.transpose(K[b], KT, batch=H, R=Spad, C=128, in_bs=Spad * 128, ld_in=128, out_bs=128 * Spad, ld_out=Spad)
            ops.gemm(dP, KT, out=dQ[b], M=S, N=128, K=Spad, batch=H, a_batch_stride=L2, lda=Spad, w_batch_stride=128 * Spad,
                     c_batch_stride=Spad * 128, ldc=128)
            ops.transpose(dP, T, batch=H, R=Spad, C=Spad, in_bs=L2, ld_in=Spad, out_bs=L2, ld_out=Spad)
            ops.transpose(Q[b], QT, batch=H, R=Spad, C=128, in_bs=Spad * 128, ld_in=128, out_bs=128 * Spad, ld_out=Spad)
            ops.gemm(T, QT, out=dK[b], M=S, N=128, K=Spad, batch=H, a_batch_stride=L2, lda=Spad, w_batch_stride=128 * Spad,
                     c_batch_stride=Spad * 128, ldc=128)
            ops.transpose(P, T, batch=H, R=Spad, C=Spad, in_bs=L2, ld_in=Spad, out_bs=L2, ld_out=Spad)
            ops.transpose(dATT, dOT, batch=H, R=S, C=128, in_bs=128, ld_in=ld_datt, out_bs=128 * Spad, ld_out=Spad,
                          in_offset=b * S * ld_datt)
            ops.gemm(T, dOT, out=dV[b], M=S, N=128, K=Spad, batch=H, a_batch_stride=L2, lda=Spad, w_batch_stride=128 * Spad,
                     c_batch_stride=Spad * 128, ldc=128)
        ops.qkv_split_bwd(qkv0, qkv1, ld, ld, dQKV0, dQKV1, ld, ld, B, S, S0, H, nq0, nk0, nq1, nk1, cos, sin, dQ, dK, dV, Spad)

    @torch.no_grad()
    def backward(self):
        """Runs the chain on the activations kept by forward_train(); returns (d_encoder_hidden_states [B, St, joint_dim] bf16,
        d_pooled_projections [B, pooled_dim] f32)."""
        sv = self.saved
        if sv is None:
            raise RuntimeError("backward: call forward_train() first")
        m = self.m
        cfg, f = m.config, m._fused
        B, St, Si, S = sv["B"], sv["St"], sv["Si"], sv["S"]
        D, H = m.inner_dim, m._H
        dev = m.device
        bf = dict(device=dev, dtype=torch.bfloat16)
        MOD = sv["MOD"]
        Ntot = m._mod_rows
        R = _gcd_rows(St, Si)
        dX = self._buf("dX", (B, S, D), zero=True)            # gradient of the residual stream
        dT = self._buf("dT", (B, S, D))
        dN = self._buf("dN", (B, S, D))
        dMOD = self._buf("dMOD", (B, Ntot), torch.float32, zero=True)
        part = self._buf("part", (B * ((S + R - 1) // R) * 2 * D,), torch.float32)

        def mod(off):
            return MOD[:, off:]

        def colsum(npart, stat, nstat, out_off):
            """dMOD[:, out_off : out_off + D] += sum over the npart wave partials of statistic `stat` (of nstat)"""
            ops.reduce_rows(part, dMOD, np_=npart, len_=D, nz=B, in_zs=npart * nstat * D, in_ps=nstat * D, out_zs=Ntot, accumulate=True,
                            in_offset=stat * D, out_offset=out_off)

        def gate_bwd(T, G, gate_off, r0, n):
            """rows [r0, r0 + n) of every sample: dT = gate * dX (+ G), d gate accumulated"""
            ops.gate_bwd(dX, T, mod(gate_off), G, dT, part, B=B, S=n, D=D, R=R, gate_bs=Ntot, dx_bs=S * D, t_bs=S * D,
                         g_bs=n * D, dt_bs=S * D, dx_offset=r0 * D, t_offset=r0 * D, dt_offset=r0 * D)
            colsum((n + R - 1) // R, 0, 1, gate_off)

        def ln_bwd(Xsaved, scale_off, shift_off, r0, n):
            """rows [r0, r0 + n): dX += LayerNorm-modulate backward of dN; d scale / d shift accumulated"""
            ops.ln_mod_bwd(Xsaved, dN, mod(scale_off), dX, dX, part, B=B, S=n, D=D, R=R, mult_bs=Ntot, x_bs=S * D, dy_bs=S * D, dx_bs=S * D,
                           x_offset=r0 * D, dy_offset=r0 * D, dx_offset=r0 * D)
            np_ = (n + R - 1) // R
            colsum(np_, 0, 2, scale_off)
            colsum(np_, 1, 2, shift_off)

        base = cfg.num_layers * 12 * D
        # ---- single-stream blocks, last to first
        for i in reversed(range(cfg.num_single_layers)):
            p = f"s{i}"
            o = base + i * 3 * D
            s_ = sv["single"][i]
            gate_bwd(s_["PO"], None, o + 2 * D, 0, S)                                     # dT = d proj_out output
            dCAT = self._buf("dCAT", (B * S, 5 * D))
            ops.gemm(dT, self._wt(p + ".proj_out.w"), out=dCAT, M=B * S)                    # [d attention | d GELU(proj_mlp)]
            if s_["G"] is not None:
                ops.gate_bwd(dCAT, None, None, s_["G"], dCAT, None, B=B, S=S, D=D, R=R, dx_bs=S * 5 * D, lddx=5 * D, g_bs=S * D, dt_bs=S * 5 * D,
                             lddt=5 * D)                                                    # d attention += tap gradient
            dIN = self._buf("dIN", (B * S, 7 * D))
            # d proj_mlp pre-activation -> columns [3D, 7D) of dIN
            dIN.view(B * S, 7 * D)[:, 3 * D:].copy_(dCAT.view(B * S, 5 * D)[:, D:])
            ops.act_bwd_(dIN, s_["IN"], ACT_GELU_TANH, rows=B * S, cols=4 * D, ldd=7 * D, ldp=7 * D, d_offset=3 * D, p_offset=3 * D)
            self._attention_bwd(None, s_["IN"], 7 * D, 0, (None, None, f[p + ".norm_q"], f[p + ".norm_k"]), dCAT, 5 * D, None, dIN, O=s_["O"], L=s_["L"])
            ops.gemm(dIN, self._wt(p + ".in.w"), out=dN, M=B * S)
            ln_bwd(s_["Xin"], o + D, o, 0, S)
        # ---- double-stream blocks
        for i in reversed(range(cfg.num_layers)):
            p = f"d{i}"
            oi = i * 12 * D
            oc = oi + 6 * D
            d_ = sv["double"][i]
            # feed-forward: x = x_mid + gate_mlp * FF
            gate_bwd(d_["FF"], None, oi + 5 * D, St, Si)
            gate_bwd(d_["FF"], None, oc + 5 * D, 0, St)
            dH = self._buf("dH", (B * S, 4 * D))
            ff_img = B * St * 4 * D
            ops.gemm_pair(dict(A=dT, W=self._wt(p + ".ff.2.w"), out=dH, M=Si, batch=B, a_batch_stride=S * D, lda=D, a_offset=St * D,
                               c_batch_stride=Si * 4 * D, ldc=4 * D, c_offset=ff_img),
                          dict(A=dT, W=self._wt(p + ".ff_context.2.w"), out=dH, M=St, batch=B, a_batch_stride=S * D, lda=D,
                               c_batch_stride=St * 4 * D, ldc=4 * D))
            ops.act_bwd_(dH, d_["PRE"], ACT_GELU_TANH)
            ops.gemm_pair(dict(A=dH, W=self._wt(p + ".ff.0.w"), out=dN, M=Si, batch=B, a_batch_stride=Si * 4 * D, lda=4 * D, a_offset=ff_img,
                               c_batch_stride=S * D, ldc=D, c_offset=St * D),
                          dict(A=dH, W=self._wt(p + ".ff_context.0.w"), out=dN, M=St, batch=B, a_batch_stride=St * 4 * D, lda=4 * D,
                               c_batch_stride=S * D, ldc=D))
            ln_bwd(d_["Xmid"], oi + 4 * D, oi + 3 * D, St, Si)
            ln_bwd(d_["Xmid"], oc + 4 * D, oc + 3 * D, 0, St)
            # attention: x_mid = x_in + gate_msa * OP, OP = to_out(attention) -- the taps sit on OP
            gate_bwd(d_["OP"], d_["Gimg"], oi + 2 * D, St, Si)
            gate_bwd(d_["OP"], d_["Gtxt"], oc + 2 * D, 0, St)
            dATT = self._buf("dATT", (B, S, D))
            ops.gemm_pair(dict(A=dT, W=self._wt(p + ".to_out.w"), out=dATT, M=Si, batch=B, a_batch_stride=S * D, lda=D, a_offset=St * D,
                               c_batch_stride=S * D, ldc=D, c_offset=St * D),
                          dict(A=dT, W=self._wt(p + ".to_add_out.w"), out=dATT, M=St, batch=B, a_batch_stride=S * D, lda=D, c_batch_stride=S * D,
                               ldc=D))
            dQKV = self._buf("dQKV", (B * S, 3 * D))
            img = B * St * 3 * D
            self._attention_bwd(d_["QKV"], d_["QKV"].view(-1)[img:], 3 * D, St,
                                (f[p + ".norm_added_q"], f[p + ".norm_added_k"], f[p + ".norm_q"], f[p + ".norm_k"]), dATT, D, dQKV,
                                dQKV.view(-1)[img:], O=d_["O"], L=d_["L"])
            ops.gemm_pair(dict(A=dQKV, W=self._wt(p + ".qkv.w"), out=dN, M=Si, batch=B, a_batch_stride=Si * 3 * D, lda=3 * D, a_offset=img,
                               c_batch_stride=S * D, ldc=D, c_offset=St * D),
                          dict(A=dQKV, W=self._wt(p + ".cqkv.w"), out=dN, M=St, batch=B, a_batch_stride=St * 3 * D, lda=3 * D,
                               c_batch_stride=S * D, ldc=D))
            ln_bwd(d_["Xin"], oi + D, oi, St, Si)
            ln_bwd(d_["Xin"], oc + D, oc, 0, St)
        # ---- embedders: text rows of dX -> context_embedder -> encoder_hidden_states
        joint = f["context_embedder.w"].shape[1]
        d_enc = torch.empty((B, St, joint), **bf)
        ops.gemm(dX, self._wt("context_embedder.w"), out=d_enc, M=St, batch=B, a_batch_stride=S * D, lda=D, c_batch_stride=St * joint, ldc=joint)
        # ---- modulation table -> temb -> text_embedder -> pooled_projections  (MOD = Linear(SiLU(temb)))
        d_act = ops.skinny_linear_bwd(dMOD, f["mod.w"])                       # d SiLU(temb)
        ops.act_bwd_(d_act, sv["temb"], ACT_SILU)                             # d temb = d cond
        pooled = sv["state"].get("pooled")
        if pooled is None:
            raise RuntimeError("backward: prepare the conditioning with DistillBackward.prepare_conditioning (it keeps the pooled input)")
        pre1 = ops.skinny_linear(pooled, f["tte.text_embedder.1.w"], f["tte.text_embedder.1.b"])
        d_h1 = ops.skinny_linear_bwd(d_act, f["tte.text_embedder.2.w"])
        ops.act_bwd_(d_h1, pre1, ACT_SILU)
        d_pooled = ops.skinny_linear_bwd(d_h1, f["tte.text_embedder.1.w"])
        self.dMOD = dMOD
        return d_enc, d_pooled

    @torch.no_grad()
    def prepare_conditioning(self, encoder_hidden_states, pooled_projections, txt_ids, img_ids, guidance=None):
        """FluxTransformer2DModel.prepare_conditioning + the pooled input kept for the backward of the text embedder."""
        st = self.m.prepare_conditioning(encoder_hidden_states, pooled_projections, txt_ids, img_ids, guidance)
        st["pooled"] = pooled_projections.to(device=self.m.device, dtype=torch.bfloat16).contiguous()
        return st


class ProjectorTrainer:
    """The trainable half of the reference's step: Proj7Exp forward with saved activations, its backward (weight gradients), gradient
    all-reduce over the data-parallel group (what DistributedDataParallel does for the reference, train/train_qwenvl.py:483),
    `clip_grad_norm_` (:628) and `torch.optim.AdamW` (:447-459, :630).  Parameters stay the module's bf16 tensors; gradients and the two
    moments are f32 in ONE flat buffer each (a single RCCL all-reduce per step; the reference's moments are bf16 -- stated difference,
    results agree with torch.optim.AdamW on bf16 parameters to bf16 rounding, tests/test_train_gpu.py)."""

    def __init__(self, proj, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=1.0, process_group=None):
        self.proj = proj
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.pg = process_group
        self.names = [n for n, _ in proj.named_parameters()]
        self.params = [p for _, p in proj.named_parameters()]
        sizes = [p.numel() for p in self.params]
        dev = self.params[0].device
        total = sum(sizes)
        self.grad = torch.zeros((total,), device=dev, dtype=torch.float32)
        self.m = torch.zeros_like(self.grad)
        self.v = torch.zeros_like(self.grad)
        self.off = {}
        o = 0
        for n, s in zip(self.names, sizes):
            self.off[n] = (o, s)
            o += s
        self.step_count = 0
        self.saved = None
        self.last_norm = None

    def g(self, name):
        o, s = self.off[name]
        return self.grad[o:o + s]

    @torch.no_grad()
    def forward(self, x):
        """Proj7Exp.forward (utils/proj.py:62-72, :28-33) keeping the activations its backward needs.  Returns (pooled, prompt_embeds)."""
        pr = self.proj
        mlp = pr.mlp
        B, Cc, S, H = x.shape
        x = x.to(torch.bfloat16).contiguous()
        if pr.use_scale:
            x0 = ops.proj_layer_mean(x, pr.cha_scale.float().reshape(-1).contiguous())
        elif pr.use_cnn:
            from .proj import _conv5x5
            x0 = _conv5x5(pr, x)
        else:
            x0 = ops.proj_layer_mean(x, None)
        xn = ops.ln_affine(x0, mlp.layernorm.weight, mlp.layernorm.bias, mlp.eps)
        W0, W2, Wf, bfc = mlp.projector[0].weight, mlp.projector[2].weight, mlp.fc[1].weight, mlp.fc[1].bias
        bf = dict(device=x.device, dtype=torch.bfloat16)
        pre0 = torch.empty((B * S, W0.shape[0]), **bf)
        h = torch.empty_like(pre0)
        ops.gemm(xn, W0, out=pre0, out2=h, act2=ACT_GELU_ERF, M=B * S)
        x2 = torch.empty((B, S, W2.shape[0]), **bf)
        g2 = torch.empty_like(x2)
        ops.gemm(h, W2, out=x2, out2=g2, act2=ACT_GELU_ERF, M=B * S)
        x1_tok = ops.gemm(g2, Wf, bfc, M=B * S, out_f32=True)
        x1 = ops.seq_mean(x1_tok.view(B, S, -1))
        self.saved = dict(x=x, x0=x0, xn=xn, pre0=pre0, h=h, x2=x2, g2=g2, B=B, S=S)
        return x1.to(torch.bfloat16), x2

    @torch.no_grad()
    def _wgrad(self, dY, X, out):
        """out f32 [N, K] = dY^T X   (dY bf16 [M, N], X bf16 [M, K]) -- the forward GEMM on transposed operands"""
        M = dY.shape[0]
        dYT, XT = ops.transpose(dY.contiguous()), ops.transpose(X.contiguous())
        ops.gemm(dYT, XT, out=out.view(dYT.shape[0], XT.shape[0]), M=dYT.shape[0], N=XT.shape[0], K=M, out_f32=True)

    @torch.no_grad()
    def backward(self, d_enc, d_pooled, keep=False):
        """Weight gradients from d loss / d prompt_embeds (bf16 [B,S,4096]) and d loss / d pooled (f32 [B,768]); gradients ACCUMULATE
        (gradient accumulation steps, :560) until step()."""
        sv, pr = self.saved, self.proj
        mlp = pr.mlp
        B, S = sv["B"], sv["S"]
        dev = sv["x"].device
        bf = dict(device=dev, dtype=torch.bfloat16)
        W0, W2, Wf = mlp.projector[0].weight, mlp.projector[2].weight, mlp.fc[1].weight
        N1 = Wf.shape[0]
        tmp = torch.empty((Wf.numel(),), device=dev, dtype=torch.float32)

        def acc(name, t):
            ops.reduce_rows(t, self.g(name), np_=1, len_=t.numel(), accumulate=True)

        # fc: x1 = mean_s(g2 Wf^T + b): every token row receives d_pooled[b] / S
        dp = torch.empty((B, N1), device=dev, dtype=torch.float32)
        ops.reduce_rows(d_pooled.contiguous(), dp, np_=1, len_=B * N1, alpha=1.0 / S)
        dtok = ops.to_bf16(dp).view(B, 1, N1).expand(B, S, N1).contiguous().view(B * S, N1)
        self._wgrad(dtok, sv["g2"].view(B * S, -1), tmp)
        acc("mlp.fc.1.weight", tmp)
        ops.reduce_rows(d_pooled.contiguous(), self.g("mlp.fc.1.bias"), np_=B, len_=N1, in_ps=N1, accumulate=True)
        dg2 = torch.empty((B * S, W2.shape[0]), **bf)
        ops.gemm(dtok, ops.transpose(Wf.contiguous()), out=dg2, M=B * S)
        ops.act_bwd_(dg2, sv["x2"].view(B * S, -1), ACT_GELU_ERF)
        # d x2 = d_enc + d g2 * gelu'(x2)
        ops.gate_bwd(dg2, None, None, d_enc.to(**bf).contiguous(), dg2, None, B=1, S=B * S, D=W2.shape[0], R=8)
        tmp2 = torch.empty((W2.numel(),), device=dev, dtype=torch.float32)
        self._wgrad(dg2, sv["h"], tmp2)
        acc("mlp.projector.2.weight", tmp2)
        dh = torch.empty((B * S, W2.shape[1]), **bf)
        ops.gemm(dg2, ops.transpose(W2.contiguous()), out=dh, M=B * S)
        ops.act_bwd_(dh, sv["pre0"], ACT_GELU_ERF)
        tmp0 = torch.empty((W0.numel(),), device=dev, dtype=torch.float32)
        self._wgrad(dh, sv["xn"].view(B * S, -1), tmp0)
        acc("mlp.projector.0.weight", tmp0)
        Hd = W0.shape[1]
        dxn = torch.empty((B * S, Hd), **bf)
        ops.gemm(dh, ops.transpose(W0.contiguous()), out=dxn, M=B * S)
        # affine LayerNorm backward: d weight / d bias column sums, d x0
        R = 8
        nw = (B * S + R - 1) // R
        part = torch.empty((nw, 2, Hd), device=dev, dtype=torch.float32)
        dx0 = torch.empty((B, S, Hd), **bf)
        ops.ln_mod_bwd(sv["x0"], dxn, mlp.layernorm.weight.float().contiguous(), None, dx0, part, B=1, S=B * S, D=Hd, R=R, mult_is_scale=False,
                       eps=mlp.eps)
        ops.reduce_rows(part, self.g("mlp.layernorm.weight"), np_=nw, len_=Hd, in_ps=2 * Hd, accumulate=True)
        ops.reduce_rows(part, self.g("mlp.layernorm.bias"), np_=nw, len_=Hd, in_ps=2 * Hd, accumulate=True, in_offset=Hd)
        # layer fusion
        Cc = sv["x"].shape[1]
        if pr.use_scale:
            acc("cha_scale", ops.plane_dot(sv["x"], dx0, alpha=1.0 / Cc))
        elif pr.use_cnn:
            acc("conv.weight", ops.conv5x5_wgrad(sv["x"], dx0).view(-1))
            ops.sum_all(dx0, out=self.g("conv.bias"), accumulate=True)
        if keep:  # intermediates for the parity tests
            self.kept = dict(dtok=dtok, dx2=dg2, dpre0=dh, dxn=dxn, dx0=dx0)

    @torch.no_grad()
    def step(self):
        """all-reduce (mean) over the data-parallel group, clip by global norm, AdamW; clears the gradients.  Returns the device tensor
        [clip coefficient, gradient norm]."""
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.pg) > 1:
            dist.all_reduce(self.grad, group=self.pg)  # one flat buffer: a single RCCL ring all-reduce per step
            ops.reduce_rows(self.grad, self.grad, np_=1, len_=self.grad.numel(), alpha=1.0 / dist.get_world_size(self.pg))  # in place: mean
        coef = ops.clip_coef(ops.sum_all(self.grad, squares=True), self.max_norm)
        self.step_count += 1
        for n, p in zip(self.names, self.params):
            o, s = self.off[n]
            ops.adamw_(p, self.grad[o:o + s], self.m[o:o + s], self.v[o:o + s], lr=self.lr, beta1=self.betas[0], beta2=self.betas[1],
                       eps=self.eps, weight_decay=self.wd, step=self.step_count, coef=coef)
        self.proj.__dict__.pop("_conv5x5_cache", None)  # the packed conv table is keyed on the weight's version counter; drop it
        self.grad.zero_()
        self.last_norm = coef
        return coef


@torch.no_grad()
def distill_step(trainer, chain, text_embeddings, latents, timestep, teacher, txt_ids, img_ids, guidance=None, temperature=3.0,
                 optimizer_step=True):
    """train/train_qwenvl.py:575-632 on the HIP path: projector forward, student transformer forward with the distillation loss
    evaluated at every attention tap, backward through the frozen transformer into the projector's outputs, projector backward,
    (clip + AdamW).  `timestep` as the reference passes it to the transformer (already / 1000).  Returns the loss (device scalar)."""
    pooled, prompt = trainer.forward(text_embeddings)
    st = chain.prepare_conditioning(prompt, pooled, txt_ids, img_ids, guidance)
    _, loss = chain.forward_train(st, latents, timestep, teacher=teacher, temperature=temperature)
    d_enc, d_pooled = chain.backward()
    trainer.backward(d_enc, d_pooled)
    if optimizer_step:
        trainer.step()
    return loss


class GraphedDistillStep:
    """distill_step with the launch sequence of everything but the optimizer captured into ONE hipGraph: projector forward, transformer
    forward with saves and the 76 loss evaluations, the activation-gradient chain, projector backward -- about 6 000 launches per step,
    replayed without the Python / launch overhead that an eager step pays (it matters at per-GPU batch 1-2, where many kernels run for
    20-100 us).  The first call runs eagerly (it also sets kernel attributes and fills the buffer pool), the second captures, later
    calls copy the batch into the captured input buffers and replay.  The optimizer (all-reduce, clip, AdamW) stays outside the graph."""

    def __init__(self, trainer, chain, txt_ids, img_ids, guidance=None, temperature=3.0):
        self.trainer, self.chain = trainer, chain
        self.txt_ids, self.img_ids, self.guidance, self.temperature = txt_ids, img_ids, guidance, temperature
        self.graph = None
        self.static = None
        self.loss = None
        self.calls = 0

    def _body(self, x, lat, ts, teacher):
        pooled, prompt = self.trainer.forward(x)
        st = self.chain.prepare_conditioning(prompt, pooled, self.txt_ids, self.img_ids, self.guidance)
        _, loss = self.chain.forward_train(st, lat, ts, teacher=teacher, temperature=self.temperature)
        d_enc, d_pooled = self.chain.backward()
        self.trainer.backward(d_enc, d_pooled)
        return loss

    @torch.no_grad()
    def __call__(self, text_embeddings, latents, timestep, teacher, optimizer_step=True):
        self.calls += 1
        shapes = (tuple(text_embeddings.shape), tuple(latents.shape)) + tuple(tuple(t.shape) for t in teacher)
        if self.static is not None and self.static["shapes"] != shapes:
            # a new batch shape: drop the graph AND run this step eagerly, so that every pooled buffer of the new shape exists before
            # the next capture (buffers first allocated inside a capture live in the old graph's private pool -- ADVICE r2)
            self.graph, self.static = None, None
            self.warm_shapes = None
        if self.calls == 1 or getattr(self, "warm_shapes", None) != shapes:
            loss = self._body(text_embeddings, latents, timestep, teacher)   # eager warm-up step (a real step)
            ops.streamk_poll()
            self.warm_shapes = shapes
        else:
            if self.graph is None:
                self.static = dict(shapes=shapes, x=text_embeddings.clone(), lat=latents.clone(), ts=timestep.clone(),
                                   teacher=[t.clone() for t in teacher])
                torch.cuda.synchronize()
                # host-side weight caches must not be satisfied from an eager call: the capture has to RECORD the conv-table pack, or a
                # replay after the next optimizer step would read a table packed from the old conv weights (freed memory) --
                # e.g. when the warm-up call ran with optimizer_step=False and nothing had popped the cache
                self.trainer.proj.__dict__.pop("_conv5x5_cache", None)
                self.graph = torch.cuda.CUDAGraph()
                self.sk_ws = ops.StreamKWorkspace(text_embeddings.device)   # the graph's own stream-K workspace (ops.streamk_scope)
                with torch.cuda.graph(self.graph), ops.streamk_scope(self.sk_ws):
                    self.loss = self._body(self.static["x"], self.static["lat"], self.static["ts"], self.static["teacher"])
            else:
                s = self.static
                s["x"].copy_(text_embeddings)
                s["lat"].copy_(latents)
                s["ts"].copy_(timestep)
                for d, t in zip(s["teacher"], teacher):
                    d.copy_(t)
            self.graph.replay()
            self.sk_ws.poll()   # asynchronous read of the graph's stream-K give-up marker; checked per optimizer step (train_distill.run)
            loss = self.loss
            # the table packed inside the graph belongs to the graph's pool and is rewritten by every replay; an eager forward must
            # not pick it up from the cache
            self.trainer.proj.__dict__.pop("_conv5x5_cache", None)
        if optimizer_step:
            self.trainer.step()
        return loss
