"""Thin torch-tensor wrappers over the C ABI (include/x2i.h).  PyTorch only supplies device memory and the current
HIP stream; every computation happens in libx2i_hip.so.  All wrappers enqueue on torch's current stream, so they can
be captured with torch.cuda.graph().
"""
import ctypes as C

import torch

from . import _lib
from ._lib import QkvDesc, ACT_GELU_ERF, ACT_GELU_TANH, ACT_NONE, ACT_SILU, GemmArgs, check  # noqa: F401


import contextlib


@contextlib.contextmanager
def option(name, value):
    """Temporarily set a library A/B switch (include/x2i.h: x2i_set_option) -- used by tests and tools, never by the product path."""
    old = _lib.set_option(name, value)
    try:
        yield
    finally:
        _lib.set_option(name, old)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---------------------------------------------------------------------------------------------------- stream-K workspaces
# The persistent GEMM's chained stream-K segments park their accumulators in CALLER-owned memory (include/x2i.h:
# x2i_gemm_args.workspace).  This module is that caller: one torch-owned workspace per (device, stream) for eager launches, and one
# per captured graph (streamk_scope, entered by whoever captures: FluxPipeline, GraphedDistillStep) -- launches that can run
# concurrently never share one.  Inside a capture with no scope there is NO workspace: the launch keeps whole tiles + the peeled
# tail (same results).
# A launch that carries a workspace ALSO assumes what every persistent launch with chained segments assumes: its workgroups (one per CU)
# are co-resident, so a segment's predecessor is always running or done.  Two such launches overlapping on two streams can each hold
# part of the CUs while waiting for workgroups that cannot start -- the bounded spin then gives up (marker, undefined results; found
# by tests/test_gemm_w4_gpu.py in round 4: separate workspaces alone do not make overlap safe).  Hence: eager launches are ORDERED
# across streams here (a stream that takes over waits for the stream that issued the previous workspace-carrying GEMM), and a graph
# captured under streamk_scope must not be replayed concurrently with other GEMM launches on its device (FluxPipeline and
# GraphedDistillStep replay on the caller's one stream); streamk_check() turns a violation into an exception.
import threading
import weakref

SK_ERR_SLOT = 256  # csrc/gemm_device.h: flags[SK_ERR_SLOT] = a chained segment gave up waiting for its predecessor
_sk_eager = {}
_sk_last_stream = {}   # device index -> torch stream of the last eager workspace-carrying GEMM
_sk_all = weakref.WeakSet()
_sk_tls = threading.local()
_SK_EAGER_MAX = 8


class StreamKWorkspace:
    """128 MB + 4 KiB of device memory (x2i_streamk_workspace_bytes), zero flags; `poll()` enqueues an asynchronous read of the give-up marker, `check()` raises
    X2IError when a completed read (or, with sync=True, a synchronising one) shows it set."""

    def __init__(self, device=None):
        n = int(_lib.load().x2i_streamk_workspace_bytes())
        self.buf = torch.empty(n, dtype=torch.uint8, device=device if device is not None else torch.device("cuda", torch.cuda.current_device()))
        self.buf[:4096].zero_()
        self.nbytes = n
        self._host = None
        self._event = None
        _sk_all.add(self)

    @property
    def marker(self):
        return self.buf[4 * SK_ERR_SLOT:4 * SK_ERR_SLOT + 4].view(torch.int32)

    def poll(self):
        if self._host is None:
            self._host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._event = torch.cuda.Event()
        self._host.copy_(self.marker, non_blocking=True)
        self._event.record()

    def check(self, sync=False):
        bad = False
        if sync:
            bad = _lib.load().x2i_streamk_workspace_status(C.c_void_p(self.buf.data_ptr()), self.nbytes) != 0
        elif self._event is not None and self._event.query():
            bad = int(self._host[0]) != 0
        if bad:
            raise _lib.X2IError("x2i_amd: a chained stream-K GEMM segment gave up waiting for its predecessor (workspace marker set): the "
                                "results of that launch are undefined.  Set option gemm_streamk = 0 and report.")


@contextlib.contextmanager
def streamk_scope(ws):
    """Every GEMM issued inside uses `ws` (a StreamKWorkspace, or None for none) -- for stream captures: one workspace per graph."""
    old = getattr(_sk_tls, "ws", False)
    _sk_tls.ws = ws
    try:
        yield ws
    finally:
        _sk_tls.ws = old


def _sk_workspace():
    ws = getattr(_sk_tls, "ws", False)
    if ws is not False:
        return ws
    if torch.cuda.is_current_stream_capturing():
        return None
    dev, cur = torch.cuda.current_device(), torch.cuda.current_stream()
    last = _sk_last_stream.get(dev)
    if last is not None and last.cuda_stream != cur.cuda_stream:
        cur.wait_stream(last)          # never two chained stream-K launches in flight on one device (see the header comment)
    _sk_last_stream[dev] = cur
    key = (dev, cur.cuda_stream)
    ws = _sk_eager.get(key)
    if ws is None:
        if len(_sk_eager) >= _SK_EAGER_MAX:   # streams come and go: drop the oldest (their launches are long enqueued; torch frees by stream order)
            _sk_eager.pop(next(iter(_sk_eager)))
        ws = _sk_eager[key] = StreamKWorkspace()
    return ws


def _set_ws(a):
    ws = _sk_workspace()
    if ws is not None:
        a.workspace, a.workspace_bytes = ws.buf.data_ptr(), ws.nbytes


def option_epoch():
    """x2i_set_option changes since import (x2i_amd/_lib.py): part of FluxPipeline's graph key."""
    return _lib.option_epoch()


def streamk_poll():
    """Enqueue an asynchronous read of every live workspace's give-up marker on the current stream (no synchronisation)."""
    for ws in list(_sk_all):
        if ws.buf.device.index == torch.cuda.current_device():
            ws.poll()


def streamk_check(sync=True):
    """Raise X2IError if any stream-K workspace carries the give-up marker.  sync=True reads the markers now (synchronises);
    sync=False only looks at reads enqueued by streamk_poll() that have completed."""
    for ws in list(_sk_all):
        ws.check(sync=sync)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _req(t, dtype, name):
    if t.device.type != "cuda":
        raise _lib.X2IError("x2i_amd: %s must live on the GPU (got %s); the HIP path has no CPU fallback" % (name, t.device))
    if t.dtype != dtype:
        raise _lib.X2IError("x2i_amd: %s must be %s (got %s)" % (name, dtype, t.dtype))


def pad128(n):
    return (n + 127) // 128 * 128


def _gemm_args(A, W, bias=None, out=None, *, M=None, batch=1, a_batch_stride=0, lda=None, c_batch_stride=0, ldc=None,
         act=ACT_NONE, gate=None, gate_batch_stride=0, res=None, res_batch_stride=0, ldr=None, out2=None, act2=ACT_NONE,
         out_f32=False, a_offset=0, c_offset=0, res_offset=0, N=None, K=None, bias2=None, bias2_batch_stride=0, w_batch_stride=0, w_group=0):
    _req(A, torch.bfloat16, "A")
    _req(W, torch.bfloat16, "W")
    N = W.shape[-2] if N is None else N
    K = W.shape[-1] if K is None else K
    if M is None:
        M = A.numel() // A.shape[-1]
    lda = A.shape[-1] if lda is None else lda
    if out is None:
        out = torch.empty((batch, M, N) if batch > 1 else (M, N), device=A.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
        if batch > 1:
            c_batch_stride = M * N
    ldc = N if ldc is None else ldc
    esz_c = 4 if out_f32 else 2
    a = GemmArgs()
    a.A = A.data_ptr() + a_offset * 2
    a.a_batch_stride = a_batch_stride
    a.lda = lda
    a.W = W.data_ptr()
    a.ldw = W.stride(-2)  # row stride of the [.., N, K] weight (a leading batch dimension is addressed by w_batch_stride)
    a.bias = bias.data_ptr() if bias is not None else None
    a.C = out.data_ptr() + c_offset * esz_c
    a.c_batch_stride = c_batch_stride
    a.ldc = ldc
    a.C2 = (out2.data_ptr() + c_offset * 2) if out2 is not None else None
    a.act2 = act2
    a.gate = gate.data_ptr() if gate is not None else None
    a.gate_batch_stride = gate_batch_stride
    a.res = (res.data_ptr() + res_offset * 2) if res is not None else None
    a.res_batch_stride = res_batch_stride
    a.ldr = (ldc if ldr is None else ldr)
    a.bias2 = bias2.data_ptr() if bias2 is not None else None
    a.bias2_batch_stride = bias2_batch_stride
    a.w_batch_stride = w_batch_stride
    a.w_group = w_group   # > 0: grouped weights, W [groups, N, K] / bias [groups, N], w_group consecutive batch items per group (include/x2i.h)
    a.M, a.N, a.K, a.batch = M, N, K, batch
    a.act = act
    a.out_f32 = 1 if out_f32 else 0
    _set_ws(a)
    return a, out


def gemm(A, W, bias=None, out=None, **kw):
    """C = epi(A W^T).  A, out, res may be sub-views addressed as (tensor, element offset, row stride, batch stride)."""
    a, out = _gemm_args(A, W, bias, out, **kw)
    check(_lib.load().x2i_gemm_bf16(C.byref(a), _stream()), "gemm")
    return out


def gemm_pair(g0, g1):
    """Two gemm() calls of the same kind (dicts of gemm()'s arguments: A, W, bias, out, ...) as one grouped launch of the persistent
    kernel when the library can (include/x2i.h: x2i_gemm_pair_bf16); bit-identical to the two calls."""
    a0, _ = _gemm_args(**g0)
    a1, _ = _gemm_args(**g1)
    check(_lib.load().x2i_gemm_pair_bf16(C.byref(a0), C.byref(a1), _stream()), "gemm_pair")


def rope_pairs(cos, sin, check=True):
    """The interleaved-pair RoPE tables f32 [S,128] (cos[s][2k] == cos[s][2k+1]: FluxPosEmbed's repeat_interleave(2)) in PAIR form:
    f32 [S,64,2] = (cos, sin) of dim pair k -- the same values in half the bytes.  Pass it as `cos` with sin=None to gemm_qkv / gemm_qkv_fp8
    (include/x2i.h, x2i_qkv_desc: sin == NULL): the fused epilogues then fetch two 16-byte pieces per token and lane instead of four."""
    # (check=False: tables that are pair-form by construction -- rope_table's -- inside a stream capture, where the comparison's sync is not allowed)
    if check and not (torch.equal(cos[:, 0::2], cos[:, 1::2]) and torch.equal(sin[:, 0::2], sin[:, 1::2])):
        raise ValueError("rope_pairs: the tables are not in interleaved-pair form")
    return torch.stack((cos[:, 0::2], sin[:, 0::2]), dim=-1).contiguous()


def gemm_qkv_pair(g0, g1):
    """Two gemm_qkv() calls (dicts of its arguments) as one grouped launch (x2i_gemm_qkv_pair_bf16)."""
    (a0, q0), (a1, q1) = _gemm_qkv_args(**g0), _gemm_qkv_args(**g1)
    check(_lib.load().x2i_gemm_qkv_pair_bf16(C.byref(a0), C.byref(q0), C.byref(a1), C.byref(q1), _stream()), "gemm_qkv_pair")


def gemm_qkv(A, W, bias, Q, K, VT, norm_q, norm_k, cos, sin, **kw):
    """QKV projection with RMSNorm(q,k) + RoPE + head split + V transpose fused into the epilogue: rows of A (row m of batch
    item z = joint token tok_off + m % rows_per_sample of sample z + m // rows_per_sample) go straight to Q/K [B,H,Spad,128]
    and VT [B,H,128,Spad]; the [M, 3*H*128] product is never written (include/x2i.h: x2i_gemm_qkv_bf16)."""
    a, q = _gemm_qkv_args(A, W, bias, Q, K, VT, norm_q, norm_k, cos, sin, **kw)
    check(_lib.load().x2i_gemm_qkv_bf16(C.byref(a), C.byref(q), _stream()), "gemm_qkv")


def _gemm_qkv_args(A, W, bias, Q, K, VT, norm_q, norm_k, cos, sin, *, M, H, Spad, tok_off, rows_per_sample, batch=1, a_batch_stride=0,
             lda=None, a_offset=0, eps=1e-6, q_scale=1.0, vt_perm=False, _act2=0, _bias2=None):
    _req(A, torch.bfloat16, "A")
    _req(W, torch.bfloat16, "W")
    a = GemmArgs()
    a.A = A.data_ptr() + a_offset * 2
    a.a_batch_stride = a_batch_stride
    a.lda = A.shape[-1] if lda is None else lda
    a.W = W.data_ptr()
    a.ldw = W.stride(-2)
    a.bias = bias.data_ptr() if bias is not None else None
    a.C = None
    a.c_batch_stride, a.ldc = 0, 3 * H * 128
    a.C2, a.act2, a.gate, a.gate_batch_stride, a.res, a.res_batch_stride, a.ldr = None, _act2, None, 0, None, 0, 0
    a.bias2, a.bias2_batch_stride, a.w_batch_stride = (_bias2.data_ptr() if _bias2 is not None else None), 0, 0   # (_act2 / _bias2: tools only)
    a.M, a.N, a.K, a.batch = M, 3 * H * 128, W.shape[-1], batch
    a.act, a.out_f32 = ACT_NONE, 0
    _set_ws(a)
    q = QkvDesc()
    q.norm_q, q.norm_k, q.cos, q.sin = norm_q.data_ptr(), norm_k.data_ptr(), cos.data_ptr(), (sin.data_ptr() if sin is not None else None)   # sin=None: `cos` is the pair-form table (rope_pairs)
    q.Q, q.K, q.VT = Q.data_ptr(), K.data_ptr(), VT.data_ptr()
    q.H, q.Spad, q.tok_off, q.rows_per_sample, q.eps, q.q_scale = H, Spad, tok_off, rows_per_sample, eps, q_scale
    q.vt_perm = 1 if vt_perm else 0
    return a, q


def attention_prefers_vt_perm(H, S, scale):
    """True when the sampling path should ask its fused QKV projections for the span-permuted V^T (vt_perm=True) and call
    attention(..., vt_perm=True): the 16 x 16 x 32 MFMA attention kernel (include/x2i.h: x2i_attention_prefers_vt_perm)."""
    return bool(_lib.load().x2i_attention_prefers_vt_perm(int(H), int(S), float(scale)))


def attention(Q, K, VT, out, B, H, S, Spad, ldo, o_batch_stride, scale, o_offset=0, vt_perm=False):
    lib = _lib.load()
    if vt_perm:
        # with the stream-K workspace of this stream / graph (the GEMMs' one): a partly filled last round of work items is cut along the key axis over
        # all CUs, chained through the workspace -- bit-identical to the undivided launch (include/x2i.h: x2i_attention_vp_ws_bf16)
        ws = _sk_workspace()
        check(lib.x2i_attention_vp_ws_bf16(_p(Q), _p(K), _p(VT), C.c_void_p(out.data_ptr() + o_offset * 2), B, H, S, Spad, ldo,
                                           o_batch_stride, scale, C.c_void_p(ws.buf.data_ptr()) if ws is not None else None,
                                           ws.nbytes if ws is not None else 0, _stream()), "attention_vp")
        return out
    check(lib.x2i_attention_bf16(_p(Q), _p(K), _p(VT), C.c_void_p(out.data_ptr() + o_offset * 2), B, H, S, Spad, ldo,
                                 o_batch_stride, scale, _stream()), "attention")
    return out


def qkv_split(qkv0, qkv1, ld0, ld1, B, S, S0, H, nq0, nk0, nq1, nk1, cos, sin, Q, K, VT, Spad, eps=1e-6):
    lib = _lib.load()
    check(lib.x2i_qkv_split_bf16(_p(qkv0), _p(qkv1), ld0, ld1, B, S, S0, H, _p(nq0), _p(nk0), _p(nq1), _p(nk1), _p(cos),
                                 _p(sin), _p(Q), _p(K), _p(VT), Spad, eps, _stream()), "qkv_split")


def ln_modulate(X, Y, B, S, D, S0, shift0, scale0, shift1, scale1, mod_bs, eps=1e-6, x_bs=None, ldx=None, y_bs=None,
                ldy=None, x_offset=0, y_offset=0):
    lib = _lib.load()
    ldx = D if ldx is None else ldx
    ldy = D if ldy is None else ldy
    x_bs = S * ldx if x_bs is None else x_bs
    y_bs = S * ldy if y_bs is None else y_bs
    check(lib.x2i_ln_modulate_bf16(C.c_void_p(X.data_ptr() + 2 * x_offset), x_bs, ldx, C.c_void_p(Y.data_ptr() + 2 * y_offset),
                                   y_bs, ldy, B, S, D, S0, _p(shift0), _p(scale0), _p(shift1), _p(scale1), mod_bs, eps,
                                   _stream()), "ln_modulate")
    return Y


def ln_affine(X, weight, bias, eps, out=None):
    lib = _lib.load()
    _req(X, torch.bfloat16, "X")
    D = X.shape[-1]
    out = torch.empty_like(X) if out is None else out
    check(lib.x2i_ln_affine_bf16(_p(X), _p(out), X.numel() // D, D, _p(weight), _p(bias), eps, _stream()), "ln_affine")
    return out


def skinny_linear(X, W, bias=None, out=None, act_in=ACT_NONE, act_out=ACT_NONE, accumulate=False, ldy=None):
    lib = _lib.load()
    _req(W, torch.bfloat16, "W")
    B, K = X.shape
    N = W.shape[0]
    if X.dtype not in (torch.float32, torch.bfloat16):
        raise _lib.X2IError("skinny_linear: X must be f32 or bf16")
    if out is None:
        out = torch.empty((B, N), device=X.device, dtype=torch.float32)
    ldy = out.stride(0) if ldy is None else ldy
    check(lib.x2i_skinny_linear(_p(X), 1 if X.dtype == torch.bfloat16 else 0, _p(W), _p(bias), _p(out), ldy, B, N, K, act_in,
                                act_out, 1 if accumulate else 0, _stream()), "skinny_linear")
    return out


def skinny_linear_grouped(X, W, bias=None, *, rows, out=None, act_in=ACT_NONE, act_out=ACT_NONE):
    """`G` independent skinny linears in one launch (x2i_skinny_linear_grouped): W bf16 [G, N, K], bias bf16 [G, N]; X f32 / bf16 [G * rows, K] (group
    g reads rows g * rows ..) or [rows, K] (one input for every group); returns f32 [G * rows, N]."""
    lib = _lib.load()
    _req(W, torch.bfloat16, "W")
    G, N, K = W.shape
    if not W.is_contiguous() or (bias is not None and not bias.is_contiguous()) or not X.is_contiguous():
        raise _lib.X2IError("skinny_linear_grouped: X, W and bias must be contiguous")
    if X.dtype not in (torch.float32, torch.bfloat16):
        raise _lib.X2IError("skinny_linear_grouped: X must be f32 or bf16")
    if X.shape[0] not in (rows, G * rows) or X.shape[1] != K:
        raise _lib.X2IError("skinny_linear_grouped: X must be [rows, K] or [G * rows, K]")
    x_gs = rows * K if X.shape[0] == G * rows and G > 1 else 0
    if out is None:
        out = torch.empty((G * rows, N), device=X.device, dtype=torch.float32)
    check(lib.x2i_skinny_linear_grouped(_p(X), 1 if X.dtype == torch.bfloat16 else 0, x_gs, _p(W), _p(bias), _p(out), out.stride(0), G, rows, N, K,
                                        act_in, act_out, 0, _stream()), "skinny_linear_grouped")
    return out


def timestep_sinusoid(t, dim, round_bf16=False):
    lib = _lib.load()
    _req(t, torch.float32, "t")
    out = torch.empty((t.shape[0], dim), device=t.device, dtype=torch.float32)
    check(lib.x2i_timestep_sinusoid(_p(t), _p(out), t.shape[0], dim, 1 if round_bf16 else 0, _stream()), "timestep_sinusoid")
    return out


def rope_table(ids, axes_dim, theta=10000.0):
    """ids: f32 [S,3] -> (cos, sin) f32 [S, sum(axes_dim)] (FluxPosEmbed; float64 inside the kernel)."""
    lib = _lib.load()
    ids = ids.to(torch.float32).contiguous()
    _req(ids, torch.float32, "ids")
    d0, d1, d2 = (list(axes_dim) + [0, 0, 0])[:3]
    S = ids.shape[0]
    cos = torch.empty((S, d0 + d1 + d2), device=ids.device, dtype=torch.float32)
    sin = torch.empty_like(cos)
    check(lib.x2i_rope_table_f32(_p(ids), S, d0, d1, d2, float(theta), _p(cos), _p(sin), _stream()), "rope_table")
    return cos, sin


def euler_step_(x, eps, dt):
    """x <- bf16(f32(x) + dt * f32(eps)) in place; dt is a 1-element f32 DEVICE tensor."""
    lib = _lib.load()
    _req(x, torch.bfloat16, "x")
    _req(eps, torch.bfloat16, "eps")
    _req(dt, torch.float32, "dt")
    check(lib.x2i_euler_step_bf16(_p(x), _p(eps), x.numel(), _p(dt), _stream()), "euler_step")
    return x


def proj_conv5x5(x, w, bias, out=None):
    lib = _lib.load()
    _req(x, torch.bfloat16, "x")
    _req(w, torch.float32, "w")
    B, Cc, S, H = x.shape
    out = torch.empty((B, S, H), device=x.device, dtype=torch.bfloat16) if out is None else out
    check(lib.x2i_proj_conv5x5_bf16(_p(x), _p(w), _p(bias), _p(out), B, Cc, S, H, _stream()), "proj_conv5x5")
    return out


def proj_conv5x5_pack(w):
    """f32 taps [C,25] -> banded-Toeplitz MFMA fragments, bf16 [C,5,64,8] (include/x2i.h: x2i_proj_conv5x5_pack)."""
    lib = _lib.load()
    _req(w, torch.float32, "w")
    Cc = w.shape[0]
    table = torch.empty((Cc, 5, 64, 8), device=w.device, dtype=torch.bfloat16)
    check(lib.x2i_proj_conv5x5_pack(_p(w), _p(table), Cc, _stream()), "proj_conv5x5_pack")
    return table


def proj_conv5x5_packed(x, table, bias, out=None):
    """Conv2d(C->1, 5, pad 2) over the (S,H) plane on the matrix cores (x2i_proj_conv5x5_packed_bf16)."""
    lib = _lib.load()
    _req(x, torch.bfloat16, "x")
    _req(table, torch.bfloat16, "table")
    B, Cc, S, H = x.shape
    if tuple(table.shape) != (Cc, 5, 64, 8):
        raise ValueError(f"proj_conv5x5_packed: table {tuple(table.shape)} does not match C={Cc}")
    out = torch.empty((B, S, H), device=x.device, dtype=torch.bfloat16) if out is None else out
    check(lib.x2i_proj_conv5x5_packed_bf16(_p(x), _p(table), _p(bias), _p(out), B, Cc, S, H, _stream()), "proj_conv5x5_packed")
    return out


def proj_layer_mean(x, scale, out=None):
    lib = _lib.load()
    _req(x, torch.bfloat16, "x")
    B, Cc, S, H = x.shape
    out = torch.empty((B, S, H), device=x.device, dtype=torch.bfloat16) if out is None else out
    check(lib.x2i_proj_layer_mean_bf16(_p(x), _p(scale), _p(out), B, Cc, S * H, _stream()), "proj_layer_mean")
    return out


def seq_mean(x):
    lib = _lib.load()
    _req(x, torch.float32, "x")
    B, S, N = x.shape
    out = torch.empty((B, N), device=x.device, dtype=torch.float32)
    check(lib.x2i_seq_mean_f32(_p(x), _p(out), B, S, N, _stream()), "seq_mean")
    return out


def to_bf16(x):
    lib = _lib.load()
    _req(x, torch.float32, "x")
    out = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    check(lib.x2i_cast_f32_to_bf16(_p(x), _p(out), x.numel(), _stream()), "cast")
    return out


def to_f32(x):
    lib = _lib.load()
    _req(x, torch.bfloat16, "x")
    out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    check(lib.x2i_cast_bf16_to_f32(_p(x), _p(out), x.numel(), _stream()), "cast")
    return out


# ---------------------------------------------------------------------------------------------------- ControlNeXt ops
from ._lib import ACT_RELU, ConvDesc  # noqa: E402


def conv2d_nhwc(x, w_packed, bias, H, W, Cin, Cout, KH, KW, stride, pad, out=None, act=ACT_NONE, bias2=None, res=None,
                c_offset=0, c_batch_stride=None, ldc=None, res_offset=0, res_batch_stride=None, ldr=None, up=False, pad_w=None, B=None,
                a_batch_stride=None, a_offset=0, out_w=None, out_h=None, out_row_pitch=0, moments=None, moments_accumulate=False, w_group=0):
    """x: bf16 NHWC [B,H,W,Cin]; w_packed: bf16 [Cout, KH*KW*Cin] (ky,kx,ci order).  Returns NHWC [B,OH,OW,Cout].
    w_group > 0: GROUPED weights -- w_packed [groups, Cout, KH*KW*Cin], bias [groups, Cout], batch item b uses group b // w_group (the models of a
    ControlNeXt bank in one launch; include/x2i.h: x2i_gemm_args.w_group).
    pad_w: padding along W when it differs from `pad` (along H).  B / a_batch_stride / a_offset: the input is a window of `H` rows of a
    larger NHWC tensor (elements between two images / in front of the window).  up: True / 1 = nearest x2 upsampling in front of the conv,
    2 = along H only; out_w / out_h: number of output columns / rows when the right-hand / bottom padding differs from pad_w / pad;
    out_row_pitch: elements between two output rows when they are not dense; moments: f32 [B, Cout, 2] that receives (or, with
    moments_accumulate, is added) the (sum, sum of squares) of the bf16 outputs per channel QUAD (entry c % 4 == 0; the others are zero) --
    the statistics input of groupnorm_nhwc_from_moments without pre_add, written by the conv epilogue (include/x2i.h: x2i_conv_desc)."""
    lib = _lib.load()
    _req(x, torch.bfloat16, "x")
    _req(w_packed, torch.bfloat16, "w")
    B = x.shape[0] if B is None else B
    up = int(up)
    uh, uw = (2 if up else 1), (2 if up == 1 else 1)  # nearest-neighbour x2 upsampling fused in front of the conv
    pw = pad if pad_w is None else pad_w
    OH = (H * uh + 2 * pad - KH) // stride + 1 if out_h is None else out_h
    OW = (W * uw + 2 * pw - KW) // stride + 1 if out_w is None else out_w
    if out is None:
        out = torch.empty((B, OH, OW, Cout), device=x.device, dtype=torch.bfloat16)
    a = GemmArgs()
    a.A = x.data_ptr() + 2 * a_offset
    a.a_batch_stride = H * W * Cin if a_batch_stride is None else a_batch_stride
    a.lda = Cin
    a.W = w_packed.data_ptr()
    a.ldw = KH * KW * Cin
    a.bias = bias.data_ptr() if bias is not None else None
    a.C = out.data_ptr() + 2 * c_offset
    a.c_batch_stride = OH * OW * Cout if c_batch_stride is None else c_batch_stride
    a.ldc = Cout if ldc is None else ldc
    a.C2 = None
    a.gate = None
    a.res = (res.data_ptr() + 2 * res_offset) if res is not None else None
    a.res_batch_stride = (OH * OW * Cout if res_batch_stride is None else res_batch_stride)
    a.ldr = (Cout if ldr is None else ldr)
    a.bias2 = bias2.data_ptr() if bias2 is not None else None
    a.bias2_batch_stride = bias2.stride(0) if bias2 is not None else 0
    a.M, a.N, a.K, a.batch = OH * OW, Cout, KH * KW * Cin, B
    a.act = act
    a.out_f32 = 0
    a.w_batch_stride = Cout * KH * KW * Cin if w_group else 0
    a.w_group = w_group
    d = ConvDesc(H, W, Cin, KH, KW, stride, pad, up, 0 if pad_w is None else pad_w + 1, 0 if out_w is None else out_w,
                 0 if out_h is None else out_h, out_row_pitch)
    if moments is not None:
        _req(moments, torch.float32, "moments")
        n = int(lib.x2i_conv_moments_scratch_floats(a.M, a.N, a.batch))
        key = (x.device, "conv_moments")
        if key not in _gn_scratch or _gn_scratch[key].numel() < n:
            _gn_scratch[key] = torch.empty(n, device=x.device, dtype=torch.float32)
        d.moments, d.moments_scratch, d.moments_accumulate = moments.data_ptr(), _gn_scratch[key].data_ptr(), 1 if moments_accumulate else 0
    check(lib.x2i_conv2d_nhwc_bf16(C.byref(a), C.byref(d), _stream()), "conv2d_nhwc")
    return out


def conv3x3_narrow(x, w_packed, bias, Cout, out=None, ldy=4):
    """Conv2d(Cin -> Cout <= 4, 3 x 3, stride 1, padding 1) on NHWC bf16 [B,H,W,Cin] (include/x2i.h: x2i_conv3x3_narrow_bf16): returns
    [B,H,W,ldy] whose channels 0 .. 3 are written (zeros behind Cout).  w_packed: bf16 [Cout, 9 * Cin] in (ky, kx, ci) order."""
    lib = _lib.load()
    _req(x, torch.bfloat16, "x")
    _req(w_packed, torch.bfloat16, "w")
    B, H, W, Cin = x.shape
    if out is None:
        out = torch.empty((B, H, W, ldy), device=x.device, dtype=torch.bfloat16)
    check(lib.x2i_conv3x3_narrow_bf16(_p(x), _p(w_packed), _p(bias), _p(out), B, H, W, Cin, Cout, ldy, _stream()), "conv3x3_narrow")
    return out


def conv_stem(x_nhwc, w, bias, Cout):
    """Conv2d(3->Cout, k3, s2, p1): x bf16 NHWC [B,H,W,3]; w f32 [Cout,3,3,3] (ky,kx,ci)."""
    lib = _lib.load()
    _req(x_nhwc, torch.bfloat16, "x")
    _req(w, torch.float32, "w")
    B, H, W, _ = x_nhwc.shape
    out = torch.empty((B, H // 2, W // 2, Cout), device=x_nhwc.device, dtype=torch.bfloat16)
    check(lib.x2i_conv_stem_bf16(_p(x_nhwc), _p(w), _p(bias), _p(out), B, H, W, Cout, _stream()), "conv_stem")
    return out


_gn_scratch = {}


def groupnorm_nhwc(x, weight, bias, G, eps, act=ACT_NONE, pre_add=None, post_add=None, out=None, w_group=0):
    """GroupNorm on NHWC bf16 [B, ..., C] with fused pre-add (f32 [B,C]), activation and post-add (bf16 like x).
    w_group > 0: weight / bias are [groups, C] and item b uses row b // w_group (x2i_groupnorm_nhwc_grouped_bf16)."""
    lib = _lib.load()
    _req(x, torch.bfloat16, "x")
    B, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * Cc)
    out = torch.empty_like(x) if out is None else out
    n = lib.x2i_groupnorm_scratch_floats(B, G)
    key = (x.device, n)
    if key not in _gn_scratch:
        _gn_scratch[key] = torch.empty(n, device=x.device, dtype=torch.float32)
    check(lib.x2i_groupnorm_nhwc_grouped_bf16(_p(x), _p(out), B, HW, Cc, G, _p(weight), _p(bias), w_group, eps, act, _p(pre_add), _p(post_add),
                                              _p(_gn_scratch[key]), _stream()), "groupnorm_nhwc")
    return out


def groupnorm_moments(x):
    """Per-channel moments f32 [B, C, 2] = (sum x, sum x^2) over the pixels of an NHWC bf16 tensor (x2i_groupnorm_moments_f32): cached
    by the caller, they give groupnorm_nhwc_from_moments its statistics for any per-channel pre_add without another pass over x."""
    lib = _lib.load()
    _req(x, torch.bfloat16, "x")
    B, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * Cc)
    mom = torch.empty((B, Cc, 2), device=x.device, dtype=torch.float32)
    scratch = torch.empty(int(lib.x2i_groupnorm_moments_scratch_floats(B, Cc)), device=x.device, dtype=torch.float32)
    check(lib.x2i_groupnorm_moments_f32(_p(x), B, HW, Cc, _p(mom), _p(scratch), _stream()), "groupnorm_moments")
    return mom


def groupnorm_nhwc_from_moments(x, moments, weight, bias, G, eps, act=ACT_NONE, pre_add=None, post_add=None, out=None, w_group=0):
    """groupnorm_nhwc(x, ..., pre_add=...) with the statistics derived from groupnorm_moments(x) instead of a pass over x."""
    lib = _lib.load()
    _req(x, torch.bfloat16, "x")
    _req(moments, torch.float32, "moments")
    B, Cc = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * Cc)
    out = torch.empty_like(x) if out is None else out
    n = lib.x2i_groupnorm_scratch_floats(B, G)
    key = (x.device, n)
    if key not in _gn_scratch:
        _gn_scratch[key] = torch.empty(n, device=x.device, dtype=torch.float32)
    check(lib.x2i_groupnorm_nhwc_from_moments_grouped_bf16(_p(x), _p(out), B, HW, Cc, G, _p(weight), _p(bias), w_group, eps, act, _p(moments),
                                                           _p(pre_add), _p(post_add), _p(_gn_scratch[key]), _stream()), "groupnorm_nhwc_from_moments")
    return out


def softmax_rows_(x, scale=1.0):
    """In-place softmax(scale * x) over the last dimension of a contiguous bf16 tensor."""
    lib = _lib.load()
    _req(x, torch.bfloat16, "x")
    cols = x.shape[-1]
    check(lib.x2i_softmax_rows_bf16(_p(x), x.numel() // cols, cols, scale, _stream()), "softmax_rows")
    return x


# ---------------------------------------------------------------------------------------------------- fp8 (e4m3) path
from ._lib import Fp8Desc  # noqa: E402

FP8 = torch.float8_e4m3fn
E4M3_MAX = 448.0


def quantize_rows_fp8(x, static_inv_scale=None):
    """bf16 [rows, cols] -> (e4m3 [rows, cols], f32 [rows] scales) with scale = amax / 448 per row; or, with
    `static_inv_scale`, (e4m3, None) with y = sat(x * static_inv_scale)."""
    lib = _lib.load()
    _req(x, torch.bfloat16, "x")
    cols = x.shape[-1]
    rows = x.numel() // cols
    y = torch.empty(x.shape, device=x.device, dtype=FP8)
    scale = None if static_inv_scale is not None else torch.empty((rows,), device=x.device, dtype=torch.float32)
    check(lib.x2i_quantize_rows_fp8(_p(x), rows, cols, x.stride(-2) if x.dim() > 1 else cols, _p(y), cols, _p(scale),
                                    1.0 if static_inv_scale is None else float(static_inv_scale), _stream()), "quantize_rows_fp8")
    return y, scale


def gemm_fp8(A8, W8, bias=None, out=None, *, M=None, N=None, K=None, batch=1, a_batch_stride=0, lda=None, a_offset=0, a_scale=None,
             a_scale_batch_stride=0, w_scale=None, alpha=1.0, c_batch_stride=0, ldc=None, c_offset=0, act=ACT_NONE, gate=None,
             gate_batch_stride=0, res=None, res_batch_stride=0, ldr=None, res_offset=0, out_fp8=False, out_inv_scale=1.0, _act2=0, _bias2=None):
    """C = epi(a_scale[m] * w_scale[n] * alpha * A8 W8^T) on e4m3 operands (include/x2i.h: x2i_gemm_fp8)."""
    lib = _lib.load()
    _req(A8, FP8, "A8")
    _req(W8, FP8, "W8")
    N = W8.shape[-2] if N is None else N
    K = W8.shape[-1] if K is None else K
    if M is None:
        M = A8.numel() // A8.shape[-1]
    lda = A8.shape[-1] if lda is None else lda
    if out is None:
        out = torch.empty((batch, M, N) if batch > 1 else (M, N), device=A8.device, dtype=FP8 if out_fp8 else torch.bfloat16)
        if batch > 1:
            c_batch_stride = M * N
    ldc = N if ldc is None else ldc
    esz = 1 if out_fp8 else 2
    a = GemmArgs()
    a.A = A8.data_ptr() + a_offset
    a.a_batch_stride, a.lda = a_batch_stride, lda
    a.W, a.ldw = W8.data_ptr(), W8.stride(-2)
    a.bias = bias.data_ptr() if bias is not None else None
    a.C = out.data_ptr() + c_offset * esz
    a.c_batch_stride, a.ldc = c_batch_stride, ldc
    a.C2, a.act2 = None, 0
    a.gate = gate.data_ptr() if gate is not None else None
    a.gate_batch_stride = gate_batch_stride
    a.res = (res.data_ptr() + res_offset * 2) if res is not None else None
    a.res_batch_stride = res_batch_stride
    a.ldr = ldc if ldr is None else ldr
    a.bias2, a.bias2_batch_stride, a.w_batch_stride = (_bias2.data_ptr() if _bias2 is not None else None), 0, 0   # (_act2 / _bias2: tools only)
    a.act2 = _act2
    a.M, a.N, a.K, a.batch = M, N, K, batch
    a.act, a.out_f32 = act, 0
    _set_ws(a)
    f = Fp8Desc()
    f.a_scale = a_scale.data_ptr() if a_scale is not None else None
    f.a_scale_batch_stride = a_scale_batch_stride
    f.w_scale = w_scale.data_ptr() if w_scale is not None else None
    f.alpha, f.out_fp8, f.out_inv_scale = float(alpha), 1 if out_fp8 else 0, float(out_inv_scale)
    check(lib.x2i_gemm_fp8(C.byref(a), C.byref(f), _stream()), "gemm_fp8")
    return out


def gemm_qkv_fp8(A8, W8, bias, Q, K, VT, norm_q, norm_k, cos, sin, *, M, H, Spad, tok_off, rows_per_sample, batch=1,
                 a_batch_stride=0, lda=None, a_offset=0, a_scale=None, a_scale_batch_stride=0, w_scale=None, alpha=1.0, eps=1e-6, q_scale=1.0,
                 vt_perm=False):
    """gemm_qkv on e4m3 operands (include/x2i.h: x2i_gemm_qkv_fp8): dequantised accumulators, then the same fused epilogue."""
    lib = _lib.load()
    _req(A8, FP8, "A8")
    _req(W8, FP8, "W8")
    a = GemmArgs()
    a.A = A8.data_ptr() + a_offset
    a.a_batch_stride = a_batch_stride
    a.lda = A8.shape[-1] if lda is None else lda
    a.W, a.ldw = W8.data_ptr(), W8.stride(-2)
    a.bias = bias.data_ptr() if bias is not None else None
    a.C = None
    a.c_batch_stride, a.ldc = 0, 3 * H * 128
    a.C2, a.act2, a.gate, a.gate_batch_stride, a.res, a.res_batch_stride, a.ldr = None, 0, None, 0, None, 0, 0
    a.bias2, a.bias2_batch_stride, a.w_batch_stride = None, 0, 0
    a.M, a.N, a.K, a.batch = M, 3 * H * 128, W8.shape[-1], batch
    a.act, a.out_f32 = ACT_NONE, 0
    _set_ws(a)
    f = Fp8Desc()
    f.a_scale = a_scale.data_ptr() if a_scale is not None else None
    f.a_scale_batch_stride = a_scale_batch_stride
    f.w_scale = w_scale.data_ptr() if w_scale is not None else None
    f.alpha, f.out_fp8, f.out_inv_scale = float(alpha), 0, 1.0
    q = QkvDesc()
    q.norm_q, q.norm_k, q.cos, q.sin = norm_q.data_ptr(), norm_k.data_ptr(), cos.data_ptr(), (sin.data_ptr() if sin is not None else None)   # sin=None: `cos` is the pair-form table (rope_pairs)
    q.Q, q.K, q.VT = Q.data_ptr(), K.data_ptr(), VT.data_ptr()
    q.H, q.Spad, q.tok_off, q.rows_per_sample, q.eps, q.q_scale = H, Spad, tok_off, rows_per_sample, eps, q_scale
    q.vt_perm = 1 if vt_perm else 0
    check(lib.x2i_gemm_qkv_fp8(C.byref(a), C.byref(f), C.byref(q), _stream()), "gemm_qkv_fp8")


def ln_modulate_fp8(X, Y, Y8, row_scale, B, S, D, S0, shift0, scale0, shift1, scale1, mod_bs, eps=1e-6, x_bs=None, ldx=None,
                    y_bs=None, ldy=None, x_offset=0, y_offset=0, y8_bs=None, ldy8=None, y8_offset=0):
    """ln_modulate with an extra e4m3 output + per-row scales (Y may be None)."""
    lib = _lib.load()
    ldx = D if ldx is None else ldx
    ldy = D if ldy is None else ldy
    ldy8 = D if ldy8 is None else ldy8
    x_bs = S * ldx if x_bs is None else x_bs
    y_bs = S * ldy if y_bs is None else y_bs
    y8_bs = S * ldy8 if y8_bs is None else y8_bs
    yp = C.c_void_p(Y.data_ptr() + 2 * y_offset) if Y is not None else C.c_void_p(0)
    check(lib.x2i_ln_modulate_fp8(C.c_void_p(X.data_ptr() + 2 * x_offset), x_bs, ldx, yp, y_bs, ldy,
                                  C.c_void_p(Y8.data_ptr() + y8_offset), y8_bs, ldy8, _p(row_scale), B, S, D, S0, _p(shift0), _p(scale0),
                                  _p(shift1), _p(scale1), mod_bs, eps, _stream()), "ln_modulate_fp8")
    return Y8


def attention_e4m3out(Q, K, VT, out8, B, H, S, Spad, ldo, o_batch_stride, scale, o_offset=0, out_inv_scale=1.0):
    """attention() writing e4m3 (include/x2i.h: x2i_attention_e4m3out); ldo / o_batch_stride / o_offset in bytes."""
    lib = _lib.load()
    check(lib.x2i_attention_e4m3out(_p(Q), _p(K), _p(VT), C.c_void_p(out8.data_ptr() + o_offset), B, H, S, Spad, ldo, o_batch_stride,
                                    scale, out_inv_scale, _stream()), "attention_e4m3out")
    return out8


def gated_residual_(X, T, gate, B, S, D, x_bs, ldx, t_bs, ldt, gate_bs, x_offset=0, t_offset=0):
    """X <- bf16(X + gate[b] * T) in place (include/x2i.h: x2i_gated_residual_bf16); offsets / strides in elements."""
    lib = _lib.load()
    check(lib.x2i_gated_residual_bf16(C.c_void_p(X.data_ptr() + 2 * x_offset), x_bs, ldx, C.c_void_p(T.data_ptr() + 2 * t_offset), t_bs, ldt,
                                      _p(gate), gate_bs, B, S, D, _stream()), "gated_residual")
    return X


# ---------------------------------------------------------------------------------------------------------------------
# N4: backward kernels of the attention-distillation step (include/x2i.h, csrc/train.hip)
def transpose(x, out=None, *, batch=1, R=None, C=None, in_bs=0, ld_in=None, out_bs=0, ld_out=None, in_offset=0, out_offset=0):
    """out[z][c][r] = in[z][r][c] (bf16).  Defaults: x is a contiguous [..., R, C] tensor, leading dims are the batch."""
    lib = _lib.load()
    _req(x, torch.bfloat16, "x")
    if R is None:
        R, C = x.shape[-2], x.shape[-1]
        batch = x.numel() // (R * C)
        in_bs, ld_in = R * C, C
    ld_in = C if ld_in is None else ld_in
    if out is None:
        out = torch.empty((batch, C, R) if batch > 1 else (C, R), device=x.device, dtype=torch.bfloat16)
        out_bs, ld_out = C * R, R
    ld_out = R if ld_out is None else ld_out
    check(lib.x2i_transpose_bf16(_off(x, in_offset), in_bs, ld_in, _off(out, out_offset),
                                 out_bs, ld_out, batch, R, C, _stream()), "transpose")
    return out


def _off(t, elems):
    return C.c_void_p(t.data_ptr() + elems * t.element_size())


def softmax_pad_(x, nz, Rt, Rv, Ct, Cv, scale, ld=None):
    check(_lib.load().x2i_softmax_pad_bf16(_p(x), Ct if ld is None else ld, nz, Rt, Rv, Ct, Cv, float(scale), _stream()), "softmax_pad")
    return x


def softmax_bwd_(P, dP, nz, Rt, Rv, Ct, Cv, scale, ld=None):
    check(_lib.load().x2i_softmax_bwd_bf16(_p(P), _p(dP), Ct if ld is None else ld, nz, Rt, Rv, Ct, Cv, float(scale), _stream()), "softmax_bwd")
    return dP


def reduce_rows(partial, out, *, np_, len_, nz=1, in_zs=0, in_ps=None, out_zs=0, accumulate=False, alpha=1.0, in_offset=0, out_offset=0):
    check(_lib.load().x2i_reduce_rows_f32(_off(partial, in_offset), in_zs, np_, len_ if in_ps is None else in_ps, _off(out, out_offset), out_zs, nz,
                                          len_, 1 if accumulate else 0, float(alpha), _stream()), "reduce_rows")
    return out


def ln_mod_bwd(X, dY, mult, dXin, dXout, partial, *, B, S, D, R, mult_is_scale=True, mult_bs=0, x_bs=None, ldx=None, dy_bs=None, ldy=None,
               dx_bs=None, lddx=None, x_offset=0, dy_offset=0, dx_offset=0, eps=1e-6):
    """Backward of LayerNorm(no affine) * mult + shift on S rows per sample; partial f32 [B, ceil(S/R), 2, D]."""
    ldx = D if ldx is None else ldx
    ldy = D if ldy is None else ldy
    lddx = D if lddx is None else lddx
    x_bs = S * ldx if x_bs is None else x_bs
    dy_bs = S * ldy if dy_bs is None else dy_bs
    dx_bs = S * lddx if dx_bs is None else dx_bs
    check(_lib.load().x2i_ln_mod_bwd_bf16(_off(X, x_offset), x_bs, ldx, _off(dY, dy_offset), dy_bs, ldy, _p(mult), mult_bs,
                                          1 if mult_is_scale else 0, _off(dXin, dx_offset) if dXin is not None else C.c_void_p(0),
                                          _off(dXout, dx_offset), dx_bs, lddx, B, S, D, R, _p(partial), float(eps), _stream()), "ln_mod_bwd")


def gate_bwd(dX, T, gate, G, dT, partial, *, B, S, D, R, gate_bs=0, dx_bs=None, lddx=None, t_bs=None, ldt=None, g_bs=None, ldg=None,
             dt_bs=None, lddt=None, dx_offset=0, t_offset=0, g_offset=0, dt_offset=0):
    """dT = gate * dX (+ G); partial f32 [B, ceil(S/R), D] = sum_rows dX * T."""
    lddx = D if lddx is None else lddx
    ldt = D if ldt is None else ldt
    ldg = D if ldg is None else ldg
    lddt = D if lddt is None else lddt
    dx_bs = S * lddx if dx_bs is None else dx_bs
    t_bs = S * ldt if t_bs is None else t_bs
    g_bs = S * ldg if g_bs is None else g_bs
    dt_bs = S * lddt if dt_bs is None else dt_bs
    check(_lib.load().x2i_gate_bwd_bf16(_off(dX, dx_offset), dx_bs, lddx, _off(T, t_offset) if T is not None else C.c_void_p(0), t_bs, ldt,
                                        _p(gate), gate_bs, _off(G, g_offset) if G is not None else C.c_void_p(0), g_bs, ldg,
                                        _off(dT, dt_offset), dt_bs, lddt, B, S, D, R, _p(partial), _stream()), "gate_bwd")


def act_bwd_(dA, pre, act, *, rows=None, cols=None, ldd=None, ldp=None, d_offset=0, p_offset=0):
    """dA <- dA * act'(pre); bf16 strided matrices, or contiguous f32 tensors."""
    f32 = dA.dtype == torch.float32
    cols = dA.shape[-1] if cols is None else cols
    rows = dA.numel() // dA.shape[-1] if rows is None else rows
    ldd = cols if ldd is None else ldd
    ldp = cols if ldp is None else ldp
    check(_lib.load().x2i_act_bwd(_off(dA, d_offset), ldd, _off(pre, p_offset), ldp, rows, cols, act, 1 if f32 else 0, _stream()), "act_bwd")
    return dA


def qkv_split_bwd(qkv0, qkv1, ld0, ld1, d0, d1, ldd0, ldd1, B, S, S0, H, nq0, nk0, nq1, nk1, cos, sin, dQ, dK, dV, Spad, eps=1e-6):
    check(_lib.load().x2i_qkv_split_bwd_bf16(_p(qkv0), _p(qkv1), ld0, ld1, _p(d0), _p(d1), ldd0, ldd1, B, S, S0, H, _p(nq0), _p(nk0), _p(nq1),
                                             _p(nk1), _p(cos), _p(sin), _p(dQ), _p(dK), _p(dV), Spad, float(eps), _stream()), "qkv_split_bwd")


def skinny_linear_bwd(dy, W, *, chunk=1024):
    """dx [B, K] f32 = dy [B, N] f32 @ W [N, K] bf16 (B <= 8): two-stage, deterministic."""
    lib = _lib.load()
    _req(dy, torch.float32, "dy")
    _req(W, torch.bfloat16, "W")
    B, N = dy.shape
    K = W.shape[1]
    nchunk = (N + chunk - 1) // chunk
    partial = torch.empty((nchunk, B, K), device=dy.device, dtype=torch.float32)
    check(lib.x2i_skinny_linear_bwd(_p(dy), dy.stride(0), _p(W), W.stride(0), _p(partial), B, N, K, chunk, _stream()), "skinny_linear_bwd")
    out = torch.empty((B, K), device=dy.device, dtype=torch.float32)
    reduce_rows(partial, out, np_=nchunk, len_=B * K)
    return out


def kd_loss_rows(teacher, student, grad, row_loss, *, rows, D, temperature, loss_scale, ldt=None, lds=None, ldg=None):
    check(_lib.load().x2i_kd_loss_bf16(_p(teacher), D if ldt is None else ldt, _p(student), D if lds is None else lds, _p(grad),
                                       D if ldg is None else ldg, _p(row_loss), rows, D, float(temperature), float(loss_scale), _stream()),
          "kd_loss")


def zero_if_nonfinite_(g, term):
    check(_lib.load().x2i_zero_if_nonfinite_bf16(_p(g), g.numel(), _p(term), _stream()), "zero_if_nonfinite")
    return g


def conv5x5_wgrad(x, dy):
    """dw f32 [C, 25] of Conv2d(C->1, 5, pad 2) over the (S, H) plane: x bf16 [B,C,S,H], dy bf16 [B,S,H] (x2i_proj_conv5x5_wgrad)."""
    lib = _lib.load()
    B, Cc, S, H = x.shape
    nchunk = (S + 15) // 16
    partial = torch.empty((Cc, B, nchunk, 25), device=x.device, dtype=torch.float32)
    check(lib.x2i_proj_conv5x5_wgrad(_p(x), _p(dy), _p(partial), B, Cc, S, H, _stream()), "conv5x5_wgrad")
    out = torch.empty((Cc, 25), device=x.device, dtype=torch.float32)
    reduce_rows(partial, out, np_=B * nchunk, len_=25, nz=Cc, in_zs=B * nchunk * 25, in_ps=25, out_zs=25)
    return out


def plane_dot(x, dy, alpha=1.0, nchunk=64):
    """out f32 [C] = alpha * sum_{b,i} dy[b][i] * x[b][c][i]   (x bf16 [B,C,S,H], dy bf16 [B,S,H])."""
    lib = _lib.load()
    B, Cc = x.shape[0], x.shape[1]
    plane = x[0, 0].numel()
    partial = torch.empty((Cc, B, nchunk), device=x.device, dtype=torch.float32)
    check(lib.x2i_plane_dot_bf16(_p(x), _p(dy), _p(partial), B, Cc, plane, nchunk, _stream()), "plane_dot")
    out = torch.empty((Cc,), device=x.device, dtype=torch.float32)
    reduce_rows(partial, out, np_=B * nchunk, len_=1, nz=Cc, in_zs=B * nchunk, in_ps=1, out_zs=1, alpha=alpha)
    return out


def sum_all(x, squares=False, out=None, accumulate=False, nblocks=256):
    """out f32 [1] (+)= sum x or sum x^2 over every element of x (f32 or bf16), two-stage."""
    lib = _lib.load()
    partial = torch.empty((nblocks,), device=x.device, dtype=torch.float32)
    check(lib.x2i_sum_partials(_p(x), 1 if x.dtype == torch.bfloat16 else 0, x.numel(), 1 if squares else 0, _p(partial), nblocks, _stream()), "sum")
    out = torch.zeros((1,), device=x.device, dtype=torch.float32) if out is None else out
    reduce_rows(partial, out, np_=nblocks, len_=1, in_ps=1, accumulate=accumulate)
    return out


def clip_coef(sumsq, max_norm):
    out = torch.empty((2,), device=sumsq.device, dtype=torch.float32)
    check(_lib.load().x2i_clip_coef_f32(_p(sumsq), float(max_norm), _p(out), _stream()), "clip_coef")
    return out


def adamw_(p, g, m, v, *, lr, beta1, beta2, eps, weight_decay, step, coef=None):
    check(_lib.load().x2i_adamw_bf16(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                                     1.0 - beta1 ** step, 1.0 - beta2 ** step, _p(coef), _stream()), "adamw")
    return p


def attention_bwd(Q, K, V, QT, KT, dOh, dOT, lse2, Dv, dQ, dK, dV, B, H, S, Spad, scale, have_lse=False):
    """Fused attention backward (include/x2i.h: x2i_attention_bwd_bf16); have_lse: lse2 was written by attention_lse()."""
    check(_lib.load().x2i_attention_bwd_bf16(_p(Q), _p(K), _p(V), _p(QT), _p(KT), _p(dOh), _p(dOT), _p(lse2), _p(Dv), _p(dQ), _p(dK), _p(dV), B, H, S,
                                             Spad, float(scale), 1 if have_lse else 0, _stream()), "attention_bwd")


def attention_lse(Q, K, VT, out, lse2, B, H, S, Spad, ldo, o_batch_stride, scale, o_offset=0):
    """attention() that also writes the log2-sum-exp statistics lse2 f32 [B,H,Spad] (x2i_attention_lse_bf16)."""
    check(_lib.load().x2i_attention_lse_bf16(_p(Q), _p(K), _p(VT), C.c_void_p(out.data_ptr() + o_offset * 2), _p(lse2), B, H, S, Spad, ldo,
                                             o_batch_stride, float(scale), _stream()), "attention_lse")


def attention_bwd_prep(dO, O, Dv, B, H, S, Spad, *, do_bs, lddo, o_bs, ldo, do_offset=0, o_offset=0):
    check(_lib.load().x2i_attention_bwd_prep_bf16(_off(dO, do_offset), do_bs, lddo, _off(O, o_offset), o_bs, ldo, _p(Dv), B, H, S, Spad, _stream()),
          "attention_bwd_prep")
