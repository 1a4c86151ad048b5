"""ctypes binding of libx2i_hip.so (C ABI: include/x2i.h).

The product path has NO CPU or eager-PyTorch fallback: if the HIP library cannot be loaded, every entry point
raises.  (The CPU oracle lives in oracle/ and is test infrastructure only.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# X2I_LIB_VARIANT=<name> (read once, at import) makes tools/ load libx2i_hip_<name>.so instead: "ablate" = the measurement-only build
# (ablation kernels, k-half-unit GEMM form), anything else = a library built from another commit for a same-box A/B.  The product
# package never sets it.
_VARIANT = os.environ.get("X2I_LIB_VARIANT", "")
LIB_PATH = os.path.join(_HERE, "libx2i_hip_%s.so" % _VARIANT if _VARIANT else "libx2i_hip.so")

ACT_NONE, ACT_GELU_TANH, ACT_GELU_ERF, ACT_SILU, ACT_RELU = 0, 1, 2, 3, 4
ABI_VERSION = 5  # include/x2i.h: X2I_ABI_VERSION


class X2IError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("a_batch_stride", C.c_int64), ("lda", C.c_int32),
        ("W", C.c_void_p), ("ldw", C.c_int32),
        ("bias", C.c_void_p),
        ("C", C.c_void_p), ("c_batch_stride", C.c_int64), ("ldc", C.c_int32),
        ("C2", C.c_void_p), ("act2", C.c_int32),
        ("gate", C.c_void_p), ("gate_batch_stride", C.c_int64),
        ("res", C.c_void_p), ("res_batch_stride", C.c_int64), ("ldr", C.c_int32),
        ("bias2", C.c_void_p), ("bias2_batch_stride", C.c_int64),
        ("w_batch_stride", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("batch", C.c_int32),
        ("act", C.c_int32), ("out_f32", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
        ("w_group", C.c_int32), ("reserved1", C.c_int32),
    ]


class ConvDesc(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32),
                ("stride", C.c_int32), ("pad", C.c_int32), ("up", C.c_int32), ("pad_w_p1", C.c_int32), ("out_w", C.c_int32), ("out_h", C.c_int32),
                ("out_row_pitch", C.c_int32), ("moments_accumulate", C.c_int32), ("reserved0", C.c_int32),
                ("moments", C.c_void_p), ("moments_scratch", C.c_void_p)]


class QkvDesc(C.Structure):
    _fields_ = [("norm_q", C.c_void_p), ("norm_k", C.c_void_p), ("cos", C.c_void_p), ("sin", C.c_void_p),
                ("Q", C.c_void_p), ("K", C.c_void_p), ("VT", C.c_void_p),
                ("H", C.c_int32), ("Spad", C.c_int32), ("tok_off", C.c_int32), ("rows_per_sample", C.c_int32),
                ("eps", C.c_float), ("q_scale", C.c_float), ("vt_perm", C.c_int32)]


class Fp8Desc(C.Structure):
    _fields_ = [("a_scale", C.c_void_p), ("a_scale_batch_stride", C.c_int64), ("w_scale", C.c_void_p), ("alpha", C.c_float),
                ("out_fp8", C.c_int32), ("out_inv_scale", C.c_float)]


_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> argtypes (restype is int for all but the two below); mirrors include/x2i.h exactly
SIGNATURES = {
    "x2i_gemm_bf16": [C.POINTER(GemmArgs), _vp],
    "x2i_conv2d_nhwc_bf16": [C.POINTER(GemmArgs), C.POINTER(ConvDesc), _vp],
    "x2i_gemm_qkv_bf16": [C.POINTER(GemmArgs), C.POINTER(QkvDesc), _vp],
    "x2i_gemm_pair_bf16": [C.POINTER(GemmArgs), C.POINTER(GemmArgs), _vp],
    "x2i_gemm_qkv_pair_bf16": [C.POINTER(GemmArgs), C.POINTER(QkvDesc), C.POINTER(GemmArgs), C.POINTER(QkvDesc), _vp],
    "x2i_gemm_fp8": [C.POINTER(GemmArgs), C.POINTER(Fp8Desc), _vp],
    "x2i_gemm_qkv_fp8": [C.POINTER(GemmArgs), C.POINTER(Fp8Desc), C.POINTER(QkvDesc), _vp],
    "x2i_quantize_rows_fp8": [_vp, _i64, _i32, _i64, _vp, _i64, _vp, _f32, _vp],
    "x2i_ln_modulate_fp8": [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i64,
                            _f32, _vp],
    "x2i_conv_stem_bf16": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "x2i_conv3x3_narrow_bf16": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "x2i_groupnorm_nhwc_bf16": [_vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _f32, _i32, _vp, _vp, _vp, _vp],
    "x2i_groupnorm_moments_f32": [_vp, _i32, _i64, _i32, _vp, _vp, _vp],
    "x2i_groupnorm_nhwc_from_moments_bf16": [_vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _f32, _i32, _vp, _vp, _vp, _vp, _vp],
    "x2i_attention_bf16": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _f32, _vp],
    "x2i_attention_vp_bf16": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _f32, _vp],
    "x2i_attention_vp_ws_bf16": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _f32, _vp, _i64, _vp],
    "x2i_attention_prefers_vt_perm": [_i32, _i32, _f32],
    "x2i_attention_e4m3out": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _f32, _f32, _vp],
    "x2i_qkv_split_bf16": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _f32, _vp],
    "x2i_ln_modulate_bf16": [_vp, _i64, _i32, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i64, _f32, _vp],
    "x2i_ln_affine_bf16": [_vp, _vp, _i64, _i32, _vp, _vp, _f32, _vp],
    "x2i_groupnorm_nhwc_grouped_bf16": [_vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _i32, _f32, _i32, _vp, _vp, _vp, _vp],
    "x2i_groupnorm_nhwc_from_moments_grouped_bf16": [_vp, _vp, _i32, _i64, _i32, _i32, _vp, _vp, _i32, _f32, _i32, _vp, _vp, _vp, _vp, _vp],
    "x2i_skinny_linear_grouped": [_vp, _i32, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "x2i_skinny_linear": [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "x2i_timestep_sinusoid": [_vp, _vp, _i32, _i32, _i32, _vp],
    "x2i_rope_table_f32": [_vp, _i32, _i32, _i32, _i32, _f32, _vp, _vp, _vp],
    "x2i_gated_residual_bf16": [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _i32, _i32, _i32, _vp],
    "x2i_euler_step_bf16": [_vp, _vp, _i64, _vp, _vp],
    "x2i_proj_conv5x5_bf16": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "x2i_proj_conv5x5_pack": [_vp, _vp, _i32, _vp],
    "x2i_transpose_bf16": [_vp, _i64, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _vp],
    "x2i_softmax_pad_bf16": [_vp, _i64, _i32, _i32, _i32, _i32, _i32, _f32, _vp],
    "x2i_softmax_bwd_bf16": [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _f32, _vp],
    "x2i_ln_mod_bwd_bf16": [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _i32, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _f32, _vp],
    "x2i_gate_bwd_bf16": [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
    "x2i_reduce_rows_f32": [_vp, _i64, _i32, _i64, _vp, _i64, _i32, _i32, _i32, _f32, _vp],
    "x2i_act_bwd": [_vp, _i64, _vp, _i64, _i64, _i32, _i32, _i32, _vp],
    "x2i_qkv_split_bwd_bf16": [_vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               _i32, _f32, _vp],
    "x2i_skinny_linear_bwd": [_vp, _i64, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp],
    "x2i_kd_loss_bf16": [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _f32, _f32, _vp],
    "x2i_zero_if_nonfinite_bf16": [_vp, _i64, _vp, _vp],
    "x2i_attention_bwd_bf16": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _i32, _vp],
    "x2i_attention_lse_bf16": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _f32, _vp],
    "x2i_attention_bwd_prep_bf16": [_vp, _i64, _i32, _vp, _i64, _i32, _vp, _i32, _i32, _i32, _i32, _vp],
    "x2i_proj_conv5x5_wgrad": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "x2i_plane_dot_bf16": [_vp, _vp, _vp, _i32, _i32, _i64, _i32, _vp],
    "x2i_sum_partials": [_vp, _i32, _i64, _i32, _vp, _i32, _vp],
    "x2i_clip_coef_f32": [_vp, _f32, _vp, _vp],
    "x2i_adamw_bf16": [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _vp, _vp],
    "x2i_proj_conv5x5_packed_bf16": [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "x2i_proj_layer_mean_bf16": [_vp, _vp, _vp, _i32, _i32, _i64, _vp],
    "x2i_seq_mean_f32": [_vp, _vp, _i32, _i32, _i32, _vp],
    "x2i_softmax_rows_bf16": [_vp, _i64, _i32, _f32, _vp],
    "x2i_cast_f32_to_bf16": [_vp, _vp, _i64, _vp],
    "x2i_cast_bf16_to_f32": [_vp, _vp, _i64, _vp],
    "x2i_streamk_workspace_status": [_vp, _i64],
    "x2i_set_option": [C.c_char_p, _i64],
    "x2i_get_option": [C.c_char_p, C.POINTER(C.c_int64)],
}

_lib = None


def load():
    """dlopen the HIP library (once).  Raises X2IError when it is missing -- there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise X2IError(
            "x2i_amd: %s not found. Build it with `python -m x2i_amd.build` (needs hipcc); "
            "there is no CPU fallback for the product path." % LIB_PATH)
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise X2IError("x2i_amd: cannot load %s: %s" % (LIB_PATH, e))
    lib.x2i_abi_version.restype = C.c_int
    lib.x2i_last_error.restype = C.c_char_p
    lib.x2i_groupnorm_scratch_floats.argtypes = [_i32, _i32]
    lib.x2i_groupnorm_scratch_floats.restype = C.c_int64
    lib.x2i_is_ablation_build.restype = C.c_int
    lib.x2i_groupnorm_moments_scratch_floats.argtypes = [_i32, _i32]
    lib.x2i_groupnorm_moments_scratch_floats.restype = C.c_int64
    lib.x2i_conv_moments_scratch_floats.argtypes = [_i32, _i32, _i32]
    lib.x2i_conv_moments_scratch_floats.restype = C.c_int64
    lib.x2i_streamk_workspace_bytes.argtypes = []
    lib.x2i_streamk_workspace_bytes.restype = C.c_int64
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.argtypes = argtypes
        fn.restype = C.c_int
    if lib.x2i_abi_version() != ABI_VERSION:
        raise X2IError("x2i_amd: %s reports ABI version %d, this binding is written against %d (stale build? run `python -m x2i_amd.build`)"
                       % (LIB_PATH, lib.x2i_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


_option_epoch = 0


def set_option(name, value):
    """A/B / tuning switch of the library (include/x2i.h: x2i_set_option); returns the previous value."""
    global _option_epoch
    lib = load()
    old = C.c_int64(0)
    check(lib.x2i_get_option(name.encode(), C.byref(old)), "get_option")
    check(lib.x2i_set_option(name.encode(), int(value)), "set_option")
    if old.value != int(value):
        _option_epoch += 1
    return old.value


def option_epoch():
    """Number of option CHANGES made through set_option since import.  Options select kernels; FluxPipeline keys its graph policy on this so
    that a capture always follows an eager pass under the same option state (pipeline.py)."""
    return _option_epoch


def get_option(name):
    v = C.c_int64(0)
    check(load().x2i_get_option(name.encode(), C.byref(v)), "get_option")
    return v.value


def check(rc, what=""):
    if rc != 0:
        msg = load().x2i_last_error().decode("utf-8", "replace")
        raise X2IError("x2i %s failed (code %d): %s" % (what, rc, msg))
