// extern "C" surface of libx2i_hip.so (declared in include/x2i.h): argument plumbing + error reporting only.
#include <stdarg.h>
#include <stdio.h>

#include "x2i_common.h"
#include "x2i_kernels.h"

static thread_local char g_err[512] = "";

int x2i_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int x2i_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return x2i_set_error(X2I_ERR_HIP, "%s: launch failed: %s", what, hipGetErrorString(e));
  return X2I_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// options: environment read once, then only x2i_set_option()
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <set>
#include <utility>

namespace {
int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v && *v ? atoi(v) : dflt;
}
X2IOptions make_options() {
  X2IOptions o;
  o.gemm_tile = env_int("X2I_GEMM_TILE", 0);
  o.gemm_min256 = env_int("X2I_GEMM_MIN256", 128);
  o.gemm_gm = env_int("X2I_GEMM_GM", 0);
  o.gemm_split_tail = env_int("X2I_GEMM_NOSPLIT", 0) ? 0 : 1;
  o.gemm_w4 = env_int("X2I_GEMM_W4", 1);
  o.gemm_persist = env_int("X2I_GEMM_PERSIST", 1);
  o.gemm_fp8_persist = env_int("X2I_GEMM_FP8_PERSIST", 1);
  o.gemm_fx = env_int("X2I_GEMM_FX", 2);
  o.gemm_fx_nk = env_int("X2I_GEMM_FX_NK", 96);
  o.gemm_r2 = env_int("X2I_GEMM_R2", 0);
  o.gemm_streamk = env_int("X2I_GEMM_STREAMK", 1);
  o.gemm_pair = env_int("X2I_GEMM_PAIR", 1);
  o.train_rows_wg = env_int("X2I_TRAIN_ROWS_WG", 1);
  o.attn_bwd_overlap = env_int("X2I_ATTN_BWD_OVERLAP", 1);
  o.attn_bwd_dq64 = env_int("X2I_ATTN_BWD_DQ64", 1);
  o.attn_bwd_pipe = env_int("X2I_ATTN_BWD_PIPE", 1);
  o.conv256 = env_int("X2I_CONV256", 1);
  o.conv_w4 = env_int("X2I_CONV_W4", 1);
  o.conv_korder = env_int("X2I_CONV_KORDER", 1);
  o.attn_variant = env_int("X2I_ATTN_VARIANT", 0);
  o.attn_w16 = env_int("X2I_ATTN_W16", 1);
  o.attn_streamk = env_int("X2I_ATTN_STREAMK", 1);
  o.conv5_variant = env_int("X2I_CONV5_VARIANT", 0);
  o.fp8 = env_int("X2I_FP8", 0);
  o.last_gemm_tile = -1;
  o.gemm_lform = env_int("X2I_GEMM_LFORM", 1);
  o.gemm_ablate = env_int("X2I_GEMM_ABLATE", 0);
  o.attn_ablate = env_int("X2I_ATTN_ABLATE", 0);
  return o;
}
std::mutex g_smem_mu;
std::set<std::pair<const void*, int>> g_smem_done;
}  // namespace

X2IOptions& x2i_options() {
  static X2IOptions o = make_options();  // thread-safe one-time initialisation
  return o;
}

int x2i_ensure_dynamic_smem(const void* kernel, int bytes) {
  int dev = 0;
  hipGetDevice(&dev);
  const std::pair<const void*, int> key(kernel, dev);
  std::lock_guard<std::mutex> lk(g_smem_mu);
  if (g_smem_done.count(key)) return X2I_OK;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return x2i_set_error(X2I_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%d): %s", bytes, hipGetErrorString(e));
  g_smem_done.insert(key);
  return X2I_OK;
}

int x2i_num_cus() {
  static int cus[64] = {0};
  int dev = 0;
  hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  if (!cus[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}

// Stream-K workspace: caller-owned (include/x2i.h).  Layout: [progress flags + give-up marker, 4 KiB][SK_MAX_TILES accumulator slabs].
bool x2i_streamk_workspace(const x2i_gemm_args* a, float** slabs, unsigned** flags, int* rc) {
  *rc = X2I_OK;
  if (!a->workspace) return false;
  const long long need = 4096 + (long long)x2i_gemm_sk_slabs() * x2i_gemm_sk_slab_bytes();
  if (a->workspace_bytes < need || (((uintptr_t)a->workspace) & 255)) {
    *rc = x2i_set_error(X2I_ERR_ARG, "gemm: stream-K workspace must be 256-byte aligned and >= x2i_streamk_workspace_bytes() = %lld bytes (got %lld)",
                        need, (long long)a->workspace_bytes);
    return false;
  }
  *flags = (unsigned*)a->workspace;
  *slabs = (float*)((char*)a->workspace + 4096);
  return true;
}

// A second stream per device for launches that are independent of each other (the dQ and the dK / dV pass of the attention backward):
// fork / join by events, which is also the form a stream capture accepts.  Created on first use outside a capture; returns false when
// it is not available (the caller then issues its launches one after the other on its own stream).
namespace {
struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  bool failed = false;
};
SideStream g_side[64];
std::mutex g_side_mu;
}  // namespace

bool x2i_side_stream(hipStream_t main, hipStream_t* side, hipEvent_t* fork, hipEvent_t* join) {
  int dev = 0;
  hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) return false;
  std::lock_guard<std::mutex> lk(g_side_mu);
  SideStream& w = g_side[dev];
  if (!w.s) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(main, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone || w.failed) { (void)hipGetLastError(); return false; }
    if (hipStreamCreateWithFlags(&w.s, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&w.fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&w.join, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      w.s = nullptr; w.failed = true;
      return false;
    }
  }
  *side = w.s; *fork = w.fork; *join = w.join;
  return true;
}

extern "C" {

int x2i_abi_version(void) { return X2I_ABI_VERSION; }

static long long* opt_slot(X2IOptions& o, const char* name, int** as_int) {
  *as_int = nullptr;
#define X2I_OPT_INT(N_) if (!strcmp(name, #N_)) { *as_int = &o.N_; return nullptr; }
  X2I_OPT_INT(gemm_tile) X2I_OPT_INT(gemm_gm) X2I_OPT_INT(gemm_split_tail) X2I_OPT_INT(gemm_w4) X2I_OPT_INT(gemm_persist) X2I_OPT_INT(gemm_fp8_persist) X2I_OPT_INT(gemm_fx) X2I_OPT_INT(gemm_fx_nk) X2I_OPT_INT(gemm_streamk) X2I_OPT_INT(gemm_pair) X2I_OPT_INT(train_rows_wg) X2I_OPT_INT(attn_bwd_overlap) X2I_OPT_INT(attn_streamk) X2I_OPT_INT(conv256) X2I_OPT_INT(conv_w4) X2I_OPT_INT(conv_korder) X2I_OPT_INT(attn_variant) X2I_OPT_INT(attn_w16) X2I_OPT_INT(conv5_variant) X2I_OPT_INT(fp8) X2I_OPT_INT(last_gemm_tile)
#ifdef X2I_ABLATION
  X2I_OPT_INT(gemm_lform) X2I_OPT_INT(gemm_ablate) X2I_OPT_INT(attn_ablate) X2I_OPT_INT(gemm_r2) X2I_OPT_INT(attn_bwd_dq64) X2I_OPT_INT(attn_bwd_pipe)
#endif
#undef X2I_OPT_INT
  if (!strcmp(name, "gemm_min256")) return &o.gemm_min256;
  return nullptr;
}

int x2i_set_option(const char* name, int64_t value) {
  if (!name) return x2i_set_error(X2I_ERR_ARG, "set_option: null name");
  int* ip;
  long long* lp = opt_slot(x2i_options(), name, &ip);
  if (ip) *ip = (int)value;
  else if (lp) *lp = (long long)value;
  else return x2i_set_error(X2I_ERR_ARG, "set_option: unknown option '%s'", name);
  return X2I_OK;
}

int x2i_get_option(const char* name, int64_t* value) {
  if (!name || !value) return x2i_set_error(X2I_ERR_ARG, "get_option: null pointer");
  int* ip;
  long long* lp = opt_slot(x2i_options(), name, &ip);
  if (ip) *value = *ip;
  else if (lp) *value = *lp;
  else return x2i_set_error(X2I_ERR_ARG, "get_option: unknown option '%s'", name);
  return X2I_OK;
}

int64_t x2i_streamk_workspace_bytes(void) { return 4096 + (int64_t)x2i_gemm_sk_slabs() * x2i_gemm_sk_slab_bytes(); }

int x2i_streamk_workspace_status(const void* workspace, int64_t workspace_bytes) {
  if (!workspace || workspace_bytes < x2i_streamk_workspace_bytes()) return x2i_set_error(X2I_ERR_ARG, "streamk_workspace_status: no / too small workspace");
  unsigned v = 0;
  const hipError_t e = hipMemcpy(&v, (const unsigned*)workspace + x2i_gemm_sk_max_tiles(), 4, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return x2i_set_error(X2I_ERR_HIP, "streamk_workspace_status: %s", hipGetErrorString(e));
  return v ? 1 : 0;
}

int x2i_is_ablation_build(void) {
#ifdef X2I_ABLATION
  return 1;
#else
  return 0;
#endif
}
const char* x2i_last_error(void) { return g_err; }

int x2i_gemm_bf16(const x2i_gemm_args* args, x2i_stream_t stream) { return x2i_launch_gemm(args, (hipStream_t)stream); }

int x2i_gemm_fp8(const x2i_gemm_args* args, const x2i_fp8_desc* fp8, x2i_stream_t stream) {
  return x2i_launch_gemm_fp8(args, fp8, (hipStream_t)stream);
}

int x2i_gemm_qkv_fp8(const x2i_gemm_args* args, const x2i_fp8_desc* fp8, const x2i_qkv_desc* qkv, x2i_stream_t stream) {
  return x2i_launch_gemm_qkv_fp8(args, fp8, qkv, (hipStream_t)stream);
}

int x2i_quantize_rows_fp8(const void* x, int64_t rows, int32_t cols, int64_t ldx, void* y, int64_t ldy, float* scale,
                          float static_inv_scale, x2i_stream_t stream) {
  return x2i_launch_quantize_rows_fp8(x, rows, cols, ldx, y, ldy, scale, static_inv_scale, (hipStream_t)stream);
}

int x2i_ln_modulate_fp8(const void* X, int64_t x_bs, int32_t ldx, void* Y, int64_t y_bs, int32_t ldy, void* Y8, int64_t y8_bs,
                        int32_t ldy8, float* row_scale, int32_t B, int32_t S, int32_t D, int32_t S0, const float* shift0,
                        const float* scale0, const float* shift1, const float* scale1, int64_t mod_bs, float eps,
                        x2i_stream_t stream) {
  return x2i_launch_ln_modulate_fp8(X, x_bs, ldx, Y, y_bs, ldy, Y8, y8_bs, ldy8, row_scale, B, S, D, S0, shift0, scale0, shift1, scale1,
                                    mod_bs, eps, (hipStream_t)stream);
}

int64_t x2i_conv_moments_scratch_floats(int32_t M, int32_t N, int32_t batch) { return (M > 0 && N > 0 && batch > 0) ? x2i_conv_moments_scratch(M, N, batch) : 0; }
int x2i_conv2d_nhwc_bf16(const x2i_gemm_args* args, const x2i_conv_desc* conv, x2i_stream_t stream) {
  if (!conv) return x2i_set_error(X2I_ERR_ARG, "conv2d: null descriptor");
  return x2i_launch_gemm_conv(args, conv, (hipStream_t)stream);
}

int x2i_gemm_pair_bf16(const x2i_gemm_args* args0, const x2i_gemm_args* args1, x2i_stream_t stream) {
  return x2i_launch_gemm_pair(args0, nullptr, args1, nullptr, (hipStream_t)stream);
}
int x2i_gemm_qkv_pair_bf16(const x2i_gemm_args* args0, const x2i_qkv_desc* qkv0, const x2i_gemm_args* args1, const x2i_qkv_desc* qkv1,
                           x2i_stream_t stream) {
  if (!qkv0 || !qkv1) return x2i_set_error(X2I_ERR_ARG, "gemm_qkv_pair: null descriptor");
  return x2i_launch_gemm_pair(args0, qkv0, args1, qkv1, (hipStream_t)stream);
}
int x2i_gemm_qkv_bf16(const x2i_gemm_args* args, const x2i_qkv_desc* qkv, x2i_stream_t stream) {
  if (!qkv) return x2i_set_error(X2I_ERR_ARG, "gemm_qkv: null descriptor");
  return x2i_launch_gemm_qkv(args, qkv, (hipStream_t)stream);
}

int x2i_conv3x3_narrow_bf16(const void* x, const void* w, const void* bias, void* y, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                            int32_t ldy, x2i_stream_t stream) {
  return x2i_launch_conv3x3_narrow(x, w, bias, y, B, H, W, Cin, Cout, ldy, (hipStream_t)stream);
}

int x2i_conv_stem_bf16(const void* x, const float* w, const float* bias, void* y, int32_t B, int32_t H, int32_t W, int32_t Cout,
                       x2i_stream_t stream) {
  return x2i_launch_conv_stem(x, w, bias, y, B, H, W, Cout, (hipStream_t)stream);
}

int64_t x2i_groupnorm_scratch_floats(int32_t B, int32_t G) { return x2i_groupnorm_scratch(B, G); }

int x2i_groupnorm_nhwc_bf16(const void* x, void* y, int32_t B, int64_t HW, int32_t C, int32_t G, const void* weight,
                            const void* bias, float eps, int32_t act, const float* pre_add, const void* post_add, float* partial,
                            x2i_stream_t stream) {
  return x2i_launch_groupnorm(x, y, B, HW, C, G, weight, bias, eps, act, pre_add, post_add, partial, (hipStream_t)stream);
}

int64_t x2i_groupnorm_moments_scratch_floats(int32_t B, int32_t C) { return x2i_groupnorm_moments_scratch(B, C); }
int x2i_groupnorm_moments_f32(const void* x, int32_t B, int64_t HW, int32_t C, float* moments, float* scratch, x2i_stream_t stream) {
  return x2i_launch_groupnorm_moments(x, B, HW, C, moments, scratch, (hipStream_t)stream);
}
int x2i_groupnorm_nhwc_from_moments_bf16(const void* x, void* y, int32_t B, int64_t HW, int32_t C, int32_t G, const void* weight, const void* bias,
                                         float eps, int32_t act, const float* moments, const float* pre_add, const void* post_add, float* partial,
                                         x2i_stream_t stream) {
  return x2i_launch_groupnorm_from_moments(x, y, B, HW, C, G, weight, bias, eps, act, moments, pre_add, post_add, partial, (hipStream_t)stream);
}

int x2i_groupnorm_nhwc_grouped_bf16(const void* x, void* y, int32_t B, int64_t HW, int32_t C, int32_t G, const void* weight, const void* bias,
                                    int32_t w_group, float eps, int32_t act, const float* pre_add, const void* post_add, float* partial,
                                    x2i_stream_t stream) {
  return x2i_launch_groupnorm(x, y, B, HW, C, G, weight, bias, eps, act, pre_add, post_add, partial, (hipStream_t)stream, w_group);
}
int x2i_groupnorm_nhwc_from_moments_grouped_bf16(const void* x, void* y, int32_t B, int64_t HW, int32_t C, int32_t G, const void* weight,
                                                 const void* bias, int32_t w_group, float eps, int32_t act, const float* moments,
                                                 const float* pre_add, const void* post_add, float* partial, x2i_stream_t stream) {
  return x2i_launch_groupnorm_from_moments(x, y, B, HW, C, G, weight, bias, eps, act, moments, pre_add, post_add, partial, (hipStream_t)stream,
                                           w_group);
}

int x2i_attention_bf16(const void* Q, const void* K, const void* VT, void* O, int32_t B, int32_t H, int32_t S, int32_t Spad,
                       int32_t ldo, int64_t o_batch_stride, float scale, x2i_stream_t stream) {
  return x2i_launch_attention(Q, K, VT, O, B, H, S, Spad, ldo, o_batch_stride, scale, (hipStream_t)stream);
}

int x2i_attention_prefers_vt_perm(int32_t H, int32_t S, float scale) {
  const float sl2 = scale * 1.4426950408889634f;
  const int mode = x2i_options().attn_w16;   // 2 = whatever the size (tests)
  return (mode && fabsf(sl2 - 1.f) < 1e-6f && S > 0 && (mode == 2 || (long long)((S + 255) / 256) * H >= 128)) ? 1 : 0;
}

int x2i_attention_vp_ws_bf16(const void* Q, const void* K, const void* VT, void* O, int32_t B, int32_t H, int32_t S, int32_t Spad,
                             int32_t ldo, int64_t o_batch_stride, float scale, void* workspace, int64_t workspace_bytes, x2i_stream_t stream) {
  if (!Q || !K || !VT || !O) return x2i_set_error(X2I_ERR_ARG, "attention_vp: null pointer");
  if (B <= 0 || H <= 0 || S <= 0 || Spad < S || Spad % 128) return x2i_set_error(X2I_ERR_SHAPE, "attention_vp: need Spad %% 128 == 0 and Spad >= S (S=%d Spad=%d)", S, Spad);
  if (workspace && (workspace_bytes < x2i_streamk_workspace_bytes() || (((uintptr_t)workspace) & 255)))
    return x2i_set_error(X2I_ERR_ARG, "attention_vp: the stream-K workspace must be 256-byte aligned and >= x2i_streamk_workspace_bytes() bytes (got %lld)", (long long)workspace_bytes);
  float sl2 = scale * 1.4426950408889634f;
  const bool unit = fabsf(sl2 - 1.f) < 1e-6f;
  if (unit) sl2 = 1.f;
  const int rc = x2i_launch_attention_w16(Q, K, VT, O, B, H, S, Spad, ldo, o_batch_stride, sl2, unit ? 0 : 1, (hipStream_t)stream, nullptr, workspace, workspace_bytes);
  if (rc == X2I_ERR_STATE)
    return x2i_set_error(X2I_ERR_SHAPE, "attention_vp: the span-permuted V^T layout is only read by the 16x16x32 kernel, which needs 16-byte aligned output rows (ldo=%d)", ldo);
  return rc;
}

int x2i_attention_vp_bf16(const void* Q, const void* K, const void* VT, void* O, int32_t B, int32_t H, int32_t S, int32_t Spad,
                          int32_t ldo, int64_t o_batch_stride, float scale, x2i_stream_t stream) {
  return x2i_attention_vp_ws_bf16(Q, K, VT, O, B, H, S, Spad, ldo, o_batch_stride, scale, nullptr, 0, stream);
}

int x2i_attention_lse_bf16(const void* Q, const void* K, const void* VT, void* O, float* lse2, int32_t B, int32_t H, int32_t S, int32_t Spad,
                           int32_t ldo, int64_t o_batch_stride, float scale, x2i_stream_t stream) {
  if (!lse2 || (((uintptr_t)lse2) & 3)) return x2i_set_error(X2I_ERR_ARG, "attention_lse: lse2 must be a valid f32 pointer");
  return x2i_launch_attention(Q, K, VT, O, B, H, S, Spad, ldo, o_batch_stride, scale, (hipStream_t)stream, 0, 1.f, lse2);
}

int x2i_attention_e4m3out(const void* Q, const void* K, const void* VT, void* O8, int32_t B, int32_t H, int32_t S, int32_t Spad,
                          int32_t ldo, int64_t o_batch_stride, float scale, float out_inv_scale, x2i_stream_t stream) {
  return x2i_launch_attention(Q, K, VT, O8, B, H, S, Spad, ldo, o_batch_stride, scale, (hipStream_t)stream, 1, out_inv_scale);
}

int x2i_qkv_split_bf16(const void* qkv0, const void* qkv1, int32_t ld0, int32_t ld1, int32_t B, int32_t S, int32_t S0, int32_t H,
                       const void* nq0, const void* nk0, const void* nq1, const void* nk1, const float* cos, const float* sin,
                       void* Q, void* K, void* VT, int32_t Spad, float eps, x2i_stream_t stream) {
  return x2i_launch_qkv_split(qkv0, qkv1, ld0, ld1, B, S, S0, H, nq0, nk0, nq1, nk1, cos, sin, Q, K, VT, Spad, eps,
                              (hipStream_t)stream);
}

int x2i_ln_modulate_bf16(const void* X, int64_t x_bs, int32_t ldx, void* Y, int64_t y_bs, int32_t ldy, int32_t B, int32_t S,
                         int32_t D, int32_t S0, const float* shift0, const float* scale0, const float* shift1,
                         const float* scale1, int64_t mod_bs, float eps, x2i_stream_t stream) {
  return x2i_launch_ln_modulate(X, x_bs, ldx, Y, y_bs, ldy, B, S, D, S0, shift0, scale0, shift1, scale1, mod_bs, eps,
                                (hipStream_t)stream);
}

int x2i_ln_affine_bf16(const void* X, void* Y, int64_t rows, int32_t D, const void* weight, const void* bias, float eps,
                       x2i_stream_t stream) {
  return x2i_launch_ln_affine(X, Y, rows, D, weight, bias, eps, (hipStream_t)stream);
}

int x2i_skinny_linear(const void* X, int32_t x_is_bf16, const void* W, const void* bias, float* Y, int32_t ldy, int32_t B,
                      int32_t N, int32_t K, int32_t act_in, int32_t act_out, int32_t accumulate, x2i_stream_t stream) {
  return x2i_launch_skinny_linear(X, x_is_bf16, W, bias, Y, ldy, B, N, K, act_in, act_out, accumulate, (hipStream_t)stream);
}

int x2i_skinny_linear_grouped(const void* X, int32_t x_is_bf16, int64_t x_group_stride, const void* W, const void* bias, float* Y, int32_t ldy,
                              int32_t groups, int32_t B, int32_t N, int32_t K, int32_t act_in, int32_t act_out, int32_t accumulate,
                              x2i_stream_t stream) {
  return x2i_launch_skinny_linear_grouped(X, x_is_bf16, x_group_stride, W, bias, Y, ldy, groups, B, N, K, act_in, act_out, accumulate,
                                          (hipStream_t)stream);
}

int x2i_timestep_sinusoid(const float* t, float* out, int32_t B, int32_t dim, int32_t round_bf16, x2i_stream_t stream) {
  return x2i_launch_timestep_sinusoid(t, out, B, dim, round_bf16, (hipStream_t)stream);
}

int x2i_rope_table_f32(const float* ids, int32_t S, int32_t d0, int32_t d1, int32_t d2, float theta, float* cos, float* sin,
                       x2i_stream_t stream) {
  return x2i_launch_rope_table(ids, S, d0, d1, d2, theta, cos, sin, (hipStream_t)stream);
}

int x2i_gated_residual_bf16(void* X, int64_t x_bs, int32_t ldx, const void* T, int64_t t_bs, int32_t ldt, const float* gate,
                            int64_t gate_bs, int32_t B, int32_t S, int32_t D, x2i_stream_t stream) {
  return x2i_launch_gated_residual(X, x_bs, ldx, T, t_bs, ldt, gate, gate_bs, B, S, D, (hipStream_t)stream);
}

int x2i_euler_step_bf16(void* x, const void* eps, int64_t n, const float* dt, x2i_stream_t stream) {
  return x2i_launch_euler_step(x, eps, n, dt, (hipStream_t)stream);
}

int x2i_proj_conv5x5_bf16(const void* x, const float* w, const float* bias, void* y, int32_t B, int32_t C, int32_t S, int32_t H,
                          x2i_stream_t stream) {
  return x2i_launch_proj_conv5x5(x, w, bias, y, B, C, S, H, (hipStream_t)stream);
}

int x2i_proj_conv5x5_pack(const float* w, void* table, int32_t C, x2i_stream_t stream) {
  return x2i_launch_proj_conv5x5_pack(w, table, C, (hipStream_t)stream);
}

int x2i_proj_conv5x5_packed_bf16(const void* x, const void* table, const float* bias, void* y, int32_t B, int32_t C, int32_t S,
                                 int32_t H, x2i_stream_t stream) {
  return x2i_launch_proj_conv5x5_packed(x, table, bias, y, B, C, S, H, (hipStream_t)stream);
}

int x2i_proj_layer_mean_bf16(const void* x, const float* scale, void* y, int32_t B, int32_t C, int64_t plane, x2i_stream_t stream) {
  return x2i_launch_layer_mean(x, scale, y, B, C, plane, (hipStream_t)stream);
}

int x2i_seq_mean_f32(const float* x, float* y, int32_t B, int32_t S, int32_t N, x2i_stream_t stream) {
  return x2i_launch_seq_mean(x, y, B, S, N, (hipStream_t)stream);
}

int x2i_softmax_rows_bf16(void* x, int64_t rows, int32_t cols, float scale, x2i_stream_t stream) {
  return x2i_launch_softmax_rows(x, rows, cols, scale, (hipStream_t)stream);
}

int x2i_cast_f32_to_bf16(const float* x, void* y, int64_t n, x2i_stream_t stream) {
  return x2i_launch_cast_f32_bf16(x, y, n, (hipStream_t)stream);
}
int x2i_cast_bf16_to_f32(const void* x, float* y, int64_t n, x2i_stream_t stream) {
  return x2i_launch_cast_bf16_f32(x, y, n, (hipStream_t)stream);
}

/* ---- N4: backward kernels of the attention-distillation step (train.hip) */
int x2i_transpose_bf16(const void* in, int64_t in_batch_stride, int64_t ld_in, void* out, int64_t out_batch_stride, int64_t ld_out,
                       int32_t batch, int32_t R, int32_t C, x2i_stream_t stream) {
  return x2i_launch_transpose(in, in_batch_stride, ld_in, out, out_batch_stride, ld_out, batch, R, C, (hipStream_t)stream);
}
int x2i_softmax_pad_bf16(void* x, int64_t ld, int32_t nz, int32_t Rt, int32_t Rv, int32_t Ct, int32_t Cv, float scale, x2i_stream_t stream) {
  return x2i_launch_softmax_pad(x, ld, nz, Rt, Rv, Ct, Cv, scale, (hipStream_t)stream);
}
int x2i_softmax_bwd_bf16(const void* P, void* dP, int64_t ld, int32_t nz, int32_t Rt, int32_t Rv, int32_t Ct, int32_t Cv, float scale,
                         x2i_stream_t stream) {
  return x2i_launch_softmax_bwd(P, dP, ld, nz, Rt, Rv, Ct, Cv, scale, (hipStream_t)stream);
}
int x2i_ln_mod_bwd_bf16(const void* X, int64_t x_bs, int32_t ldx, const void* dY, int64_t dy_bs, int32_t ldy, const float* mult, int64_t mult_bs,
                        int32_t mult_is_scale, const void* dXin, void* dXout, int64_t dx_bs, int32_t lddx, int32_t B, int32_t S, int32_t D,
                        int32_t rows_per_wave, float* partial, float eps, x2i_stream_t stream) {
  return x2i_launch_ln_mod_bwd(X, x_bs, ldx, dY, dy_bs, ldy, mult, mult_bs, mult_is_scale, dXin, dXout, dx_bs, lddx, B, S, D, rows_per_wave, partial,
                               eps, (hipStream_t)stream);
}
int x2i_gate_bwd_bf16(const void* dX, int64_t dx_bs, int32_t lddx, const void* T, int64_t t_bs, int32_t ldt, const float* gate, int64_t gate_bs,
                      const void* G, int64_t g_bs, int32_t ldg, void* dT, int64_t dt_bs, int32_t lddt, int32_t B, int32_t S, int32_t D,
                      int32_t rows_per_wave, float* partial, x2i_stream_t stream) {
  return x2i_launch_gate_bwd(dX, dx_bs, lddx, T, t_bs, ldt, gate, gate_bs, G, g_bs, ldg, dT, dt_bs, lddt, B, S, D, rows_per_wave, partial,
                             (hipStream_t)stream);
}
int x2i_reduce_rows_f32(const float* in, int64_t in_z_stride, int32_t np, int64_t in_p_stride, float* out, int64_t out_z_stride, int32_t nz,
                        int32_t len, int32_t accumulate, float alpha, x2i_stream_t stream) {
  return x2i_launch_reduce_rows(in, in_z_stride, np, in_p_stride, out, out_z_stride, nz, len, accumulate, alpha, (hipStream_t)stream);
}
int x2i_act_bwd(void* dA, int64_t ldd, const void* pre, int64_t ldp, int64_t rows, int32_t cols, int32_t act, int32_t is_f32, x2i_stream_t stream) {
  return x2i_launch_act_bwd(dA, ldd, pre, ldp, rows, cols, act, is_f32, (hipStream_t)stream);
}
int x2i_qkv_split_bwd_bf16(const void* qkv0, const void* qkv1, int32_t ld0, int32_t ld1, void* d0, void* d1, int32_t ldd0, int32_t ldd1, int32_t B,
                           int32_t S, int32_t S0, int32_t H, const void* nq0, const void* nk0, const void* nq1, const void* nk1, const float* cosp,
                           const float* sinp, const void* dQ, const void* dK, const void* dV, int32_t Spad, float eps, x2i_stream_t stream) {
  return x2i_launch_qkv_split_bwd(qkv0, qkv1, ld0, ld1, d0, d1, ldd0, ldd1, B, S, S0, H, nq0, nk0, nq1, nk1, cosp, sinp, dQ, dK, dV, Spad, eps,
                                  (hipStream_t)stream);
}
int x2i_skinny_linear_bwd(const float* dy, int64_t dy_bs, const void* W, int32_t ldw, float* partial, int32_t B, int32_t N, int32_t K,
                          int32_t chunk, x2i_stream_t stream) {
  return x2i_launch_skinny_bwd(dy, dy_bs, W, ldw, partial, B, N, K, chunk, (hipStream_t)stream);
}
int x2i_kd_loss_bf16(const void* teacher, int64_t ldt, const void* student, int64_t lds, void* grad, int64_t ldg, float* row_loss, int64_t rows,
                     int32_t D, float temperature, float loss_scale, x2i_stream_t stream) {
  return x2i_launch_kd_loss(teacher, ldt, student, lds, grad, ldg, row_loss, rows, D, temperature, loss_scale, (hipStream_t)stream);
}
int x2i_zero_if_nonfinite_bf16(void* g, int64_t n, const float* term, x2i_stream_t stream) {
  return x2i_launch_zero_if_nonfinite(g, n, term, (hipStream_t)stream);
}

int x2i_proj_conv5x5_wgrad(const void* x, const void* dy, float* partial, int32_t B, int32_t C, int32_t S, int32_t H, x2i_stream_t stream) {
  return x2i_launch_conv5x5_wgrad(x, dy, partial, B, C, S, H, (hipStream_t)stream);
}
int x2i_plane_dot_bf16(const void* x, const void* dy, float* partial, int32_t B, int32_t C, int64_t plane, int32_t nchunk, x2i_stream_t stream) {
  return x2i_launch_plane_dot(x, dy, partial, B, C, plane, nchunk, (hipStream_t)stream);
}
int x2i_sum_partials(const void* x, int32_t is_bf16, int64_t n, int32_t squares, float* partial, int32_t nblocks, x2i_stream_t stream) {
  return x2i_launch_sum(x, is_bf16, n, squares, partial, nblocks, (hipStream_t)stream);
}
int x2i_clip_coef_f32(const float* sumsq, float max_norm, float* out, x2i_stream_t stream) {
  return x2i_launch_clip_coef(sumsq, max_norm, out, (hipStream_t)stream);
}
int x2i_adamw_bf16(void* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                   float bias_correction1, float bias_correction2, const float* grad_coef, x2i_stream_t stream) {
  return x2i_launch_adamw(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2, grad_coef, (hipStream_t)stream);
}

int x2i_attention_bwd_bf16(const void* Q, const void* K, const void* V, const void* QT, const void* KT, const void* dO, const void* dOT, float* lse2,
                           const float* D, void* dQ, void* dK, void* dV, int32_t B, int32_t H, int32_t S, int32_t Spad, float scale,
                           int32_t have_lse, x2i_stream_t stream) {
  return x2i_launch_attention_bwd(Q, K, V, QT, KT, dO, dOT, lse2, D, dQ, dK, dV, B, H, S, Spad, scale, have_lse, (hipStream_t)stream);
}
int x2i_attention_bwd_prep_bf16(const void* dO, int64_t do_bs, int32_t lddo, const void* O, int64_t o_bs, int32_t ldo, float* D, int32_t B, int32_t H,
                                int32_t S, int32_t Spad, x2i_stream_t stream) {
  return x2i_launch_attention_bwd_prep(dO, do_bs, lddo, O, o_bs, ldo, D, B, H, S, Spad, (hipStream_t)stream);
}

}  // extern "C"
