// Shared device/host helpers for the X2I HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define X2I_OK 0
#define X2I_ERR_ARG (-1)
#define X2I_ERR_SHAPE (-2)
#define X2I_ERR_ALIGN (-3)
#define X2I_ERR_HIP (-4)
#define X2I_ERR_STATE (-5)

// host side: record an error string (thread-local) and return the code
int x2i_set_error(int code, const char* fmt, ...);
int x2i_check_launch(const char* what);

// A/B and tuning switches (include/x2i.h: x2i_set_option).  Resolved ONCE, on first use, from the X2I_* environment
// variables; afterwards only x2i_set_option() changes them -- no getenv on the launch path.
struct X2IOptions {
  int gemm_tile;          // 0 = automatic tile choice; 128 / 256 force a kernel (A/B, race screen)        X2I_GEMM_TILE
  long long gemm_min256;  // minimum number of 256^2 tiles for the 256^2 kernel (default 128)            X2I_GEMM_MIN256
  int gemm_gm;            // 0 = per-shape XCD patch height; > 0 forces it                                 X2I_GEMM_GM
  int gemm_split_tail;    // 1 = peel a thin last round into a 128^2 launch (default)                      X2I_GEMM_NOSPLIT=1 -> 0
  int gemm_w4;            // 1 = plain 256^2 launches take the 4-wave hand-scheduled kernel (gemm256w.hip, default); 0 = 8-wave gemm256.hip   X2I_GEMM_W4
  int attn_bwd_overlap;   // 1 = the dQ and dK / dV passes fill each other's partly filled last rounds: one fused launch (pipelined kernels) or a side stream (the older ones)
  int attn_bwd_dq64;      // 1 = the dQ pass of the attention backward keeps 64 query rows per wave (0: 32, A/B; bit-identical)
  int attn_bwd_pipe;      // 1 = the dK / dV pass runs the software-pipelined kernel (attn_bwd_dkdv_kernel: element-wise section under the MFMAs; default);
                          // 0 = attn_bwd_kernel<1> (A/B; bit-identical)                                          X2I_ATTN_BWD_PIPE
  int train_rows_wg;      // 1 = ln_mod_bwd / gate_bwd of the training step run a workgroup per row group, a thread per eight columns (default);
                          // 0 = the wave-per-row forms (A/B; same values up to the summation order of the row statistics)    X2I_TRAIN_ROWS_WG
  int gemm_pair;          // 1 = x2i_gemm_pair_bf16 / x2i_gemm_qkv_pair_bf16 issue ONE grouped persistent launch when they can (0: always two launches)
  int gemm_streamk;       // 1 = the persistent kernel splits the tiles of a partly filled last round along K (chained partial accumulators,
                          // bit-identical results; default); 0 = whole tiles only (+ the peeled 128^2 tail launch)          X2I_GEMM_STREAMK
  int gemm_fx_nk;         // K-tiles from which the split of gemm_fx applies (default 96 = K >= 6144)   X2I_GEMM_FX_NK
  int gemm_fx;            // launches with fewer 256^2 tiles per batch item than CUs and a deep K (>= 96 K-tiles) are cut along K over all CUs (parallel split
                          // with fix-up, gemm256p.hip FX; not bit-identical to whole tiles -- the K sum is associated differently; decided by the
                          // item's shape alone).  2 (DEFAULT) = items with at most half a round of tiles (512^2 samples); 1 = every item below one
                          // round; 0 = never                                                                                 X2I_GEMM_FX
  int gemm_r2;            // A/B: 1 = plain bf16 launches take the "two residents" kernel (gemm_r2.hip: 256 x 128 tiles, two workgroups per CU) when
                          // their shape allows (K % 256 == 0); bit-identical results; default 0                              X2I_GEMM_R2
  int gemm_fp8_persist;   // 1 = x2i_gemm_fp8 / x2i_gemm_qkv_fp8 take the persistent four-wave form when they can (default); 0 = the one-tile 8-wave kernel (A/B)
  int gemm_persist;       // 1 = batch-1 launches with whole-line epilogues take the persistent form (gemm256p.hip, default)   X2I_GEMM_PERSIST
  int conv256;            // 1 = >= 256-channel convolutions use the 256^2 kernel (default)                X2I_CONV256
  int conv_w4;            // 1 = those convolutions take the persistent four-wave kernel with the hand-scheduled K-loop (gemm256c.hip, default); 0 = the
                          // eight-wave one-tile form (gemm256.hip); bit-identical results                   X2I_CONV_W4
  int conv_korder;        // persistent conv kernels: 1 = K runs (filter row, channel slice, kx): a filter row's taps back to back, their shifted re-reads hit
                          // L2 (default); 0 = (filter row, kx, channel slice), the other kernels' order: bit-identical to them              X2I_CONV_KORDER
  int attn_variant;       // 0 = automatic (8-wave ping-pong when the grid fills the chip, else 4-wave); 1..8 = A/B   X2I_ATTN_VARIANT
  int attn_w16;           // 1 = x2i_attention_prefers_vt_perm may say yes (sampling path on attention_w16.hip; default); 0 = never; 2 = at any size   X2I_ATTN_W16
  int attn_streamk;       // 1 = x2i_attention_vp_ws_bf16 cuts the items of a partly filled last round along the key axis over all CUs (chained through the
                          // caller's workspace: bit-identical; default); 0 = whole items only                                X2I_ATTN_STREAMK
  int conv5_variant;      // matrix-core projector conv: 0 = automatic form choice; 1 = plain stages, 2 = pipelined, 3 = two row blocks   X2I_CONV5_VARIANT
  int fp8;                // 2 = x2i_ln_modulate_fp8 keeps its per-row kernel at the model's width (A/B against the four-rows-per-wave form); else unused   X2I_FP8
  int last_gemm_tile;     // read-only introspection for the parity tests: tile edge of the kernel the last GEMM / conv launch used
                          // (256, 128, 0 = generic kernel), + 1000 when a peeled 128^2 tail launch followed the 256^2 launch
  // measurement-only library (libx2i_hip_ablate.so, -DX2I_ABLATION); ignored by the product library
  int gemm_lform;         // 0 = k-half-unit 256^2 kernel                                                   X2I_GEMM_LFORM
  int gemm_ablate;        // ablation bit mask, wrong results by design                                     X2I_GEMM_ABLATE
  int attn_ablate;        //                                                                                X2I_ATTN_ABLATE
};
X2IOptions& x2i_options();
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device) instead of once per launch
int x2i_ensure_dynamic_smem(const void* kernel, int bytes);
int x2i_num_cus();  // compute units of the current device (cached)
// a second stream + fork / join events per device for x2i_attention_bwd_bf16 (created on first use outside a capture; false: not available)
bool x2i_side_stream(hipStream_t main, hipStream_t* side, hipEvent_t* fork, hipEvent_t* join);
struct x2i_gemm_args;
bool x2i_streamk_workspace(const x2i_gemm_args* a, float** slabs, unsigned** flags, int* rc);  // the caller's workspace, validated
long long x2i_conv_moments_scratch(int M, int N, int batch);   // gemm.hip: floats of x2i_conv_desc.moments_scratch
int x2i_gemm_sk_slabs();                          // slabs per workspace (2 x max tiles: FX double buffer)
int x2i_gemm_sk_max_tiles();                      // (gemm.hip: the constants of gemm_device.h)
long long x2i_gemm_sk_slab_bytes();

// Position of token / key t along the Spad axis of a V^T row when V^T is written "span-permuted" (x2i_qkv_desc.vt_perm, x2i_attention_vp_bf16):
// within every 32-key span position kk holds key 16 ((kk >> 2) & 1) + 4 (kk >> 3) + (kk & 3), i.e. key t sits at 8 ((t >> 2) & 3) + 4 ((t >> 4) & 1) + (t & 3)
// -- the order in which a lane of the 16 x 16 x 32 attention kernel finds a PV MFMA's eight k-positions in its own score registers
// (attention_w16.hip).  Four consecutive tokens stay consecutive; an aligned run of eight becomes two 8-byte pieces 8 positions apart.
__device__ __forceinline__ int x2i_vt_pos(int t, int perm) {
  return perm ? ((t & ~31) | (((t >> 2) & 3) << 3) | (((t >> 4) & 1) << 2) | (t & 3)) : t;
}
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// float -> bf16, round-to-nearest-even (== torch .to(bfloat16)); hipcc lowers these casts to v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  const __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw;
  typedef __attribute__((ext_vector_type(2))) float f32x2_hw;
  const f32x2_hw v = {lo, hi};
  const bf16x2_hw h = __builtin_convertvector(v, bf16x2_hw);
  return __builtin_bit_cast(uint32_t, h);
}
// value held by lane ^ 32, via v_permlane32_swap (no LDS round trip); returns max / sum of own and partner
__device__ __forceinline__ float xhalf_max(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float gelu_tanh_f(float x) {
  // 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))  ==  x * sigmoid(2u)  ==  x / (1 + exp2(x (a + b x^2))) with
  // a = -2 log2(e) sqrt(2/pi), b = 0.044715 a: seven instructions per element (mul, fma, mul, v_exp_f32, add, v_rcp_f32 (1 ulp), mul) --
  // this runs 64 K times per output tile in the GEMM epilogues with nothing to hide behind
  const float z = x * __builtin_fmaf(x * x, -0.10294324f, -2.3022082f);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
}
// LayerNorm(no affine) * (1 + scale) + shift of one element, operation by operation (shared by ln_kernel, ln_rows_kernel and
// ln_fp8_kernel, whose bf16 outputs must agree bit for bit)
__device__ __forceinline__ float ln_mod1(float v, float mean, float rstd, float sc, float sh) {
  return __builtin_fmaf(__fmul_rn(__fsub_rn(v, mean), rstd), __fadd_rn(1.f, sc), sh);
}
// RMSNorm statistics and normalise + interleaved-pair RoPE of a lane's 8 dims, with every fused / unfused operation spelled out: the
// fused QKV epilogues of the GEMM kernels and x2i_qkv_split_bf16 must round identically (bit-exact tests, batch independence), which
// a contractable `a * c - b * s` only does as long as the compiler happens to contract it the same way in every surrounding
__device__ __forceinline__ float sumsq8(const float (&x)[8]) {
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) ss = __builtin_fmaf(x[j], x[j], ss);
  return ss;
}
__device__ __forceinline__ float rms_rsqrt128(float ss, float eps) { return rsqrtf(__builtin_fmaf(ss, 1.f / 128.f, eps)); }
__device__ __forceinline__ void norm_rope8(const float (&x)[8], float r, const float (&w)[8], const float (&cs)[8], const float (&sn)[8],
                                           float (&o)[8]) {
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const float a = __fmul_rn(__fmul_rn(x[j], r), w[j]), bb = __fmul_rn(__fmul_rn(x[j + 1], r), w[j + 1]);
    o[j] = __builtin_fmaf(a, cs[j], -__fmul_rn(bb, sn[j]));
    o[j + 1] = __builtin_fmaf(bb, cs[j + 1], __fmul_rn(a, sn[j + 1]));
  }
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

// activation codes shared by every kernel and by the C ABI (include/x2i.h)
enum { X2I_ACT_NONE = 0, X2I_ACT_GELU_TANH = 1, X2I_ACT_GELU_ERF = 2, X2I_ACT_SILU = 3, X2I_ACT_RELU = 4 };

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case X2I_ACT_GELU_TANH: return gelu_tanh_f(v);
    case X2I_ACT_GELU_ERF: return gelu_erf_f(v);
    case X2I_ACT_SILU: return silu_f(v);
    case X2I_ACT_RELU: return fmaxf(v, 0.f);
    default: return v;
  }
}

// Eight tanh-GELUs at once, for the GEMM epilogues that run with ONE wave per SIMD (nothing else to hide a latency behind): hipcc emits the
// seven-instruction chain of gelu_tanh_f element after element through one temporary -- mul, fma, mul, v_exp_f32, add, v_rcp_f32, wait state,
// mul, each waiting for the one before: ~62 cycles per element (profiles/r04r_fp8_unit_timeline.log).  Here the two transcendental steps are
// two asm blocks of eight independent instructions (their latencies overlap; the closing s_nop is the wait state a VALU read of a
// transcendental result needs, which the compiler cannot know about behind an asm block) and the polynomial / sum / product steps are packed
// two-wide.  Same operations on the same values in the same order per element: bit-identical to gelu_tanh_f.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_tanh8(float (&x)[8]) {
  f32x2_t v[4], t[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = (f32x2_t){x[2 * k], x[2 * k + 1]};
    t[k] = v[k] * __builtin_elementwise_fma(v[k] * v[k], (f32x2_t){-0.10294324f, -0.10294324f}, (f32x2_t){-2.3022082f, -2.3022082f});
  }
  float e0 = t[0][0], e1 = t[0][1], e2 = t[1][0], e3 = t[1][1], e4 = t[2][0], e5 = t[2][1], e6 = t[3][0], e7 = t[3][1];
  asm("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\t"
      "v_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\ts_nop 0"
      : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7));
  t[0] = (f32x2_t){e0, e1} + 1.0f; t[1] = (f32x2_t){e2, e3} + 1.0f; t[2] = (f32x2_t){e4, e5} + 1.0f; t[3] = (f32x2_t){e6, e7} + 1.0f;
  e0 = t[0][0]; e1 = t[0][1]; e2 = t[1][0]; e3 = t[1][1]; e4 = t[2][0]; e5 = t[2][1]; e6 = t[3][0]; e7 = t[3][1];
  asm("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3\n\tv_rcp_f32 %4, %4\n\tv_rcp_f32 %5, %5\n\t"
      "v_rcp_f32 %6, %6\n\tv_rcp_f32 %7, %7\n\ts_nop 0"
      : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7));
  t[0] = v[0] * (f32x2_t){e0, e1}; t[1] = v[1] * (f32x2_t){e2, e3}; t[2] = v[2] * (f32x2_t){e4, e5}; t[3] = v[3] * (f32x2_t){e6, e7};
#pragma unroll
  for (int k = 0; k < 4; ++k) { x[2 * k] = t[k][0]; x[2 * k + 1] = t[k][1]; }
}
// the activation of a quarter chunk (eight values) of the pipelined GEMM epilogues
__device__ __forceinline__ void apply_act8(float (&x)[8], int act) {
  if (act == X2I_ACT_GELU_TANH) { gelu_tanh8(x); return; }
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = apply_act(x[e], act);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
