// e4m3 (OCP fp8) quantisation kernels of the fp8 path: row-wise weight / activation quantisation and the LayerNorm+modulate
// variant that emits the following GEMM's A operand in e4m3 together with its per-row scale.  HBM-bound, one wave per row,
// 16-byte loads, wave-shuffle reductions.  (GEMM: gemm256_fp8.hip; contract: include/x2i.h "fp8 path".)
#include "x2i_common.h"
#include "x2i_kernels.h"

namespace {

constexpr float E4M3_MAX = 448.f;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// 8 floats -> 8 e4m3 bytes (round to nearest even, saturating at +-448)
__device__ __forceinline__ uint2 pack_e4m3x8(const float (&f)[8], float inv) {
  float c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_fmed3f(f[i] * inv, -E4M3_MAX, E4M3_MAX);
  int lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], 0, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], lo, true);
  int hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], 0, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], hi, true);
  return make_uint2((uint32_t)lo, (uint32_t)hi);
}
__device__ __forceinline__ void unpack8(const bf16x8_t& v, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = bf16_to_f32((bf16_t)v[i]);
}

// one wave per row, any number of 16-byte chunks (two passes over the row: amax, then quantise; the row stays in L2)
__global__ __launch_bounds__(256) void quantize_rows_kernel(const bf16_t* __restrict__ x, long long rows, int cols, long long ldx,
                                                            uint8_t* __restrict__ y, long long ldy, float* __restrict__ scale,
                                                            float static_inv) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + row * ldx;
  const int nv = cols >> 3;
  float inv = static_inv;
  if (scale) {
    float amax = 0.f;
    for (int c = lane; c < nv; c += 64) {
      float f[8];
      unpack8(*(const bf16x8_t*)(xr + c * 8), f);
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(f[j]));
    }
    amax = wave_max(amax);
    const float s = amax > 0.f ? amax * (1.f / E4M3_MAX) : 1.f;
    inv = 1.f / s;
    if (lane == 0) scale[row] = s;
  }
  for (int c = lane; c < nv; c += 64) {
    float f[8];
    unpack8(*(const bf16x8_t*)(xr + c * 8), f);
    *(uint2*)(y + row * ldy + c * 8) = pack_e4m3x8(f, inv);
  }
}

constexpr int LN_MAXV = 8;  // up to D = 4096

// LayerNorm (no affine) + modulate with bf16 (optional) and e4m3 + per-row scale outputs; same arithmetic as ln_kernel<false>
// (elementwise.hip) up to the output conversion.
__global__ __launch_bounds__(256) void ln_fp8_kernel(const bf16_t* __restrict__ X, long long x_bs, int ldx, bf16_t* __restrict__ Y,
                                                     long long y_bs, int ldy, uint8_t* __restrict__ Y8, long long y8_bs, int ldy8,
                                                     float* __restrict__ row_scale, int S, int D, int S0, const float* shift0,
                                                     const float* scale0, const float* shift1, const float* scale1, long long mod_bs,
                                                     float eps, long long total_rows) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= total_rows) return;
  const int b = (int)(row / S);
  const int s = (int)(row - (long long)b * S);
  const bf16_t* x = X + (long long)b * x_bs + (long long)s * ldx;
  const int nv = D >> 3;
  float v[LN_MAXV][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      unpack8(*(const bf16x8_t*)(x + c * 8), v[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[i][j];
    }
  }
  const float mean = wave_sum(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = __fsub_rn(v[i][j], mean);
        sq = __builtin_fmaf(d, d, sq);
      }
    }
  }
  const float rstd = rsqrtf(__fadd_rn(wave_sum(sq) / (float)D, eps));
  const float* sh = (s < S0 ? shift0 : shift1) + (long long)b * mod_bs;
  const float* sc = (s < S0 ? scale0 : scale1) + (long long)b * mod_bs;
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const f32x4_t s0 = *(const f32x4_t*)(sc + c * 8), s1 = *(const f32x4_t*)(sc + c * 8 + 4);
      const f32x4_t h0 = *(const f32x4_t*)(sh + c * 8), h1 = *(const f32x4_t*)(sh + c * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[i][j] = ln_mod1(v[i][j], mean, rstd, s0[j], h0[j]);
        v[i][j + 4] = ln_mod1(v[i][j + 4], mean, rstd, s1[j], h1[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf(v[i][j]));
    }
  }
  amax = wave_max(amax);
  const float qs = amax > 0.f ? amax * (1.f / E4M3_MAX) : 1.f;
  const float inv = 1.f / qs;
  if (lane == 0) row_scale[row] = qs;
  uint8_t* y8 = Y8 + (long long)b * y8_bs + (long long)s * ldy8;
  bf16_t* y = Y ? Y + (long long)b * y_bs + (long long)s * ldy : nullptr;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      *(uint2*)(y8 + c * 8) = pack_e4m3x8(v[i], inv);
      if (y) {
        union { bf16x8_t v8; uint32_t u[4]; } r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r.u[j] = pack_bf16x2(v[i][2 * j], v[i][2 * j + 1]);
        *(bf16x8_t*)(y + c * 8) = r.v8;
      }
    }
  }
}

// The same operator for D = 512 * CPL with RW consecutive rows per wave: the (1 + scale) / shift vectors of a sample (four times a row's
// bf16 bytes) stay in registers across the rows instead of coming back from the L2 for every row -- ln_rows_kernel's form (elementwise.hip;
// 57.8 -> the bf16 kernel's ~45 us per call is what the per-row fetch cost).  Same arithmetic as ln_fp8_kernel, element for element.
template <int CPL, int RW>
__global__ __launch_bounds__(256) void ln_fp8_rows_kernel(const bf16_t* __restrict__ X, long long x_bs, int ldx, bf16_t* __restrict__ Y,
                                                          long long y_bs, int ldy, uint8_t* __restrict__ Y8, long long y8_bs, int ldy8,
                                                          float* __restrict__ row_scale, int S, int S0, const float* shift0,
                                                          const float* scale0, const float* shift1, const float* scale1, long long mod_bs,
                                                          float eps, long long total_rows) {
  constexpr int D = CPL * 512;
  const int lane = threadIdx.x & 63;
  const long long row0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
  float sc[CPL][8], sh[CPL][8];
  int cb = -1, cside = -1;
#pragma unroll 1
  for (int r = 0; r < RW; ++r) {
    const long long row = row0 + r;
    if (row >= total_rows) return;
    const int b = (int)(row / S);
    const int s = (int)(row - (long long)b * S);
    const int side = s < S0 ? 0 : 1;
    const bf16_t* x = X + (long long)b * x_bs + (long long)s * ldx;
    float v[CPL][8];
#pragma unroll
    for (int i = 0; i < CPL; ++i) unpack8(*(const bf16x8_t*)(x + (lane + i * 64) * 8), v[i]);
    if (b != cb || side != cside) {  // (wave-uniform) a new sample or stream: fetch its modulation vectors once
      const float* shp = (side ? shift1 : shift0) + (long long)b * mod_bs;
      const float* scp = (side ? scale1 : scale0) + (long long)b * mod_bs;
#pragma unroll
      for (int i = 0; i < CPL; ++i) {
        const int c = (lane + i * 64) * 8;
        const f32x4_t s0 = *(const f32x4_t*)(scp + c), s1 = *(const f32x4_t*)(scp + c + 4);
        const f32x4_t h0 = *(const f32x4_t*)(shp + c), h1 = *(const f32x4_t*)(shp + c + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { sc[i][j] = s0[j]; sc[i][j + 4] = s1[j]; sh[i][j] = h0[j]; sh[i][j + 4] = h1[j]; }
      }
      cb = b; cside = side;
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[i][j];
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = __fsub_rn(v[i][j], mean);
        sq = __builtin_fmaf(d, d, sq);
      }
    const float rstd = rsqrtf(__fadd_rn(wave_sum(sq) / (float)D, eps));
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[i][j] = ln_mod1(v[i][j], mean, rstd, sc[i][j], sh[i][j]);
        amax = fmaxf(amax, fabsf(v[i][j]));
      }
    amax = wave_max(amax);
    const float qs = amax > 0.f ? amax * (1.f / E4M3_MAX) : 1.f;
    const float inv = 1.f / qs;
    if (lane == 0) row_scale[row] = qs;
    uint8_t* y8 = Y8 + (long long)b * y8_bs + (long long)s * ldy8;
    bf16_t* y = Y ? Y + (long long)b * y_bs + (long long)s * ldy : nullptr;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = lane + i * 64;
      *(uint2*)(y8 + c * 8) = pack_e4m3x8(v[i], inv);
      if (y) {
        union { bf16x8_t v8; uint32_t u[4]; } rr;
#pragma unroll
        for (int j = 0; j < 4; ++j) rr.u[j] = pack_bf16x2(v[i][2 * j], v[i][2 * j + 1]);
        *(bf16x8_t*)(y + c * 8) = rr.v8;
      }
    }
  }
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

int x2i_launch_quantize_rows_fp8(const void* x, long long rows, int cols, long long ldx, void* y, long long ldy, float* scale,
                                 float static_inv_scale, hipStream_t stream) {
  if (!x || !y) return x2i_set_error(X2I_ERR_ARG, "quantize_rows_fp8: null pointer");
  if (rows <= 0 || cols <= 0 || cols % 8) return x2i_set_error(X2I_ERR_SHAPE, "quantize_rows_fp8: cols=%d must be a positive multiple of 8", cols);
  if (ldx % 8 || ldy % 8 || !al16(x) || (((uintptr_t)y) & 7)) return x2i_set_error(X2I_ERR_ALIGN, "quantize_rows_fp8: rows must be 16-byte (x) / 8-byte (y) aligned");
  hipLaunchKernelGGL(quantize_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, (const bf16_t*)x, rows, cols, ldx,
                     (uint8_t*)y, ldy, scale, static_inv_scale);
  return x2i_check_launch("quantize_rows_fp8");
}

int x2i_launch_ln_modulate_fp8(const void* X, long long x_bs, int ldx, void* Y, long long y_bs, int ldy, void* Y8, long long y8_bs,
                               int ldy8, float* row_scale, int B, int S, int D, int S0, const float* shift0, const float* scale0,
                               const float* shift1, const float* scale1, long long mod_bs, float eps, hipStream_t stream) {
  if (!X || !Y8 || !row_scale || !shift1 || !scale1 || (S0 > 0 && (!shift0 || !scale0))) return x2i_set_error(X2I_ERR_ARG, "ln_modulate_fp8: null pointer");
  if (D % 8 || D > 64 * 8 * LN_MAXV || B <= 0 || S <= 0) return x2i_set_error(X2I_ERR_SHAPE, "ln_modulate_fp8: D=%d must be a multiple of 8 and <= %d", D, 64 * 8 * LN_MAXV);
  if (ldx % 8 || x_bs % 8 || mod_bs % 4 || !al16(X) || (Y && (ldy % 8 || y_bs % 8 || !al16(Y))) || ldy8 % 8 || y8_bs % 8 || (((uintptr_t)Y8) & 7))
    return x2i_set_error(X2I_ERR_ALIGN, "ln_modulate_fp8: rows must be 16-byte (bf16) / 8-byte (e4m3) aligned");
  const long long rows = (long long)B * S;
  if (D == 3072 && rows >= 4096 && x2i_options().fp8 != 2) {  // the model's width: four rows per wave, modulation vectors in registers (option fp8 = 2: the per-row form, A/B)
    hipLaunchKernelGGL((ln_fp8_rows_kernel<6, 4>), dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, stream, (const bf16_t*)X, x_bs, ldx, (bf16_t*)Y, y_bs,
                       ldy, (uint8_t*)Y8, y8_bs, ldy8, row_scale, S, S0, shift0 ? shift0 : shift1, scale0 ? scale0 : scale1, shift1, scale1, mod_bs, eps, rows);
    return x2i_check_launch("ln_modulate_fp8");
  }
  hipLaunchKernelGGL(ln_fp8_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, (const bf16_t*)X, x_bs, ldx, (bf16_t*)Y, y_bs,
                     ldy, (uint8_t*)Y8, y8_bs, ldy8, row_scale, S, D, S0, shift0 ? shift0 : shift1, scale0 ? scale0 : scale1, shift1, scale1,
                     mod_bs, eps, rows);
  return x2i_check_launch("ln_modulate_fp8");
}
