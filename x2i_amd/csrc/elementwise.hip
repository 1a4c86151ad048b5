// HBM-bound kernels of the DiT step: LayerNorm+modulate, QKV split (RMSNorm + RoPE + V transpose), skinny
// linears (AdaLN modulation / embedders), sinusoid, Euler step, casts.  All loads are 16-byte vectors, all
// reductions are 64-lane wave shuffles (no LDS round trips) -- CDNA guide G13 / Appendix B.
#include "x2i_common.h"
#include "x2i_kernels.h"

namespace {

__device__ __forceinline__ void unpack8(const bf16x8_t& v, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = bf16_to_f32((bf16_t)v[i]);
}
__device__ __forceinline__ bf16x8_t pack8(const float (&f)[8]) {
  union { bf16x8_t v; uint32_t u[4]; } r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.u[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
  return r.v;
}

// ------------------------------------------------------------------------------------------------------------
// LayerNorm (no affine) + modulate.  One wave per row; D % 8 == 0, D <= 64*8*MAXV.
// ------------------------------------------------------------------------------------------------------------
constexpr int LN_MAXV = 8;  // up to D = 4096

template <bool AFFINE>
__global__ __launch_bounds__(256) void ln_kernel(const bf16_t* __restrict__ X, long long x_bs, int ldx, bf16_t* __restrict__ Y,
                                                 long long y_bs, int ldy, int S, int D, int S0, const float* shift0,
                                                 const float* scale0, const float* shift1, const float* scale1,
                                                 long long mod_bs, const bf16_t* aw, const bf16_t* ab, float eps,
                                                 long long total_rows) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= total_rows) return;
  const int b = (int)(row / S);
  const int s = (int)(row - (long long)b * S);
  const bf16_t* x = X + (long long)b * x_bs + (long long)s * ldx;
  bf16_t* y = Y + (long long)b * y_bs + (long long)s * ldy;
  const int nv = D >> 3;  // 16-byte chunks per row
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
  const float* sh = nullptr;
  const float* sc = nullptr;
  if (!AFFINE) {
    sh = (s < S0 ? shift0 : shift1) + (long long)b * mod_bs;
    sc = (s < S0 ? scale0 : scale1) + (long long)b * mod_bs;
  }
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      float o[8];
      if (AFFINE) {
        float w[8], bb[8];
        unpack8(*(const bf16x8_t*)(aw + c * 8), w);
        unpack8(*(const bf16x8_t*)(ab + c * 8), bb);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * w[j] + bb[j];
      } else {
        const f32x4_t s0 = *(const f32x4_t*)(sc + c * 8), s1 = *(const f32x4_t*)(sc + c * 8 + 4);
        const f32x4_t h0 = *(const f32x4_t*)(sh + c * 8), h1 = *(const f32x4_t*)(sh + c * 8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] = ln_mod1(v[i][j], mean, rstd, s0[j], h0[j]);
          o[j + 4] = ln_mod1(v[i][j + 4], mean, rstd, s1[j], h1[j]);
        }
      }
      *(bf16x8_t*)(y + c * 8) = pack8(o);
    }
  }
}

// The modulated form for D = 512 * CPL, several consecutive rows per wave: the (1 + scale) / shift vectors of a sample (2 x 4 D bytes,
// four times a row's bf16 bytes) stay in registers across the rows instead of coming back from L2 for every row.  Same arithmetic
// as ln_kernel<false>, element for element.
template <int CPL, int RW>
__global__ __launch_bounds__(256) void ln_rows_kernel(const bf16_t* __restrict__ X, long long x_bs, int ldx, bf16_t* __restrict__ Y,
                                                      long long y_bs, int ldy, int S, int S0, const float* shift0, const float* scale0,
                                                      const float* shift1, const float* scale1, long long mod_bs, float eps,
                                                      long long total_rows) {
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
    bf16_t* y = Y + (long long)b * y_bs + (long long)s * ldy;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = ln_mod1(v[i][j], mean, rstd, sc[i][j], sh[i][j]);
      *(bf16x8_t*)(y + (lane + i * 64) * 8) = pack8(o);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// q/k: RMSNorm(128) * w -> RoPE -> Q/K [B,H,Spad,128].  16 lanes x 8 elements per (token, head).
// grid: (S, B); block 256 threads loops over 2*H*16 chunk-units of its token.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(const bf16_t* __restrict__ qkv0, const bf16_t* __restrict__ qkv1, int ld0,
                                                           int ld1, int S, int S0, int H, const bf16_t* nq0, const bf16_t* nk0,
                                                           const bf16_t* nq1, const bf16_t* nk1, const float* __restrict__ cosp,
                                                           const float* __restrict__ sinp, bf16_t* __restrict__ Q,
                                                           bf16_t* __restrict__ K, int Spad, float eps) {
  const int s = blockIdx.x, b = blockIdx.y;
  const bool src0 = s < S0;
  const bf16_t* row = src0 ? qkv0 + ((long long)b * S0 + s) * ld0 : qkv1 + ((long long)b * (S - S0) + (s - S0)) * ld1;
  const int D = H * 128;
  const int units = 2 * H * 16;
  for (int u = threadIdx.x; u < units; u += 256) {
    const int isk = u / (H * 16);
    const int rem = u - isk * H * 16;
    const int h = rem >> 4, c = rem & 15;
    float x[8];
    unpack8(*(const bf16x8_t*)(row + isk * D + h * 128 + c * 8), x);
    float ss = sumsq8(x);
    // reduce over the 16 lanes of this (token, head)
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float r = rms_rsqrt128(ss, eps);
    const bf16_t* wn = isk ? (src0 ? nk0 : nk1) : (src0 ? nq0 : nq1);
    float w[8];
    unpack8(*(const bf16x8_t*)(wn + c * 8), w);
    const float* cp = cosp + (long long)s * 128 + c * 8;
    const float* sp = sinp + (long long)s * 128 + c * 8;
    const f32x4_t c0 = *(const f32x4_t*)cp, c1 = *(const f32x4_t*)(cp + 4);
    const f32x4_t s0 = *(const f32x4_t*)sp, s1 = *(const f32x4_t*)(sp + 4);
    float cs[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
    float sn[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
    float o[8];
    norm_rope8(x, r, w, cs, sn, o);
    bf16_t* dst = (isk ? K : Q) + (((long long)b * H + h) * Spad + s) * 128 + c * 8;
    *(bf16x8_t*)dst = pack8(o);
  }
}

// V [token][h*128+d] -> VT [B,H,128,Spad] through an LDS transpose; tile = 64 tokens x 128 d of one head.
// grid: (Spad/64, H, B)
__global__ __launch_bounds__(256) void v_transpose_kernel(const bf16_t* __restrict__ qkv0, const bf16_t* __restrict__ qkv1, int ld0,
                                                          int ld1, int S, int S0, int H, bf16_t* __restrict__ VT, int Spad) {
  __shared__ bf16_t tile[64][136];  // +8 pad: row stride 272 B
  const int t0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const int D = H * 128;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = j * 256 + threadIdx.x;
    const int tok = p >> 4, c = p & 15;
    const int s = t0 + tok;
    bf16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (s < S) {
      const bf16_t* row = (s < S0) ? qkv0 + ((long long)b * S0 + s) * ld0 : qkv1 + ((long long)b * (S - S0) + (s - S0)) * ld1;
      v = *(const bf16x8_t*)(row + 2 * D + h * 128 + c * 8);
    }
    *(bf16x8_t*)&tile[tok][c * 8] = v;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = j * 256 + threadIdx.x;
    const int d = p >> 3, c = p & 7;  // output row d, tokens c*8 .. c*8+7
    bf16x8_t v;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (short)tile[c * 8 + k][d];
    *(bf16x8_t*)(VT + (((long long)b * H + h) * 128 + d) * Spad + t0 + c * 8) = v;
  }
}

// ------------------------------------------------------------------------------------------------------------
// Skinny linear: Y[b][n] = act_out(bias[n] + sum_k W[n][k] act_in(X[b][k])), B <= 16 per launch chunk.
// Each wave owns output rows n (strided); activations live in LDS as f32; weights stream once from HBM.
// ------------------------------------------------------------------------------------------------------------
constexpr int SK_MAXB = 8;

template <int NB>
__global__ __launch_bounds__(256) void skinny_linear_kernel(const void* __restrict__ Xv, int x_is_bf16, const bf16_t* __restrict__ W,
                                                            const bf16_t* __restrict__ bias, float* __restrict__ Y, int ldy, int N,
                                                            int K, int act_in, int act_out, int accumulate, int rpw, long long x_gs, int rows_g) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  // grouped form (x2i_skinny_linear_grouped): blockIdx.y = group g with its own W [N][K], bias [N], rows_g output rows and input rows x_gs elements apart
  if (blockIdx.y) {
    const long long g = blockIdx.y;
    Xv = (const char*)Xv + g * x_gs * (x_is_bf16 ? 2 : 4);
    W += g * N * K;
    if (bias) bias += g * N;
    Y += g * rows_g * ldy;
  }
  // [K][NBP]: the NB samples' values of one k side by side; the row stride is rounded up to an EVEN count so that every sample pair
  // read below (8-byte vector) is 8-byte aligned also for odd NB (the pad column is never read)
  constexpr int NBP = NB == 1 ? 1 : (NB + 1) & ~1;
  extern __shared__ __attribute__((aligned(16))) float xs[];
  for (int i = threadIdx.x; i < NB * K; i += 256) {
    const int b = i / K, k = i - b * K;
    float v = x_is_bf16 ? bf16_to_f32(((const bf16_t*)Xv)[i]) : ((const float*)Xv)[i];
    xs[k * NBP + b] = apply_act(v, act_in);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n_base = (blockIdx.x * 4 + wave) * rpw;  // 4 waves x rpw output rows each
  // R = 4 weight rows are streamed together: 4 x (K/512) independent 16-byte loads in flight per lane (a single row
  // would leave the wave latency-bound on HBM)
  constexpr int R = 4;
  constexpr int NP = NB / 2;  // sample pairs: one packed fma (v_pk_fma_f32) carries two samples' chains
#pragma unroll 1
  for (int rr = 0; rr < rpw; rr += R) {
    const int n0 = n_base + rr;
    if (n0 >= N) break;
    float acc[R][NB];
    f32x2 acc2[R][NP > 0 ? NP : 1];
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[r][b] = 0.f;
#pragma unroll
      for (int q = 0; q < NP; ++q) acc2[r][q] = f32x2{0.f, 0.f};
    }
    for (int k = lane * 8; k < K; k += 512) {
      bf16x8_t wv[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int n = min(n0 + r, N - 1);  // clamp: rows past N are computed and discarded
        wv[r] = __builtin_nontemporal_load((const bf16x8_t*)(W + (long long)n * K + k));  // read once: do not keep it in the caches
      }
      float wf[R][8];
#pragma unroll
      for (int r = 0; r < R; ++r) unpack8(wv[r], wf[r]);
      // explicit fma chains, k ascending: the arithmetic of one sample must not depend on how many samples share the launch (a packed
      // fma is two independent fmas)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float* xk = xs + (k + j) * NBP;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
          const f32x2 x2 = *(const f32x2*)(xk + 2 * q);
#pragma unroll
          for (int r = 0; r < R; ++r) acc2[r][q] = __builtin_elementwise_fma(f32x2{wf[r][j], wf[r][j]}, x2, acc2[r][q]);
        }
        if constexpr (NB & 1) {
          const float x1 = xk[NB - 1];
#pragma unroll
          for (int r = 0; r < R; ++r) acc[r][NB - 1] = fmaf(wf[r][j], x1, acc[r][NB - 1]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int q = 0; q < NP; ++q) { acc[r][2 * q] = acc2[r][q][0]; acc[r][2 * q + 1] = acc2[r][q][1]; }
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[r][b] = wave_sum(acc[r][b]);
      const int n = n0 + r;
      if (lane == 0 && n < N && rr + r < rpw) {
        const float bv = bias ? bf16_to_f32(bias[n]) : 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          float v = apply_act(acc[r][b] + bv, act_out);
          float* yp = Y + (long long)b * ldy + n;
          *yp = accumulate ? (*yp + v) : v;
        }
      }
    }
  }
}

__global__ void sinusoid_kernel(const float* __restrict__ t, float* __restrict__ out, int B, int dim, int round_bf16) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dim / 2;
  if (i >= B * dim) return;
  const int b = i / dim, j = i - b * dim;
  const int k = j < half ? j : j - half;
  const float f = expf(-logf(10000.f) * (float)k / (float)half);
  const float a = t[b] * f;
  float v = j < half ? cosf(a) : sinf(a);
  if (round_bf16) v = bf16_to_f32(f32_to_bf16(v));
  out[i] = v;
}

// FluxPosEmbed table: column c of axis a (dims d_a, pair index k = (c - start_a) / 2) holds cos / sin of
// pos[s][a] * theta^(-2k / d_a), evaluated in float64 as diffusers does, stored as fp32.
__global__ void rope_table_kernel(const float* __restrict__ ids, int S, int d0, int d1, int d2, double theta, float* __restrict__ cosp,
                                  float* __restrict__ sinp) {
  const int D = d0 + d1 + d2;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)S * D) return;
  const int s = (int)(i / D), c = (int)(i - (long long)s * D);
  int a = 0, start = 0, d = d0;
  if (c >= d0 + d1) a = 2, start = d0 + d1, d = d2;
  else if (c >= d0) a = 1, start = d0, d = d1;
  const int k = (c - start) >> 1;
  const double freq = 1.0 / pow(theta, (double)(2 * k) / (double)d);
  const double ang = (double)ids[(long long)s * 3 + a] * freq;
  cosp[i] = (float)cos(ang);
  sinp[i] = (float)sin(ang);
}

__global__ void euler_kernel(bf16_t* __restrict__ x, const bf16_t* __restrict__ eps, long long n8, const float* __restrict__ dt) {
  const float d = dt[0];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    float a[8], e[8];
    unpack8(*(const bf16x8_t*)(x + i * 8), a);
    unpack8(*(const bf16x8_t*)(eps + i * 8), e);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = a[j] + d * e[j];
    *(bf16x8_t*)(x + i * 8) = pack8(a);
  }
}

// x[b][s][:] = bf16(x + gate[b][:] * t[b][s][:])  -- `hidden + gate.unsqueeze(1) * attn_output` (lightcontrol_flux.py:180-181,193-194) as
// its own pass, used only when the per-block attention outputs have to be materialised for forward hooks (x2i_gated_residual_bf16)
__global__ __launch_bounds__(256) void gated_residual_kernel(bf16_t* __restrict__ X, long long x_bs, int ldx, const bf16_t* __restrict__ T,
                                                             long long t_bs, int ldt, const float* __restrict__ gate, long long g_bs, int S,
                                                             int D8, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % D8);
    const long long row = i / D8;
    const int b = (int)(row / S), s = (int)(row - (long long)b * S);
    bf16_t* x = X + (long long)b * x_bs + (long long)s * ldx + c * 8;
    float a[8], t[8];
    unpack8(*(const bf16x8_t*)x, a);
    unpack8(*(const bf16x8_t*)(T + (long long)b * t_bs + (long long)s * ldt + c * 8), t);
    const float* g = gate + (long long)b * g_bs + c * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = fmaf(g[j], t[j], a[j]);
    *(bf16x8_t*)x = pack8(a);
  }
}

__global__ void euler_tail_kernel(bf16_t* x, const bf16_t* eps, long long start, long long n, const float* dt) {
  const long long i = start + blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = f32_to_bf16(bf16_to_f32(x[i]) + dt[0] * bf16_to_f32(eps[i]));
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = f32_to_bf16(x[i]);
}
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = bf16_to_f32(x[i]);
}

// x f32 [B,S,N] -> y f32 [B,N] mean over S.  grid (ceil(N/256), B)
__global__ void seq_mean_kernel(const float* __restrict__ x, float* __restrict__ y, int S, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (n >= N) return;
  const float* p = x + (long long)b * S * N + n;
  float acc = 0.f;
  for (int s = 0; s < S; ++s) acc += p[(long long)s * N];
  y[(long long)b * N + n] = acc / (float)S;
}

inline bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

int x2i_launch_ln_modulate(const void* X, long long x_bs, int ldx, void* Y, long long y_bs, int ldy, int B, int S, int D,
                           int S0, const float* shift0, const float* scale0, const float* shift1, const float* scale1,
                           long long mod_bs, float eps, hipStream_t stream) {
  if (!X || !Y || !shift1 || !scale1 || (S0 > 0 && (!shift0 || !scale0))) return x2i_set_error(X2I_ERR_ARG, "ln_modulate: null pointer");
  if (D % 8 || D > 64 * 8 * LN_MAXV || B <= 0 || S <= 0) return x2i_set_error(X2I_ERR_SHAPE, "ln_modulate: D=%d must be a multiple of 8 and <= %d", D, 64 * 8 * LN_MAXV);
  if (ldx % 8 || ldy % 8 || x_bs % 8 || y_bs % 8 || mod_bs % 4 || !al16(X) || !al16(Y)) return x2i_set_error(X2I_ERR_ALIGN, "ln_modulate: rows must be 16-byte aligned");
  const long long rows = (long long)B * S;
  if (D == 3072 && rows >= 4096) {  // the model's width, enough rows to fill the chip with 4 waves x 4 rows per workgroup
    hipLaunchKernelGGL((ln_rows_kernel<6, 4>), dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, stream, (const bf16_t*)X, x_bs, ldx, (bf16_t*)Y, y_bs,
                       ldy, S, S0, shift0 ? shift0 : shift1, scale0 ? scale0 : scale1, shift1, scale1, mod_bs, eps, rows);
    return x2i_check_launch("ln_modulate");
  }
  hipLaunchKernelGGL(ln_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, (const bf16_t*)X, x_bs, ldx,
                     (bf16_t*)Y, y_bs, ldy, S, D, S0, shift0 ? shift0 : shift1, scale0 ? scale0 : scale1, shift1, scale1, mod_bs,
                     (const bf16_t*)nullptr, (const bf16_t*)nullptr, eps, rows);
  return x2i_check_launch("ln_modulate");
}

int x2i_launch_ln_affine(const void* X, void* Y, long long rows, int D, const void* w, const void* b, float eps, hipStream_t stream) {
  if (!X || !Y || !w || !b) return x2i_set_error(X2I_ERR_ARG, "ln_affine: null pointer");
  if (D % 8 || D > 64 * 8 * LN_MAXV || rows <= 0) return x2i_set_error(X2I_ERR_SHAPE, "ln_affine: D=%d unsupported", D);
  if (!al16(X) || !al16(Y) || !al16(w) || !al16(b)) return x2i_set_error(X2I_ERR_ALIGN, "ln_affine: 16-byte alignment required");
  hipLaunchKernelGGL(ln_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, (const bf16_t*)X, (long long)0, D,
                     (bf16_t*)Y, (long long)0, D, (int)1 << 30, D, 0, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, (long long)0, (const bf16_t*)w, (const bf16_t*)b, eps, rows);
  return x2i_check_launch("ln_affine");
}

int x2i_launch_qkv_split(const void* qkv0, const void* qkv1, int ld0, int ld1, int B, int S, int S0, int H, const void* nq0,
                         const void* nk0, const void* nq1, const void* nk1, const float* cosp, const float* sinp, void* Q,
                         void* K, void* VT, int Spad, float eps, hipStream_t stream) {
  if (S0 < 0 || S0 > S || B <= 0 || S <= 0 || H <= 0) return x2i_set_error(X2I_ERR_SHAPE, "qkv_split: bad shape");
  if ((S0 > 0 && (!qkv0 || !nq0 || !nk0)) || (S0 < S && (!qkv1 || !nq1 || !nk1)) || !cosp || !sinp || !Q || !K || !VT)
    return x2i_set_error(X2I_ERR_ARG, "qkv_split: null pointer");
  if (Spad % 128 || Spad < S) return x2i_set_error(X2I_ERR_SHAPE, "qkv_split: Spad=%d must be a multiple of 128 and >= S=%d", Spad, S);
  if ((S0 > 0 && ld0 % 8) || (S0 < S && ld1 % 8)) return x2i_set_error(X2I_ERR_ALIGN, "qkv_split: ld must be a multiple of 8");
  hipLaunchKernelGGL(qk_norm_rope_kernel, dim3(S, B), dim3(256), 0, stream, (const bf16_t*)qkv0, (const bf16_t*)qkv1, ld0, ld1, S,
                     S0, H, (const bf16_t*)nq0, (const bf16_t*)nk0, (const bf16_t*)nq1, (const bf16_t*)nk1, cosp, sinp, (bf16_t*)Q,
                     (bf16_t*)K, Spad, eps);
  int rc = x2i_check_launch("qk_norm_rope");
  if (rc) return rc;
  hipLaunchKernelGGL(v_transpose_kernel, dim3((S + 63) / 64, H, B), dim3(256), 0, stream, (const bf16_t*)qkv0, (const bf16_t*)qkv1,
                     ld0, ld1, S, S0, H, (bf16_t*)VT, Spad);
  return x2i_check_launch("v_transpose");
}

int x2i_launch_skinny_linear(const void* X, int x_is_bf16, const void* W, const void* bias, float* Y, int ldy, int B, int N, int K,
                             int act_in, int act_out, int accumulate, hipStream_t stream) {
  return x2i_launch_skinny_linear_grouped(X, x_is_bf16, 0, W, bias, Y, ldy, 1, B, N, K, act_in, act_out, accumulate, stream);
}

// G independent skinny linears in one launch: group g has W[g] [N][K], bias[g] [N], input rows X + g * x_gs (x_gs = 0: one input for all groups) and
// output rows g * B .. g * B + B - 1 of Y
int x2i_launch_skinny_linear_grouped(const void* X, int x_is_bf16, long long x_gs, const void* W, const void* bias, float* Y, int ldy, int G, int B,
                                     int N, int K, int act_in, int act_out, int accumulate, hipStream_t stream) {
  if (!X || !W || !Y) return x2i_set_error(X2I_ERR_ARG, "skinny_linear: null pointer");
  if (G <= 0 || G > 65535 || x_gs < 0) return x2i_set_error(X2I_ERR_SHAPE, "skinny_linear: bad group count / stride (G=%d)", G);
  if (G > 1 && B > SK_MAXB) return x2i_set_error(X2I_ERR_SHAPE, "skinny_linear_grouped: at most %d rows per group (B=%d)", SK_MAXB, B);
  if (B <= 0 || N <= 0 || K <= 0 || K % 8) return x2i_set_error(X2I_ERR_SHAPE, "skinny_linear: K=%d must be a multiple of 8", K);
  if (!al16(W)) return x2i_set_error(X2I_ERR_ALIGN, "skinny_linear: W must be 16-byte aligned");
  const int esz = x_is_bf16 ? 2 : 4;
  for (int b0 = 0; b0 < B; b0 += SK_MAXB) {
    const int nb = (B - b0) < SK_MAXB ? (B - b0) : SK_MAXB;
    const void* xp = (const char*)X + (long long)b0 * K * esz;
    float* yp = Y + (long long)b0 * ldy;
    // many rows per block when N is huge (the 1M-row AdaLN modulation table) so the activations are staged once
    const int rpw = N >= 65536 ? 16 : 4;
    const dim3 grid((N + 4 * rpw - 1) / (4 * rpw), G), block(256);
    const size_t shm = (size_t)(nb == 1 ? 1 : (nb + 1) & ~1) * K * 4;   // (row stride padded to an even sample count)
    if (shm > 160 * 1024) return x2i_set_error(X2I_ERR_SHAPE, "skinny_linear: B*K too large for LDS");
#define SK_CASE(NB)                                                                                                   \
  case NB: {                                                                                                          \
    hipError_t e = hipFuncSetAttribute((const void*)skinny_linear_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       (int)shm);                                                                     \
    if (e != hipSuccess) return x2i_set_error(X2I_ERR_HIP, "skinny_linear: %s", hipGetErrorString(e));                \
    hipLaunchKernelGGL(skinny_linear_kernel<NB>, grid, block, shm, stream, xp, x_is_bf16, (const bf16_t*)W,           \
                       (const bf16_t*)bias, yp, ldy, N, K, act_in, act_out, accumulate, rpw, x_gs, B);               \
  } break;
    switch (nb) {
      SK_CASE(1) SK_CASE(2) SK_CASE(3) SK_CASE(4) SK_CASE(5) SK_CASE(6) SK_CASE(7) SK_CASE(8)
    }
#undef SK_CASE
    int rc = x2i_check_launch("skinny_linear");
    if (rc) return rc;
  }
  return X2I_OK;
}

int x2i_launch_timestep_sinusoid(const float* t, float* out, int B, int dim, int round_bf16, hipStream_t stream) {
  if (!t || !out || B <= 0 || dim <= 0 || dim % 2) return x2i_set_error(X2I_ERR_ARG, "timestep_sinusoid: bad argument");
  hipLaunchKernelGGL(sinusoid_kernel, dim3((B * dim + 255) / 256), dim3(256), 0, stream, t, out, B, dim, round_bf16);
  return x2i_check_launch("timestep_sinusoid");
}

int x2i_launch_rope_table(const float* ids, int S, int d0, int d1, int d2, float theta, float* cosp, float* sinp, hipStream_t stream) {
  if (!ids || !cosp || !sinp || S <= 0) return x2i_set_error(X2I_ERR_ARG, "rope_table: bad argument");
  if (d0 < 0 || d1 < 0 || d2 < 0 || (d0 | d1 | d2) & 1 || d0 + d1 + d2 <= 0 || theta <= 0.f)
    return x2i_set_error(X2I_ERR_SHAPE, "rope_table: axis dims must be even and non-negative (%d,%d,%d)", d0, d1, d2);
  const long long n = (long long)S * (d0 + d1 + d2);
  hipLaunchKernelGGL(rope_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ids, S, d0, d1, d2, (double)theta, cosp, sinp);
  return x2i_check_launch("rope_table");
}

int x2i_launch_gated_residual(void* X, long long x_bs, int ldx, const void* T, long long t_bs, int ldt, const float* gate, long long g_bs,
                              int B, int S, int D, hipStream_t stream) {
  if (!X || !T || !gate || B <= 0 || S <= 0 || D <= 0) return x2i_set_error(X2I_ERR_ARG, "gated_residual: bad argument");
  if (D % 8 || ldx % 8 || ldt % 8 || x_bs % 8 || t_bs % 8 || g_bs % 4 || !al16(X) || !al16(T) || !al16(gate))
    return x2i_set_error(X2I_ERR_ALIGN, "gated_residual: rows must be 16-byte aligned, D %% 8 == 0");
  const long long total = (long long)B * S * (D / 8);
  const long long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(gated_residual_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, stream, (bf16_t*)X, x_bs, ldx,
                     (const bf16_t*)T, t_bs, ldt, gate, g_bs, S, D / 8, total);
  return x2i_check_launch("gated_residual");
}

int x2i_launch_euler_step(void* x, const void* eps, long long n, const float* dt, hipStream_t stream) {
  if (!x || !eps || !dt || n <= 0) return x2i_set_error(X2I_ERR_ARG, "euler_step: bad argument");
  const long long n8 = (al16(x) && al16(eps)) ? n / 8 : 0;
  if (n8 > 0) {
    const long long blocks = (n8 + 255) / 256;
    hipLaunchKernelGGL(euler_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream, (bf16_t*)x,
                       (const bf16_t*)eps, n8, dt);
  }
  if (n8 * 8 < n) {
    const long long rem = n - n8 * 8;
    hipLaunchKernelGGL(euler_tail_kernel, dim3((unsigned)((rem + 255) / 256)), dim3(256), 0, stream, (bf16_t*)x, (const bf16_t*)eps,
                       n8 * 8, n, dt);
  }
  return x2i_check_launch("euler_step");
}

int x2i_launch_seq_mean(const float* x, float* y, int B, int S, int N, hipStream_t stream) {
  if (!x || !y || B <= 0 || S <= 0 || N <= 0) return x2i_set_error(X2I_ERR_ARG, "seq_mean: bad argument");
  hipLaunchKernelGGL(seq_mean_kernel, dim3((N + 255) / 256, B), dim3(256), 0, stream, x, y, S, N);
  return x2i_check_launch("seq_mean");
}

int x2i_launch_cast_f32_bf16(const float* x, void* y, long long n, hipStream_t stream) {
  if (!x || !y || n <= 0) return x2i_set_error(X2I_ERR_ARG, "cast: bad argument");
  const long long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, stream, x, (bf16_t*)y, n);
  return x2i_check_launch("cast_f32_bf16");
}
int x2i_launch_cast_bf16_f32(const void* x, float* y, long long n, hipStream_t stream) {
  if (!x || !y || n <= 0) return x2i_set_error(X2I_ERR_ARG, "cast: bad argument");
  const long long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, stream, (const bf16_t*)x, y, n);
  return x2i_check_launch("cast_bf16_f32");
}
