// Backward kernels of the attention-distillation step (reference train/train_qwenvl.py:556-654:
// `loss.backward()` with only the projector trainable -- the gradient runs back through the frozen transformer's ACTIVATIONS, there
// are no weight gradients inside the DiT).  Every matrix product of the backward pass is a launch of the forward MFMA GEMM
// (x2i_gemm_bf16) on transposed operands; this file holds what surrounds them, all HBM-bound row kernels:
//
//   transpose            batched bf16 2-D transpose (frozen weights once; K, Q, dO, P, dS per attention backward)
//   softmax_fwd / bwd    P = softmax(scale * S) over the valid keys, dS = scale * P * (dP - rowsum(P * dP)), padding zero-filled
//   ln_mod_bwd           LayerNorm(no affine) * mult + shift backward: dx (added to the residual-stream gradient) and the column
//                        sums that become d(scale) / d(shift) of the AdaLN modulation (or d(weight) / d(bias) of an affine LN)
//   gate_bwd             x + gate * t backward: dt = gate * dx (+ the distillation gradient tapped at t), d(gate) column sums
//   act_bwd              d(pre) = d(act) * act'(pre) for GELU-tanh / GELU-erf / SiLU
//   qkv_split_bwd        RoPE^T, RMSNorm backward and the head gather: dQ, dK, dV [B,H,Spad,128] -> d(q|k|v) rows
//   skinny_bwd           dx[b][:] = sum_n dy[b][n] W[n][:] for the AdaLN / embedder linears (tiny batch, 1e6 rows)
//   kd_loss              KL(softmax(normalize(teacher)/T) || softmax(normalize(student)/T)) per row and its gradient (:613-634)
//
// Column sums are two-stage (per-wave partials in a scratch buffer, then reduce_rows) so that results do not depend on atomics order.
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
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
constexpr int TR_MAXV = 8;  // row kernels keep up to 64 lanes x 8 chunks x 8 elements = 4096 columns in registers

// ---------------------------------------------------------------------------------------------------------------------
// out[z][c][r] = in[z][r][c]; 64 x 64 tiles through LDS, 16-byte accesses on both sides.  R % 8 == 0, C % 8 == 0.
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ in, long long in_bs, long long ld_in,
                                                        bf16_t* __restrict__ out, long long out_bs, long long ld_out, int R, int C) {
  __shared__ bf16_t tile[64][72];  // +8 pad
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const bf16_t* src = in + (long long)blockIdx.z * in_bs;
  bf16_t* dst = out + (long long)blockIdx.z * out_bs;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int p = j * 256 + threadIdx.x, r = p >> 3, c = (p & 7) * 8;
    bf16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (r0 + r < R && c0 + c < C) v = *(const bf16x8_t*)(src + (long long)(r0 + r) * ld_in + c0 + c);
    *(bf16x8_t*)&tile[r][c] = v;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int p = j * 256 + threadIdx.x, c = p >> 3, r = (p & 7) * 8;
    if (c0 + c < C && r0 + r < R) {
      bf16x8_t v;
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = (short)tile[r + k][c];
      *(bf16x8_t*)(dst + (long long)(c0 + c) * ld_out + r0 + r) = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Row softmax with padding: x [nz][Rt][ld]; rows < Rv, cols < Cv are valid, everything up to (Rt, Ct) is written as zero.
// One wave per row, three passes over the (L2-resident) row.
__global__ __launch_bounds__(256) void softmax_pad_kernel(bf16_t* __restrict__ x, long long ld, int Rt, int Rv, int Ct, int Cv, float scale_log2,
                                                          long long total_rows) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= total_rows) return;
  const int r = (int)(row % Rt);
  bf16_t* p = x + row * ld;
  const int nct = Ct >> 3;
  if (r >= Rv) {
    for (int c = lane; c < nct; c += 64) *(bf16x8_t*)(p + c * 8) = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
    return;
  }
  float mx = -3.0e38f;
  for (int c = lane; c < nct; c += 64) {
    float v[8];
    unpack8(*(const bf16x8_t*)(p + c * 8), v);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (c * 8 + j < Cv) mx = fmaxf(mx, v[j]);
  }
  mx = wave_max(mx) * scale_log2;
  float sum = 0.f;
  for (int c = lane; c < nct; c += 64) {
    float v[8];
    unpack8(*(const bf16x8_t*)(p + c * 8), v);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (c * 8 + j < Cv) sum += __builtin_amdgcn_exp2f(v[j] * scale_log2 - mx);
  }
  const float inv = 1.f / wave_sum(sum);
  for (int c = lane; c < nct; c += 64) {
    float v[8];
    unpack8(*(const bf16x8_t*)(p + c * 8), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (c * 8 + j < Cv) ? __builtin_amdgcn_exp2f(v[j] * scale_log2 - mx) * inv : 0.f;
    *(bf16x8_t*)(p + c * 8) = pack8(v);
  }
}

// dS = scale * P * (dP - sum_j P_j dP_j), written over dP; same padding rules (P's padding is already zero)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const bf16_t* __restrict__ P, bf16_t* __restrict__ dP, long long ld, int Rt, int Rv, int Ct,
                                                          int Cv, float scale, long long total_rows) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= total_rows) return;
  const int r = (int)(row % Rt);
  const bf16_t* p = P + row * ld;
  bf16_t* d = dP + row * ld;
  const int nct = Ct >> 3;
  if (r >= Rv) {
    for (int c = lane; c < nct; c += 64) *(bf16x8_t*)(d + c * 8) = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
    return;
  }
  float dot = 0.f;
  for (int c = lane; c < nct; c += 64) {
    float a[8], g[8];
    unpack8(*(const bf16x8_t*)(p + c * 8), a);
    unpack8(*(const bf16x8_t*)(d + c * 8), g);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (c * 8 + j < Cv) dot = fmaf(a[j], g[j], dot);
  }
  dot = wave_sum(dot);
  for (int c = lane; c < nct; c += 64) {
    float a[8], g[8];
    unpack8(*(const bf16x8_t*)(p + c * 8), a);
    unpack8(*(const bf16x8_t*)(d + c * 8), g);
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = (c * 8 + j < Cv) ? scale * a[j] * (g[j] - dot) : 0.f;
    *(bf16x8_t*)(d + c * 8) = pack8(g);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// y = xhat * mult + shift (xhat = LayerNorm without affine).  A wave owns R consecutive rows of one sample:
//   dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * mult;   dXout = dXin + dx   (dXin may be NULL)
//   partial[b][w][0][:] = sum_rows dy * xhat   (-> d mult: d scale of the modulation / d weight of an affine LN)
//   partial[b][w][1][:] = sum_rows dy          (-> d shift / d bias)
// mult_is_scale: mult = 1 + m[b][col] (AdaLN) else mult = m[col] (affine weight as f32).
__global__ __launch_bounds__(256) void ln_mod_bwd_kernel(const bf16_t* __restrict__ X, long long x_bs, int ldx, const bf16_t* __restrict__ dY,
                                                         long long dy_bs, int ldy, const float* __restrict__ m, long long m_bs,
                                                         int mult_is_scale, const bf16_t* dXin, bf16_t* __restrict__ dXout, long long dx_bs,
                                                         int lddx, int S, int D, int R, float* __restrict__ partial, float eps, int nwaves_per_b,
                                                         int B) {
  const int lane = threadIdx.x & 63;
  const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (long long)B * nwaves_per_b) return;
  const int b = (int)(wid / nwaves_per_b), w = (int)(wid % nwaves_per_b);
  const int nv = D >> 3;
  float mul[TR_MAXV][8], a0[TR_MAXV][8], a1[TR_MAXV][8];
#pragma unroll
  for (int i = 0; i < TR_MAXV; ++i) {
    const int c = lane + i * 64;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a0[i][j] = 0.f; a1[i][j] = 0.f;
      mul[i][j] = c < nv ? (mult_is_scale ? 1.f + m[(long long)b * m_bs + c * 8 + j] : m[c * 8 + j]) : 0.f;
    }
  }
  for (int rr = 0; rr < R; ++rr) {
    const int s = w * R + rr;
    if (s >= S) break;
    const bf16_t* x = X + (long long)b * x_bs + (long long)s * ldx;
    const bf16_t* dy = dY + (long long)b * dy_bs + (long long)s * ldy;
    float v[TR_MAXV][8], g[TR_MAXV][8];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < TR_MAXV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        unpack8(*(const bf16x8_t*)(x + c * 8), v[i]);
        unpack8(*(const bf16x8_t*)(dy + c * 8), g[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += v[i][j];
      }
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < TR_MAXV; ++i)
      if (lane + i * 64 < nv) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = v[i][j] - mean;
          sq += d * d;
        }
      }
    const float rstd = rsqrtf(wave_sum(sq) / (float)D + eps);
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < TR_MAXV; ++i)
      if (lane + i * 64 < nv) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = (v[i][j] - mean) * rstd;
          a0[i][j] = fmaf(g[i][j], xh, a0[i][j]);
          a1[i][j] += g[i][j];
          const float gm = g[i][j] * mul[i][j];
          v[i][j] = xh;      // keep xhat
          g[i][j] = gm;      // keep g
          sg += gm;
          sgx = fmaf(gm, xh, sgx);
        }
      }
    const float mg = wave_sum(sg) / (float)D, mgx = wave_sum(sgx) / (float)D;
    bf16_t* dxo = dXout + (long long)b * dx_bs + (long long)s * lddx;
    const bf16_t* dxi = dXin ? dXin + (long long)b * dx_bs + (long long)s * lddx : nullptr;
#pragma unroll
    for (int i = 0; i < TR_MAXV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        float o[8];
        if (dxi) unpack8(*(const bf16x8_t*)(dxi + c * 8), o);
        else {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += rstd * (g[i][j] - mg - v[i][j] * mgx);
        *(bf16x8_t*)(dxo + c * 8) = pack8(o);
      }
    }
  }
  float* pp = partial + ((long long)b * nwaves_per_b + w) * 2 * D;
#pragma unroll
  for (int i = 0; i < TR_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pp[c * 8 + j] = a0[i][j];
        pp[D + c * 8 + j] = a1[i][j];
      }
    }
  }
}

// x_out = x + gate * t backward:  dT = gate[b][:] * dX (+ G, the gradient injected at t by the distillation loss);
// partial[b][w][:] = sum_rows dX * T  (-> d gate).  gate == NULL: plain residual (dT = dX (+ G)), no partials.
__global__ __launch_bounds__(256) void gate_bwd_kernel(const bf16_t* __restrict__ dX, long long dx_bs, int lddx, const bf16_t* __restrict__ T,
                                                       long long t_bs, int ldt, const float* __restrict__ gate, long long g_bs,
                                                       const bf16_t* __restrict__ G, long long gg_bs, int ldg, bf16_t* __restrict__ dT,
                                                       long long dt_bs, int lddt, int S, int D, int R, float* __restrict__ partial,
                                                       int nwaves_per_b, int B) {
  const int lane = threadIdx.x & 63;
  const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= (long long)B * nwaves_per_b) return;
  const int b = (int)(wid / nwaves_per_b), w = (int)(wid % nwaves_per_b);
  const int nv = D >> 3;
  float gt[TR_MAXV][8], acc[TR_MAXV][8];
#pragma unroll
  for (int i = 0; i < TR_MAXV; ++i) {
    const int c = lane + i * 64;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc[i][j] = 0.f;
      gt[i][j] = (c < nv) ? (gate ? gate[(long long)b * g_bs + c * 8 + j] : 1.f) : 0.f;
    }
  }
  for (int rr = 0; rr < R; ++rr) {
    const int s = w * R + rr;
    if (s >= S) break;
#pragma unroll
    for (int i = 0; i < TR_MAXV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        float d[8], t[8], o[8];
        unpack8(*(const bf16x8_t*)(dX + (long long)b * dx_bs + (long long)s * lddx + c * 8), d);
        if (gate) {
          unpack8(*(const bf16x8_t*)(T + (long long)b * t_bs + (long long)s * ldt + c * 8), t);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[i][j] = fmaf(d[j], t[j], acc[i][j]);
        }
        if (G) unpack8(*(const bf16x8_t*)(G + (long long)b * gg_bs + (long long)s * ldg + c * 8), o);
        else {
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = fmaf(gt[i][j], d[j], o[j]);
        *(bf16x8_t*)(dT + (long long)b * dt_bs + (long long)s * lddt + c * 8) = pack8(o);
      }
    }
  }
  if (gate && partial) {
    float* pp = partial + ((long long)b * nwaves_per_b + w) * D;
#pragma unroll
    for (int i = 0; i < TR_MAXV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
#pragma unroll
        for (int j = 0; j < 8; ++j) pp[c * 8 + j] = acc[i][j];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same two operators with a WORKGROUP per R rows and a THREAD per eight columns (D / 8 <= 512 threads): the wave-per-row forms above
// keep a whole row and its column accumulators in one wave's registers (320+ registers: one wave per SIMD, one row's loads in flight),
// and at batch 1 their 576 waves leave half the SIMDs empty -- 0.9-1.4 TB/s on kernels that only stream rows
// (profiles/r03i_train_b1_kernel_stats.csv: 90 / 62 us per call for 112 / 84 MB).  Here a thread owns its eight columns for all R rows
// (its slice of the column sums stays in 8-16 registers), loads RB rows at a time, and the row statistics of ln_mod_bwd are
// workgroup reductions (wave sums, one LDS exchange per round, fixed order).  Same partial layout: partial[b][w][stat][:], w = row group.
// wave sum on the VALU's DPP path (four steps inside a row of 16 lanes, two row broadcasts): the total arrives in lane 63.  (wave_sum's six
// __shfl_xor steps are six dependent LDS-crossbar round trips; a row kernel that needs eight sums per row step spends its time there.)
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_step(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_lane63(float v) {
  v = dpp_step<0xB1>(v);        // quad_perm [1,0,3,2]
  v = dpp_step<0x4E>(v);        // quad_perm [2,3,0,1]
  v = dpp_step<0x141>(v);       // row_half_mirror
  v = dpp_step<0x140>(v);       // row_mirror: every lane of a 16-lane row holds the row's sum
  v = dpp_step<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v = dpp_step<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}
// every thread returns the workgroup's sums of x[0..NV); `red` is one of the cyclically used LDS buffers (one barrier per call: a buffer is
// rewritten only two calls later, behind the barrier of the call in between)
template <int NV>
__device__ __forceinline__ void block_sums(float (&x)[NV], float* red, int nwaves) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) x[k] = wave_sum_lane63(x[k]);
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < NV; ++k) red[wv * NV + k] = x[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    float a = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) a += red[u * NV + k];   // (slots of waves the workgroup does not have hold zeros: cleared once by the kernel)
    x[k] = a;
  }
}

constexpr int LB_RB = 2;   // rows per step of ln_mod_bwd_rows_kernel (measured: 4 rows = 227 registers, two waves per SIMD, 3.4 TB/s; 2 rows = 120 registers, four
                           // waves, 4.2-4.6 TB/s; 8 rows 1.2 TB/s: tools/train_rows_probe.py)
constexpr int GB_RB = 4;   // ... of gate_bwd_rows_kernel (no reductions: loads in flight are all that matters)
__global__ __launch_bounds__(512) void ln_mod_bwd_rows_kernel(const bf16_t* __restrict__ X, long long x_bs, int ldx, const bf16_t* __restrict__ dY,
                                                               long long dy_bs, int ldy, const float* __restrict__ m, long long m_bs,
                                                               int mult_is_scale, const bf16_t* dXin, bf16_t* __restrict__ dXout, long long dx_bs,
                                                               int lddx, int S, int D, int R, float* __restrict__ partial, float eps,
                                                               int ngroups_per_b) {
  __shared__ float red[3][8 * 2 * LB_RB];   // one exchange buffer per reduction round of a step (0: sum x, 1: sum (x - mean)^2, 2: sum g, sum g xhat)
  for (int i = threadIdx.x; i < 3 * 8 * 2 * LB_RB; i += blockDim.x) (&red[0][0])[i] = 0.f;
  __syncthreads();
  const int b = blockIdx.x / ngroups_per_b, w = blockIdx.x % ngroups_per_b;
  const int c = threadIdx.x, nv = D >> 3, nwaves = blockDim.x >> 6;
  const bool act = c < nv;
  const float invD = 1.f / (float)D;
  float mul[8], a0[8], a1[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a0[j] = a1[j] = 0.f;
    mul[j] = act ? (mult_is_scale ? 1.f + m[(long long)b * m_bs + c * 8 + j] : m[c * 8 + j]) : 0.f;
  }
  const int s_end = min(S, (w + 1) * R);
  for (int s0 = w * R; s0 < s_end; s0 += LB_RB) {
    float v[LB_RB][8], g[LB_RB][8], st[2 * LB_RB];
    bf16x8_t dxi[LB_RB];
#pragma unroll
    for (int k = 0; k < LB_RB; ++k) {   // every load of the step first
      const bool live = act && s0 + k < s_end;
      bf16x8_t xv = {0, 0, 0, 0, 0, 0, 0, 0}, gv = {0, 0, 0, 0, 0, 0, 0, 0};
      dxi[k] = xv;
      if (live) {
        xv = *(const bf16x8_t*)(X + (long long)b * x_bs + (long long)(s0 + k) * ldx + c * 8);
        gv = *(const bf16x8_t*)(dY + (long long)b * dy_bs + (long long)(s0 + k) * ldy + c * 8);
        if (dXin) dxi[k] = *(const bf16x8_t*)(dXin + (long long)b * dx_bs + (long long)(s0 + k) * lddx + c * 8);
      }
      unpack8(xv, v[k]);
      unpack8(gv, g[k]);
    }
    // rounds 1a / 1b: the row mean, then the variance as mean((x - mean)^2) -- the two-pass form of the forward kernel and of the
    // wave-per-row backward (E[x^2] - mean^2 loses the variance of a row whose mean is large against its spread: massive-activation rows)
    float mean[LB_RB], rstd[LB_RB], s1[LB_RB];
#pragma unroll
    for (int k = 0; k < LB_RB; ++k) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) a += v[k][j];
      s1[k] = a;
    }
    block_sums<LB_RB>(s1, red[0], nwaves);
#pragma unroll
    for (int k = 0; k < LB_RB; ++k) {
      mean[k] = s1[k] * invD;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = act ? v[k][j] - mean[k] : 0.f;
        q = fmaf(d, d, q);
      }
      s1[k] = q;
    }
    block_sums<LB_RB>(s1, red[1], nwaves);
#pragma unroll
    for (int k = 0; k < LB_RB; ++k) {
      rstd[k] = rsqrtf(s1[k] * invD + eps);
      float sg = 0.f, sgx = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = act ? (v[k][j] - mean[k]) * rstd[k] : 0.f;
        a0[j] = fmaf(g[k][j], xh, a0[j]);
        a1[j] += g[k][j];
        const float gm = g[k][j] * mul[j];
        v[k][j] = xh;   // keep xhat
        g[k][j] = gm;   // keep g * mult
        sg += gm;
        sgx = fmaf(gm, xh, sgx);
      }
      st[2 * k] = sg;
      st[2 * k + 1] = sgx;
    }
    block_sums<2 * LB_RB>(st, red[2], nwaves);   // round 2: sum g and sum g * xhat
#pragma unroll
    for (int k = 0; k < LB_RB; ++k) {
      if (act && s0 + k < s_end) {
        const float mg = st[2 * k] * invD, mgx = st[2 * k + 1] * invD;
        float o[8];
        unpack8(dxi[k], o);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += rstd[k] * (g[k][j] - mg - v[k][j] * mgx);
        *(bf16x8_t*)(dXout + (long long)b * dx_bs + (long long)(s0 + k) * lddx + c * 8) = pack8(o);
      }
    }
  }
  if (act) {
    float* pp = partial + ((long long)b * ngroups_per_b + w) * 2 * D + c * 8;
    *(f32x4_t*)pp = (f32x4_t){a0[0], a0[1], a0[2], a0[3]};
    *(f32x4_t*)(pp + 4) = (f32x4_t){a0[4], a0[5], a0[6], a0[7]};
    *(f32x4_t*)(pp + D) = (f32x4_t){a1[0], a1[1], a1[2], a1[3]};
    *(f32x4_t*)(pp + D + 4) = (f32x4_t){a1[4], a1[5], a1[6], a1[7]};
  }
}

__global__ __launch_bounds__(512) void gate_bwd_rows_kernel(const bf16_t* __restrict__ dX, long long dx_bs, int lddx, const bf16_t* __restrict__ T,
                                                             long long t_bs, int ldt, const float* __restrict__ gate, long long g_bs,
                                                             const bf16_t* __restrict__ G, long long gg_bs, int ldg, bf16_t* __restrict__ dT,
                                                             long long dt_bs, int lddt, int S, int D, int R, float* __restrict__ partial,
                                                             int ngroups_per_b) {
  const int b = blockIdx.x / ngroups_per_b, w = blockIdx.x % ngroups_per_b;
  const int c = threadIdx.x;
  if (c >= (D >> 3)) return;
  float gt[8], acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    acc[j] = 0.f;
    gt[j] = gate ? gate[(long long)b * g_bs + c * 8 + j] : 1.f;
  }
  const int s_end = min(S, (w + 1) * R);
  for (int s0 = w * R; s0 < s_end; s0 += GB_RB) {
    bf16x8_t dv[GB_RB], tv[GB_RB], ov[GB_RB];
#pragma unroll
    for (int k = 0; k < GB_RB; ++k) {   // every load of the step first
      const bool live = s0 + k < s_end;
      dv[k] = tv[k] = ov[k] = (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
      if (live) {
        dv[k] = *(const bf16x8_t*)(dX + (long long)b * dx_bs + (long long)(s0 + k) * lddx + c * 8);
        if (gate) tv[k] = *(const bf16x8_t*)(T + (long long)b * t_bs + (long long)(s0 + k) * ldt + c * 8);
        if (G) ov[k] = *(const bf16x8_t*)(G + (long long)b * gg_bs + (long long)(s0 + k) * ldg + c * 8);
      }
    }
#pragma unroll
    for (int k = 0; k < GB_RB; ++k) {
      if (s0 + k < s_end) {
        float d[8], t[8], o[8];
        unpack8(dv[k], d);
        unpack8(tv[k], t);
        unpack8(ov[k], o);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          acc[j] = fmaf(d[j], t[j], acc[j]);
          o[j] = fmaf(gt[j], d[j], o[j]);
        }
        *(bf16x8_t*)(dT + (long long)b * dt_bs + (long long)(s0 + k) * lddt + c * 8) = pack8(o);
      }
    }
  }
  if (gate && partial) {
    float* pp = partial + ((long long)b * ngroups_per_b + w) * D + c * 8;
    *(f32x4_t*)pp = (f32x4_t){acc[0], acc[1], acc[2], acc[3]};
    *(f32x4_t*)(pp + 4) = (f32x4_t){acc[4], acc[5], acc[6], acc[7]};
  }
}

// out[z][i] (op)= sum_{p < np} in[z][p][i]   (second stage of every column sum); accumulate != 0: add to what is there.
// A block owns 64 consecutive i and splits the partial rows over its 4 waves (fixed order: deterministic), four loads in flight per lane.
__global__ __launch_bounds__(1024) void reduce_rows_kernel(const float* __restrict__ in, long long in_zs, int np, long long in_ps, float* __restrict__ out,
                                                           long long out_zs, int len, int accumulate, float alpha) {
  // 16 waves per block, wave w takes partial rows w, w + 16, ...; eight loads in flight per lane (the first form -- four waves, four
  // loads -- walked 576 partial rows in 36 dependent steps: 16.8 us for 7 MB, 432 calls per training step)
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (i < len) {
    const float* p = in + (long long)blockIdx.y * in_zs + i;
    float acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = 0.f;
    int k = w;
    for (; k + 7 * 16 < np; k += 8 * 16) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += p[(long long)(k + u * 16) * in_ps];
    }
    for (; k < np; k += 16) acc[0] += p[(long long)k * in_ps];
    s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  }
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && i < len) {
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) t += red[u][lane];
    float* o = out + (long long)blockIdx.y * out_zs + i;
    *o = accumulate ? *o + alpha * t : alpha * t;
  }
}

// d(pre) = d(act) * act'(pre), written over d(act); rows x cols with independent row strides
__device__ __forceinline__ float act_grad(float x, int act) {
  if (act == X2I_ACT_GELU_TANH) {
    const float k = 0.7978845608028654f, u = k * (x + 0.044715f * x * x * x), t = tanhf(u);
    return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * k * (1.f + 3.f * 0.044715f * x * x);
  }
  if (act == X2I_ACT_GELU_ERF) return 0.5f * (1.f + erff(x * 0.7071067811865476f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
  if (act == X2I_ACT_SILU) {
    const float sg = 1.f / (1.f + expf(-x));
    return sg * (1.f + x * (1.f - sg));
  }
  return 1.f;
}
__global__ __launch_bounds__(256) void act_bwd_kernel(bf16_t* __restrict__ dA, long long ldd, const bf16_t* __restrict__ pre, long long ldp,
                                                      long long rows, int cols8, int act) {
  const long long total = rows * cols8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / cols8;
    const int c = (int)(i - r * cols8);
    float d[8], p[8];
    unpack8(*(const bf16x8_t*)(dA + r * ldd + c * 8), d);
    unpack8(*(const bf16x8_t*)(pre + r * ldp + c * 8), p);
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] *= act_grad(p[j], act);
    *(bf16x8_t*)(dA + r * ldd + c * 8) = pack8(d);
  }
}
__global__ void act_bwd_f32_kernel(float* __restrict__ dA, const float* __restrict__ pre, long long n, int act) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dA[i] *= act_grad(pre[i], act);
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of qk_norm_rope + the V gather (elementwise.hip forward kernels): token s of sample b, head h:
//   dn = RoPE^T dQ;  dx_i = r w_i dn_i - x_i r^3 / 128 * sum_k dn_k w_k x_k   (x = the saved pre-norm q / k row, r = rsqrt(mean x^2 + eps))
//   dv = dV[b][h][s][:]
// written into d(qkv) with the forward's two-source row addressing (text rows / image rows).
__global__ __launch_bounds__(256) void qkv_split_bwd_kernel(const bf16_t* __restrict__ qkv0, const bf16_t* __restrict__ qkv1, int ld0, int ld1,
                                                            bf16_t* __restrict__ d0, bf16_t* __restrict__ d1, int ldd0, int ldd1, int S, int S0,
                                                            int H, const bf16_t* nq0, const bf16_t* nk0, const bf16_t* nq1, const bf16_t* nk1,
                                                            const float* __restrict__ cosp, const float* __restrict__ sinp,
                                                            const bf16_t* __restrict__ dQ, const bf16_t* __restrict__ dK,
                                                            const bf16_t* __restrict__ dV, int Spad, float eps) {
  const int s = blockIdx.x, b = blockIdx.y;
  const bool src0 = s < S0;
  const long long rowi = src0 ? ((long long)b * S0 + s) : ((long long)b * (S - S0) + (s - S0));
  const bf16_t* row = src0 ? qkv0 + rowi * ld0 : qkv1 + rowi * ld1;
  bf16_t* drow = src0 ? d0 + rowi * ldd0 : d1 + rowi * ldd1;
  const int D = H * 128;
  const int units = 3 * H * 16;
  for (int u = threadIdx.x; u < units; u += 256) {
    const int part = u / (H * 16);  // 0 q, 1 k, 2 v
    const int rem = u - part * H * 16;
    const int h = rem >> 4, c = rem & 15;
    const long long hoff = (((long long)b * H + h) * Spad + s) * 128 + c * 8;
    if (part == 2) {  // (all 16 lanes of a (token, head) take the same branch: part is uniform over them)
      *(bf16x8_t*)(drow + 2 * D + h * 128 + c * 8) = *(const bf16x8_t*)(dV + hoff);
      continue;
    }
    float x[8], dy[8], w[8];
    unpack8(*(const bf16x8_t*)(row + part * D + h * 128 + c * 8), x);
    unpack8(*(const bf16x8_t*)((part ? dK : dQ) + hoff), dy);
    const bf16_t* wn = part ? (src0 ? nk0 : nk1) : (src0 ? nq0 : nq1);
    unpack8(*(const bf16x8_t*)(wn + c * 8), w);
    const float* cp = cosp + (long long)s * 128 + c * 8;
    const float* sp = sinp + (long long)s * 128 + c * 8;
    float dn[8];
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      dn[j] = dy[j] * cp[j] + dy[j + 1] * sp[j + 1];
      dn[j + 1] = -dy[j] * sp[j] + dy[j + 1] * cp[j + 1];
    }
    float ss = 0.f, dot = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      ss += x[j] * x[j];
      dot += dn[j] * w[j] * x[j];
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      ss += __shfl_xor(ss, o, 64);
      dot += __shfl_xor(dot, o, 64);
    }
    const float r = rsqrtf(ss * (1.f / 128.f) + eps);
    const float k = r * r * r * (1.f / 128.f) * dot;
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = r * w[j] * dn[j] - x[j] * k;
    *(bf16x8_t*)(drow + part * D + h * 128 + c * 8) = pack8(o);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// dx[b][k] = sum_n dy[b][n] * W[n][k]   (W bf16 [N][K] row-major, dy f32 [B][N], B <= 8): a block takes `chunk` rows of W and writes
// its partial [B][K] sums; reduce_rows finishes.  K <= 256 * 8 * 2.
__global__ __launch_bounds__(256) void skinny_bwd_kernel(const float* __restrict__ dy, long long dy_bs, const bf16_t* __restrict__ W, int ldw,
                                                         float* __restrict__ partial, int B, int N, int K, int chunk) {
  __shared__ float dys[8][256];
  const int n0 = blockIdx.x * chunk;
  const int nv = K >> 3;
  float acc[2][8][8];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][b][j] = 0.f;
  for (int base = 0; base < chunk; base += 256) {
    __syncthreads();
    for (int b = 0; b < B; ++b) {
      const int n = n0 + base + threadIdx.x;
      dys[b][threadIdx.x] = (base + threadIdx.x < chunk && n < N) ? dy[(long long)b * dy_bs + n] : 0.f;
    }
    __syncthreads();
    const int lim = min(256, min(chunk - base, N - n0 - base));
    for (int t = 0; t < lim; ++t) {
      const bf16_t* wr = W + (long long)(n0 + base + t) * ldw;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = threadIdx.x + i * 256;
        if (c < nv) {
          float w[8];
          unpack8(*(const bf16x8_t*)(wr + c * 8), w);
#pragma unroll
          for (int b = 0; b < 8; ++b) {
            if (b < B) {
              const float d = dys[b][t];
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[i][b][j] = fmaf(d, w[j], acc[i][b][j]);
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = threadIdx.x + i * 256;
    if (c < nv) {
#pragma unroll
      for (int b = 0; b < 8; ++b)
        if (b < B) {
#pragma unroll
          for (int j = 0; j < 8; ++j) partial[((long long)blockIdx.x * B + b) * K + c * 8 + j] = acc[i][b][j];
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Distillation loss of one tapped tensor (train/train_qwenvl.py:58-61,613-634), one wave per row of D columns:
//   that = (t - mean t) / (1e-7 + std t),  shat likewise (unbiased std);  q = softmax(that / T), p = softmax(shat / T)
//   row_loss = sum_j p_j (log p_j - log q_j)                      (F.kl_div(q.log(), p, 'batchmean') sums these and divides by B)
//   d row_loss / d z_k = p_k ((log p_k - log q_k) - row_loss),  z = shat / T;  then through the normalisation to s
// grad (bf16) = d(loss_scale * row_loss) / d s;  row_loss[row] is written unscaled for the deterministic second-stage sum.
__global__ __launch_bounds__(256) void kd_loss_kernel(const bf16_t* __restrict__ Tt, long long ldt, const bf16_t* __restrict__ Ss, long long lds,
                                                      bf16_t* __restrict__ grad, long long ldg, float* __restrict__ row_loss, long long rows, int D,
                                                      float inv_temp, float loss_scale) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nv = D >> 3;
  float t[TR_MAXV][8], s[TR_MAXV][8];
  float st = 0.f, ssum = 0.f;
#pragma unroll
  for (int i = 0; i < TR_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      unpack8(*(const bf16x8_t*)(Tt + row * ldt + c * 8), t[i]);
      unpack8(*(const bf16x8_t*)(Ss + row * lds + c * 8), s[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) { st += t[i][j]; ssum += s[i][j]; }
    }
  }
  const float mt = wave_sum(st) / (float)D, ms = wave_sum(ssum) / (float)D;
  float vt = 0.f, vs = 0.f;
#pragma unroll
  for (int i = 0; i < TR_MAXV; ++i)
    if (lane + i * 64 < nv) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        t[i][j] -= mt; s[i][j] -= ms;  // u
        vt = fmaf(t[i][j], t[i][j], vt);
        vs = fmaf(s[i][j], s[i][j], vs);
      }
    }
  const float sdt = sqrtf(wave_sum(vt) / (float)(D - 1)), sds = sqrtf(wave_sum(vs) / (float)(D - 1));
  const float ct = 1.f / (1e-7f + sdt), cs = 1.f / (1e-7f + sds);
  // z = u * c / T for both; log-softmax of each
  float mxt = -3.0e38f, mxs = -3.0e38f;
  const float kt = ct * inv_temp, ks = cs * inv_temp;
#pragma unroll
  for (int i = 0; i < TR_MAXV; ++i)
    if (lane + i * 64 < nv) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        mxt = fmaxf(mxt, t[i][j] * kt);
        mxs = fmaxf(mxs, s[i][j] * ks);
      }
    }
  mxt = wave_max(mxt); mxs = wave_max(mxs);
  float et = 0.f, es = 0.f;
#pragma unroll
  for (int i = 0; i < TR_MAXV; ++i)
    if (lane + i * 64 < nv) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        et += expf(t[i][j] * kt - mxt);
        es += expf(s[i][j] * ks - mxs);
      }
    }
  const float lset = mxt + logf(wave_sum(et)), lses = mxs + logf(wave_sum(es));
  // row loss; keep delta_j = log p_j - log q_j in t[], p_j in a second pass
  float rl = 0.f;
#pragma unroll
  for (int i = 0; i < TR_MAXV; ++i)
    if (lane + i * 64 < nv) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float lp = s[i][j] * ks - lses, lq = t[i][j] * kt - lset;
        const float p = expf(lp);
        t[i][j] = lp - lq;
        rl = fmaf(p, t[i][j], rl);
      }
    }
  rl = wave_sum(rl);
  if (lane == 0) row_loss[row] = rl;
  if (!grad) return;
  // dshat_k = p_k (delta_k - rl) / T ; then ds_i = cs (dshat_i - mean dshat) - cs^2 u_i (sum_j dshat_j u_j) / ((D-1) sd)
  float sd1 = 0.f, sdu = 0.f;
#pragma unroll
  for (int i = 0; i < TR_MAXV; ++i)
    if (lane + i * 64 < nv) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float p = expf(s[i][j] * ks - lses);
        const float dsh = p * (t[i][j] - rl) * inv_temp;
        t[i][j] = dsh;
        sd1 += dsh;
        sdu = fmaf(dsh, s[i][j], sdu);
      }
    }
  const float md = wave_sum(sd1) / (float)D;
  const float kk = sds > 0.f ? cs * cs * wave_sum(sdu) / ((float)(D - 1) * sds) : 0.f;
#pragma unroll
  for (int i = 0; i < TR_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = loss_scale * (cs * (t[i][j] - md) - kk * s[i][j]);
      *(bf16x8_t*)(grad + row * ldg + c * 8) = pack8(o);
    }
  }
}

// zero a gradient tensor when its block's loss term is not finite (the reference skips such terms, :617-620): flag read on the device
__global__ void zero_if_nonfinite_kernel(bf16_t* __restrict__ g, long long n8, const float* __restrict__ term) {
  const float v = term[0];
  if (isfinite(v)) return;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n8) *(bf16x8_t*)(g + i * 8) = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
}

// ---------------------------------------------------------------------------------------------------------------------
// Projector side (the trainable module): weight gradients of the layer fusion, gradient norm, AdamW.
//
// conv5x5 weight gradient: dw[c][ds][dh] = sum_{b,s,h} dy[b][s][h] * x[b][c][s + ds - 2][h + dh - 2]  (utils/proj.py:50,68-69 backward).
// A block takes 16 input rows of one (b, c) plane; a thread owns 8 columns per pass and the 25 running sums; dy rows come from L2.
// partial: [C][B][nchunk][25].
__global__ __launch_bounds__(256) void conv5x5_wgrad_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, float* __restrict__ partial,
                                                            int C, int S, int H, int nchunk) {
  __shared__ float red[4][25];
  const int chunk = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
  const bf16_t* xp = x + ((long long)b * C + c) * S * H;
  const bf16_t* dp = dy + (long long)b * S * H;
  float acc[5][5];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[i][j] = 0.f;
  const int nh = H >> 3;
  for (int rr = 0; rr < 16; ++rr) {
    const int sp = chunk * 16 + rr;   // input row s'
    if (sp >= S) break;
    for (int hc = threadIdx.x; hc < nh; hc += 256) {
      float xv[8];
      unpack8(*(const bf16x8_t*)(xp + (long long)sp * H + hc * 8), xv);
#pragma unroll
      for (int ds = 0; ds < 5; ++ds) {
        const int r = sp - ds + 2;   // output row that sees x row s' through kernel row ds
        if (r < 0 || r >= S) continue;
        // dy[r][hc*8 - 2 .. hc*8 + 9]: the aligned chunk and one neighbour element pair on each side
        float d[12];
        float mid[8];
        unpack8(*(const bf16x8_t*)(dp + (long long)r * H + hc * 8), mid);
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j + 2] = mid[j];
        const long long base = (long long)r * H + hc * 8;
        d[0] = hc > 0 ? bf16_to_f32(dp[base - 2]) : 0.f;
        d[1] = hc > 0 ? bf16_to_f32(dp[base - 1]) : 0.f;
        d[10] = hc + 1 < nh ? bf16_to_f32(dp[base + 8]) : 0.f;
        d[11] = hc + 1 < nh ? bf16_to_f32(dp[base + 9]) : 0.f;
        // x[s'][h'] pairs with dy[r][h' - dh + 2]
#pragma unroll
        for (int dh = 0; dh < 5; ++dh)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[ds][dh] = fmaf(xv[j], d[j + 4 - dh], acc[ds][dh]);
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const float v = wave_sum(acc[i][j]);
      if (lane == 0) red[wave][i * 5 + j] = v;
    }
  __syncthreads();
  if (threadIdx.x < 25)
    partial[(((long long)c * gridDim.z + b) * nchunk + chunk) * 25 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// partial[c][b][chunk] = sum_i dy[b][i] * x[b][c][i] over a chunk of the (S, H) plane  (cha_scale gradient, utils/proj.py:66-67)
__global__ __launch_bounds__(256) void plane_dot_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, float* __restrict__ partial, int C,
                                                        long long plane8, int nchunk, long long per_chunk) {
  __shared__ float red[4];
  const int chunk = blockIdx.x, c = blockIdx.y, b = blockIdx.z;
  const bf16_t* xp = x + ((long long)b * C + c) * plane8 * 8;
  const bf16_t* dp = dy + (long long)b * plane8 * 8;
  float acc = 0.f;
  const long long lo = (long long)chunk * per_chunk, hi = min(plane8, lo + per_chunk);
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    float a[8], d[8];
    unpack8(*(const bf16x8_t*)(xp + i * 8), a);
    unpack8(*(const bf16x8_t*)(dp + i * 8), d);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = fmaf(a[j], d[j], acc);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[((long long)c * gridDim.z + b) * nchunk + chunk] = red[0] + red[1] + red[2] + red[3];
}

// partial[block] = sum x (mode 0) or sum x^2 (mode 1) over a grid-strided slice; x is f32 or bf16
__global__ __launch_bounds__(256) void sum_kernel(const void* __restrict__ xv, int is_bf16, long long n, int mode, float* __restrict__ partial) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float v = is_bf16 ? bf16_to_f32(((const bf16_t*)xv)[i]) : ((const float*)xv)[i];
    acc += mode ? v * v : v;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// torch.nn.utils.clip_grad_norm_ (train/train_qwenvl.py:628): coef = min(1, max_norm / (sqrt(sumsq) + 1e-6)), on the device
__global__ void clip_coef_kernel(const float* __restrict__ sumsq, float max_norm, float* __restrict__ out /* [2]: coef, total norm */) {
  const float nrm = sqrtf(sumsq[0]);
  out[1] = nrm;
  out[0] = fminf(1.f, max_norm / (nrm + 1e-6f));
}

// AdamW (torch.optim.AdamW semantics, decoupled weight decay) on bf16 parameters with f32 gradients and f32 moments:
//   g' = coef * g;  m = b1 m + (1 - b1) g';  v = b2 v + (1 - b2) g'^2;  p = p (1 - lr wd) - lr (m / bc1) / (sqrt(v / bc2) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(bf16_t* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    long long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2,
                                                    const float* __restrict__ coef) {
  const float cf = coef ? coef[0] : 1.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float gg = cf * g[i];
    const float mm = b1 * m[i] + (1.f - b1) * gg;
    const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
    m[i] = mm; v[i] = vv;
    float pw = bf16_to_f32(p[i]);
    pw = pw * (1.f - lr * wd) - lr * (mm / bc1) / (sqrtf(vv / bc2) + eps);
    p[i] = f32_to_bf16(pw);
  }
}

bool al16p(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace

int x2i_launch_conv5x5_wgrad(const void* x, const void* dy, float* partial, int B, int C, int S, int H, hipStream_t stream) {
  if (!x || !dy || !partial || B <= 0 || C <= 0 || S <= 0 || H <= 0 || H % 8) return x2i_set_error(X2I_ERR_ARG, "conv5x5_wgrad: bad argument (H %% 8 == 0)");
  const int nchunk = (S + 15) / 16;
  hipLaunchKernelGGL(conv5x5_wgrad_kernel, dim3(nchunk, C, B), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)dy, partial, C, S, H, nchunk);
  return x2i_check_launch("conv5x5_wgrad");
}

int x2i_launch_plane_dot(const void* x, const void* dy, float* partial, int B, int C, long long plane, int nchunk, hipStream_t stream) {
  if (!x || !dy || !partial || B <= 0 || C <= 0 || plane <= 0 || plane % 8 || nchunk <= 0) return x2i_set_error(X2I_ERR_ARG, "plane_dot: bad argument");
  const long long plane8 = plane / 8, per = (plane8 + nchunk - 1) / nchunk;
  hipLaunchKernelGGL(plane_dot_kernel, dim3(nchunk, C, B), dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)dy, partial, C, plane8, nchunk, per);
  return x2i_check_launch("plane_dot");
}

int x2i_launch_sum(const void* x, int is_bf16, long long n, int mode, float* partial, int nblocks, hipStream_t stream) {
  if (!x || !partial || n <= 0 || nblocks <= 0) return x2i_set_error(X2I_ERR_ARG, "sum: bad argument");
  hipLaunchKernelGGL(sum_kernel, dim3(nblocks), dim3(256), 0, stream, x, is_bf16, n, mode, partial);
  return x2i_check_launch("sum");
}

int x2i_launch_clip_coef(const float* sumsq, float max_norm, float* out, hipStream_t stream) {
  if (!sumsq || !out) return x2i_set_error(X2I_ERR_ARG, "clip_coef: null pointer");
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, stream, sumsq, max_norm, out);
  return x2i_check_launch("clip_coef");
}

int x2i_launch_adamw(void* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, float wd, float bc1,
                     float bc2, const float* coef, hipStream_t stream) {
  if (!p || !g || !m || !v || n <= 0) return x2i_set_error(X2I_ERR_ARG, "adamw: bad argument");
  const long long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, stream, (bf16_t*)p, g, m, v, n, lr, b1, b2, eps, wd,
                     bc1, bc2, coef);
  return x2i_check_launch("adamw");
}

int x2i_launch_transpose(const void* in, long long in_bs, long long ld_in, void* out, long long out_bs, long long ld_out, int batch, int R, int C,
                         hipStream_t stream) {
  if (!in || !out || batch <= 0 || R <= 0 || C <= 0) return x2i_set_error(X2I_ERR_ARG, "transpose: bad argument");
  if (R % 8 || C % 8 || ld_in % 8 || ld_out % 8 || in_bs % 8 || out_bs % 8 || !al16p(in) || !al16p(out))
    return x2i_set_error(X2I_ERR_ALIGN, "transpose: R, C, leading dims and batch strides must be multiples of 8, pointers 16-byte aligned");
  if (batch > 65535 || (R + 63) / 64 > 65535) return x2i_set_error(X2I_ERR_SHAPE, "transpose: batch / rows beyond the grid limits");
  hipLaunchKernelGGL(transpose_kernel, dim3((C + 63) / 64, (R + 63) / 64, batch), dim3(256), 0, stream, (const bf16_t*)in, in_bs, ld_in,
                     (bf16_t*)out, out_bs, ld_out, R, C);
  return x2i_check_launch("transpose");
}

int x2i_launch_softmax_pad(void* x, long long ld, int nz, int Rt, int Rv, int Ct, int Cv, float scale, hipStream_t stream) {
  if (!x || nz <= 0 || Rt <= 0 || Ct <= 0 || Rv < 0 || Rv > Rt || Cv <= 0 || Cv > Ct) return x2i_set_error(X2I_ERR_ARG, "softmax_pad: bad argument");
  if (Ct % 8 || ld % 8 || !al16p(x)) return x2i_set_error(X2I_ERR_ALIGN, "softmax_pad: Ct and ld must be multiples of 8");
  const long long rows = (long long)nz * Rt;
  hipLaunchKernelGGL(softmax_pad_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, (bf16_t*)x, ld, Rt, Rv, Ct, Cv,
                     scale * 1.4426950408889634f, rows);
  return x2i_check_launch("softmax_pad");
}

int x2i_launch_softmax_bwd(const void* P, void* dP, long long ld, int nz, int Rt, int Rv, int Ct, int Cv, float scale, hipStream_t stream) {
  if (!P || !dP || nz <= 0 || Rt <= 0 || Ct <= 0 || Rv < 0 || Rv > Rt || Cv <= 0 || Cv > Ct) return x2i_set_error(X2I_ERR_ARG, "softmax_bwd: bad argument");
  if (Ct % 8 || ld % 8 || !al16p(P) || !al16p(dP)) return x2i_set_error(X2I_ERR_ALIGN, "softmax_bwd: Ct and ld must be multiples of 8");
  const long long rows = (long long)nz * Rt;
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, (const bf16_t*)P, (bf16_t*)dP, ld, Rt, Rv, Ct, Cv,
                     scale, rows);
  return x2i_check_launch("softmax_bwd");
}

int x2i_launch_ln_mod_bwd(const void* X, long long x_bs, int ldx, const void* dY, long long dy_bs, int ldy, const float* m, long long m_bs,
                          int mult_is_scale, const void* dXin, void* dXout, long long dx_bs, int lddx, int B, int S, int D, int R,
                          float* partial, float eps, hipStream_t stream) {
  if (!X || !dY || !m || !dXout || !partial || B <= 0 || S <= 0 || D <= 0 || R <= 0) return x2i_set_error(X2I_ERR_ARG, "ln_mod_bwd: bad argument");
  if (D % 8 || D > 64 * 8 * TR_MAXV || ldx % 8 || ldy % 8 || lddx % 8 || x_bs % 8 || dy_bs % 8 || dx_bs % 8 || !al16p(X) || !al16p(dY) ||
      !al16p(dXout) || (dXin && !al16p(dXin)))
    return x2i_set_error(X2I_ERR_ALIGN, "ln_mod_bwd: D %% 8 == 0, D <= 4096, 16-byte aligned rows");
  const int nw = (S + R - 1) / R;
  const long long waves = (long long)B * nw;
  if (x2i_options().train_rows_wg && D / 8 <= 512 && (((uintptr_t)partial) & 15) == 0 && waves < 0x7fffffffLL) {   // a workgroup per row group (see ln_mod_bwd_rows_kernel)
    const int nt = ((D / 8 + 63) / 64) * 64;
    hipLaunchKernelGGL(ln_mod_bwd_rows_kernel, dim3((unsigned)waves), dim3(nt), 0, stream, (const bf16_t*)X, x_bs, ldx, (const bf16_t*)dY, dy_bs, ldy, m,
                       m_bs, mult_is_scale, (const bf16_t*)dXin, (bf16_t*)dXout, dx_bs, lddx, S, D, R, partial, eps, nw);
    return x2i_check_launch("ln_mod_bwd");
  }
  hipLaunchKernelGGL(ln_mod_bwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, (const bf16_t*)X, x_bs, ldx, (const bf16_t*)dY,
                     dy_bs, ldy, m, m_bs, mult_is_scale, (const bf16_t*)dXin, (bf16_t*)dXout, dx_bs, lddx, S, D, R, partial, eps, nw, B);
  return x2i_check_launch("ln_mod_bwd");
}

int x2i_launch_gate_bwd(const void* dX, long long dx_bs, int lddx, const void* T, long long t_bs, int ldt, const float* gate, long long g_bs,
                        const void* G, long long gg_bs, int ldg, void* dT, long long dt_bs, int lddt, int B, int S, int D, int R,
                        float* partial, hipStream_t stream) {
  if (!dX || !dT || B <= 0 || S <= 0 || D <= 0 || R <= 0 || (gate && (!T || !partial))) return x2i_set_error(X2I_ERR_ARG, "gate_bwd: bad argument");
  if (D % 8 || D > 64 * 8 * TR_MAXV || lddx % 8 || lddt % 8 || (T && ldt % 8) || (G && ldg % 8) || !al16p(dX) || !al16p(dT) || (T && !al16p(T)) ||
      (G && !al16p(G)) || dx_bs % 8 || dt_bs % 8 || t_bs % 8 || gg_bs % 8)
    return x2i_set_error(X2I_ERR_ALIGN, "gate_bwd: D %% 8 == 0, D <= 4096, 16-byte aligned rows");
  const int nw = (S + R - 1) / R;
  const long long waves = (long long)B * nw;
  if (x2i_options().train_rows_wg && D / 8 <= 512 && (!partial || (((uintptr_t)partial) & 15) == 0) && waves < 0x7fffffffLL) {
    const int nt = ((D / 8 + 63) / 64) * 64;
    hipLaunchKernelGGL(gate_bwd_rows_kernel, dim3((unsigned)waves), dim3(nt), 0, stream, (const bf16_t*)dX, dx_bs, lddx, (const bf16_t*)T, t_bs, ldt, gate,
                       g_bs, (const bf16_t*)G, gg_bs, ldg, (bf16_t*)dT, dt_bs, lddt, S, D, R, partial, nw);
    return x2i_check_launch("gate_bwd");
  }
  hipLaunchKernelGGL(gate_bwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, (const bf16_t*)dX, dx_bs, lddx, (const bf16_t*)T,
                     t_bs, ldt, gate, g_bs, (const bf16_t*)G, gg_bs, ldg, (bf16_t*)dT, dt_bs, lddt, S, D, R, partial, nw, B);
  return x2i_check_launch("gate_bwd");
}

int x2i_launch_reduce_rows(const float* in, long long in_zs, int np, long long in_ps, float* out, long long out_zs, int nz, int len,
                           int accumulate, float alpha, hipStream_t stream) {
  if (!in || !out || np <= 0 || nz <= 0 || len <= 0) return x2i_set_error(X2I_ERR_ARG, "reduce_rows: bad argument");
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((len + 63) / 64, nz), dim3(1024), 0, stream, in, in_zs, np, in_ps, out, out_zs, len, accumulate, alpha);
  return x2i_check_launch("reduce_rows");
}

int x2i_launch_act_bwd(void* dA, long long ldd, const void* pre, long long ldp, long long rows, int cols, int act, int is_f32,
                       hipStream_t stream) {
  if (!dA || !pre || rows <= 0 || cols <= 0) return x2i_set_error(X2I_ERR_ARG, "act_bwd: bad argument");
  if (act != X2I_ACT_GELU_TANH && act != X2I_ACT_GELU_ERF && act != X2I_ACT_SILU) return x2i_set_error(X2I_ERR_ARG, "act_bwd: unknown activation %d", act);
  if (is_f32) {
    if (ldd != cols || ldp != cols) return x2i_set_error(X2I_ERR_ARG, "act_bwd: f32 form is contiguous only");
    const long long n = rows * cols;
    hipLaunchKernelGGL(act_bwd_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (float*)dA, (const float*)pre, n, act);
    return x2i_check_launch("act_bwd");
  }
  if (cols % 8 || ldd % 8 || ldp % 8 || !al16p(dA) || !al16p(pre)) return x2i_set_error(X2I_ERR_ALIGN, "act_bwd: cols and leading dims %% 8 == 0");
  const long long total = rows * (cols / 8), blocks = (total + 255) / 256;
  hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, stream, (bf16_t*)dA, ldd, (const bf16_t*)pre,
                     ldp, rows, cols / 8, act);
  return x2i_check_launch("act_bwd");
}

int x2i_launch_qkv_split_bwd(const void* qkv0, const void* qkv1, int ld0, int ld1, void* d0, void* d1, int ldd0, int ldd1, int B, int S, int S0,
                             int H, const void* nq0, const void* nk0, const void* nq1, const void* nk1, const float* cosp, const float* sinp,
                             const void* dQ, const void* dK, const void* dV, int Spad, float eps, hipStream_t stream) {
  if (!qkv1 || !d1 || !nq1 || !nk1 || !cosp || !sinp || !dQ || !dK || !dV || (S0 > 0 && (!qkv0 || !d0 || !nq0 || !nk0)))
    return x2i_set_error(X2I_ERR_ARG, "qkv_split_bwd: null pointer");
  if (B <= 0 || S <= 0 || S0 < 0 || S0 > S || H <= 0 || Spad < S || ld1 % 8 || ldd1 % 8 || (S0 > 0 && (ld0 % 8 || ldd0 % 8)))
    return x2i_set_error(X2I_ERR_SHAPE, "qkv_split_bwd: bad shape");
  hipLaunchKernelGGL(qkv_split_bwd_kernel, dim3(S, B), dim3(256), 0, stream, (const bf16_t*)qkv0, (const bf16_t*)qkv1, ld0, ld1, (bf16_t*)d0,
                     (bf16_t*)d1, ldd0, ldd1, S, S0, H, (const bf16_t*)nq0, (const bf16_t*)nk0, (const bf16_t*)nq1, (const bf16_t*)nk1, cosp, sinp,
                     (const bf16_t*)dQ, (const bf16_t*)dK, (const bf16_t*)dV, Spad, eps);
  return x2i_check_launch("qkv_split_bwd");
}

int x2i_launch_skinny_bwd(const float* dy, long long dy_bs, const void* W, int ldw, float* partial, int B, int N, int K, int chunk,
                          hipStream_t stream) {
  if (!dy || !W || !partial || B <= 0 || B > 8 || N <= 0 || K <= 0 || chunk <= 0) return x2i_set_error(X2I_ERR_ARG, "skinny_bwd: bad argument (B <= 8)");
  if (K % 8 || K > 4096 || ldw % 8 || !al16p(W)) return x2i_set_error(X2I_ERR_ALIGN, "skinny_bwd: K %% 8 == 0, K <= 4096");
  hipLaunchKernelGGL(skinny_bwd_kernel, dim3((N + chunk - 1) / chunk), dim3(256), 0, stream, dy, dy_bs, (const bf16_t*)W, ldw, partial, B, N, K, chunk);
  return x2i_check_launch("skinny_bwd");
}

int x2i_launch_kd_loss(const void* teacher, long long ldt, const void* student, long long lds, void* grad, long long ldg, float* row_loss,
                       long long rows, int D, float temperature, float loss_scale, hipStream_t stream) {
  if (!teacher || !student || !row_loss || rows <= 0 || D <= 1 || temperature <= 0.f) return x2i_set_error(X2I_ERR_ARG, "kd_loss: bad argument");
  if (D % 8 || D > 64 * 8 * TR_MAXV || ldt % 8 || lds % 8 || (grad && ldg % 8) || !al16p(teacher) || !al16p(student) || (grad && !al16p(grad)))
    return x2i_set_error(X2I_ERR_ALIGN, "kd_loss: D %% 8 == 0, D <= 4096, 16-byte aligned rows");
  hipLaunchKernelGGL(kd_loss_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, (const bf16_t*)teacher, ldt, (const bf16_t*)student, lds,
                     (bf16_t*)grad, ldg, row_loss, rows, D, 1.f / temperature, loss_scale);
  return x2i_check_launch("kd_loss");
}

int x2i_launch_zero_if_nonfinite(void* g, long long n, const float* term, hipStream_t stream) {
  if (!g || !term || n <= 0 || n % 8 || !al16p(g)) return x2i_set_error(X2I_ERR_ARG, "zero_if_nonfinite: bad argument");
  const long long n8 = n / 8;
  hipLaunchKernelGGL(zero_if_nonfinite_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, stream, (bf16_t*)g, n8, term);
  return x2i_check_launch("zero_if_nonfinite");
}
