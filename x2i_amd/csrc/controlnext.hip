// ControlNeXt hint-encoder kernels that are not GEMM-shaped (the 3x3 / 2x2 / 1x1 convolutions with Cin >= 64 run as
// implicit GEMMs in gemm.hip):
//   conv_stem_kernel   Conv2d(3 -> 64, k3, s2, p1) on the NHWC hint image     (lightcontrol_flux.py:594)
//   gn_partial_kernel  per-(sample, group) partial sums over a slab of pixels   } nn.GroupNorm on NHWC bf16 with fused
//   gn_apply_kernel    finish statistics, normalise, affine, activation, adds   } pre-add / activation / residual
// All HBM-bound: 16-byte loads, fp32 statistics, deterministic two-stage reduction (no atomics).
#include "x2i_common.h"
#include "x2i_kernels.h"

namespace {

constexpr int GN_SLABS = 256;  // partial-sum slabs per sample

// one thread = one output pixel x 16 output channels
__global__ __launch_bounds__(256) void conv_stem_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, bf16_t* __restrict__ y, int H, int W,
                                                        int Cout, long long total) {
  __shared__ float wsh[64 * 27];
  for (int i = threadIdx.x; i < Cout * 27; i += 256) wsh[i] = w[i];
  __syncthreads();
  const int groups = Cout / 16;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= total) return;
  const int cg = (int)(gid % groups);
  const long long pix = gid / groups;
  const int OH = H / 2, OW = W / 2;
  const int ox = (int)(pix % OW);
  const int oy = (int)((pix / OW) % OH);
  const int b = (int)(pix / ((long long)OW * OH));
  float in[27];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = oy * 2 + ky - 1, ix = ox * 2 + kx - 1;
      const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
      const bf16_t* px = x + (((long long)b * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) in[(ky * 3 + kx) * 3 + c] = ok ? bf16_to_f32(px[c]) : 0.f;
    }
  float o[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const float* wc = wsh + (cg * 16 + c) * 27;
    float a = bias ? bias[cg * 16 + c] : 0.f;
#pragma unroll
    for (int k = 0; k < 27; ++k) a += wc[k] * in[k];
    o[c] = a;
  }
  bf16_t* yp = y + pix * Cout + cg * 16;
  union { bf16x8_t v; uint32_t u[4]; } lo, hi;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    lo.u[j] = pack_bf16x2(o[2 * j], o[2 * j + 1]);
    hi.u[j] = pack_bf16x2(o[8 + 2 * j], o[8 + 2 * j + 1]);
  }
  *(bf16x8_t*)yp = lo.v;
  *(bf16x8_t*)(yp + 8) = hi.v;
}

// grid (GN_SLABS, B); each thread owns one 8-channel chunk position (tid % (C/8)) and strides over pixels, four 16-byte
// loads in flight.  A chunk lies in one group (channels per group a multiple of 8) or spans exactly two (channels per group
// == 4, the VAE's GroupNorm(32, 128)): two partial-sum pairs per thread cover both cases.
// scratch layout: stats [B][G][2] (mean, rstd) followed by partial sums [B][GN_SLABS][G][2]
__global__ __launch_bounds__(256) void gn_partial_kernel(const bf16_t* __restrict__ x, const float* __restrict__ pre_add,
                                                         float* __restrict__ partial, long long HW, int C, int G) {
  __shared__ float red[256][4];
  const int b = blockIdx.y, slab = blockIdx.x;
  const int cpp = C / 8;                   // chunks per pixel
  const int ppi = 256 / cpp;               // pixels per block iteration
  const int chunk = threadIdx.x % cpp, psub = threadIdx.x / cpp;
  const long long per = (HW + GN_SLABS - 1) / GN_SLABS;
  const long long p0 = (long long)slab * per, p1 = min(HW, p0 + per);
  float add[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (pre_add) {
#pragma unroll
    for (int j = 0; j < 8; ++j) add[j] = pre_add[(long long)b * C + chunk * 8 + j];
  }
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;  // first / second half of the chunk
  const bf16_t* xb = x + (long long)b * HW * C + chunk * 8;
  for (long long pbase = p0 + psub; pbase < p1; pbase += 4 * ppi) {
    bf16x8_t v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long pp = pbase + (long long)u * ppi;
      v[u] = (pp < p1) ? *(const bf16x8_t*)(xb + pp * C) : (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (pbase + (long long)u * ppi < p1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float f = bf16_to_f32((bf16_t)v[u][j]) + add[j];
          s0 += f;
          q0 = fmaf(f, f, q0);
          const float g = bf16_to_f32((bf16_t)v[u][j + 4]) + add[j + 4];
          s1 += g;
          q1 = fmaf(g, g, q1);
        }
      }
    }
  }
  // block reduction in three short steps (the first form had G threads walk all 256 entries with two integer divisions each: ~10 us per
  // block, which made this HBM-bound pass run at 1.6 TB/s -- profiles/r04i_bench_config5_kernel_stats.csv): (1) the threads that hold the
  // same chunk are cpp apart: thread c < cpp sums them (ppi <= 32 entries); (2) chunk-half sums -> LDS; (3) thread g < G sums the chunk
  // halves of its group (cpg / 4 consecutive halves; a 4-channel group is exactly one half).  Fixed order: deterministic.
  red[threadIdx.x][0] = s0;
  red[threadIdx.x][1] = q0;
  red[threadIdx.x][2] = s1;
  red[threadIdx.x][3] = q1;
  __syncthreads();
  __shared__ float half[512][2];  // [chunk * 2 + half][sum, sum of squares]; cpp <= 256
  if ((int)threadIdx.x < cpp) {
    float a0 = 0.f, b0 = 0.f, a1 = 0.f, b1 = 0.f;
    for (int t = threadIdx.x; t < 256; t += cpp) {
      a0 += red[t][0]; b0 += red[t][1]; a1 += red[t][2]; b1 += red[t][3];
    }
    half[threadIdx.x * 2][0] = a0; half[threadIdx.x * 2][1] = b0;
    half[threadIdx.x * 2 + 1][0] = a1; half[threadIdx.x * 2 + 1][1] = b1;
  }
  __syncthreads();
  if ((int)threadIdx.x < G) {
    const int hpg = (C / G) / 4;  // chunk halves per group (channels per group: 4 or a multiple of 8)
    float a = 0.f, bq = 0.f;
    for (int h = 0; h < hpg; ++h) {
      a += half[threadIdx.x * hpg + h][0];
      bq += half[threadIdx.x * hpg + h][1];
    }
    float* o = partial + (long long)gridDim.y * G * 2 + (((long long)b * GN_SLABS + slab) * G + threadIdx.x) * 2;
    o[0] = a;
    o[1] = bq;
  }
}

// grid B, 256 threads: thread (part = t / 32, g = t % 32) sums every 8th slab; parts are combined in a fixed order
__global__ __launch_bounds__(256) void gn_finish_kernel(float* __restrict__ partial, long long HW, int C, int G, float eps) {
  __shared__ float red[8][32][2];
  const int b = blockIdx.x, g = threadIdx.x & 31, part = threadIdx.x >> 5;
  float a = 0.f, q = 0.f;
  if (g < G) {
    const float* pp = partial + (long long)gridDim.x * G * 2 + ((long long)b * GN_SLABS * G + g) * 2;
    // eight slabs' loads in flight (as one dependent chain of 32 the kernel took 9.7 us, 80 calls per LightControl step)
    float a4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s0 = 0; s0 < GN_SLABS; s0 += 32) {
      float2 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *(const float2*)(pp + (long long)(s0 + part + 8 * u) * G * 2);
#pragma unroll
      for (int u = 0; u < 4; ++u) { a4[u] += v[u].x; q4[u] += v[u].y; }
    }
    a = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    q = (q4[0] + q4[1]) + (q4[2] + q4[3]);
  }
  red[part][g][0] = a;
  red[part][g][1] = q;
  __syncthreads();
  if (threadIdx.x < G) {
    a = 0.f, q = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      a += red[k][g][0];
      q += red[k][g][1];
    }
    const float n = (float)HW * (float)(C / G);
    const float m = a / n;
    const float var = fmaxf(q / n - m * m, 0.f);
    partial[((long long)b * G + g) * 2] = m;
    partial[((long long)b * G + g) * 2 + 1] = rsqrtf(var + eps);
  }
}

// Per-channel moments of an NHWC tensor: mom[b][c] = (sum_p x, sum_p x^2).  GroupNorm statistics of x + v[b][c] for ANY per-channel
// vector v follow from them without touching x again (gn_finish_moments_kernel): ResnetBlock2D's norm2 runs on conv1(...) + the time
// embedding term (lightcontrol_flux.py:620-640, diffusers ResnetBlock2D), and in the ControlNeXt branch conv1's output does not depend on
// the timestep -- its moments are taken once per hint, and the statistics pass over the 64 MB-per-image tensor leaves the denoising loop.
// grid (GN_SLABS, B): partial moments per slab, then gn_moments_finish_kernel (grid B) sums the slabs in a fixed order.
__global__ __launch_bounds__(256) void gn_moments_partial_kernel(const bf16_t* __restrict__ x, float* __restrict__ part, long long HW, int C) {
  __shared__ float red[256][17];
  const int b = blockIdx.y, slab = blockIdx.x;
  const int cpp = C / 8, ppi = 256 / cpp;
  const int chunk = threadIdx.x % cpp, psub = threadIdx.x / cpp;
  const long long per = (HW + GN_SLABS - 1) / GN_SLABS;
  const long long p0 = (long long)slab * per, p1 = min(HW, p0 + per);
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
  const bf16_t* xb = x + (long long)b * HW * C + chunk * 8;
  for (long long pp = p0 + psub; pp < p1; pp += ppi) {
    const bf16x8_t v = *(const bf16x8_t*)(xb + pp * C);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float f = bf16_to_f32((bf16_t)v[j]);
      s[j] += f;
      q[j] = fmaf(f, f, q[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = s[j], red[threadIdx.x][8 + j] = q[j];
  __syncthreads();
  // thread t < C: channel t = chunk (t / 8), element t % 8: sum over the ppi threads that hold this chunk
  for (int c = threadIdx.x; c < C; c += 256) {
    const int ch = c >> 3, j = c & 7;
    float a = 0.f, bq = 0.f;
    for (int t = ch; t < 256; t += cpp) a += red[t][j], bq += red[t][8 + j];
    float* o = part + (((long long)b * GN_SLABS + slab) * C + c) * 2;
    o[0] = a;
    o[1] = bq;
  }
}
__global__ __launch_bounds__(256) void gn_moments_finish_kernel(const float* __restrict__ part, float* __restrict__ mom, int C) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, q = 0.f;
    for (int sl = 0; sl < GN_SLABS; ++sl) {
      const float* pp = part + (((long long)b * GN_SLABS + sl) * C + c) * 2;
      a += pp[0];
      q += pp[1];
    }
    mom[((long long)b * C + c) * 2] = a;
    mom[((long long)b * C + c) * 2 + 1] = q;
  }
}
// stats[b][g] = (mean, rstd) of x + v over group g from the channel moments: sum (x + v) = S1 + HW v, sum (x + v)^2 = S2 + 2 v S1 + HW v^2
__global__ __launch_bounds__(256) void gn_finish_moments_kernel(const float* __restrict__ mom, const float* __restrict__ pre_add, float* __restrict__ stats,
                                                                long long HW, int C, int G, float eps) {
  __shared__ float cs[2048][2];
  const int b = blockIdx.x;
  const float n1 = (float)HW;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float s1 = mom[((long long)b * C + c) * 2], s2 = mom[((long long)b * C + c) * 2 + 1];
    const float v = pre_add ? pre_add[(long long)b * C + c] : 0.f;
    cs[c][0] = fmaf(n1, v, s1);
    cs[c][1] = fmaf(n1 * v, v, fmaf(2.f * v, s1, s2));
  }
  __syncthreads();
  if ((int)threadIdx.x < G) {
    const int cpg = C / G;
    float a = 0.f, q = 0.f;
    for (int c = threadIdx.x * cpg; c < ((int)threadIdx.x + 1) * cpg; ++c) a += cs[c][0], q += cs[c][1];
    const float n = n1 * (float)cpg;
    const float m = a / n;
    const float var = fmaxf(q / n - m * m, 0.f);
    stats[((long long)b * G + threadIdx.x) * 2] = m;
    stats[((long long)b * G + threadIdx.x) * 2 + 1] = rsqrtf(var + eps);
  }
}

// every thread keeps its channel chunk for the whole grid-stride loop (256 % (C/8) == 0), so the affine parameters, the
// pre-add vector and the group statistics live in registers; two independent 16-byte loads in flight per thread
__global__ __launch_bounds__(256) void gn_apply_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long long HW, int C, int G,
                                                       const bf16_t* __restrict__ w, const bf16_t* __restrict__ bias,
                                                       int act, const float* __restrict__ pre_add,
                                                       const bf16_t* __restrict__ post_add, const float* __restrict__ partial, int wdiv) {
  const int b = blockIdx.y;
  w += (long long)(b / wdiv) * C;      // grouped affine parameters (x2i_groupnorm_*_grouped_bf16): wdiv consecutive items share one [C] pair
  bias += (long long)(b / wdiv) * C;
  const int cpp = C / 8, cpg = C / G;
  const long long total = HW * cpp;
  const int chunk = threadIdx.x % cpp, c0 = chunk * 8;
  const float* st = partial + (long long)b * G * 2;
  const int g0 = c0 / cpg, g1 = (c0 + 4) / cpg;
  float sc[8], sh[8];  // y = act(x * sc + sh)
  {
    const bf16x8_t wv = *(const bf16x8_t*)(w + c0), bv = *(const bf16x8_t*)(bias + c0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float m = st[(j < 4 ? g0 : g1) * 2], r = st[(j < 4 ? g0 : g1) * 2 + 1];
      const float pa = pre_add ? pre_add[(long long)b * C + c0 + j] : 0.f;
      sc[j] = r * bf16_to_f32((bf16_t)wv[j]);
      sh[j] = fmaf(pa - m, sc[j], bf16_to_f32((bf16_t)bv[j]));
    }
  }
  const bf16_t* xb = x + (long long)b * HW * C;
  bf16_t* yb = y + (long long)b * HW * C;
  const bf16_t* ab = post_add ? post_add + (long long)b * HW * C : nullptr;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += 2 * stride) {
    const long long i1 = i + stride;
    const bool has1 = i1 < total;
    const bf16x8_t v0 = *(const bf16x8_t*)(xb + i * 8);
    const bf16x8_t v1 = has1 ? *(const bf16x8_t*)(xb + i1 * 8) : v0;
    bf16x8_t a0 = {0, 0, 0, 0, 0, 0, 0, 0}, a1 = a0;
    if (ab) {
      a0 = *(const bf16x8_t*)(ab + i * 8);
      if (has1) a1 = *(const bf16x8_t*)(ab + i1 * 8);
    }
    union { bf16x8_t v8; uint32_t u[4]; } r0, r1;
    float o0[8], o1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o0[j] = apply_act(fmaf(bf16_to_f32((bf16_t)v0[j]), sc[j], sh[j]), act);
      o1[j] = apply_act(fmaf(bf16_to_f32((bf16_t)v1[j]), sc[j], sh[j]), act);
      if (ab) {
        o0[j] += bf16_to_f32((bf16_t)a0[j]);
        o1[j] += bf16_to_f32((bf16_t)a1[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r0.u[j] = pack_bf16x2(o0[2 * j], o0[2 * j + 1]);
      r1.u[j] = pack_bf16x2(o1[2 * j], o1[2 * j + 1]);
    }
    *(bf16x8_t*)(yb + i * 8) = r0.v8;
    if (has1) *(bf16x8_t*)(yb + i1 * 8) = r1.v8;
  }
}

// In-place row softmax over bf16 rows (<= 16384 columns live in registers: 256 threads x 8 chunks x 8 values).
constexpr int SM_MAXC = 8;
__global__ __launch_bounds__(256) void softmax_rows_kernel(bf16_t* __restrict__ x, int cols, float scale_log2) {
  __shared__ float red[4];
  bf16_t* row = x + (long long)blockIdx.x * cols;
  const int nchunk = cols >> 3;
  float v[SM_MAXC][8];
  float mx = -1.0e30f;
#pragma unroll
  for (int c = 0; c < SM_MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
    if (ch < nchunk) {
      const bf16x8_t t = *(const bf16x8_t*)(row + ch * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[c][j] = bf16_to_f32((bf16_t)t[j]) * scale_log2;
        mx = fmaxf(mx, v[c][j]);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < SM_MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
    if (ch < nchunk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[c][j] = __builtin_amdgcn_exp2f(v[c][j] - mx);
        sum += v[c][j];
      }
    }
  }
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
#pragma unroll
  for (int c = 0; c < SM_MAXC; ++c) {
    const int ch = threadIdx.x + c * 256;
    if (ch < nchunk) {
      union { bf16x8_t v8; uint32_t u[4]; } r;
#pragma unroll
      for (int j = 0; j < 4; ++j) r.u[j] = pack_bf16x2(v[c][2 * j] * inv, v[c][2 * j + 1] * inv);
      *(bf16x8_t*)(row + ch * 8) = r.v8;
    }
  }
}

}  // namespace

int x2i_launch_conv_stem(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int Cout, hipStream_t stream) {
  if (!x || !w || !y) return x2i_set_error(X2I_ERR_ARG, "conv_stem: null pointer");
  if (B <= 0 || H <= 0 || W <= 0 || H % 2 || W % 2 || Cout % 16 || Cout > 64 || Cout <= 0)
    return x2i_set_error(X2I_ERR_SHAPE, "conv_stem: need even H,W and Cout in {16,32,48,64}");
  const long long total = (long long)B * (H / 2) * (W / 2) * (Cout / 16);
  hipLaunchKernelGGL(conv_stem_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const bf16_t*)x, w, bias,
                     (bf16_t*)y, H, W, Cout, total);
  return x2i_check_launch("conv_stem");
}

long long x2i_groupnorm_scratch(int B, int G) { return (long long)B * G * 2 * (GN_SLABS + 1); }

int x2i_launch_groupnorm(const void* x, void* y, int B, long long HW, int C, int G, const void* w, const void* b, float eps, int act,
                         const float* pre_add, const void* post_add, float* partial, hipStream_t stream, int w_group) {
  if (w_group < 0) return x2i_set_error(X2I_ERR_ARG, "groupnorm: w_group < 0");
  if (!x || !y || !w || !b || !partial) return x2i_set_error(X2I_ERR_ARG, "groupnorm: null pointer");
  if (B <= 0 || HW <= 0 || C % 8 || 256 % (C / 8) || G <= 0 || G > 32 || C % G || !((C / G) % 8 == 0 || (C / G) == 4))
    return x2i_set_error(X2I_ERR_SHAPE, "groupnorm: unsupported C=%d G=%d (need C/8 | 256 and C/G == 4 or a multiple of 8)", C, G);
  hipLaunchKernelGGL(gn_partial_kernel, dim3(GN_SLABS, B), dim3(256), 0, stream, (const bf16_t*)x, pre_add, partial, HW, C, G);
  int rc = x2i_check_launch("groupnorm_partial");
  if (rc) return rc;
  hipLaunchKernelGGL(gn_finish_kernel, dim3(B), dim3(256), 0, stream, partial, HW, C, G, eps);
  rc = x2i_check_launch("groupnorm_finish");
  if (rc) return rc;
  const long long total = HW * (C / 8);
  long long blocks = (total + 511) / 512;  // two chunks per thread per iteration
  const long long cap = 8192 / B > 256 ? 8192 / B : 256;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)blocks, B), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, HW, C, G,
                     (const bf16_t*)w, (const bf16_t*)b, act, pre_add, (const bf16_t*)post_add, partial, w_group > 0 ? w_group : B);
  return x2i_check_launch("groupnorm_apply");
}

long long x2i_groupnorm_moments_scratch(int B, int C) { return (long long)B * GN_SLABS * C * 2; }

int x2i_launch_groupnorm_moments(const void* x, int B, long long HW, int C, float* moments, float* scratch, hipStream_t stream) {
  if (!x || !moments || !scratch) return x2i_set_error(X2I_ERR_ARG, "groupnorm_moments: null pointer");
  if (B <= 0 || HW <= 0 || C % 8 || 256 % (C / 8) || C > 2048) return x2i_set_error(X2I_ERR_SHAPE, "groupnorm_moments: unsupported C=%d (need C/8 | 256)", C);
  hipLaunchKernelGGL(gn_moments_partial_kernel, dim3(GN_SLABS, B), dim3(256), 0, stream, (const bf16_t*)x, scratch, HW, C);
  int rc = x2i_check_launch("groupnorm_moments_partial");
  if (rc) return rc;
  hipLaunchKernelGGL(gn_moments_finish_kernel, dim3(B), dim3(256), 0, stream, scratch, moments, C);
  return x2i_check_launch("groupnorm_moments_finish");
}

int x2i_launch_groupnorm_from_moments(const void* x, void* y, int B, long long HW, int C, int G, const void* w, const void* b, float eps, int act,
                                      const float* moments, const float* pre_add, const void* post_add, float* partial, hipStream_t stream, int w_group) {
  if (w_group < 0) return x2i_set_error(X2I_ERR_ARG, "groupnorm_from_moments: w_group < 0");
  if (!x || !y || !w || !b || !partial || !moments) return x2i_set_error(X2I_ERR_ARG, "groupnorm_from_moments: null pointer");
  if (B <= 0 || HW <= 0 || C % 8 || 256 % (C / 8) || C > 2048 || G <= 0 || G > 32 || C % G || !((C / G) % 8 == 0 || (C / G) == 4))
    return x2i_set_error(X2I_ERR_SHAPE, "groupnorm_from_moments: unsupported C=%d G=%d", C, G);
  hipLaunchKernelGGL(gn_finish_moments_kernel, dim3(B), dim3(256), 0, stream, moments, pre_add, partial, HW, C, G, eps);
  int rc = x2i_check_launch("groupnorm_finish_moments");
  if (rc) return rc;
  const long long total = HW * (C / 8);
  long long blocks = (total + 511) / 512;
  const long long cap = 8192 / B > 256 ? 8192 / B : 256;
  if (blocks > cap) blocks = cap;
  hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)blocks, B), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, HW, C, G,
                     (const bf16_t*)w, (const bf16_t*)b, act, pre_add, (const bf16_t*)post_add, partial, w_group > 0 ? w_group : B);
  return x2i_check_launch("groupnorm_apply");
}

int x2i_launch_softmax_rows(void* x, long long rows, int cols, float scale, hipStream_t stream) {
  if (!x || rows <= 0 || cols <= 0) return x2i_set_error(X2I_ERR_ARG, "softmax_rows: bad argument");
  if (cols % 8 || cols > 256 * 8 * SM_MAXC) return x2i_set_error(X2I_ERR_SHAPE, "softmax_rows: cols=%d must be a multiple of 8 and <= %d", cols, 256 * 8 * SM_MAXC);
  if (rows > 0x7fffffffLL) return x2i_set_error(X2I_ERR_SHAPE, "softmax_rows: too many rows");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, stream, (bf16_t*)x, cols, scale * 1.4426950408889634f);
  return x2i_check_launch("softmax_rows");
}
