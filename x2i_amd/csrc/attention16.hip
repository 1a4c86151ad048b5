// Flash attention forward on v_mfma_f32_16x16x32_bf16 -- an A/B kernel (option attn_variant = 10), compiler-scheduled, built in round 5 to
// answer ONE question at kernel level: the matrix pipe alone sustains 11-14 % more FLOP/s at the power cap with the 16 x 16 x 32 shape than with
// the 32 x 32 x 16 shape every other attention kernel here uses (tools/ubench/mfma_power.hip, DESIGN.md section 4 "What the matrix pipe alone
// can do") -- does an attention kernel keep that advantage?  Same organisation as attention.hip's 4-wave kernel (the comparison partner:
// attn_variant = 4): NW waves x 32 query rows, 64-key tiles through the same double-buffered LDS images (K [64][128], V^T [128][64], same
// swizzles, same LDS-DMA staging), fp32 statistics, defer-max, same output layout; only the MFMA shape and what follows from it differ:
//   * S^T = K Q^T in 16 x 16 blocks: lane (c = lane & 15, g = lane >> 4) holds query c of a query block and keys 4g .. 4g+3 of each of the
//     tile's four key blocks;
//   * two key blocks' values of a lane are exactly the eight k-positions 8g .. 8g+7 a PV MFMA's B operand wants IF the k-position kk stands
//     for key 16 ((kk >> 2) & 1) + 4 (kk >> 3) + (kk & 3) of the 32-key span -- so P^T leaves the lane's own registers (no shuffle), and the
//     matching V^T A operand is gathered as two 8-byte LDS reads (keys 4g .. 4g+3 of both blocks).  (A product kernel would have the fused QKV
//     epilogue write V^T with that key permutation and read one 16-byte fragment.)
//   * a query's keys are spread over four lanes (g = 0 .. 3): row maxima and row sums are combined across them by two lane exchanges.
// Reference call sites: lightcontrol/lightcontrol_flux.py:92-95,173-177 (F.scaled_dot_product_attention in FluxAttnProcessor2_0).
#include "x2i_common.h"
#include "x2i_kernels.h"

namespace {

constexpr int KVB = 64;
constexpr int KTILE = KVB * 128 * 2;
constexpr int VTILE = 128 * KVB * 2;
constexpr float NEG_BIG = -1.0e30f;

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16,
                                   0, 0);
}
// over the four lanes that hold one query (lane ^ 16, lane ^ 32): v_permlane16_swap / v_permlane32_swap (gfx950), no LDS round trip
__device__ __forceinline__ float xg_max(float x) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float xg_sum(float x) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// VPERM: V^T arrives with its keys permuted within every 32-key span (position kk holds key 16 ((kk >> 2) & 1) + 4 (kk >> 3) + (kk & 3)): one
// 16-byte fragment read instead of two 8-byte gathers (what a product kernel would have the QKV epilogue write)
template <int NW, int THR, bool VPERM>
__global__ __launch_bounds__(NW * 64, 2) void attn16_fwd_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                const bf16_t* __restrict__ VT, bf16_t* __restrict__ O, int H, int S, int Spad,
                                                                int ldo, long long o_bs, float scale_log2, int nbatch, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][K 16K | VT 16K]
  constexpr int NT = NW * 64;
  constexpr int CH = 1024 / NT;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int nqt = gridDim.x / (H * nbatch);
  int bid = blockIdx.x;
  {
    const int T = gridDim.x, q = T >> 3, r = T & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int qt = bid % nqt, h = (bid / nqt) % H, b = bid / (nqt * H);
  const int q0 = qt * (32 * NW) + wave * 32;
  const long long bh = (long long)b * H + h;
  const bf16_t* Qh = Q + bh * Spad * 128;
  const bf16_t* Kh = K + bh * Spad * 128;
  const bf16_t* Vh = VT + bh * 128 * Spad;

  // Q fragments (second MFMA operand of S^T = K Q^T): lane holds Q[q0 + 16 qb + c][32 ds + 8 g .. + 8]
  bf16x8_t qf[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) qf[qb][ds] = *(const bf16x8_t*)(Qh + (long long)(q0 + qb * 16 + c) * 128 + ds * 32 + g * 8);

  int k_src[CH], v_src[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int p = j * NT + tid;
    {
      const int row = p >> 4, cphys = p & 15;
      k_src[j] = row * 128 + ((cphys ^ (row & 15)) << 3);
    }
    {
      const int row = p >> 3, cphys = p & 7;
      v_src[j] = row * Spad + ((cphys ^ ((row >> 1) & 7)) << 3);
    }
  }
  auto stage = [&](int buf, int kv0) {
    char* kb = smem + buf * (KTILE + VTILE);
    char* vb = kb + KTILE;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      glds16(Kh + (long long)kv0 * 128 + k_src[j], kb + (j * NT + wave * 64) * 16);
      glds16(Vh + kv0 + v_src[j], vb + (j * NT + wave * 64) * 16);
    }
  };
  // K fragment (first operand; key block kb, d-step ds): row 16 kb + c, logical 16-byte chunk 4 ds + g, physical chunk ^ (row & 15) = ^ c
  const int k_row_off = c * 256;
  // V^T fragment (first operand; d-block db, key span sp): row 16 db + c; the lane's eight k-positions are keys 32 sp + 4g .. + 3 and
  // 32 sp + 16 + 4g .. + 3: two 8-byte pieces; 16-byte chunk of key offset ko = ko >> 3, physical chunk ^ ((row >> 1) & 7)
  const int v_row_off = c * 128;

  f32x4_t oacc[8][2];
#pragma unroll
  for (int db = 0; db < 8; ++db)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) oacc[db][qb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {NEG_BIG, NEG_BIG}, l_run[2] = {0.f, 0.f};

  const int ntiles = (S + KVB - 1) / KVB;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    if (t + 1 < ntiles) stage(buf ^ 1, (t + 1) * KVB);
    const char* kbuf = smem + buf * (KTILE + VTILE);
    const char* vbuf = kbuf + KTILE;
    // ---- S^T: four key blocks x two query blocks, d in four steps of 32
    f32x4_t sacc[4][2];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) sacc[kb][qb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    {
      bf16x8_t kf[2][4];
      auto kload = [&](int ds, int kb) { return *(const bf16x8_t*)(kbuf + kb * 16 * 256 + k_row_off + (((ds * 4 + g) ^ c) << 4)); };
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) kf[0][kb] = kload(0, kb);
#pragma unroll
      for (int ds = 0; ds < 4; ++ds) {
        if (ds < 3) {
#pragma unroll
          for (int kb = 0; kb < 4; ++kb) kf[(ds + 1) & 1][kb] = kload(ds + 1, kb);
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) sacc[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ds & 1][kb], qf[qb][ds], sacc[kb][qb], 0, 0, 0);
        if (ds < 3) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // the next step's four fragment reads first ...
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                // ... then this step's eight MFMAs
      }
    }
    // lane (query c of block qb; g), key block kb, register r  <->  key = kv0 + 16 kb + 4 g + r
    const int kv0 = t * KVB;
    if (kv0 + KVB > S) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (kv0 + kb * 16 + 4 * g + r >= S) sacc[kb][0][r] = sacc[kb][1][r] = NEG_BIG;
    }
    // ---- online softmax (exp2 domain), one state per query block
    float mnew[2], alpha[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float mx = NEG_BIG;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sacc[kb][qb][r]);
      mx = xg_max(mx);
      mnew[qb] = fmaxf(m_run[qb], mx * scale_log2);
    }
    if (THR > 0 && __all(mnew[0] - m_run[0] <= (float)THR && mnew[1] - m_run[1] <= (float)THR)) {
      mnew[0] = m_run[0];
      mnew[1] = m_run[1];
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      alpha[qb] = __builtin_amdgcn_exp2f(m_run[qb] - mnew[qb]);
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = __builtin_amdgcn_exp2f(sacc[kb][qb][r] * scale_log2 - mnew[qb]);
          sacc[kb][qb][r] = pv;
          psum += pv;
        }
      l_run[qb] = l_run[qb] * alpha[qb] + psum;   // (this lane's keys only; the four lanes of a query are summed in the epilogue -- alpha is shared)
    }
    if (!__all(mnew[0] == m_run[0] && mnew[1] == m_run[1])) {
#pragma unroll
      for (int db = 0; db < 8; ++db)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
          for (int r = 0; r < 4; ++r) oacc[db][qb][r] *= alpha[qb];
    }
    m_run[0] = mnew[0];
    m_run[1] = mnew[1];
    // ---- P^T fragments: span sp (keys 32 sp ..), query block qb: k-positions 8g + 4b + r = sacc[2 sp + b][qb][r]
    bf16x8_t pf[2][2];
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        union { bf16x8_t v; uint32_t w[4]; } cv;
        cv.w[0] = pack_bf16x2(sacc[2 * sp][qb][0], sacc[2 * sp][qb][1]);
        cv.w[1] = pack_bf16x2(sacc[2 * sp][qb][2], sacc[2 * sp][qb][3]);
        cv.w[2] = pack_bf16x2(sacc[2 * sp + 1][qb][0], sacc[2 * sp + 1][qb][1]);
        cv.w[3] = pack_bf16x2(sacc[2 * sp + 1][qb][2], sacc[2 * sp + 1][qb][3]);
        pf[sp][qb] = cv.v;
      }
    // ---- O^T += V^T P^T: eight d-blocks x two spans x two query blocks; fragments of the next group of four d-blocks are read ahead
    {
      auto vload = [&](int sp, int db) {
        const int row = db * 16 + c;
        const char* rb = vbuf + db * 16 * 128 + v_row_off;
        const int sw = (row >> 1) & 7;
        if constexpr (VPERM) {
          return *(const bf16x8_t*)(rb + (((sp * 4 + g) ^ sw) << 4));
        } else {
          union { bf16x8_t v; uint2 h[2]; } vf;
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) {
            const int ko = sp * 32 + bb * 16 + 4 * g;
            vf.h[bb] = *(const uint2*)(rb + (((ko >> 3) ^ sw) << 4) + ((ko & 4) << 1));
          }
          return vf.v;
        }
      };
      bf16x8_t vf[2][4];
#pragma unroll
      for (int d4 = 0; d4 < 4; ++d4) vf[0][d4] = vload(0, d4);
#pragma unroll
      for (int st = 0; st < 4; ++st) {   // step = (span sp = st >> 1, d-blocks 4 (st & 1) .. + 3)
        if (st < 3) {
#pragma unroll
          for (int d4 = 0; d4 < 4; ++d4) vf[(st + 1) & 1][d4] = vload((st + 1) >> 1, 4 * ((st + 1) & 1) + d4);
        }
#pragma unroll
        for (int d4 = 0; d4 < 4; ++d4)
#pragma unroll
          for (int qb = 0; qb < 2; ++qb)
            oacc[4 * (st & 1) + d4][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[st & 1][d4], pf[st >> 1][qb], oacc[4 * (st & 1) + d4][qb], 0, 0, 0);
        if (st < 3) __builtin_amdgcn_sched_group_barrier(0x100, VPERM ? 4 : 8, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: lane (query c of block qb; g) holds d = 16 db + 4 g + r
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const float l = xg_sum(l_run[qb]);
    const float inv = 1.f / l;
    const int q = q0 + qb * 16 + c;
    if (lse && g == 0 && q < Spad) lse[bh * Spad + q] = q < S ? m_run[qb] + __log2f(l) : 1.0e30f;
    if (q < S) {
      bf16_t* orow = O + (long long)b * o_bs + (long long)q * ldo + h * 128;
#pragma unroll
      for (int db = 0; db < 8; ++db)
        *(uint2*)(orow + db * 16 + 4 * g) = make_uint2(pack_bf16x2(oacc[db][qb][0] * inv, oacc[db][qb][1] * inv),
                                                       pack_bf16x2(oacc[db][qb][2] * inv, oacc[db][qb][3] * inv));
    }
  }
}

}  // namespace

// attn_variant = 10 (A/B): returns X2I_ERR_STATE when the shape is not served (the caller falls through to the product kernels)
int x2i_launch_attention_16(const void* Q, const void* K, const void* VT, void* O, int B, int H, int S, int Spad, int ldo, long long o_bs,
                            float scale_log2, hipStream_t stream, float* lse, int vperm) {
  if (ldo % 4 || o_bs % 4 || (((uintptr_t)O) & 7)) return X2I_ERR_STATE;
  const size_t shm = 2 * (KTILE + VTILE);
  auto kern = vperm ? attn16_fwd_kernel<4, 8, true> : attn16_fwd_kernel<4, 8, false>;
  const int rc = x2i_ensure_dynamic_smem((const void*)kern, (int)shm);
  if (rc) return rc;
  dim3 grid(((S + 127) / 128) * H * B);
  hipLaunchKernelGGL(kern, grid, dim3(256), shm, stream, (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)VT, (bf16_t*)O, H, S, Spad, ldo, o_bs,
                     scale_log2, B, lse);
  return x2i_check_launch("attention (16x16x32)");
}
