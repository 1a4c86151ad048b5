// Flash attention forward for the FLUX joint text+image attention: head_dim 128, no mask, non-causal, bf16 in/out,
// fp32 softmax statistics and accumulation.  Stands behind F.scaled_dot_product_attention as called by diffusers'
// FluxAttnProcessor2_0 (reference call sites lightcontrol/lightcontrol_flux.py:92-95,173-177).
//
// CDNA4 mapping
//   * one workgroup = NW waves, each wave owns 32 query rows (QBLK = 32*NW); K/V stream through LDS in 64-key tiles
//   * "swapped" QK^T: S^T = K Q^T with v_mfma_f32_32x32x16_bf16, so every lane ends up holding 32 scores of ONE
//     query row (the other 32 live in lane^32): row max / row sum are 31 in-register ops + one cross-half exchange,
//     never an LDS round trip
//   * the key index fed to MFMA row i is kvmap(i) = i with bits 2,3 swapped: that makes each lane's 8-register groups
//     contiguous in the key axis, so P^T is directly the B operand of the PV MFMA (no permlane / LDS shuffle) and the
//     matching V^T A-operand fragment is one contiguous ds_read_b128
//   * V arrives pre-transposed (VT [B,H,128,Spad], written by x2i_qkv_split), so both K and V^T tiles are plain
//     row-major images filled by LDS-DMA (global_load_lds dwordx4); bank conflicts are removed with an XOR swizzle
//     applied on the DMA source address and on the ds_read_b128 address (CDNA guide rule 21)
//   * double-buffered tiles: DMA of tile t+1 overlaps the 32 MFMAs of tile t; one barrier per tile
//   * O^T accumulates in 4 x f32x16; epilogue normalises by 1/l and writes token-major bf16 (16-byte stores after a half-wave exchange)
#include "x2i_common.h"
#include "x2i_kernels.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int KVB = 64;                  // keys per tile
constexpr int KTILE = KVB * 128 * 2;     // 16 KiB  K  tile: [64 keys][128 d]
constexpr int VTILE = 128 * KVB * 2;     // 16 KiB  V^T tile: [128 d][64 keys]
constexpr float NEG_BIG = -1.0e30f;

template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (N > 0) {
    sfor<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ABL (measurement only, wrong results; tools/microbench.py): 1 = no softmax VALU, 2 = no barrier/wait after the first
// tile, 4 = no DMA after the first tile.  ABL = 0 is the product kernel.
// OUT8: the output is e4m3 (sat(o * oinv)), 8 bytes per lane and d-group -- the A operand of an fp8 projection (x2i_attention_e4m3out)
template <int NW, int THR, int ABL = 0, bool OUT8 = false>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                           const bf16_t* __restrict__ VT, bf16_t* __restrict__ O, int H, int S,
                                                           int Spad, int ldo, long long o_bs, float scale_log2, int nbatch,
                                                           float oinv, float* __restrict__ lse) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][K 16K | VT 16K]
  constexpr int NT = NW * 64;
  constexpr int CH = 1024 / NT;  // 16-byte chunks per thread per tile (1024 chunks per 16 KiB tile)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;     // which half-wave
  const int li = lane & 31;     // MFMA row/col index owned by this lane
  // XCD-aware block order (workgroup id % 8 = XCD): each XCD walks a contiguous range of (batch, head, q-tile) triples,
  // so all query tiles of one head -- which stream the same 2.4 MB of K / V^T -- hit the same 4 MiB L2.
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

  // ---- Q fragments (B operand of S^T = K Q^T): lane holds Q[q0+li][ds*16 + hi*8 .. +8]
  bf16x8_t qf[8];
#pragma unroll
  for (int ds = 0; ds < 8; ++ds) qf[ds] = *(const bf16x8_t*)(Qh + (long long)(q0 + li) * 128 + ds * 16 + hi * 8);

  // ---- DMA source offsets (elements) for this thread's chunks; LDS image is linear, swizzle goes on the source
  int k_src[CH], v_src[CH];
#pragma unroll
  for (int j = 0; j < CH; ++j) {
    const int p = j * NT + tid;
    {  // K tile: row = key (256 B = 16 chunks); physical chunk c holds logical chunk c ^ (row & 15)
      const int row = p >> 4, cphys = p & 15;
      k_src[j] = row * 128 + ((cphys ^ (row & 15)) << 3);
    }
    {  // V^T tile: row = d (128 B = 8 chunks); physical chunk c holds logical chunk c ^ ((row >> 1) & 7)
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

  // ---- per-lane LDS read offsets
  // K fragment (A operand, sub-tile u, d-step ds): row = u*32 + kvmap(li), logical chunk = ds*2 + hi
  const int kvm = (li & 0x13) | ((li & 4) << 1) | ((li & 8) >> 1);  // swap bits 2 and 3
  const int k_row_off = kvm * 256;
  const int k_swz = kvm & 15;
  // V^T fragment (A operand, d-block db, sub-tile u, k-step t): row = db*32 + li, logical chunk = 4u + 2t + hi
  const int v_row_off = li * 128;
  const int v_swz = (li >> 1) & 7;

  f32x16_t oacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
  float m_run = NEG_BIG, l_run = 0.f;

  const int ntiles = (S + KVB - 1) / KVB;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // One key/value tile.  LAST = false: a successor tile exists and every key is valid, so the prefetch and the score path
  // carry no branches (one basic block up to the rescale test); the final tile handles the ragged tail.
  auto kv_tile = [&](int t, auto last_c) {
    constexpr int MODE = decltype(last_c)::value;  // 0 steady, 1 last, 2 general (runtime tests; A/B reference, ABL & 32)
    constexpr bool LAST = MODE != 0;
    const int buf = t & 1;
    if ((MODE == 0 || (MODE == 2 && t + 1 < ntiles)) && !((ABL & 4) && t > 0)) stage(buf ^ 1, (t + 1) * KVB);
    const char* kb = smem + buf * (KTILE + VTILE);
    const char* vb = kb + KTILE;

    // ---- S^T = K Q^T : two 32-key sub-tiles.  K fragments are read in groups of four, one group ahead of the MFMAs that
    // consume them (hipcc otherwise serialises ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma per fragment); consecutive MFMAs
    // alternate between the two sub-tile accumulators.  Fragment f of group g: ds = 2*g + (f>>1), sub-tile u = f & 1.
    f32x16_t sacc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[u][r] = 0.f;
    {
      bf16x8_t kf[2][4];
      auto kload = [&](int g, int f) {
        const int ds = 2 * g + (f >> 1), u = f & 1;
        return *(const bf16x8_t*)(kb + u * 32 * 256 + k_row_off + (((ds * 2 + hi) ^ k_swz) << 4));
      };
#pragma unroll
      for (int f = 0; f < 4; ++f) kf[0][f] = kload(0, f);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (g < 3) {
#pragma unroll
          for (int f = 0; f < 4; ++f) kf[(g + 1) & 1][f] = kload(g + 1, f);
        }
#pragma unroll
        for (int f = 0; f < 4; ++f)
          sacc[f & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[g & 1][f], qf[2 * g + (f >> 1)], sacc[f & 1], 0, 0, 0);
        if (g < 3) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);  // the next group's 4 DS reads first ...
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);             // ... then this group's 4 MFMAs
      }
    }
    // lane (q = li, hi), sub-tile u, reg r  <->  key = kv0 + u*32 + 16*(r>>3) + 8*hi + (r&7)
    const int kv0 = t * KVB;
    if (LAST && kv0 + KVB > S) {  // ragged last tile: mask keys >= S
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + u * 32 + 16 * (r >> 3) + 8 * hi + (r & 7);
          if (key >= S) sacc[u][r] = NEG_BIG;
        }
    }
    if (ABL & 1) {  // ablation: skip the softmax VALU work (keep the data dependence QK -> P -> PV)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; r += 4) sacc[u][r] = sacc[u][r] * 0.001f;
    }
    // ---- online softmax (scores scaled into the exp2 domain)
    float mx = NEG_BIG;
    if (!(ABL & 1)) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[u][r]);
    mx = xhalf_max(mx);
    // defer-max (T13): keep the old running max while no row of this wave grew by more than THR (exp2 domain), so the
    // O rescale pass is skipped on most tiles; P is then bounded by 2^THR instead of 1 (fp32 accumulate, bf16 P).
    float m_new = fmaxf(m_run, mx * scale_log2);
    if (THR > 0 && __all(m_new - m_run <= (float)THR)) m_new = m_run;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f(sacc[u][r] * scale_log2 - m_new);
        sacc[u][r] = pv;
        psum += pv;
      }
    l_run = l_run * alpha + psum;
    if (!__all(m_new == m_run)) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] *= alpha;
    }
    m_run = m_new;
    }

    // ---- P^T fragments (B operand): sub-tile u, k-step kt uses regs 8kt..8kt+7  (keys u*32+16kt+8hi+0..7)
    bf16x8_t pf[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
        union { bf16x8_t v; uint32_t w[4]; } cv;
#pragma unroll
        for (int j = 0; j < 4; ++j) cv.w[j] = pack_bf16x2(sacc[u][kt * 8 + 2 * j], sacc[u][kt * 8 + 2 * j + 1]);
        pf[u][kt] = cv.v;
      }
    // ---- O^T += V^T P^T : same one-group-ahead fragment pipeline; group g = (u, kt) feeds the four d-block accumulators
    {
      bf16x8_t vf[2][4];
      auto vload = [&](int g, int db) {
        const int u = g >> 1, kt = g & 1;
        return *(const bf16x8_t*)(vb + db * 32 * 128 + v_row_off + (((4 * u + 2 * kt + hi) ^ v_swz) << 4));
      };
#pragma unroll
      for (int db = 0; db < 4; ++db) vf[0][db] = vload(0, db);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (g < 3) {
#pragma unroll
          for (int db = 0; db < 4; ++db) vf[(g + 1) & 1][db] = vload(g + 1, db);
        }
#pragma unroll
        for (int db = 0; db < 4; ++db)
          oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[g & 1][db], pf[g >> 1][g & 1], oacc[db], 0, 0, 0);
        if (g < 3) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
    }

    if (!((ABL & 2) && t > 0)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // next tile's DMA (issued by this wave) has landed
      __syncthreads();
    }
  };
  if (ABL & 32) {
    for (int t = 0; t < ntiles; ++t) kv_tile(t, std::integral_constant<int, 2>{});
  } else {
    for (int t = 0; t < ntiles - 1; ++t) kv_tile(t, std::integral_constant<int, 0>{});
    kv_tile(ntiles - 1, std::integral_constant<int, 1>{});
  }

  // ---- epilogue: O[q][d] = O^T[d][q] / l ; lane (q = li, hi) holds d = db*32 + 8*(r>>2) + 4*hi + (r&3)
  l_run = xhalf_sum(l_run);
  const float inv = 1.f / l_run;
  const int q = q0 + li;
  // log2-sum-exp of the scaled scores (the statistic the backward pass needs, x2i_attention_lse_bf16); +big on the padding rows
  if (lse && hi == 0 && q < Spad) lse[bh * Spad + q] = q < S ? m_run + __log2f(l_run) : 1.0e30f;
  if constexpr (OUT8) {
    uint8_t* orow8 = (uint8_t*)O + (long long)b * o_bs + (long long)q * ldo + h * 128;
    const float sc = inv * oinv;
    auto pk4 = [&](float a, float bq, float c, float d) {
      int v = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(a * sc, -448.f, 448.f), __builtin_amdgcn_fmed3f(bq * sc, -448.f, 448.f), 0, false);
      return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(c * sc, -448.f, 448.f), __builtin_amdgcn_fmed3f(d * sc, -448.f, 448.f), v, true);
    };
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        const uint32_t a4 = pk4(oacc[db][4 * g], oacc[db][4 * g + 1], oacc[db][4 * g + 2], oacc[db][4 * g + 3]);
        const uint32_t b4 = pk4(oacc[db][4 * g + 4], oacc[db][4 * g + 5], oacc[db][4 * g + 6], oacc[db][4 * g + 7]);
        const auto s0 = __builtin_amdgcn_permlane32_swap(a4, b4, false, false);
        if (q < S) *(uint2*)(orow8 + db * 32 + 8 * (g + hi)) = make_uint2(s0[0], s0[1]);
      }
    return;
  }
  bf16_t* orow = O + (long long)b * o_bs + (long long)q * ldo + h * 128;
  if ((((uintptr_t)O) & 15) == 0 && (ldo & 7) == 0 && (o_bs & 7) == 0) {
    // half-wave exchange (v_permlane32_swap) turns two 8-byte fragments of neighbouring d-groups into one 16-byte
    // store per lane: lanes 0-31 end up with d = 8g..8g+7, lanes 32-63 with d = 8(g+1)..8(g+1)+7
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        const uint32_t a0 = pack_bf16x2(oacc[db][4 * g] * inv, oacc[db][4 * g + 1] * inv);
        const uint32_t a1 = pack_bf16x2(oacc[db][4 * g + 2] * inv, oacc[db][4 * g + 3] * inv);
        const uint32_t b0 = pack_bf16x2(oacc[db][4 * g + 4] * inv, oacc[db][4 * g + 5] * inv);
        const uint32_t b1 = pack_bf16x2(oacc[db][4 * g + 6] * inv, oacc[db][4 * g + 7] * inv);
        const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
        if (q < S) *(uint4*)(orow + db * 32 + 8 * (g + hi)) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
      }
  } else if (q < S) {
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = db * 32 + 8 * g + 4 * hi;
        const uint2 o2 = make_uint2(pack_bf16x2(oacc[db][4 * g] * inv, oacc[db][4 * g + 1] * inv),
                                    pack_bf16x2(oacc[db][4 * g + 2] * inv, oacc[db][4 * g + 3] * inv));
        *(uint2*)(orow + d) = o2;
      }
  }
}

}  // namespace

int x2i_launch_attention(const void* Q, const void* K, const void* VT, void* O, int B, int H, int S, int Spad, int ldo,
                         long long o_bs, float scale, hipStream_t stream, int out8, float oinv, float* lse) {
  if (!Q || !K || !VT || !O) return x2i_set_error(X2I_ERR_ARG, "attention: null pointer");
  if (B <= 0 || H <= 0 || S <= 0 || Spad < S || Spad % 128) return x2i_set_error(X2I_ERR_SHAPE, "attention: need Spad %% 128 == 0 and Spad >= S (S=%d Spad=%d)", S, Spad);
  if (out8 ? (ldo % 8 || o_bs % 8 || (((uintptr_t)O) & 7)) : (ldo % 4 || o_bs % 4 || (((uintptr_t)O) & 7)))
    return x2i_set_error(X2I_ERR_ALIGN, "attention: output rows must be 8-byte aligned");
  const size_t shm = 2 * (KTILE + VTILE);
  float scale_log2 = scale * 1.4426950408889634f;
  // scale = ln 2: the caller folded softmax_scale * log2(e) into Q (x2i_qkv_desc.q_scale); the exp2-domain multiplier is exactly 1
  const bool unit = fabsf(scale_log2 - 1.f) < 1e-6f;
  if (unit) scale_log2 = 1.f;
  const X2IOptions& opt = x2i_options();
  const int var = opt.attn_variant;  // 0 = automatic; A/B: 1 = 8 lock-step waves, 2 = no defer-max, 3 = both, 4 = 4-wave kernel, 5 / 6 = ping-pong schedule 0 (defer-max / none), 7 / 8 = ping-pong schedules 1 / 2
  // a Q that already carries the scale and sequences long enough that ONE sample's 256-row workgroups fill half the chip: the
  // hand-scheduled one-wave-per-SIMD kernel (attention_w4.hip), whose softmax has no multiply.  The rule looks at the sequence, not at
  // the batch: this kernel rounds differently from the 4-wave / ping-pong pair (which are bit-identical to each other), and a
  // sample's result must not depend on how many other samples share its launch (tests/test_fullsize_gpu.py).  Variant 9 forces it
  // for any scale and size (the kernel then rescales its bf16 Q fragments itself, at the price of a second rounding of Q); 5..8
  // select the 8-wave ping-pong kernel
  if ((var == 0 && unit && (long long)((S + 255) / 256) * H >= 128) || var == 9) {
    const int rc = x2i_launch_attention_w4(Q, K, VT, O, B, H, S, Spad, ldo, o_bs, scale_log2, unit ? 0 : 1, stream, lse, out8, oinv);
    if (rc != X2I_ERR_STATE) return rc;
  }
  if (var == 12 && !out8) {   // A/B: the hand-scheduled kernel on 16 x 16 x 32 MFMAs (attention_w16.hip); V^T pre-permuted by the caller -- tools / tests only
    const int rc = x2i_launch_attention_w16(Q, K, VT, O, B, H, S, Spad, ldo, o_bs, scale_log2, unit ? 0 : 1, stream, lse);
    if (rc != X2I_ERR_STATE) return rc;
  }
#ifdef X2I_ABLATION   // (measurement library only since round 6)
  if ((var == 10 || var == 11) && !out8) {   // (11: V^T arrives with the 32-key-span permutation of attention16.hip -- tools only)   // A/B: the 16 x 16 x 32 MFMA shape (attention16.hip; compare with variant 4, the same organisation on 32 x 32 x 16)
    const int rc = x2i_launch_attention_16(Q, K, VT, O, B, H, S, Spad, ldo, o_bs, scale_log2, stream, lse, var == 11);
    if (rc != X2I_ERR_STATE) return rc;
  }
#endif
  // the 8-wave ping-pong kernel (attention_pp.hip)
  if ((var == 0 && (long long)((S + 255) / 256) * H * B >= 256) || var == 5 || var == 6 || var == 7 || var == 8) {
    const int rc = x2i_launch_attention_pp(Q, K, VT, O, B, H, S, Spad, ldo, o_bs, scale_log2, stream, out8, oinv, var == 6 ? 0 : 8, lse);
    if (rc != X2I_ERR_STATE) return rc;  // X2I_ERR_STATE: shape / alignment not served by that kernel -> fall through
  }
#define X2I_ATTN_LAUNCH(NW_, THR_)                                                                                          \
  {                                                                                                                         \
    const int rc_ = x2i_ensure_dynamic_smem((const void*)attn_fwd_kernel<NW_, THR_>, (int)shm);                             \
    if (rc_) return rc_;                                                                                                    \
    dim3 grid(((S + 32 * NW_ - 1) / (32 * NW_)) * H * B);                                                                   \
    hipLaunchKernelGGL((attn_fwd_kernel<NW_, THR_>), grid, dim3(NW_ * 64), shm, stream, (const bf16_t*)Q, (const bf16_t*)K, \
                       (const bf16_t*)VT, (bf16_t*)O, H, S, Spad, ldo, o_bs, scale_log2, B, 1.f, lse);                      \
  }
  if (out8) {
    const int rc_ = x2i_ensure_dynamic_smem((const void*)attn_fwd_kernel<4, 8, 0, true>, (int)shm);
    if (rc_) return rc_;
    dim3 grid(((S + 127) / 128) * H * B);
    hipLaunchKernelGGL((attn_fwd_kernel<4, 8, 0, true>), grid, dim3(256), shm, stream, (const bf16_t*)Q, (const bf16_t*)K,
                       (const bf16_t*)VT, (bf16_t*)O, H, S, Spad, ldo, o_bs, scale_log2, B, oinv, lse);
    return x2i_check_launch("attention");
  }
#ifdef X2I_ABLATION
  const int abl = opt.attn_ablate;  // measurement-only variants (tools/microbench.py), wrong results by design
#define X2I_ATTN_LAUNCH_ABL(A_)                                                                                              \
  {                                                                                                                          \
    const int rc_ = x2i_ensure_dynamic_smem((const void*)attn_fwd_kernel<4, 8, A_>, (int)shm);                               \
    if (rc_) return rc_;                                                                                                     \
    dim3 grid(((S + 127) / 128) * H * B);                                                                                    \
    hipLaunchKernelGGL((attn_fwd_kernel<4, 8, A_>), grid, dim3(256), shm, stream, (const bf16_t*)Q, (const bf16_t*)K,        \
                       (const bf16_t*)VT, (bf16_t*)O, H, S, Spad, ldo, o_bs, scale_log2, B, 1.f, lse);                       \
  }
  if (abl == 1) X2I_ATTN_LAUNCH_ABL(1)
  else if (abl == 2) X2I_ATTN_LAUNCH_ABL(2)
  else if (abl == 4) X2I_ATTN_LAUNCH_ABL(4)
  else if (abl == 7) X2I_ATTN_LAUNCH_ABL(7)
  else if (abl == 32) X2I_ATTN_LAUNCH_ABL(32)
  else
#endif
#ifdef X2I_ABLATION   // (A/B forms 1 = eight lock-step waves, 2 = no defer-max, 3 = both: measurement library only since round 6)
  if (var == 1) X2I_ATTN_LAUNCH(8, 8)
  else if (var == 2) X2I_ATTN_LAUNCH(4, 0)
  else if (var == 3) X2I_ATTN_LAUNCH(8, 0)
  else
#endif
  X2I_ATTN_LAUNCH(4, 8)
#undef X2I_ATTN_LAUNCH
  return x2i_check_launch("attention");
}
