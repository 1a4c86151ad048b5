// Flash-style attention BACKWARD for the distillation step: gradients of
// O = softmax(scale Q K^T) V with respect to Q, K, V from dO, without materialising the [S, S] matrices.  Same head geometry as the
// forward kernels (attention.hip: head_dim 128, bf16 operands, fp32 statistics, 64-row streamed tiles through LDS, 32 persistent rows
// per wave held as MFMA B operands, v_mfma_f32_32x32x16_bf16 with the bit-2/3 row permutation that makes the probability registers
// the next MFMA's B operand directly).
//
// With L2[q] = log2 sum_j exp(scale s_qj) (pass 2 below) and D[q] = sum_d dO[q][d] O[q][d] (attn_bwd_prep_kernel):
//     P = exp2(scale_log2 S - L2),   dP = dO V^T,   dS = P * (dP - D),   dQ = scale dS K,   dK = scale dS^T Q,   dV = P^T dO
//
//   MODE 2  statistics : persistent 128-query block, streams K tiles, online max / sum            -> L2            (1 MFMA group / tile)
//   MODE 0  dQ         : persistent queries (Q and dO rows as B operands), streams K, V, K^T      -> dQ            (3 groups / tile)
//   MODE 1  dK, dV     : persistent keys (K and V rows as B operands), streams Q, dO, dO^T, Q^T   -> dV, dK        (4 groups / tile)
//
// Two passes over the score matrix instead of atomics on dQ: deterministic, and each pass is the forward kernel's loop with other
// operands.  Operand layouts: row-major [B,H,Spad,128] for Q, K, V, dO; transposed [B,H,128,Spad] for K^T, Q^T, dO^T -- produced
// by x2i_transpose_bf16 from what the forward keeps (the transposes are ~1 % of the pass).  Rows >= S are neutralised through the
// statistics (L2 = +big -> P = 0), keys >= S of the last tile by an explicit mask.
#include "x2i_common.h"
#include "x2i_kernels.h"
#include <type_traits>

namespace {

constexpr int KVB = 64;
constexpr int TILE = 16384;  // 64 x 128 bf16, either orientation
constexpr float BIG = 1.0e30f;

// LDS-DMA through the MUBUF path (buffer_load_dwordx4 ... lds): counted in vmcnt only.  The flat form (global_load_lds) also counts in
// lgkmcnt, so the first fragment wait of a tile (lgkmcnt(0)) would wait for the NEXT tile's DMA as well.
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff_bytes, char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff_bytes, 0, 0, 0);
}

struct Geo {
  int hi, li, k_row_off, k_swz, v_row_off, v_swz;
};

// A-operand fragments, exactly the forward's addressing: row tile [64 rows][128 d] (group g, fragment f: ds = 2g + (f >> 1), sub-tile f & 1)
__device__ __forceinline__ bf16x8_t row_frag(const char* tb, const Geo& G, int g, int f) {
  const int ds = 2 * g + (f >> 1), u = f & 1;
  return *(const bf16x8_t*)(tb + u * 32 * 256 + G.k_row_off + (((ds * 2 + G.hi) ^ G.k_swz) << 4));
}
// transposed tile [128 d][64 rows] (group g = (u, kt), d-block db)
__device__ __forceinline__ bf16x8_t col_frag(const char* tb, const Geo& G, int g, int db) {
  const int u = g >> 1, kt = g & 1;
  return *(const bf16x8_t*)(tb + db * 32 * 128 + G.v_row_off + (((4 * u + 2 * kt + G.hi) ^ G.v_swz) << 4));
}

// acc[sub-tile] = tile rows x persistent fragments  (the forward's S^T = K Q^T loop)
__device__ __forceinline__ void score_mma(const char* tb, const Geo& G, const bf16x8_t (&pf)[8], f32x16_t (&acc)[2]) {
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
  bf16x8_t fr[2][4];
#pragma unroll
  for (int f = 0; f < 4; ++f) fr[0][f] = row_frag(tb, G, 0, f);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (g < 3) {
#pragma unroll
      for (int f = 0; f < 4; ++f) fr[(g + 1) & 1][f] = row_frag(tb, G, g + 1, f);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[f & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[g & 1][f], pf[2 * g + (f >> 1)], acc[f & 1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// oacc[d-block] += transposed tile x probability-like fragments  (the forward's O^T += V^T P^T loop)
__device__ __forceinline__ void accum_mma(const char* tb, const Geo& G, const bf16x8_t (&pf)[2][2], f32x16_t (&oacc)[4]) {
  bf16x8_t fr[2][4];
#pragma unroll
  for (int db = 0; db < 4; ++db) fr[0][db] = col_frag(tb, G, 0, db);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    if (g < 3) {
#pragma unroll
      for (int db = 0; db < 4; ++db) fr[(g + 1) & 1][db] = col_frag(tb, G, g + 1, db);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int db = 0; db < 4; ++db) oacc[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[g & 1][db], pf[g >> 1][g & 1], oacc[db], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// The two score products of a tile as ONE fragment pipeline (steps 0-3: rows of tile A x pa -> sa, steps 4-7: rows of tile B x pb -> sb):
// no LDS-latency bubble between them.  NS = 4 runs the first product only.
template <int NS>
__device__ __forceinline__ void score_phase(const char* ta, const char* tb, const Geo& G, const bf16x8_t (&pa)[8], const bf16x8_t (&pb)[8],
                                            f32x16_t (&sa)[2], f32x16_t (&sb)[2]) {
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) { sa[u][r] = 0.f; sb[u][r] = 0.f; }
  bf16x8_t fr[2][4];
#pragma unroll
  for (int f = 0; f < 4; ++f) fr[0][f] = row_frag(ta, G, 0, f);
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    if (st + 1 < NS) {
#pragma unroll
      for (int f = 0; f < 4; ++f) fr[(st + 1) & 1][f] = row_frag(st + 1 < 4 ? ta : tb, G, (st + 1) & 3, f);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int g = st & 3;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      if (st < 4) sa[f & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[st & 1][f], pa[2 * g + (f >> 1)], sa[f & 1], 0, 0, 0);
      else sb[f & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[st & 1][f], pb[2 * g + (f >> 1)], sb[f & 1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
// The accumulation products of a tile as one pipeline whose first fragments (`fr0`) were read BEFORE the element-wise section:
// steps 0-3: transposed tile C x f0 -> o0; steps 4-7 (NS = 8): transposed tile D x f1 -> o1.
template <int NS>
__device__ __forceinline__ void accum_phase(const char* tc, const char* td, const Geo& G, const bf16x8_t (&f0)[2][2], const bf16x8_t (&f1)[2][2],
                                            f32x16_t (&o0)[4], f32x16_t (&o1)[4], bf16x8_t (&fr)[2][4]) {
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    if (st + 1 < NS) {
#pragma unroll
      for (int db = 0; db < 4; ++db) fr[(st + 1) & 1][db] = col_frag(st + 1 < 4 ? tc : td, G, (st + 1) & 3, db);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int g = st & 3;
#pragma unroll
    for (int db = 0; db < 4; ++db) {
      if (st < 4) o0[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[st & 1][db], f0[g >> 1][g & 1], o0[db], 0, 0, 0);
      else o1[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[st & 1][db], f1[g >> 1][g & 1], o1[db], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
__device__ __forceinline__ void pack_frags(const f32x16_t (&a)[2], bf16x8_t (&pf)[2][2]) {
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      union { bf16x8_t v; uint32_t w[4]; } cv;
#pragma unroll
      for (int j = 0; j < 4; ++j) cv.w[j] = pack_bf16x2(a[u][kt * 8 + 2 * j], a[u][kt * 8 + 2 * j + 1]);
      pf[u][kt] = cv.v;
    }
}
// out[row][d] = alpha * acc^T: lane (row = li, hi) holds d = db*32 + 8*(r>>2) + 4*hi + (r&3); half-wave exchange -> 16-byte stores
__device__ __forceinline__ void store_rows(bf16_t* orow, const f32x16_t (&acc)[4], float alpha, int hi, bool live) {
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
      const uint32_t a0 = pack_bf16x2(acc[db][4 * g] * alpha, acc[db][4 * g + 1] * alpha);
      const uint32_t a1 = pack_bf16x2(acc[db][4 * g + 2] * alpha, acc[db][4 * g + 3] * alpha);
      const uint32_t b0 = pack_bf16x2(acc[db][4 * g + 4] * alpha, acc[db][4 * g + 5] * alpha);
      const uint32_t b1 = pack_bf16x2(acc[db][4 * g + 6] * alpha, acc[db][4 * g + 7] * alpha);
      const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
      const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
      if (live) *(uint4*)(orow + db * 32 + 8 * (g + hi)) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
    }
}

struct Rsrc4 {
  __amdgpu_buffer_rsrc_t a, b, c, d;
};

// One streamed tile: start the DMA of the next tile into `nxt`, then run this tile's MFMA groups on `cur`.  cur / nxt are __restrict__
// parameters of one function so that hipcc can tell the fragment reads from the buffer being filled (else every ds_read behind a DMA
// issue gets a conservative vmcnt(0)).
template <int MODE>
__device__ __forceinline__ void bwd_tile(const char* __restrict__ cur, char* __restrict__ nxt, bool issue, int s_next, int s0, const Rsrc4& R,
                                         const int (&row_src)[4], const int (&col_src)[4], int wave, const Geo& G, const bf16x8_t (&pa)[8],
                                         const bf16x8_t (&pb)[8], float myL, float myD, const float* __restrict__ L2h,
                                         const float* __restrict__ Dh, int S, float scale_log2, f32x16_t (&oacc0)[4], f32x16_t (&oacc1)[4],
                                         float& m_run, float& l_run) {
  if (issue) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      char* dst = nxt + (j * 256 + wave * 64) * 16;
      dma16(R.a, (uint32_t)(s_next * 256 + row_src[j] * 2), dst);
      if (MODE != 2) {
        dma16(R.b, (uint32_t)(s_next * 256 + row_src[j] * 2), dst + TILE);
        dma16(R.c, (uint32_t)(s_next * 2 + col_src[j] * 2), dst + 2 * TILE);
        if (MODE == 1) dma16(R.d, (uint32_t)(s_next * 2 + col_src[j] * 2), dst + 3 * TILE);
      }
    }
  }
  f32x16_t sacc[2], dacc[2];
  // per streamed row statistics (MODE 1): lane (hi), sub-tile u, reg r <-> row s0 + u*32 + 16*(r>>3) + 8*hi + (r&7)
  float Lr[2][16], Dr[2][16];
  if (MODE == 1) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int o = s0 + u * 32 + 16 * a + 8 * G.hi;
        const f32x4_t l0 = *(const f32x4_t*)(L2h + o), l1 = *(const f32x4_t*)(L2h + o + 4);
        const f32x4_t d0 = *(const f32x4_t*)(Dh + o), d1 = *(const f32x4_t*)(Dh + o + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          Lr[u][8 * a + j] = l0[j]; Lr[u][8 * a + 4 + j] = l1[j];
          Dr[u][8 * a + j] = d0[j]; Dr[u][8 * a + 4 + j] = d1[j];
        }
      }
  }
  if (MODE == 2) {
    score_phase<4>(cur, cur, G, pa, pa, sacc, dacc);
    // online max / sum of scale_log2 * s over the valid keys
    float mx = -BIG;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = s0 + u * 32 + 16 * (r >> 3) + 8 * G.hi + (r & 7);
        sacc[u][r] = key < S ? sacc[u][r] * scale_log2 : -BIG;
        mx = fmaxf(mx, sacc[u][r]);
      }
    mx = xhalf_max(mx);
    const float m_new = fmaxf(m_run, mx);
    float ps = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) ps += __builtin_amdgcn_exp2f(sacc[u][r] - m_new);
    l_run = l_run * __builtin_amdgcn_exp2f(m_run - m_new) + ps;
    m_run = m_new;
  } else {
    score_phase<8>(cur, cur + TILE, G, pa, pb, sacc, dacc);
    // first fragments of the accumulation pipeline: in flight while the element-wise section runs
    bf16x8_t fr[2][4];
#pragma unroll
    for (int db = 0; db < 4; ++db) fr[0][db] = col_frag(cur + 2 * TILE, G, 0, db);
    f32x16_t pv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p;
        if (MODE == 0) {
          const int key = s0 + u * 32 + 16 * (r >> 3) + 8 * G.hi + (r & 7);
          p = key < S ? __builtin_amdgcn_exp2f(sacc[u][r] * scale_log2 - myL) : 0.f;
          dacc[u][r] = p * (dacc[u][r] - myD);
        } else {
          p = __builtin_amdgcn_exp2f(sacc[u][r] * scale_log2 - Lr[u][r]);
          dacc[u][r] = p * (dacc[u][r] - Dr[u][r]);
        }
        pv[u][r] = p;
      }
    bf16x8_t dsf[2][2];
    pack_frags(dacc, dsf);
    if (MODE == 0) {
      accum_phase<4>(cur + 2 * TILE, cur + 2 * TILE, G, dsf, dsf, oacc0, oacc1, fr);   // dQ^T += K^T dS^T
    } else {
      bf16x8_t pf[2][2];
      pack_frags(pv, pf);
      accum_phase<8>(cur + 2 * TILE, cur + 3 * TILE, G, pf, dsf, oacc0, oacc1, fr);   // dV^T += dO^T P;  dK^T += Q^T dS
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void attn_bwd_kernel(const bf16_t* __restrict__ PA, const bf16_t* __restrict__ PB,
                                                         const bf16_t* __restrict__ TA, const bf16_t* __restrict__ TB,
                                                         const bf16_t* __restrict__ TC, const bf16_t* __restrict__ TD, float* __restrict__ L2,
                                                         const float* __restrict__ Dv, bf16_t* __restrict__ OUT0, bf16_t* __restrict__ OUT1,
                                                         int H, int S, int Spad, float scale, float scale_log2, int nbatch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NTILE = MODE == 2 ? 1 : MODE == 0 ? 3 : 4;  // tiles per stage
  constexpr int STAGE = NTILE * TILE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  Geo G;
  G.hi = lane >> 5; G.li = lane & 31;
  {
    const int kvm = (G.li & 0x13) | ((G.li & 4) << 1) | ((G.li & 8) >> 1);
    G.k_row_off = kvm * 256; G.k_swz = kvm & 15; G.v_row_off = G.li * 128; G.v_swz = (G.li >> 1) & 7;
  }
  const int nblk = Spad / 128;
  int bid = blockIdx.x;
  {
    const int T = gridDim.x, q = T >> 3, r = T & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int blk = bid % nblk, h = (bid / nblk) % H, b = bid / (nblk * H);
  const long long bh = (long long)b * H + h;
  const long long hoff = bh * Spad * 128;
  const int r0 = blk * 128 + wave * 32 + G.li;   // this lane's persistent row (query in MODE 0 / 2, key in MODE 1)

  bf16x8_t pa[8], pb[8];
#pragma unroll
  for (int ds = 0; ds < 8; ++ds) {
    pa[ds] = *(const bf16x8_t*)(PA + hoff + (long long)r0 * 128 + ds * 16 + G.hi * 8);
    if (MODE != 2) pb[ds] = *(const bf16x8_t*)(PB + hoff + (long long)r0 * 128 + ds * 16 + G.hi * 8);
  }
  float myL = 0.f, myD = 0.f;
  if (MODE == 0) { myL = L2[bh * Spad + r0]; myD = Dv[bh * Spad + r0]; }

  // DMA source offsets: 1024 chunks per tile, 4 per thread
  int row_src[4], col_src[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = j * 256 + tid;
    { const int row = p >> 4, c = p & 15; row_src[j] = row * 128 + ((c ^ (row & 15)) << 3); }
    { const int row = p >> 3, c = p & 7; col_src[j] = row * Spad + ((c ^ ((row >> 1) & 7)) << 3); }
  }
  Rsrc4 R;
  {
    const uint32_t bytes = (uint32_t)Spad * 256u;
    R.a = __builtin_amdgcn_make_buffer_rsrc((void*)(TA + hoff), 0, bytes, 0x00020000);
    R.b = __builtin_amdgcn_make_buffer_rsrc((void*)((MODE != 2 ? TB : TA) + hoff), 0, bytes, 0x00020000);
    R.c = __builtin_amdgcn_make_buffer_rsrc((void*)((MODE != 2 ? TC : TA) + hoff), 0, bytes, 0x00020000);
    R.d = __builtin_amdgcn_make_buffer_rsrc((void*)((MODE == 1 ? TD : TA) + hoff), 0, bytes, 0x00020000);
  }
  f32x16_t oacc0[4], oacc1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc0[i][r] = 0.f; oacc1[i][r] = 0.f; }
  float m_run = -BIG, l_run = 0.f;
  const float* L2h = L2 + bh * Spad;
  const float* Dh = Dv + bh * Spad;

  const int nt = (S + KVB - 1) / KVB;
  {  // prologue: tile 0 (bwd_tile with nothing to compute would need a third form)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      char* dst = smem + (j * 256 + wave * 64) * 16;
      dma16(R.a, (uint32_t)(row_src[j] * 2), dst);
      if (MODE != 2) {
        dma16(R.b, (uint32_t)(row_src[j] * 2), dst + TILE);
        dma16(R.c, (uint32_t)(col_src[j] * 2), dst + 2 * TILE);
        if (MODE == 1) dma16(R.d, (uint32_t)(col_src[j] * 2), dst + 3 * TILE);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    bwd_tile<MODE>(smem + buf * STAGE, smem + (buf ^ 1) * STAGE, t + 1 < nt, (t + 1) * KVB, t * KVB, R, row_src, col_src, wave, G, pa, pb, myL, myD,
                   L2h, Dh, S, scale_log2, oacc0, oacc1, m_run, l_run);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (MODE == 2) {
    l_run = xhalf_sum(l_run);
    // (both half-waves hold the same row: identical values, either writes)
    if (G.hi == 0) L2[bh * Spad + r0] = r0 < S ? m_run + __log2f(l_run) : BIG;
    return;
  }
  const bool live = r0 < S;
  if (MODE == 0) {
    store_rows(OUT0 + hoff + (long long)r0 * 128, oacc0, scale, G.hi, live);
  } else {
    store_rows(OUT0 + hoff + (long long)r0 * 128, oacc0, 1.f, G.hi, live);
    store_rows(OUT1 + hoff + (long long)r0 * 128, oacc1, scale, G.hi, live);
  }
}

#ifdef X2I_ABLATION   // (measurement library only since the pipelined passes: the A/B reference of attn_bwd_dq_kernel)
// dQ with 64 persistent query rows per wave (two 32-row blocks qb): every streamed fragment feeds TWO MFMAs, so the LDS traffic per
// MFMA is half that of attn_bwd_kernel<0> (whose one-fragment-per-MFMA stream sits at the LDS roof with the matrix pipe half idle).
// 256 query rows per workgroup; same tiles, fragments, arithmetic and rounding per row as the 32-row form (bit-identical results).
__device__ __forceinline__ void dq64_tile(const char* __restrict__ cur, char* __restrict__ nxt, bool issue, int s_next, int s0, const Rsrc4& R,
                                          const int (&row_src)[4], const int (&col_src)[4], int wave, const Geo& G, const bf16x8_t (&pa)[2][8],
                                          const bf16x8_t (&pb)[2][8], const float (&myL)[2], const float (&myD)[2], int S, float scale_log2,
                                          f32x16_t (&oacc)[2][4]) {
  if (issue) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      char* dst = nxt + (j * 256 + wave * 64) * 16;
      dma16(R.a, (uint32_t)(s_next * 256 + row_src[j] * 2), dst);
      dma16(R.b, (uint32_t)(s_next * 256 + row_src[j] * 2), dst + TILE);
      dma16(R.c, (uint32_t)(s_next * 2 + col_src[j] * 2), dst + 2 * TILE);
    }
  }
  f32x16_t sacc[2][2], dacc[2][2];   // [qb][sub-tile u]
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[qb][u][r] = 0.f; dacc[qb][u][r] = 0.f; }
  {  // scores S^T = K Q^T and dP^T = V dO^T as one fragment pipeline (steps 0-3: K tile, 4-7: V tile)
    bf16x8_t fr[2][4];
#pragma unroll
    for (int f = 0; f < 4; ++f) fr[0][f] = row_frag(cur, G, 0, f);
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      if (st + 1 < 8) {
#pragma unroll
        for (int f = 0; f < 4; ++f) fr[(st + 1) & 1][f] = row_frag(st + 1 < 4 ? cur : cur + TILE, G, (st + 1) & 3, f);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int g = st & 3;
#pragma unroll
      for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          if (st < 4) sacc[qb][f & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[st & 1][f], pa[qb][2 * g + (f >> 1)], sacc[qb][f & 1], 0, 0, 0);
          else dacc[qb][f & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[st & 1][f], pb[qb][2 * g + (f >> 1)], dacc[qb][f & 1], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  bf16x8_t fr[2][4];   // first fragments of the accumulation pipeline: in flight while the element-wise section runs
#pragma unroll
  for (int db = 0; db < 4; ++db) fr[0][db] = col_frag(cur + 2 * TILE, G, 0, db);
  bf16x8_t dsf[2][2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = s0 + u * 32 + 16 * (r >> 3) + 8 * G.hi + (r & 7);
        const float p = key < S ? __builtin_amdgcn_exp2f(sacc[qb][u][r] * scale_log2 - myL[qb]) : 0.f;
        dacc[qb][u][r] = p * (dacc[qb][u][r] - myD[qb]);
      }
    pack_frags(dacc[qb], dsf[qb]);
  }
#pragma unroll
  for (int st = 0; st < 4; ++st) {   // dQ^T += K^T dS^T
    if (st + 1 < 4) {
#pragma unroll
      for (int db = 0; db < 4; ++db) fr[(st + 1) & 1][db] = col_frag(cur + 2 * TILE, G, st + 1, db);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
        oacc[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[st & 1][db], dsf[qb][st >> 1][st & 1], oacc[qb][db], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__global__ __launch_bounds__(256, 1) void attn_bwd_dq64_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ dO, const bf16_t* __restrict__ K,
                                                              const bf16_t* __restrict__ V, const bf16_t* __restrict__ KT, const float* __restrict__ L2,
                                                              const float* __restrict__ Dv, bf16_t* __restrict__ dQ, int H, int S, int Spad,
                                                              float scale, float scale_log2, int nbatch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int STAGE = 3 * TILE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  Geo G;
  G.hi = lane >> 5; G.li = lane & 31;
  {
    const int kvm = (G.li & 0x13) | ((G.li & 4) << 1) | ((G.li & 8) >> 1);
    G.k_row_off = kvm * 256; G.k_swz = kvm & 15; G.v_row_off = G.li * 128; G.v_swz = (G.li >> 1) & 7;
  }
  const int nblk = (Spad + 255) / 256;
  int bid = blockIdx.x;
  {
    const int T = gridDim.x, q = T >> 3, r = T & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int blk = bid % nblk, h = (bid / nblk) % H, b = bid / (nblk * H);
  const long long bh = (long long)b * H + h;
  const long long hoff = bh * Spad * 128;
  bf16x8_t pa[2][8], pb[2][8];
  float myL[2], myD[2];
  int r0[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    r0[qb] = blk * 256 + wave * 64 + qb * 32 + G.li;   // this lane's persistent query rows
    const int rr = min(r0[qb], Spad - 1);              // (rows behind Spad -- a last, half block -- read a valid row and are never stored)
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) {
      pa[qb][ds] = *(const bf16x8_t*)(Q + hoff + (long long)rr * 128 + ds * 16 + G.hi * 8);
      pb[qb][ds] = *(const bf16x8_t*)(dO + hoff + (long long)rr * 128 + ds * 16 + G.hi * 8);
    }
    myL[qb] = L2[bh * Spad + rr];
    myD[qb] = Dv[bh * Spad + rr];
  }
  int row_src[4], col_src[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = j * 256 + tid;
    { const int row = p >> 4, c = p & 15; row_src[j] = row * 128 + ((c ^ (row & 15)) << 3); }
    { const int row = p >> 3, c = p & 7; col_src[j] = row * Spad + ((c ^ ((row >> 1) & 7)) << 3); }
  }
  Rsrc4 R;
  {
    const uint32_t bytes = (uint32_t)Spad * 256u;
    R.a = __builtin_amdgcn_make_buffer_rsrc((void*)(K + hoff), 0, bytes, 0x00020000);
    R.b = __builtin_amdgcn_make_buffer_rsrc((void*)(V + hoff), 0, bytes, 0x00020000);
    R.c = __builtin_amdgcn_make_buffer_rsrc((void*)(KT + hoff), 0, bytes, 0x00020000);
    R.d = R.c;
  }
  f32x16_t oacc[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[qb][i][r] = 0.f;
  const int nt = (S + KVB - 1) / KVB;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    char* dst = smem + (j * 256 + wave * 64) * 16;
    dma16(R.a, (uint32_t)(row_src[j] * 2), dst);
    dma16(R.b, (uint32_t)(row_src[j] * 2), dst + TILE);
    dma16(R.c, (uint32_t)(col_src[j] * 2), dst + 2 * TILE);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    dq64_tile(smem + buf * STAGE, smem + (buf ^ 1) * STAGE, t + 1 < nt, (t + 1) * KVB, t * KVB, R, row_src, col_src, wave, G, pa, pb, myL, myD, S,
              scale_log2, oacc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) store_rows(dQ + hoff + (long long)r0[qb] * 128, oacc[qb], scale, G.hi, r0[qb] < S);
}

#endif

// ---------------------------------------------------------------------------------------------------------------------------------------
// dK / dV pass, SOFTWARE-PIPELINED over the two 32-row halves u of a streamed tile (round 6).  attn_bwd_kernel<1> runs a tile as
// [S, dP: 32 MFMAs] [element-wise section: ~330 VALU instructions with the matrix pipe idle] [dV, dK: 32 MFMAs]; with one wave per SIMD nothing
// else fills the middle, the per-row statistics come as 16 global loads per tile that sit BEHIND the next tile's DMA in the vmcnt queue (so
// their wait is a wait for the whole DMA), and the 16 DMA pieces issue in one burst.  Here a tile is 64 slots, one MFMA each, in four phases
//     A: S(u0), dP(u0)                    + the next tile's 17 DMA pieces, one per slot
//     B: S(u1), dP(u1)                    + the element-wise values of half u0, one per slot
//     C: dV(u0), dK(u0)                   + the element-wise values of half u1
//     D: dV(u1), dK(u1)
// with every slot closed by a scheduling fence, so the order written here is the order issued: the fragment of slot m + LEAD is requested in
// slot m (a ring of eight), L2 / D rows of the tile come through LDS with the tile (one 256-byte DMA piece each, read back as broadcasts).
// Same MFMA order per accumulator and the same element-wise operations as attn_bwd_kernel<1>: bit-identical results (tested).
template <int N, class F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (N > 0) {
    sfor<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}
// ... with the tile's byte offset as the instruction's scalar offset: the per-lane offset stays loop-invariant (no VALU per piece and tile)
__device__ __forceinline__ void dma16s(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff_bytes, uint32_t soff_bytes, char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff_bytes, soff_bytes, 0, 0);
}
__device__ __forceinline__ void dma4(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff_bytes, char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 4, voff_bytes, 0, 0, 0);
}

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
// MFMAs of the pipelined passes as asm statements with the register FILE of every operand fixed: the persistent operands (K / V or Q / dO rows)
// and the output accumulators in the accumulator file, the scores in VGPRs (the element-wise section reads them without v_accvgpr_read).  Left to
// itself hipcc keeps three of the eight output accumulators in VGPRs and copies them into the accumulator file and back around their MFMAs
// (48 + 48 moves per tile).  The compiler's hazard recogniser does not see into these statements: the slot order keeps every VALU read of a score
// two MFMAs behind the MFMA that wrote it (>= 16 quad-cycles; 11 required), and an accumulator's consecutive MFMAs are exact SrcC = vDst chains.
__device__ __forceinline__ void mfma_s0(f32x16_t& acc, const bf16x8_t& a, const bf16x8_t& b) {   // scores, first MFMA: C = 0
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma_s(f32x16_t& acc, const bf16x8_t& a, const bf16x8_t& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma_o(f32x16_t& acc, const bf16x8_t& a, const u32x4_t& b) {     // output accumulators
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

constexpr int KV_STAGE = 4 * TILE + 1024;  // Q rows | dO rows | dO^T | Q^T | L2[64] | D[64] | (waves 2, 3: the same rows again -- no branch in the loop)
constexpr int KV_LEAD = 4;

struct KvState {
  f32x16_t sacc[2], dacc[2];
  bf16x8_t fr[8];
  u32x4_t pf[2][2], dsf[2][2];          // P^T / dS^T fragments [u][kt] as packed bf16 pairs
  f32x4_t Lr[2][4], Dr[2][4];           // statistics of the lane's 16 rows per half: [u][a * 2 + half of eight]
};

// Fragment addresses of a tile: the per-lane part (row, swizzled chunk) of the eight row-fragment and four column-fragment positions is the same for every
// tile (KvOfs, once per kernel); a tile adds its buffer base (twelve adds) and every read carries the rest -- tile, half, d-block -- as an immediate offset.
// (Left to itself hipcc re-derived 40 addresses per tile.)
struct KvOfs { int r[8], c[4]; };
struct KvAddr { const char* r[8]; const char* c[4]; };
__device__ __forceinline__ KvOfs kv_ofs(const Geo& G) {
  KvOfs o;
#pragma unroll
  for (int ds = 0; ds < 8; ++ds) o.r[ds] = G.k_row_off + (((ds * 2 + G.hi) ^ G.k_swz) << 4);
#pragma unroll
  for (int g = 0; g < 4; ++g) o.c[g] = G.v_row_off + (((4 * (g >> 1) + 2 * (g & 1) + G.hi) ^ G.v_swz) << 4);
  return o;
}
__device__ __forceinline__ KvAddr kv_addr(const char* cur, const KvOfs& o) {
  KvAddr a;
#pragma unroll
  for (int ds = 0; ds < 8; ++ds) a.r[ds] = cur + o.r[ds];
#pragma unroll
  for (int g = 0; g < 4; ++g) a.c[g] = cur + o.c[g];
  return a;
}

template <int M>
__device__ __forceinline__ bf16x8_t kv_frag(const KvAddr& A) {
  constexpr int ph = M >> 4, i = M & 15, u = ph & 1;
  if constexpr (ph < 2) {
    constexpr int ds = i >> 1, which = i & 1;
    return *(const bf16x8_t*)(A.r[ds] + which * TILE + u * 32 * 256);
  } else {
    constexpr int kt = i >> 3, db = (i >> 1) & 3, which = i & 1;
    return *(const bf16x8_t*)(A.c[u * 2 + kt] + (2 + which) * TILE + db * 32 * 128);
  }
}

template <int U, int R>
__device__ __forceinline__ void kv_ew(KvState& st, float scale_log2) {
  const float L = st.Lr[U][R >> 2][R & 3], D = st.Dr[U][R >> 2][R & 3];
  const float p = __builtin_amdgcn_exp2f(st.sacc[U][R] * scale_log2 - L);
  const float d = p * (st.dacc[U][R] - D);
  st.sacc[U][R] = p;   // (the score registers are free: the values wait there for their pair partner)
  st.dacc[U][R] = d;
  if constexpr ((R & 1) == 1) {
    st.pf[U][R >> 3][(R & 7) >> 1] = pack_bf16x2(st.sacc[U][R - 1], p);
    st.dsf[U][R >> 3][(R & 7) >> 1] = pack_bf16x2(st.dacc[U][R - 1], d);
  }
}

constexpr int kv_u0_value(int m) {
  for (int r = 0; r < 16; ++r)
    if (18 + r * 11 / 8 == m) return r;
  return -1;
}

template <int M>
__device__ __forceinline__ void kv_slot(const char* __restrict__ cur, char* __restrict__ nxt, int s_next, const Rsrc4& R,
                                        __amdgpu_buffer_rsrc_t rL, __amdgpu_buffer_rsrc_t rD, const int (&row_src)[4], const int (&col_src)[4],
                                        int wave, int lane, const Geo& G, const bf16x8_t (&pa)[8], const bf16x8_t (&pb)[8], float scale_log2,
                                        f32x16_t (&oacc0)[4], f32x16_t (&oacc1)[4], KvState& st, const KvAddr& A) {
  constexpr int ph = M >> 4, i = M & 15, u = ph & 1;
  if constexpr (M + KV_LEAD < 64) st.fr[(M + KV_LEAD) & 7] = kv_frag<M + KV_LEAD>(A);
  if constexpr (ph == 0) {   // the next tile: pieces j = i >> 2 of tile i & 3 (unconditional: behind the last tile they are never read)
    constexpr int j = i >> 2, tl = i & 3;
    char* dst = nxt + tl * TILE + (j * 256 + wave * 64) * 16;
    if constexpr (tl == 0) dma16s(R.a, (uint32_t)(row_src[j] * 2), (uint32_t)(s_next * 256), dst);
    if constexpr (tl == 1) dma16s(R.b, (uint32_t)(row_src[j] * 2), (uint32_t)(s_next * 256), dst);
    if constexpr (tl == 2) dma16s(R.c, (uint32_t)(col_src[j] * 2), (uint32_t)(s_next * 2), dst);
    if constexpr (tl == 3) dma16s(R.d, (uint32_t)(col_src[j] * 2), (uint32_t)(s_next * 2), dst);
    if constexpr (i == 15) dma4((wave & 1) ? rD : rL, (uint32_t)((s_next + lane) * 4), nxt + 4 * TILE + wave * 256);
    // statistics of half u0 (needed from slot 18 on), half u1 in phase B
    if constexpr (i >= 8) {
      constexpr int q = i - 8, a = (q >> 1) & 1, h8 = q & 1;
      const char* base = cur + 4 * TILE + (q >> 2) * 256 + (16 * a + 8 * G.hi) * 4 + h8 * 16;
      if constexpr (q < 4) st.Lr[0][a * 2 + h8] = *(const f32x4_t*)base;
      else st.Dr[0][a * 2 + h8] = *(const f32x4_t*)base;
    }
  }
  if constexpr (ph == 1 && i >= 8) {
    constexpr int q = i - 8, a = (q >> 1) & 1, h8 = q & 1;
    const char* base = cur + 4 * TILE + (q >> 2) * 256 + (32 + 16 * a + 8 * G.hi) * 4 + h8 * 16;
    if constexpr (q < 4) st.Lr[1][a * 2 + h8] = *(const f32x4_t*)base;
    else st.Dr[1][a * 2 + h8] = *(const f32x4_t*)base;
  }
  if constexpr (ph < 2) {
    constexpr int ds = i >> 1, which = i & 1;
    if constexpr (which == 0) { if constexpr (ds == 0) mfma_s0(st.sacc[u], st.fr[M & 7], pa[ds]); else mfma_s(st.sacc[u], st.fr[M & 7], pa[ds]); }
    else { if constexpr (ds == 0) mfma_s0(st.dacc[u], st.fr[M & 7], pb[ds]); else mfma_s(st.dacc[u], st.fr[M & 7], pb[ds]); }
  } else {
    constexpr int kt = i >> 3, db = (i >> 1) & 3, which = i & 1;
    if constexpr (which == 0) mfma_o(oacc0[db], st.fr[M & 7], st.pf[u][kt]);
    else mfma_o(oacc1[db], st.fr[M & 7], st.dsf[u][kt]);
  }
  // element-wise values (from two slots behind the MFMAs that finish their scores; a half's rows r < 8 before its kt = 0 MFMAs, r >= 8 before its
  // kt = 1 MFMAs): half u0 spread over slots 18 .. 39 (value r in slot 18 + 11 r / 8), half u1 one per slot in 40 .. 55
  if constexpr (kv_u0_value(M) >= 0) kv_ew<0, (kv_u0_value(M) >= 0 ? kv_u0_value(M) : 0)>(st, scale_log2);
  if constexpr (M >= 40 && M < 56) kv_ew<1, M - 40>(st, scale_log2);
  __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void dkdv_tile(const char* __restrict__ cur, char* __restrict__ nxt, int s_next, const Rsrc4& R, __amdgpu_buffer_rsrc_t rL,
                                          __amdgpu_buffer_rsrc_t rD, const int (&row_src)[4], const int (&col_src)[4], int wave, int lane,
                                          const Geo& G, const bf16x8_t (&pa)[8], const bf16x8_t (&pb)[8], float scale_log2,
                                          f32x16_t (&oacc0)[4], f32x16_t (&oacc1)[4], const KvOfs& ofs) {
  KvState st;
  const KvAddr A = kv_addr(cur, ofs);
  sfor<KV_LEAD>([&](auto mc) { st.fr[decltype(mc)::value & 7] = kv_frag<decltype(mc)::value>(A); });
  __builtin_amdgcn_sched_barrier(0);
  sfor<64>([&](auto mc) {
    kv_slot<decltype(mc)::value>(cur, nxt, s_next, R, rL, rD, row_src, col_src, wave, lane, G, pa, pb, scale_log2, oacc0, oacc1, st, A);
  });
}

// (bid0 of T: the block's place among the pass's blocks -- blockIdx.x of gridDim.x in a launch of its own, an offset place in the fused launch)
__device__ __forceinline__ void dkdv_body(char* smem, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V, const bf16_t* __restrict__ Q,
                                          const bf16_t* __restrict__ dO, const bf16_t* __restrict__ dOT, const bf16_t* __restrict__ QT,
                                          const float* __restrict__ L2, const float* __restrict__ Dv, bf16_t* __restrict__ dV, bf16_t* __restrict__ dK,
                                          int H, int S, int Spad, float scale, float scale_log2, int bid0, int T) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  Geo G;
  G.hi = lane >> 5; G.li = lane & 31;
  {
    const int kvm = (G.li & 0x13) | ((G.li & 4) << 1) | ((G.li & 8) >> 1);
    G.k_row_off = kvm * 256; G.k_swz = kvm & 15; G.v_row_off = G.li * 128; G.v_swz = (G.li >> 1) & 7;
  }
  const int nblk = Spad / 128;
  int bid = bid0;
  {
    const int q = T >> 3, r = T & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int blk = bid % nblk, h = (bid / nblk) % H, b = bid / (nblk * H);
  const long long bh = (long long)b * H + h;
  const long long hoff = bh * Spad * 128;
  const int r0 = blk * 128 + wave * 32 + G.li;   // this lane's persistent key
  bf16x8_t pa[8], pb[8];
#pragma unroll
  for (int ds = 0; ds < 8; ++ds) {
    pa[ds] = *(const bf16x8_t*)(K + hoff + (long long)r0 * 128 + ds * 16 + G.hi * 8);
    pb[ds] = *(const bf16x8_t*)(V + hoff + (long long)r0 * 128 + ds * 16 + G.hi * 8);
  }
  int row_src[4], col_src[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = j * 256 + tid;
    { const int row = p >> 4, c = p & 15; row_src[j] = row * 128 + ((c ^ (row & 15)) << 3); }
    { const int row = p >> 3, c = p & 7; col_src[j] = row * Spad + ((c ^ ((row >> 1) & 7)) << 3); }
  }
  Rsrc4 R;
  const uint32_t bytes = (uint32_t)Spad * 256u;
  R.a = __builtin_amdgcn_make_buffer_rsrc((void*)(Q + hoff), 0, bytes, 0x00020000);
  R.b = __builtin_amdgcn_make_buffer_rsrc((void*)(dO + hoff), 0, bytes, 0x00020000);
  R.c = __builtin_amdgcn_make_buffer_rsrc((void*)(dOT + hoff), 0, bytes, 0x00020000);
  R.d = __builtin_amdgcn_make_buffer_rsrc((void*)(QT + hoff), 0, bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rL = __builtin_amdgcn_make_buffer_rsrc((void*)(L2 + bh * Spad), 0, (uint32_t)Spad * 4u, 0x00020000);
  __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc((void*)(Dv + bh * Spad), 0, (uint32_t)Spad * 4u, 0x00020000);
  f32x16_t oacc0[4], oacc1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { oacc0[i][r] = 0.f; oacc1[i][r] = 0.f; }
  const int nt = (S + KVB - 1) / KVB;
  const KvOfs ofs = kv_ofs(G);
#pragma unroll
  for (int j = 0; j < 4; ++j) {   // tile 0
    char* dst = smem + (j * 256 + wave * 64) * 16;
    dma16(R.a, (uint32_t)(row_src[j] * 2), dst);
    dma16(R.b, (uint32_t)(row_src[j] * 2), dst + TILE);
    dma16(R.c, (uint32_t)(col_src[j] * 2), dst + 2 * TILE);
    dma16(R.d, (uint32_t)(col_src[j] * 2), dst + 3 * TILE);
  }
  dma4((wave & 1) ? rD : rL, (uint32_t)(lane * 4), smem + 4 * TILE + wave * 256);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    dkdv_tile(smem + buf * KV_STAGE, smem + (buf ^ 1) * KV_STAGE, (t + 1) * KVB, R, rL, rD, row_src, col_src, wave, lane, G, pa, pb, scale_log2, oacc0,
              oacc1, ofs);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  const bool live = r0 < S;
  store_rows(dV + hoff + (long long)r0 * 128, oacc0, 1.f, G.hi, live);
  store_rows(dK + hoff + (long long)r0 * 128, oacc1, scale, G.hi, live);
}

__global__ __launch_bounds__(256, 1) void attn_bwd_dkdv_kernel(const bf16_t* __restrict__ K, const bf16_t* __restrict__ V, const bf16_t* __restrict__ Q,
                                                              const bf16_t* __restrict__ dO, const bf16_t* __restrict__ dOT,
                                                              const bf16_t* __restrict__ QT, const float* __restrict__ L2, const float* __restrict__ Dv,
                                                              bf16_t* __restrict__ dV, bf16_t* __restrict__ dK, int H, int S, int Spad, float scale,
                                                              float scale_log2, int nbatch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  dkdv_body(smem, K, V, Q, dO, dOT, QT, L2, Dv, dV, dK, H, S, Spad, scale, scale_log2, (int)blockIdx.x, (int)gridDim.x);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// dQ pass, software-pipelined the same way (64 persistent query rows per wave = two 32-row blocks qb, so a streamed fragment feeds two
// MFMAs): a tile is 96 one-MFMA slots
//     A (0..31):  S(u0), dP(u0) for qb 0, 1         + the next tile's 12 DMA pieces
//     B (32..63): S(u1), dP(u1)                     + the 32 element-wise values of half u0 (slots 34..65)
//     C (64..95): dQ^T += K^T dS^T, groups (u0,kt0) (u0,kt1) (u1,kt0) (u1,kt1)   + the values of half u1: rows r < 8 of both blocks by slot 79,
//                                                     rows r >= 8 by slot 87 (two values per slot there: the groups that need them follow)
// Same MFMA order per accumulator and element-wise operations as attn_bwd_dq64_kernel: bit-identical (tested).  MASKED: the last tile of a
// sequence with S % 64 != 0 (keys at or behind S get p = 0); every other tile runs without the compare / select per value.
constexpr int DQ_LEAD = 3;   // fragments (= pairs of slots) a read runs ahead of its MFMAs
struct DqState {
  f32x16_t sacc[2][2], dacc[2][2];   // [qb][u]
  bf16x8_t fr[8];
  u32x4_t dsf[2][2][2];              // dS^T fragments [qb][u][kt]
};

template <int F>   // fragment F (0..47) of a tile: one per two slots
__device__ __forceinline__ bf16x8_t dq_frag(const char* cur, const Geo& G) {
  if constexpr (F < 32) {
    constexpr int u = F >> 4, i = F & 15, ds = i >> 1, which = i & 1;   // which: 0 = K rows (S), 1 = V rows (dP)
    return row_frag(cur + which * TILE, G, ds >> 1, u + 2 * (ds & 1));
  } else {
    constexpr int i = F - 32, g = i >> 2, db = i & 3;
    return col_frag(cur + 2 * TILE, G, g, db);
  }
}

template <int U, int J, bool MASKED>   // value J (0..31) of half U: rows r < 8 of block 0, of block 1, then rows r >= 8 of block 0, of block 1
__device__ __forceinline__ void dq_ew(DqState& st, float scale_log2, const float (&myL)[2], const float (&myD)[2], int kbase, int S) {
  constexpr int h = J >> 4, qb = (J >> 3) & 1, R = h * 8 + (J & 7);
  float p = __builtin_amdgcn_exp2f(st.sacc[qb][U][R] * scale_log2 - myL[qb]);
  if constexpr (MASKED) {
    const int key = kbase + U * 32 + 16 * (R >> 3) + (R & 7);   // kbase = s0 + 8 hi
    p = key < S ? p : 0.f;
  }
  const float d = p * (st.dacc[qb][U][R] - myD[qb]);
  st.dacc[qb][U][R] = d;
  if constexpr ((R & 1) == 1) st.dsf[qb][U][R >> 3][(R & 7) >> 1] = pack_bf16x2(st.dacc[qb][U][R - 1], d);
}

template <int M, bool MASKED>
__device__ __forceinline__ void dq_slot(const char* __restrict__ cur, char* __restrict__ nxt, int s_next, const Rsrc4& R, const int (&row_src)[4],
                                        const int (&col_src)[4], int wave, const Geo& G, const bf16x8_t (&pa)[2][8], const bf16x8_t (&pb)[2][8],
                                        const float (&myL)[2], const float (&myD)[2], int kbase, int S, float scale_log2, f32x16_t (&oacc)[2][4],
                                        DqState& st) {
  constexpr int F = M >> 1, qb = M & 1;
  if constexpr ((M & 1) == 0 && F + DQ_LEAD < 48) st.fr[(F + DQ_LEAD) & 7] = dq_frag<F + DQ_LEAD>(cur, G);
  if constexpr (M < 24 && (M & 1) == 0) {   // the next tile: piece j = (M / 2) >> 2 ... of tile (M / 2) % 3
    constexpr int pc = M >> 1, j = pc / 3, tl = pc % 3;
    char* dst = nxt + tl * TILE + (j * 256 + wave * 64) * 16;
    if constexpr (tl == 0) dma16s(R.a, (uint32_t)(row_src[j] * 2), (uint32_t)(s_next * 256), dst);
    if constexpr (tl == 1) dma16s(R.b, (uint32_t)(row_src[j] * 2), (uint32_t)(s_next * 256), dst);
    if constexpr (tl == 2) dma16s(R.c, (uint32_t)(col_src[j] * 2), (uint32_t)(s_next * 2), dst);
  }
  if constexpr (M < 64) {
    constexpr int u = F >> 4, i = F & 15, ds = i >> 1, which = i & 1;
    if constexpr (which == 0) { if constexpr (ds == 0) mfma_s0(st.sacc[qb][u], st.fr[F & 7], pa[qb][ds]); else mfma_s(st.sacc[qb][u], st.fr[F & 7], pa[qb][ds]); }
    else { if constexpr (ds == 0) mfma_s0(st.dacc[qb][u], st.fr[F & 7], pb[qb][ds]); else mfma_s(st.dacc[qb][u], st.fr[F & 7], pb[qb][ds]); }
  } else {
    constexpr int i = F - 32, g = i >> 2, db = i & 3;
    mfma_o(oacc[qb][db], st.fr[F & 7], st.dsf[qb][g >> 1][g & 1]);
  }
  if constexpr (M >= 34 && M < 66) dq_ew<0, M - 34, MASKED>(st, scale_log2, myL, myD, kbase, S);
  // half u1: values 0..15 in slots 66..79 (the first two slots carry two), values 16..31 two per slot in slots 80..87
  if constexpr (M == 66 || M == 67) {
    dq_ew<1, 2 * (M - 66), MASKED>(st, scale_log2, myL, myD, kbase, S);
    dq_ew<1, 2 * (M - 66) + 1, MASKED>(st, scale_log2, myL, myD, kbase, S);
  }
  if constexpr (M >= 68 && M < 80) dq_ew<1, M - 64, MASKED>(st, scale_log2, myL, myD, kbase, S);
  if constexpr (M >= 80 && M < 88) {
    dq_ew<1, 16 + 2 * (M - 80), MASKED>(st, scale_log2, myL, myD, kbase, S);
    dq_ew<1, 17 + 2 * (M - 80), MASKED>(st, scale_log2, myL, myD, kbase, S);
  }
  __builtin_amdgcn_sched_barrier(0);
}

template <bool MASKED>
__device__ __forceinline__ void dq_tile(const char* __restrict__ cur, char* __restrict__ nxt, int s_next, int s0, const Rsrc4& R, const int (&row_src)[4],
                                        const int (&col_src)[4], int wave, const Geo& G, const bf16x8_t (&pa)[2][8], const bf16x8_t (&pb)[2][8],
                                        const float (&myL)[2], const float (&myD)[2], int S, float scale_log2, f32x16_t (&oacc)[2][4]) {
  DqState st;
  sfor<DQ_LEAD>([&](auto fc) { st.fr[decltype(fc)::value & 7] = dq_frag<decltype(fc)::value>(cur, G); });
  __builtin_amdgcn_sched_barrier(0);
  const int kbase = s0 + 8 * G.hi;
  sfor<96>([&](auto mc) {
    dq_slot<decltype(mc)::value, MASKED>(cur, nxt, s_next, R, row_src, col_src, wave, G, pa, pb, myL, myD, kbase, S, scale_log2, oacc, st);
  });
}

__device__ __forceinline__ void dq_body(char* smem, const bf16_t* __restrict__ Q, const bf16_t* __restrict__ dO, const bf16_t* __restrict__ K,
                                        const bf16_t* __restrict__ V, const bf16_t* __restrict__ KT, const float* __restrict__ L2,
                                        const float* __restrict__ Dv, bf16_t* __restrict__ dQ, int H, int S, int Spad, float scale, float scale_log2,
                                        int bid0, int T) {
  constexpr int STAGE = 3 * TILE;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  Geo G;
  G.hi = lane >> 5; G.li = lane & 31;
  {
    const int kvm = (G.li & 0x13) | ((G.li & 4) << 1) | ((G.li & 8) >> 1);
    G.k_row_off = kvm * 256; G.k_swz = kvm & 15; G.v_row_off = G.li * 128; G.v_swz = (G.li >> 1) & 7;
  }
  const int nblk = (Spad + 255) / 256;
  int bid = bid0;
  {
    const int q = T >> 3, r = T & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int blk = bid % nblk, h = (bid / nblk) % H, b = bid / (nblk * H);
  const long long bh = (long long)b * H + h;
  const long long hoff = bh * Spad * 128;
  bf16x8_t pa[2][8], pb[2][8];
  float myL[2], myD[2];
  int r0[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    r0[qb] = blk * 256 + wave * 64 + qb * 32 + G.li;
    const int rr = min(r0[qb], Spad - 1);
#pragma unroll
    for (int ds = 0; ds < 8; ++ds) {
      pa[qb][ds] = *(const bf16x8_t*)(Q + hoff + (long long)rr * 128 + ds * 16 + G.hi * 8);
      pb[qb][ds] = *(const bf16x8_t*)(dO + hoff + (long long)rr * 128 + ds * 16 + G.hi * 8);
    }
    myL[qb] = L2[bh * Spad + rr];
    myD[qb] = Dv[bh * Spad + rr];
  }
  int row_src[4], col_src[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = j * 256 + tid;
    { const int row = p >> 4, c = p & 15; row_src[j] = row * 128 + ((c ^ (row & 15)) << 3); }
    { const int row = p >> 3, c = p & 7; col_src[j] = row * Spad + ((c ^ ((row >> 1) & 7)) << 3); }
  }
  Rsrc4 R;
  {
    const uint32_t bytes = (uint32_t)Spad * 256u;
    R.a = __builtin_amdgcn_make_buffer_rsrc((void*)(K + hoff), 0, bytes, 0x00020000);
    R.b = __builtin_amdgcn_make_buffer_rsrc((void*)(V + hoff), 0, bytes, 0x00020000);
    R.c = __builtin_amdgcn_make_buffer_rsrc((void*)(KT + hoff), 0, bytes, 0x00020000);
    R.d = R.c;
  }
  f32x16_t oacc[2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[qb][i][r] = 0.f;
  const int nt = (S + KVB - 1) / KVB;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    char* dst = smem + (j * 256 + wave * 64) * 16;
    dma16(R.a, (uint32_t)(row_src[j] * 2), dst);
    dma16(R.b, (uint32_t)(row_src[j] * 2), dst + TILE);
    dma16(R.c, (uint32_t)(col_src[j] * 2), dst + 2 * TILE);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int nfull = (S & 63) ? nt - 1 : nt;
  for (int t = 0; t < nfull; ++t) {
    const int buf = t & 1;
    dq_tile<false>(smem + buf * STAGE, smem + (buf ^ 1) * STAGE, (t + 1) * KVB, t * KVB, R, row_src, col_src, wave, G, pa, pb, myL, myD, S, scale_log2, oacc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (nfull < nt) {
    const int t = nt - 1, buf = t & 1;
    dq_tile<true>(smem + buf * STAGE, smem + (buf ^ 1) * STAGE, (t + 1) * KVB, t * KVB, R, row_src, col_src, wave, G, pa, pb, myL, myD, S, scale_log2, oacc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) store_rows(dQ + hoff + (long long)r0[qb] * 128, oacc[qb], scale, G.hi, r0[qb] < S);
}

__global__ __launch_bounds__(256, 1) void attn_bwd_dq_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ dO, const bf16_t* __restrict__ K,
                                                            const bf16_t* __restrict__ V, const bf16_t* __restrict__ KT, const float* __restrict__ L2,
                                                            const float* __restrict__ Dv, bf16_t* __restrict__ dQ, int H, int S, int Spad,
                                                            float scale, float scale_log2, int nbatch) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  dq_body(smem, Q, dO, K, V, KT, L2, Dv, dQ, H, S, Spad, scale, scale_log2, (int)blockIdx.x, (int)gridDim.x);
}

// Both passes as ONE launch: the dQ blocks (the longer ones: 96 MFMAs per tile) in front, the dK / dV blocks behind them.  Each pass on its own ends in a
// partly filled round of one-workgroup-per-CU blocks (B = 1: 1.7 and 3.4 rounds); in one launch the dispatcher fills a CU as soon as any block ends.
__global__ __launch_bounds__(256, 1) void attn_bwd_fused_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                                                               const bf16_t* __restrict__ QT, const bf16_t* __restrict__ KT,
                                                               const bf16_t* __restrict__ dO, const bf16_t* __restrict__ dOT,
                                                               const float* __restrict__ L2, const float* __restrict__ Dv, bf16_t* __restrict__ dQ,
                                                               bf16_t* __restrict__ dK, bf16_t* __restrict__ dV, int H, int S, int Spad, float scale,
                                                               float scale_log2, int n_dq) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = __builtin_amdgcn_readfirstlane((int)blockIdx.x);
  if (b < n_dq) dq_body(smem, Q, dO, K, V, KT, L2, Dv, dQ, H, S, Spad, scale, scale_log2, b, n_dq);
  else dkdv_body(smem, K, V, Q, dO, dOT, QT, L2, Dv, dV, dK, H, S, Spad, scale, scale_log2, b - n_dq, (int)gridDim.x - n_dq);
}

// D[b][h][s] = sum_d dO[b][s][h*128 + d] * O[b][s][h*128 + d] for s < S, 0 for the padding rows; 16 lanes x 8 elements per (token, head)
__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(const bf16_t* __restrict__ dO, long long do_bs, int lddo, const bf16_t* __restrict__ O,
                                                            long long o_bs, int ldo, float* __restrict__ Dv, int H, int S, int Spad) {
  const int s = blockIdx.x, b = blockIdx.y;
  for (int u = threadIdx.x; u < H * 16; u += 256) {
    const int h = u >> 4, c = u & 15;
    float acc = 0.f;
    if (s < S) {
      const bf16x8_t a = *(const bf16x8_t*)(dO + (long long)b * do_bs + (long long)s * lddo + h * 128 + c * 8);
      const bf16x8_t o = *(const bf16x8_t*)(O + (long long)b * o_bs + (long long)s * ldo + h * 128 + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = fmaf(bf16_to_f32((bf16_t)a[j]), bf16_to_f32((bf16_t)o[j]), acc);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (c == 0) Dv[((long long)b * H + h) * Spad + s] = acc;
  }
}

}  // namespace

int x2i_launch_attention_bwd(const void* Q, const void* K, const void* V, const void* QT, const void* KT, const void* dOh, const void* dOT,
                             float* L2, const float* Dv, void* dQ, void* dK, void* dV, int B, int H, int S, int Spad, float scale,
                             int have_lse, hipStream_t stream) {
  if (!Q || !K || !V || !QT || !KT || !dOh || !dOT || !L2 || !Dv || !dQ || !dK || !dV) return x2i_set_error(X2I_ERR_ARG, "attention_bwd: null pointer");
  if (B <= 0 || H <= 0 || S <= 0 || Spad < S || Spad % 128) return x2i_set_error(X2I_ERR_SHAPE, "attention_bwd: Spad must be a multiple of 128 and >= S");
  const uintptr_t all = (uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)QT | (uintptr_t)KT | (uintptr_t)dOh | (uintptr_t)dOT |
                        (uintptr_t)L2 | (uintptr_t)Dv | (uintptr_t)dQ | (uintptr_t)dK | (uintptr_t)dV;
  if (all & 15) return x2i_set_error(X2I_ERR_ALIGN, "attention_bwd: pointers must be 16-byte aligned");
  const float scale_log2 = scale * 1.4426950408889634f;
  dim3 grid((Spad / 128) * H * B);
  if (!have_lse) {  // statistics pass; skipped when the forward already wrote them (x2i_attention_lse_bf16)
    const int shm = 2 * 1 * TILE;
    hipLaunchKernelGGL((attn_bwd_kernel<2>), grid, dim3(256), shm, stream, (const bf16_t*)Q, (const bf16_t*)nullptr, (const bf16_t*)K,
                       (const bf16_t*)nullptr, (const bf16_t*)nullptr, (const bf16_t*)nullptr, L2, Dv, (bf16_t*)nullptr, (bf16_t*)nullptr, H, S,
                       Spad, scale, scale_log2, B);
  }
  // The dQ pass and the dK / dV pass are independent (both read Q, K, V, dO and the statistics; they write different tensors) and each
  // ends in a partly filled round of one-workgroup-per-CU blocks (B = 1: 1.7 and 3.4 rounds).  Product: the software-pipelined kernels, as ONE launch
  // (attn_bwd_fused_kernel: dQ blocks in front, dK / dV blocks behind them -- a CU takes the next block as soon as one ends, whichever pass it belongs
  // to) or, option attn_bwd_overlap = 0, one after the other.  The round-2 kernels (phase after phase; 32-row dQ form) and the two-stream form are
  // compiled into the measurement library only (options attn_bwd_pipe / attn_bwd_dq64 there; bit-identical, tests behind the `ablation` marker).
#ifdef X2I_ABLATION
  const bool pipe = x2i_options().attn_bwd_pipe != 0, dq64 = x2i_options().attn_bwd_dq64 != 0;
#else
  const bool pipe = true, dq64 = true;
#endif
  if (pipe && dq64 && x2i_options().attn_bwd_overlap) {
    const int shm = 2 * KV_STAGE;
    const int rc = x2i_ensure_dynamic_smem((const void*)attn_bwd_fused_kernel, shm);
    if (rc) return rc;
    const int n_dq = ((Spad + 255) / 256) * H * B, n_kv = (Spad / 128) * H * B;
    hipLaunchKernelGGL(attn_bwd_fused_kernel, dim3(n_dq + n_kv), dim3(256), shm, stream, (const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, (const bf16_t*)QT,
                       (const bf16_t*)KT, (const bf16_t*)dOh, (const bf16_t*)dOT, (const float*)L2, Dv, (bf16_t*)dQ, (bf16_t*)dK, (bf16_t*)dV, H, S, Spad, scale,
                       scale_log2, n_dq);
    return x2i_check_launch("attention_bwd (fused passes)");
  }
  hipStream_t qs = stream;
#ifdef X2I_ABLATION
  hipStream_t side = stream;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  const bool overlap = x2i_options().attn_bwd_overlap && x2i_side_stream(stream, &side, &ev_fork, &ev_join) &&
                       hipEventRecord(ev_fork, stream) == hipSuccess && hipStreamWaitEvent(side, ev_fork, 0) == hipSuccess;
  qs = overlap ? side : stream;
#endif
  if (dq64 && pipe) {   // dQ: 64 query rows per wave, software-pipelined (attn_bwd_dq_kernel)
    const int shm = 2 * 3 * TILE;
    const int rc = x2i_ensure_dynamic_smem((const void*)attn_bwd_dq_kernel, shm);
    if (rc) return rc;
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(((Spad + 255) / 256) * H * B), dim3(256), shm, qs, (const bf16_t*)Q, (const bf16_t*)dOh,
                       (const bf16_t*)K, (const bf16_t*)V, (const bf16_t*)KT, (const float*)L2, Dv, (bf16_t*)dQ, H, S, Spad, scale, scale_log2, B);
  }
#ifdef X2I_ABLATION
  else if (dq64) {  // phase after phase (attn_bwd_dq64_kernel); option attn_bwd_dq64 = 0: the 32-row form (A/B, bit-identical)
    const int shm = 2 * 3 * TILE;
    const int rc = x2i_ensure_dynamic_smem((const void*)attn_bwd_dq64_kernel, shm);
    if (rc) return rc;
    hipLaunchKernelGGL(attn_bwd_dq64_kernel, dim3(((Spad + 255) / 256) * H * B), dim3(256), shm, qs, (const bf16_t*)Q, (const bf16_t*)dOh,
                       (const bf16_t*)K, (const bf16_t*)V, (const bf16_t*)KT, L2, Dv, (bf16_t*)dQ, H, S, Spad, scale, scale_log2, B);
  } else {
    const int shm = 2 * 3 * TILE;
    const int rc = x2i_ensure_dynamic_smem((const void*)attn_bwd_kernel<0>, shm);
    if (rc) return rc;
    hipLaunchKernelGGL((attn_bwd_kernel<0>), grid, dim3(256), shm, qs, (const bf16_t*)Q, (const bf16_t*)dOh, (const bf16_t*)K, (const bf16_t*)V,
                       (const bf16_t*)KT, (const bf16_t*)nullptr, L2, Dv, (bf16_t*)dQ, (bf16_t*)nullptr, H, S, Spad, scale, scale_log2, B);
  }
  if (overlap) (void)hipEventRecord(ev_join, side);
#endif
  if (pipe) {   // dK / dV: software-pipelined (attn_bwd_dkdv_kernel)
    const int shm = 2 * KV_STAGE;
    const int rc = x2i_ensure_dynamic_smem((const void*)attn_bwd_dkdv_kernel, shm);
    if (rc) return rc;
    hipLaunchKernelGGL(attn_bwd_dkdv_kernel, grid, dim3(256), shm, stream, (const bf16_t*)K, (const bf16_t*)V, (const bf16_t*)Q, (const bf16_t*)dOh,
                       (const bf16_t*)dOT, (const bf16_t*)QT, (const float*)L2, Dv, (bf16_t*)dV, (bf16_t*)dK, H, S, Spad, scale, scale_log2, B);
  }
#ifdef X2I_ABLATION
  else {
    const int shm = 2 * 4 * TILE;
    const int rc = x2i_ensure_dynamic_smem((const void*)attn_bwd_kernel<1>, shm);
    if (rc) return rc;
    hipLaunchKernelGGL((attn_bwd_kernel<1>), grid, dim3(256), shm, stream, (const bf16_t*)K, (const bf16_t*)V, (const bf16_t*)Q, (const bf16_t*)dOh,
                       (const bf16_t*)dOT, (const bf16_t*)QT, L2, Dv, (bf16_t*)dV, (bf16_t*)dK, H, S, Spad, scale, scale_log2, B);
  }
  if (overlap) (void)hipStreamWaitEvent(stream, ev_join, 0);
#endif
  return x2i_check_launch("attention_bwd");
}

int x2i_launch_attention_bwd_prep(const void* dO, long long do_bs, int lddo, const void* O, long long o_bs, int ldo, float* Dv, int B, int H,
                                  int S, int Spad, hipStream_t stream) {
  if (!dO || !O || !Dv || B <= 0 || H <= 0 || S <= 0 || Spad < S) return x2i_set_error(X2I_ERR_ARG, "attention_bwd_prep: bad argument");
  if (lddo % 8 || ldo % 8 || do_bs % 8 || o_bs % 8 || (((uintptr_t)dO | (uintptr_t)O) & 15)) return x2i_set_error(X2I_ERR_ALIGN, "attention_bwd_prep: 16-byte aligned rows");
  hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3(Spad, B), dim3(256), 0, stream, (const bf16_t*)dO, do_bs, lddo, (const bf16_t*)O, o_bs, ldo, Dv, H, S, Spad);
  return x2i_check_launch("attention_bwd_prep");
}
