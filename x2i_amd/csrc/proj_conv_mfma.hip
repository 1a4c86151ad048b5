// Projector layer fusion, conv5x5 on the matrix cores (round 2, third form of this kernel; reference utils/proj.py:50,68-69:
// Conv2d(C -> 1, kernel 5, padding 2) over the (S,H) plane of the stacked MLLM hidden states x[B,C,S,H]).
//
// The VALU form (projector.hip) needs 25 * C / 2 dot2 instructions per output and is bound by them (2.0 TB/s of input at C = 37).
// Here the five taps of one kernel row become a banded Toeplitz matrix and ride v_mfma_f32_16x16x32_bf16:
//
//   for a layer c and a kernel row ds:   P_ds[n][R] += sum_k T_{c,ds}[n][k] * X_c[k][R]
//     X_c[k][R] = x[c][s0 - 2 + R][hb - 8 + k]      16 input rows R, a 32-column window (one ds_read_b128 per lane)
//     T_{c,ds}[n][k] = w[c][ds][k - n - 6]          (0 outside the five taps): output column hb + n sees input columns hb + n - 2 .. + 2
//   y[s0 + o][hb + n] = bias + sum_ds P_ds[n][R = o + ds]                     12 complete output rows per 16 input rows
//
// so one X fragment feeds five MFMAs (one per kernel row, five accumulator sets) and the shift over rows happens ONCE, in the
// epilogue: with the Toeplitz fragment as the MFMA's A operand a lane owns four output columns of ONE input row, and row o + ds sits
// ds lanes to the right inside the 16-lane DPP row -- `row_shl` moves, no LDS.  The MFMA does 32 x 16 MACs per output row for 80
// useful ones, but at 16 cycles per instruction that is 30 us for the Qwen2.5-VL-3B slab (B = 4) against 39 us for its bytes at
// 8 TB/s: the kernel is back on the HBM roof.
//
// Staging: LDS-DMA into a ring of stages (one layer per stage, three deep, in the two product forms), counted vmcnt waits and a raw
// s_barrier per stage (__syncthreads() would drain vmcnt to 0).  A layer's image = rows x [32 chunks of 16 B: columns h0 - 8 ..
// h0 + 247] with chunk q of row R stored in slot q ^ (R & 15) (the ds_read_b128 lane groups pair rows {0-3, 12-15} at chunk 2t + g
// with rows {4-11} at chunk 2t + g + 1: the XOR makes the sixteen 16-byte slots distinct modulo 256 B; SQ_LDS_BANK_CONFLICT = 0),
// plus the two right-halo chunks per row, plus the layer's five Toeplitz fragments (1 KiB each, lane-linear, from the table
// x2i_proj_conv5x5_pack builds once per weight).  Rows / columns / layers outside the tensor are buffer-descriptor zero fill.
//
// Three forms, same arithmetic in the same order (bit-identical, tested): conv5x5_mfma_kernel<PCS, NBUF> (plain stages; A/B
// reference), conv5x5_mfma_pipe_kernel (fragments of layer c + 1 are read while layer c's MFMAs run) and conv5x5_mfma_rb2_kernel
// (two row blocks per wave: 28 staged rows per 24 output rows).  Measured (B = 4, Qwen2.5-VL-3B / MiniCPM slabs): 78 / 108 us =
// 3.97 / 3.95 TB/s of input against 170 / 207 us for the VALU form; what bounds them is the FETCHED stream (x 1.3-1.45 the
// algorithmic bytes: row halo, chunk-granular column halo; ~5.7 TB/s at the L2<->fabric boundary), not MFMA issue, LDS or DMA
// latency: deeper rings and the register pipeline moved the time by < 5 %, the XCD-contiguous tile order by 8-20 % where the
// plain grid had put vertical neighbours on different XCDs.
#include "x2i_common.h"
#include "x2i_kernels.h"

namespace {

constexpr int PR = 16;                    // input rows per block = MFMA N
constexpr int PO = PR - 4;                // complete output rows per block
constexpr int PTH = 256;                  // output columns per block: 4 waves x 4 tiles x 16
constexpr int P_MAIN = PR * 512;          // 16 rows x 32 chunks
constexpr int P_TAB = 5 * 1024;           // five Toeplitz fragments
constexpr int P_CH = P_MAIN + P_TAB;      // per layer
constexpr int P_TAIL = 1024;              // right halo, one DMA piece: [layer (<= 2)][row][2 chunks]
// PCS = layers per stage (1 or 2), NBUF = ring depth (NBUF - 1 stages in flight while one is consumed)
template <int PCS> constexpr int stage_bytes() { return PCS * P_CH + P_TAIL; }
constexpr uint32_t OOB = 0x7fffffffu;

typedef __attribute__((ext_vector_type(8))) __bf16 pbf16x8_t;
typedef __attribute__((ext_vector_type(4))) float pf32x4_t;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_piece, uint32_t voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_piece, 16, voff, 0, 0, 0);
}

template <int N>
__device__ __forceinline__ float row_shl(float v) {  // lane i of a 16-lane row <- lane i + N (0 beyond the row)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 | N, 0xf, 0xf, true));
}


// Workgroup -> tile mapping.  The hardware deals consecutive workgroup ids round-robin to the 8 XCDs; id -> v = (id % 8) * per + id / 8
// hands every XCD a CONTIGUOUS run of tiles in (sample, column block, row block) order with the row block fastest, so the
// workgroups an XCD runs at the same time are vertical neighbours and the 4 halo rows they share are fetched from HBM once
// and found in that XCD's L2 the second time (with the plain 3-D grid that only happened when gridDim.x % 8 == 0).
struct TileId { int x, y, b; bool live; };
__device__ __forceinline__ TileId tile_of_block(int GX, int GY, int B) {
  const int total = GX * GY * B, per = (total + 7) >> 3;
  const int id = blockIdx.x, v = (id & 7) * per + (id >> 3);
  TileId t;
  t.live = v < total && (id >> 3) < per;
  t.y = v % GY;
  t.x = (v / GY) % GX;
  t.b = v / (GY * GX);
  return t;
}

// Toeplitz fragments: table[c][ds][lane = n + 16 g][j] = w[c][ds][8 g + j - n - 6] as bf16 (RNE), 0 outside the taps
__global__ void conv5x5_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ table) {
  const int cd = blockIdx.x, lane = threadIdx.x, n = lane & 15, g = lane >> 4;
  bf16_t v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int dh = 8 * g + j - n - 6;
    v[j] = (dh >= 0 && dh <= 4) ? f32_to_bf16(w[cd * 5 + dh]) : (bf16_t)0;
  }
  uint4 o;
  o.x = (uint32_t)v[0] | ((uint32_t)v[1] << 16); o.y = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
  o.z = (uint32_t)v[4] | ((uint32_t)v[5] << 16); o.w = (uint32_t)v[6] | ((uint32_t)v[7] << 16);
  ((uint4*)table)[cd * 64 + lane] = o;
}

struct StageSrc {
  uint32_t main0, main1;   // this lane's byte offsets (layer 0 of the slab) in the wave's two main pieces, or OOB
  uint32_t tail;           // wave 2 / 3 only
  uint32_t plane_bytes;    // S * H * 2
};

// byte offset of chunk q (8 columns from h0 - 8 + 8 q) of staged row R (token s0 - 2 + R) inside one layer's plane, or OOB (zero fill)
__device__ __forceinline__ uint32_t chunk_src(int R, int q, int s0, int h0, int S, int H, int rows) {
  const int s = s0 - 2 + R, h = h0 - 8 + 8 * q;
  return (R < rows && s >= 0 && s < S && h >= 0 && h < H) ? (uint32_t)(((long long)s * H + h) * 2) : OOB;
}
// this lane's DMA sources of the 16-row forms: the wave's two main pieces (piece p = rows 2p, 2p + 1: lane -> row 2p + (lane >> 5), slot
// lane & 31 holds chunk slot ^ (R & 15)) and the halo piece ([layer = lane >> 5][row = (lane & 31) >> 1][chunk 32 + (lane & 1)])
__device__ __forceinline__ StageSrc make_stage_src(int wave, int lane, int s0, int h0, int S, int H) {
  StageSrc src;
  src.plane_bytes = (uint32_t)((long long)S * H * 2);
  const int R0 = 2 * (2 * wave) + (lane >> 5), R1 = 2 * (2 * wave + 1) + (lane >> 5), slot = lane & 31;
  src.main0 = chunk_src(R0, slot ^ (R0 & 15), s0, h0, S, H, PR);
  src.main1 = chunk_src(R1, slot ^ (R1 & 15), s0, h0, S, H, PR);
  src.tail = chunk_src((lane & 31) >> 1, 32 + (lane & 1), s0, h0, S, H, PR);
  return src;
}
// LDS offset of the X fragment of tile t = 4 wave + tt for row block `blk` (rows 12 blk ...): lane (R, g = lane >> 4) reads chunk 2t + g;
// chunks 32, 33 live in the halo piece at tail_off ([row][2 chunks])
__device__ __forceinline__ uint32_t frag_off(int wave, int lane, int tt, int blk, uint32_t tail_off) {
  const int R = 12 * blk + (lane & 15), q = 2 * (4 * wave + tt) + (lane >> 4);
  return q < 32 ? (uint32_t)(R * 512 + ((q ^ (R & 15)) * 16)) : tail_off + (uint32_t)(R * 32 + (q - 32) * 16);
}

// LDS-DMA of stage `st` (layers st * PCS ...) into dst: per wave 4 main pieces, 2-3 Toeplitz pieces, wave 3 the halo piece
template <int PCS>
__device__ __forceinline__ void issue_stage(char* __restrict__ dst, int st, int C, __amdgpu_buffer_rsrc_t xr, __amdgpu_buffer_rsrc_t tr,
                                            const StageSrc& src, int wave, int lane) {
#pragma unroll
  for (int ch = 0; ch < PCS; ++ch) {
    const int c = st * PCS + ch;
    const bool live = c < C;
    const uint32_t cb = (uint32_t)c * src.plane_bytes;
    dma16(xr, dst + ch * P_CH + (2 * wave) * 1024, (live && src.main0 != OOB) ? src.main0 + cb : OOB);
    dma16(xr, dst + ch * P_CH + (2 * wave + 1) * 1024, (live && src.main1 != OOB) ? src.main1 + cb : OOB);
  }
  // 5 * PCS table pieces (layer, ds) and the halo piece: wave takes items e = wave, wave + 4, wave + 8 of that list
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int e = wave + 4 * i;
    if (e < 5 * PCS) {
      const int ch = e / 5, ds = e % 5, c = st * PCS + ch;
      dma16(tr, dst + ch * P_CH + P_MAIN + ds * 1024, c < C ? (uint32_t)((c * 5 + ds) * 1024 + lane * 16) : OOB);
    } else if (e == 5 * PCS) {  // right halo: lanes 0-31 layer 0, lanes 32-63 layer 1 (zero fill when PCS == 1)
      const int ch = lane >> 5, c = st * PCS + ch;
      dma16(xr, dst + PCS * P_CH, (ch < PCS && c < C && src.tail != OOB) ? src.tail + (uint32_t)c * src.plane_bytes : OOB);
    }
  }
}
// DMA instructions a wave issues per stage (for the counted vmcnt waits)
template <int PCS> __device__ __forceinline__ int pieces_per_stage(int wave) { return 2 * PCS + (5 * PCS + 1 - wave + 3) / 4; }

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is six bits");
  __builtin_amdgcn_s_waitcnt(0x0f70 | (N & 15) | ((N >> 4) << 14));
}

// One stage: issue the LDS-DMA of stage `nxt` into dma_dst, then run the 2 x 4 x 5 MFMAs of the stage held in img.  img / dma_dst
// are __restrict__ parameters of one function so that hipcc can tell the fragment reads from the buffer being filled (no vmcnt(0)
// in front of the reads).
template <int PCS>
__device__ __forceinline__ void conv_stage(const char* __restrict__ img, char* __restrict__ dma_dst, bool issue, int nxt, int C,
                                           __amdgpu_buffer_rsrc_t xr, __amdgpu_buffer_rsrc_t tr, const StageSrc& src, int wave, int lane,
                                           const uint32_t (&xoff)[4], pf32x4_t (&acc)[5][4]) {
  if (issue) issue_stage<PCS>(dma_dst, nxt, C, xr, tr, src, wave, lane);
#pragma unroll
  for (int ch = 0; ch < PCS; ++ch) {
    const char* base = img + ch * P_CH;
    pbf16x8_t tf[5], xf[4];
#pragma unroll
    for (int ds = 0; ds < 5; ++ds) tf[ds] = *(const pbf16x8_t*)(base + P_MAIN + ds * 1024 + lane * 16);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      // offsets >= PCS * P_CH address the halo piece (shared by the layers: add the layer's 512-byte half)
      const uint32_t o = xoff[tt];
      xf[tt] = *(const pbf16x8_t*)(o >= (uint32_t)(PCS * P_CH) ? img + o + ch * 512 : base + o);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int ds = 0; ds < 5; ++ds) acc[ds][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tf[ds], xf[tt], acc[ds][tt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int PCS, int NBUF>
__global__ __launch_bounds__(256, 2) void conv5x5_mfma_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ table,
                                                              const float* __restrict__ bias, bf16_t* __restrict__ y, int C, int S,
                                                              int H) {
  constexpr int P_STAGE = stage_bytes<PCS>();
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [NBUF][P_STAGE]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h0 = blockIdx.x * PTH, s0 = blockIdx.y * PO, b = blockIdx.z;
  const long long plane = (long long)S * H;
  const bf16_t* xb = x + (long long)b * C * plane;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (int)(C * plane * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t tr = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, C * 5 * 1024, 0x00020000);

  const StageSrc src = make_stage_src(wave, lane, s0, h0, S, H);
  uint32_t xoff[4];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) xoff[tt] = frag_off(wave, lane, tt, 0, PCS * P_CH);
  pf32x4_t acc[5][4];
#pragma unroll
  for (int ds = 0; ds < 5; ++ds)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) acc[ds][tt] = pf32x4_t{0.f, 0.f, 0.f, 0.f};

  const int NS = (C + PCS - 1) / PCS;
  // waves 0.. issue PER_A DMA pieces per stage, the last wave(s) PER_B (pieces_per_stage)
  constexpr int PER_A = 2 * PCS + (5 * PCS + 1 + 3) / 4, PER_B = 2 * PCS + (5 * PCS + 1 - 3 + 3) / 4;
  const bool per_a = pieces_per_stage<PCS>(wave) == PER_A;
#pragma unroll
  for (int i = 0; i < NBUF - 1; ++i)
    if (i < NS) issue_stage<PCS>(smem + i * P_STAGE, i, C, xr, tr, src, wave, lane);
  int cur = 0, nxt_buf = NBUF - 1;
  for (int st = 0; st < NS; ++st) {
    // stage st has landed when only the NBUF - 2 stages issued behind it are outstanding; the last stages drain fully
    if (st + NBUF - 1 <= NS) {
      if (per_a) wait_vmcnt<(NBUF - 2) * PER_A>();
      else wait_vmcnt<(NBUF - 2) * PER_B>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();  // (not __syncthreads(): its fence would drain vmcnt to 0) everyone's pieces have landed, and everyone is
                                   // done reading the buffer consumed in the previous iteration
    conv_stage<PCS>(smem + cur * P_STAGE, smem + nxt_buf * P_STAGE, st + NBUF - 1 < NS, st + NBUF - 1, C, xr, tr, src, wave, lane, xoff, acc);
    cur = cur + 1 == NBUF ? 0 : cur + 1;
    nxt_buf = nxt_buf + 1 == NBUF ? 0 : nxt_buf + 1;
  }

  // epilogue: y[s0 + o][hb + 4G + r] = bias + sum_ds P_ds[4G + r][R = o + ds]; lane (R = o, G) pulls row o + ds with row_shl
  const float bv = bias ? bias[0] : 0.f;
  const int o = lane & 15, G = lane >> 4;
  const int s = s0 + o;
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      v[r] = bv + acc[0][tt][r] + row_shl<1>(acc[1][tt][r]) + row_shl<2>(acc[2][tt][r]) + row_shl<3>(acc[3][tt][r]) +
             row_shl<4>(acc[4][tt][r]);
    const int h = h0 + 16 * (4 * wave + tt) + 4 * G;
    if (o < PO && s < S && h < H) {
      uint2 pk;
      pk.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
      pk.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
      *(uint2*)(y + ((long long)b * S + s) * H + h) = pk;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Register-pipelined form (the default): one layer per stage, ring of NBUF stages, and the fragments of layer c + 1 are read
// from LDS into a second register set BEFORE the 20 MFMAs of layer c are issued -- a wave alone on its SIMD no longer pays
// read latency + MFMA time + barrier back to back for every layer (measured with one workgroup per CU: ~1100 cycles per layer
// against 320 of MFMA work in the form above).
struct Frags {
  pbf16x8_t tf[5], xf[4];
};

// issue the DMA of stage `nxt` into dma_dst and read the fragments of the stage held in img (restrict: see conv_stage)
__device__ __forceinline__ void issue_and_load(const char* __restrict__ img, char* __restrict__ dma_dst, bool issue, bool load, int nxt,
                                               int C, __amdgpu_buffer_rsrc_t xr, __amdgpu_buffer_rsrc_t tr, const StageSrc& src, int wave,
                                               int lane, const uint32_t (&xoff)[4], Frags& f) {
  if (issue) issue_stage<1>(dma_dst, nxt, C, xr, tr, src, wave, lane);
  if (load) {
#pragma unroll
    for (int ds = 0; ds < 5; ++ds) f.tf[ds] = *(const pbf16x8_t*)(img + P_MAIN + ds * 1024 + lane * 16);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) f.xf[tt] = *(const pbf16x8_t*)(img + xoff[tt]);
  }
}

template <int NBUF>
__global__ __launch_bounds__(256, 2) void conv5x5_mfma_pipe_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ table,
                                                                   const float* __restrict__ bias, bf16_t* __restrict__ y, int C,
                                                                   int S, int H, int B) {
  constexpr int P_STAGE = stage_bytes<1>();
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [NBUF][P_STAGE]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const TileId tile = tile_of_block((H + PTH - 1) / PTH, (S + PO - 1) / PO, B);
  if (!tile.live) return;
  const int h0 = tile.x * PTH, s0 = tile.y * PO, b = tile.b;
  const long long plane = (long long)S * H;
  const bf16_t* xb = x + (long long)b * C * plane;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (int)(C * plane * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t tr = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, C * 5 * 1024, 0x00020000);
  const StageSrc src = make_stage_src(wave, lane, s0, h0, S, H);
  uint32_t xoff[4];
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) xoff[tt] = frag_off(wave, lane, tt, 0, P_CH);
  pf32x4_t acc[5][4];
#pragma unroll
  for (int ds = 0; ds < 5; ++ds)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) acc[ds][tt] = pf32x4_t{0.f, 0.f, 0.f, 0.f};

  const int NS = C;
  constexpr int PER_A = 2 + (5 + 1 + 3) / 4, PER_B = 2 + (5 + 1 - 3 + 3) / 4;  // DMA pieces per stage: waves 0-1 / 2-3
  const bool per_a = pieces_per_stage<1>(wave) == PER_A;
  // prologue: stages 0 .. NBUF-1 go out, stage 0 is awaited and read
#pragma unroll
  for (int i = 0; i < NBUF; ++i)
    if (i < NS) issue_stage<1>(smem + i * P_STAGE, i, C, xr, tr, src, wave, lane);
  if (NS >= NBUF) {
    if (per_a) wait_vmcnt<(NBUF - 1) * PER_A>();
    else wait_vmcnt<(NBUF - 1) * PER_B>();
  } else {
    wait_vmcnt<0>();
  }
  __builtin_amdgcn_s_barrier();
  Frags fa, fb;
  issue_and_load(smem, smem, false, true, 0, C, xr, tr, src, wave, lane, xoff, fa);

  int cur = 0;  // ring slot of stage st
  auto iter = [&](int st, Frags& fc, Frags& fn) {
    // stage st + 1 has landed when only the stages issued behind it are outstanding (steady state NBUF - 2); the tail drains fully
    if (st + NBUF <= NS) {
      if (per_a) wait_vmcnt<(NBUF - 2) * PER_A>();
      else wait_vmcnt<(NBUF - 2) * PER_B>();
    } else {
      wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's reads of stage st (issued last iteration) are in registers
    __builtin_amdgcn_s_barrier();        // -> slot `cur` is free for stage st + NBUF, and stage st + 1 is visible to everyone
    const int nslot = cur + 1 == NBUF ? 0 : cur + 1;
    issue_and_load(smem + nslot * P_STAGE, smem + cur * P_STAGE, st + NBUF < NS, st + 1 < NS, st + NBUF, C, xr, tr, src, wave, lane, xoff, fn);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int ds = 0; ds < 5; ++ds) acc[ds][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc.tf[ds], fc.xf[tt], acc[ds][tt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    cur = nslot;
  };
  for (int st = 0; st < NS; st += 2) {
    iter(st, fa, fb);
    if (st + 1 < NS) iter(st + 1, fb, fa);
  }

  const float bv = bias ? bias[0] : 0.f;
  const int o = lane & 15, G = lane >> 4;
  const int s = s0 + o;
#pragma unroll
  for (int tt = 0; tt < 4; ++tt) {
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
      v[r] = bv + acc[0][tt][r] + row_shl<1>(acc[1][tt][r]) + row_shl<2>(acc[2][tt][r]) + row_shl<3>(acc[3][tt][r]) +
             row_shl<4>(acc[4][tt][r]);
    const int h = h0 + 16 * (4 * wave + tt) + 4 * G;
    if (o < PO && s < S && h < H) {
      uint2 pk;
      pk.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
      pk.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
      *(uint2*)(y + ((long long)b * S + s) * H + h) = pk;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Two row blocks per wave (the default).  PMC (profiles/r02c_pmc_proj.json): the forms above fetch 1.45x the algorithmic bytes
// (16 staged rows per 12 output rows, chunk-granular column halo, the tables) and run at 5.7 TB/s of fetched bytes -- they are
// HBM-bound on the amplified stream, and neighbouring workgroups' halo re-reads mostly miss L2 (hit rate 47 %, tables
// included).  Here a workgroup stages 28 rows for 24 output rows (x 1.17 instead of x 1.33): every wave runs two MFMA row
// blocks (input rows 0-15 and 12-27 of the image) against the same Toeplitz fragments -- 40 MFMAs per layer and wave, 5 + 8
// fragment reads -- and every wave issues exactly five DMA pieces per layer (14 row pairs + 5 table pieces + 1 halo piece).
constexpr int R2 = 28, O2 = 24;
constexpr int P2_MAIN = R2 * 512, P2_TAIL_OFF = P2_MAIN + P_TAB, P2_STAGE = P2_TAIL_OFF + 1024;

struct Stage2Src {
  uint32_t main[4];  // row pairs w, w + 4, w + 8, w + 12 (the last one: waves 0 and 1 only)
  uint32_t tail;     // wave 3
  uint32_t plane_bytes;
};

__device__ __forceinline__ void issue_stage2(char* __restrict__ dst, int c, int C, __amdgpu_buffer_rsrc_t xr, __amdgpu_buffer_rsrc_t tr,
                                             const Stage2Src& src, int wave, int lane) {
  const bool live = c < C;
  const uint32_t cb = (uint32_t)c * src.plane_bytes;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int p = wave + 4 * i;
    if (p < R2 / 2) dma16(xr, dst + p * 1024, (live && src.main[i] != OOB) ? src.main[i] + cb : OOB);
  }
  const uint32_t toff = live ? (uint32_t)(c * 5 * 1024 + lane * 16) : OOB;
  if (wave < 2) {
    dma16(tr, dst + P2_MAIN + wave * 1024, live ? toff + wave * 1024 : OOB);
  } else if (wave == 2) {
    dma16(tr, dst + P2_MAIN + 2 * 1024, live ? toff + 2 * 1024 : OOB);
    dma16(tr, dst + P2_MAIN + 3 * 1024, live ? toff + 3 * 1024 : OOB);
  } else {
    dma16(tr, dst + P2_MAIN + 4 * 1024, live ? toff + 4 * 1024 : OOB);
    dma16(xr, dst + P2_TAIL_OFF, (live && src.tail != OOB) ? src.tail + cb : OOB);
  }
}

__device__ __forceinline__ void conv_stage2(const char* __restrict__ img, char* __restrict__ dma_dst, bool issue, int nxt, int C,
                                            __amdgpu_buffer_rsrc_t xr, __amdgpu_buffer_rsrc_t tr, const Stage2Src& src, int wave,
                                            int lane, const uint32_t (&xoff)[2][4], pf32x4_t (&acc)[2][5][4]) {
  if (issue) issue_stage2(dma_dst, nxt, C, xr, tr, src, wave, lane);
  pbf16x8_t tf[5], xf[2][4];
#pragma unroll
  for (int ds = 0; ds < 5; ++ds) tf[ds] = *(const pbf16x8_t*)(img + P2_MAIN + ds * 1024 + lane * 16);
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) xf[blk][tt] = *(const pbf16x8_t*)(img + xoff[blk][tt]);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int ds = 0; ds < 5; ++ds)
        acc[blk][ds][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tf[ds], xf[blk][tt], acc[blk][ds][tt], 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
}

template <int NBUF>
__global__ __launch_bounds__(256, 2) void conv5x5_mfma_rb2_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ table,
                                                                  const float* __restrict__ bias, bf16_t* __restrict__ y, int C,
                                                                  int S, int H, int B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [NBUF][P2_STAGE]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const TileId tile = tile_of_block((H + PTH - 1) / PTH, (S + O2 - 1) / O2, B);
  if (!tile.live) return;
  const int h0 = tile.x * PTH, s0 = tile.y * O2, b = tile.b;
  const long long plane = (long long)S * H;
  const bf16_t* xb = x + (long long)b * C * plane;
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (int)(C * plane * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t tr = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, C * 5 * 1024, 0x00020000);
  Stage2Src src;
  src.plane_bytes = (uint32_t)(plane * 2);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int R = 2 * (wave + 4 * i) + (lane >> 5), slot = lane & 31;
    src.main[i] = chunk_src(R, slot ^ (R & 15), s0, h0, S, H, R2);
  }
  src.tail = chunk_src(lane >> 1, 32 + (lane & 1), s0, h0, S, H, R2);  // halo piece: [row = lane >> 1][chunk 32 + (lane & 1)], rows >= 28 zero fill
  uint32_t xoff[2][4];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) xoff[blk][tt] = frag_off(wave, lane, tt, blk, P2_TAIL_OFF);
  pf32x4_t acc[2][5][4];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int ds = 0; ds < 5; ++ds)
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) acc[blk][ds][tt] = pf32x4_t{0.f, 0.f, 0.f, 0.f};

  const int NS = C;
  constexpr int PER = 5;  // DMA pieces per wave and stage
#pragma unroll
  for (int i = 0; i < NBUF - 1; ++i)
    if (i < NS) issue_stage2(smem + i * P2_STAGE, i, C, xr, tr, src, wave, lane);
  int cur = 0, nxt_buf = NBUF - 1;
  for (int st = 0; st < NS; ++st) {
    if (st + NBUF - 1 <= NS) wait_vmcnt<(NBUF - 2) * PER>();  // only the NBUF - 2 stages issued behind stage st are outstanding
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    conv_stage2(smem + cur * P2_STAGE, smem + nxt_buf * P2_STAGE, st + NBUF - 1 < NS, st + NBUF - 1, C, xr, tr, src, wave, lane, xoff, acc);
    cur = cur + 1 == NBUF ? 0 : cur + 1;
    nxt_buf = nxt_buf + 1 == NBUF ? 0 : nxt_buf + 1;
  }

  const float bv = bias ? bias[0] : 0.f;
  const int o = lane & 15, G = lane >> 4;
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int s = s0 + 12 * blk + o;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        v[r] = bv + acc[blk][0][tt][r] + row_shl<1>(acc[blk][1][tt][r]) + row_shl<2>(acc[blk][2][tt][r]) + row_shl<3>(acc[blk][3][tt][r]) +
               row_shl<4>(acc[blk][4][tt][r]);
      const int h = h0 + 16 * (4 * wave + tt) + 4 * G;
      if (o < PO && s < S && h < H) {
        uint2 pk;
        pk.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
        pk.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
        *(uint2*)(y + ((long long)b * S + s) * H + h) = pk;
      }
    }
  }
}

}  // namespace

int x2i_launch_proj_conv5x5_pack(const float* w, void* table, int C, hipStream_t stream) {
  if (!w || !table) return x2i_set_error(X2I_ERR_ARG, "proj_conv5x5_pack: null pointer");
  if (C <= 0) return x2i_set_error(X2I_ERR_SHAPE, "proj_conv5x5_pack: C=%d", C);
  if (((uintptr_t)table) & 15) return x2i_set_error(X2I_ERR_ALIGN, "proj_conv5x5_pack: table must be 16-byte aligned");
  hipLaunchKernelGGL(conv5x5_pack_kernel, dim3(C * 5), dim3(64), 0, stream, w, (bf16_t*)table);
  return x2i_check_launch("proj_conv5x5_pack");
}

int x2i_launch_proj_conv5x5_packed(const void* x, const void* table, const float* bias, void* y, int B, int C, int S, int H,
                                   hipStream_t stream) {
  if (!x || !table || !y) return x2i_set_error(X2I_ERR_ARG, "proj_conv5x5_packed: null pointer");
  if (B <= 0 || C <= 0 || S <= 0 || H <= 0 || H % 8) return x2i_set_error(X2I_ERR_SHAPE, "proj_conv5x5_packed: need H %% 8 == 0 (H=%d)", H);
  if ((((uintptr_t)x) & 15) || (((uintptr_t)table) & 15) || (((uintptr_t)y) & 7))
    return x2i_set_error(X2I_ERR_ALIGN, "proj_conv5x5_packed: x / table must be 16-byte, y 8-byte aligned");
  if ((long long)C * S * H * 2 >= 0x7f000000LL) return x2i_set_error(X2I_ERR_SHAPE, "proj_conv5x5_packed: one sample's [C,S,H] slab must stay below 2 GB");
  dim3 grid((H + PTH - 1) / PTH, (S + PO - 1) / PO, B);
  // Form choice (measured, tools/conv_bench.py): the two-row-block form fetches less (x 1.17 rows instead of x 1.33) but has half as
  // many, twice as large workgroups -- it wins when those still fill the chip (one partial round, or a last round >= 75 % full at
  // 2 workgroups per CU), otherwise the register-pipelined one-block form does.  conv5_variant: 1 = plain 2-layer stages
  // (A/B reference), 2 / 3 force the pipelined / two-row-block form.
  const int variant = x2i_options().conv5_variant;
  const long long n2 = (long long)((H + PTH - 1) / PTH) * ((S + O2 - 1) / O2) * B;
  const double eff2 = (double)n2 / (double)(((n2 + 511) / 512) * 512);
  const bool rb2 = variant == 3 || (variant == 0 && n2 >= 256 && (n2 <= 512 || eff2 >= 0.75));
#define X2I_CONV5_LAUNCH(PCS_, NBUF_)                                                                                              \
  {                                                                                                                                 \
    const int shm = NBUF_ * stage_bytes<PCS_>();                                                                                   \
    const int rc = x2i_ensure_dynamic_smem((const void*)conv5x5_mfma_kernel<PCS_, NBUF_>, shm);                                    \
    if (rc) return rc;                                                                                                              \
    hipLaunchKernelGGL((conv5x5_mfma_kernel<PCS_, NBUF_>), grid, dim3(256), shm, stream, (const bf16_t*)x, (const bf16_t*)table,  \
                       bias, (bf16_t*)y, C, S, H);                                                                                  \
  }
  if (variant == 1) X2I_CONV5_LAUNCH(2, 2)
  else if (!rb2) {
    constexpr int NB = 3;
    const int shm = NB * stage_bytes<1>();
    const int rc = x2i_ensure_dynamic_smem((const void*)conv5x5_mfma_pipe_kernel<NB>, shm);
    if (rc) return rc;
    const int total = grid.x * grid.y * grid.z;
    hipLaunchKernelGGL((conv5x5_mfma_pipe_kernel<NB>), dim3(((total + 7) / 8) * 8), dim3(256), shm, stream, (const bf16_t*)x,
                       (const bf16_t*)table, bias, (bf16_t*)y, C, S, H, B);
  } else {
    constexpr int NB = 3;
    const int shm = NB * P2_STAGE;
    const int rc = x2i_ensure_dynamic_smem((const void*)conv5x5_mfma_rb2_kernel<NB>, shm);
    if (rc) return rc;
    hipLaunchKernelGGL((conv5x5_mfma_rb2_kernel<NB>), dim3((unsigned)(((n2 + 7) / 8) * 8)), dim3(256), shm, stream, (const bf16_t*)x,
                       (const bf16_t*)table, bias, (bf16_t*)y, C, S, H, B);
  }
#undef X2I_CONV5_LAUNCH
  return x2i_check_launch("proj_conv5x5_packed");
}
