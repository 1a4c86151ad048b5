// Layer-fusion stage of the alignment projector (reference utils/proj.py:62-72): HBM-bound streaming over the
// stacked MLLM hidden states x[B,C,S,H] (106 MB per sample for Qwen2.5-VL-7B).
//
//   conv5x5   : Conv2d(C -> 1, kernel 5, padding 2) over the (S,H) plane   (utils/proj.py:50,68-69)
//   layer_mean: (cha_scale * x).mean(dim=1) / x.mean(dim=1)                (utils/proj.py:66-67,70-71)
//
// conv5x5 (v2, round 2): VALU- and HBM-balanced (2 * 925 FLOP and 2 * 37 bytes per output for Qwen2.5-VL-3B), so both have to run
// near their rates:
//   * a block owns NSTR = 2 stacked [TS = 16 rows] x [TH = 512 columns] output tiles and streams DOWN the rows of both at once: for
//     every input row it walks the C layers in groups of 8, and the 5 x 5 taps of a layer update a ring of 5 output-row
//     accumulators per thread and stream (input row r, kernel row di -> output row r - di): nothing but 2 x 10 accumulators per
//     thread persists, so the halo costs staging only (20 / 16 rows), never arithmetic.  Two streams share every tap read: with
//     one stream the uniform-address LDS reads of the taps (4 LDS cycles per 16 bytes, whatever the lanes do with them) outweigh
//     the dot products;
//   * staging is LDS-DMA (buffer_load_dwordx4 ... lds): a sub-step image is [2 streams][8 layers][66 chunks of 16 B] = the
//     512 + 2 x 8 columns of one row of 8 layers, lane-linear; rows / columns / layers outside the tensor are buffer-descriptor
//     zero fill; double buffered, one barrier per sub-step (2 x 8 layers x 60 dot products per thread);
//   * a thread owns two adjacent columns and reads its 6-element window as three consecutive dwords (stride-1 across lanes: no
//     bank conflicts); the taps are applied with v_dot2c_f32_bf16 on the packed bf16 pairs as they lie in memory -- 3 dot2 per
//     output and kernel row instead of 5 conversions + 5 FMAs -- against tap pairs packed once per block into LDS
//     ((w0,w1) (w2,w3) (w4,0) for the even column, (0,w0) (w1,w2) (w3,w4) for the odd one).  The taps are bf16 in the reference
//     module (Conv2d under .to(bfloat16)); f32 taps handed to the entry point are rounded to bf16 (RNE) here, products are exact
//     in fp32 and accumulated in fp32.
#include "x2i_common.h"
#include "x2i_kernels.h"

namespace {

constexpr int TS = 16;            // output rows per block
constexpr int TH = 512;           // output columns per block (2 per thread)
constexpr int CL = 8;             // layers per sub-step
constexpr int WCH = TH / 8 + 2;   // staged 16-byte chunks per layer row (66): columns [h0 - 8, h0 + TH + 8)
constexpr int STR_CHUNKS = CL * WCH;                   // 528 chunks per stream and sub-step
// NSTR = row streams (stacked output tiles) per block: 2 when the grid is large enough to fill the chip with half as many blocks
// (taps are read once for both streams), 1 for small batches
template <int NSTR> struct Geo {
  static constexpr int STAGE_CHUNKS = NSTR * STR_CHUNKS;
  static constexpr int NPIECE = ((STAGE_CHUNKS + 63) / 64 + 3) / 4 * 4;  // wave-instructions per sub-step, equal for each of the 4 waves
  static constexpr int STAGE_BYTES = NPIECE * 1024;
};
constexpr int MAXC = 64;

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

// (non-template helper on purpose: with the LDS-DMA builtin called directly inside the kernel TEMPLATE, hipcc 7.2 silently dropped the
// host-side instantiation of the kernel -- an undefined __device_stub__ symbol at load time, no diagnostic)
__device__ __forceinline__ void dma_piece(__amdgpu_buffer_rsrc_t rsrc, char* lds_piece, uint32_t voff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_piece, 16, voff, 0, 0, 0);
}

__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}

// One sub-step: issue the LDS-DMA of the NEXT sub-step into `dma_dst`, then apply 8 layers of one input row from `img`.  The three
// LDS regions are __restrict__ parameters of ONE function on purpose: after inlining that is what gives hipcc the alias-scope
// information to see that the window / tap reads cannot touch the buffer the DMA is filling -- without it every ds_read behind an
// LDS-DMA issue gets a conservative s_waitcnt vmcnt(0) and the prefetch is serialised.
// K = r mod 5 (compile time) fixes which ring slot each kernel row feeds; kernel rows di in [LO, HI] are live (the first four input
// rows feed no output above the tile, the last four none below it).
template <int K, int LO, int HI, int NJ_, int NSTR>
__device__ __forceinline__ void conv_substep(const char* __restrict__ img, const uint32_t* __restrict__ wg, char* __restrict__ dma_dst,
                                             __amdgpu_buffer_rsrc_t rsrc, const uint32_t (&voff)[NJ_], int wave, uint32_t win_off,
                                             float (&acc)[NSTR][5][2]) {
#pragma unroll
  for (int j = 0; j < NJ_; ++j) dma_piece(rsrc, dma_dst + (j * 4 + wave) * 1024, voff[j]);
  // taps and windows of layer l + 1 are read (into a second register set) BEFORE the dot products of layer l are issued;
  // sched_barrier pins that order -- left alone hipcc sinks every LDS read to just in front of its first use and the loop
  // runs at LDS latency
  uint4 wa[2][5];
  uint2 wb[2][5];
  uint32_t pw[2][NSTR][3];
  auto load_layer = [&](int l, int set) {
#pragma unroll
    for (int di = LO; di <= HI; ++di) {
      wa[set][di] = *(const uint4*)(wg + (l * 5 + di) * 8);       // uniform address: LDS broadcast, shared by the streams
      wb[set][di] = *(const uint2*)(wg + (l * 5 + di) * 8 + 4);
    }
#pragma unroll
    for (int st = 0; st < NSTR; ++st) {
      const uint32_t* px = (const uint32_t*)(img + (st * CL + l) * (WCH * 16) + win_off);
      pw[set][st][0] = px[0]; pw[set][st][1] = px[1]; pw[set][st][2] = px[2];
    }
  };
  load_layer(0, 0);
#pragma unroll
  for (int l = 0; l < CL; ++l) {
    if (l + 1 < CL) load_layer(l + 1, (l + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int di = LO; di <= HI; ++di) {
      const int slot = (K - di + 5) % 5;
      const uint4 a = wa[l & 1][di];
      const uint2 bq = wb[l & 1][di];
#pragma unroll
      for (int st = 0; st < NSTR; ++st) {
        const uint32_t p0 = pw[l & 1][st][0], p1 = pw[l & 1][st][1], p2 = pw[l & 1][st][2];
        acc[st][slot][0] = dot2(p2, a.z, dot2(p1, a.y, dot2(p0, a.x, acc[st][slot][0])));
        acc[st][slot][1] = dot2(p2, bq.y, dot2(p1, bq.x, dot2(p0, a.w, acc[st][slot][1])));
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int NSTR>
__global__ __launch_bounds__(256) void conv5x5_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, bf16_t* __restrict__ y, int C, int S, int H) {
  constexpr int STAGE_CHUNKS = Geo<NSTR>::STAGE_CHUNKS, NPIECE = Geo<NSTR>::NPIECE, STAGE_BYTES = Geo<NSTR>::STAGE_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][STAGE_BYTES] staging | packed taps [G*CL][5][8] dwords
  uint32_t* wpk = (uint32_t*)(smem + 2 * STAGE_BYTES);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h0 = blockIdx.x * TH, s0 = blockIdx.y * (TS * NSTR), b = blockIdx.z;
  const int G = (C + CL - 1) / CL;
  const long long plane = (long long)S * H;

  // ---- taps: fp32 -> bf16 (RNE) pairs; layout [layer][di][8]: {w0w1, w2w3, w4.0, 0.w0, w1w2, w3w4, 0, 0} (zero for layers >= C)
  for (int i = tid; i < G * CL * 5; i += 256) {
    const int c = i / 5, di = i - c * 5;
    float t[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < C) {
#pragma unroll
      for (int j = 0; j < 5; ++j) t[j] = w[c * 25 + di * 5 + j];
    }
    uint32_t* d = wpk + i * 8;
    d[0] = pack_bf16x2(t[0], t[1]); d[1] = pack_bf16x2(t[2], t[3]); d[2] = pack_bf16x2(t[4], 0.f);
    d[3] = pack_bf16x2(0.f, t[0]); d[4] = pack_bf16x2(t[1], t[2]); d[5] = pack_bf16x2(t[3], t[4]);
    d[6] = 0; d[7] = 0;
  }

  // ---- DMA geometry: piece q = j * 4 + wave (j = 0..2), lane -> chunk p = q * 64 + lane of the sub-step image
  const bf16_t* xb = x + (long long)b * C * plane;
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, (uint32_t)((long long)C * plane * 2), 0x00020000);
  constexpr int NJ = NPIECE / 4;  // pieces per wave
  uint32_t coff[NJ];   // byte offset of (layer-in-group, column) for this lane's chunk; 0x80000000 = outside
  int clayer[NJ], crow0[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int p = (j * 4 + wave) * 64 + lane;
    const int st = p / STR_CHUNKS, pp = p - st * STR_CHUNKS;
    const int l = pp / WCH, ch = pp - l * WCH;
    const int col = h0 - 8 + ch * 8;
    clayer[j] = l;
    crow0[j] = s0 + st * TS - 2;  // first input row of this chunk's stream
    coff[j] = (p < STAGE_CHUNKS && col >= 0 && col < H) ? (uint32_t)(((long long)l * plane + col) * 2) : 0x80000000u;
  }
  // DMA source offsets of sub-step (input row r, layer group g); out-of-range chunks -- and the whole sub-step past the end -- are
  // descriptor zero fill, so the issue is unconditional
  auto offsets = [&](int r, int g, uint32_t (&voff)[NJ]) {
    const uint32_t gbase = (uint32_t)((long long)g * CL * plane * 2);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int s_in = crow0[j] + r;
      const bool ok = s_in >= 0 && s_in < S && coff[j] != 0x80000000u && g * CL + clayer[j] < C;
      voff[j] = ok ? gbase + (uint32_t)((long long)s_in * H * 2) + coff[j] : 0x80000000u;
    }
  };

  float acc[NSTR][5][2];
#pragma unroll
  for (int st = 0; st < NSTR; ++st)
#pragma unroll
    for (int i = 0; i < 5; ++i) acc[st][i][0] = acc[st][i][1] = 0.f;
  const int hcol = h0 + 2 * tid;
  const bool col_ok = hcol < H;  // H % 8 == 0: both columns of the pair are inside together
  const float bv = bias ? bias[0] : 0.f;
  const uint32_t win_off = (uint32_t)(2 * tid + 6) * 2;  // staged column of x[hcol - 2]

  {
    uint32_t v0[NJ];
    offsets(0, 0, v0);
#pragma unroll
    for (int j = 0; j < NJ; ++j) dma_piece(rsrc, smem + (j * 4 + wave) * 1024, v0[j]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // also publishes the packed taps
  int q = 0;
  // rows in groups of five so that the ring slot of (row, kernel row) is a compile-time constant
  auto row = [&](int r, auto kc, auto lo_c, auto hi_c) {
    constexpr int K = decltype(kc)::value;
    constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
    for (int g = 0; g < G; ++g, ++q) {
      uint32_t voff[NJ];
      offsets((g + 1 < G) ? r : r + 1, (g + 1 < G) ? g + 1 : 0, voff);  // (input row TS + 4, past the end: all zero fill)
      conv_substep<K, LO, HI, NJ, NSTR>(smem + (q & 1) * STAGE_BYTES, wpk + g * CL * 40, smem + ((q + 1) & 1) * STAGE_BYTES, rsrc, voff, wave,
                                  win_off, acc);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    // output row o = r - 4 is complete (its last contribution was kernel row 4 of this input row): store it, free its slot
    constexpr int slot = (K + 1) % 5;
    const int o = r - 4;
#pragma unroll
    for (int st = 0; st < NSTR; ++st) {
      const int s = s0 + st * TS + o;
      if (o >= 0 && s < S && col_ok)
        *(uint32_t*)(y + ((long long)b * S + s) * H + hcol) = pack_bf16x2(acc[st][slot][0] + bv, acc[st][slot][1] + bv);
      acc[st][slot][0] = acc[st][slot][1] = 0.f;
    }
  };
  using std::integral_constant;
#define X2I_ROW(R_, K_, LO_, HI_) row(R_, integral_constant<int, K_>{}, integral_constant<int, LO_>{}, integral_constant<int, HI_>{})
  // input row r feeds output row r - di: rows 0..3 only kernel rows di <= r; rows TS..TS+3 only di >= r - TS + 1
  X2I_ROW(0, 0, 0, 0); X2I_ROW(1, 1, 0, 1); X2I_ROW(2, 2, 0, 2); X2I_ROW(3, 3, 0, 3); X2I_ROW(4, 4, 0, 4);
  for (int rr = 1; rr < (TS + 4) / 5 - 1; ++rr) {
    X2I_ROW(5 * rr + 0, 0, 0, 4); X2I_ROW(5 * rr + 1, 1, 0, 4); X2I_ROW(5 * rr + 2, 2, 0, 4); X2I_ROW(5 * rr + 3, 3, 0, 4); X2I_ROW(5 * rr + 4, 4, 0, 4);
  }
  X2I_ROW(TS - 1, 0, 0, 4); X2I_ROW(TS, 1, 1, 4); X2I_ROW(TS + 1, 2, 2, 4); X2I_ROW(TS + 2, 3, 3, 4); X2I_ROW(TS + 3, 4, 4, 4);
#undef X2I_ROW
}

// y[b][i] = (1/C) sum_c scale[c] * x[b][c][i]   (i over the S*H plane), 8 elements per thread
__global__ __launch_bounds__(256) void layer_mean_kernel(const bf16_t* __restrict__ x, const float* __restrict__ scale,
                                                         bf16_t* __restrict__ y, int C, long long plane8) {
  const int b = blockIdx.y;
  const float invC = 1.f / (float)C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < plane8; i += (long long)gridDim.x * 256) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) {
      const bf16x8_t v = *(const bf16x8_t*)(x + (((long long)b * C + c) * plane8 + i) * 8);
      const float sc = scale ? scale[c] : 1.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += sc * bf16_to_f32((bf16_t)v[j]);
    }
    bf16x8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (short)f32_to_bf16(acc[j] * invC);
    *(bf16x8_t*)(y + ((long long)b * plane8 + i) * 8) = o;
  }
}

}  // namespace

int x2i_launch_proj_conv5x5(const void* x, const float* w, const float* bias, void* y, int B, int C, int S, int H,
                            hipStream_t stream) {
  if (!x || !w || !y) return x2i_set_error(X2I_ERR_ARG, "proj_conv5x5: null pointer");
  if (B <= 0 || C <= 0 || C > MAXC || S <= 0 || H <= 0 || H % 8) return x2i_set_error(X2I_ERR_SHAPE, "proj_conv5x5: need C <= 64 and H %% 8 == 0 (C=%d H=%d)", C, H);
  if ((((uintptr_t)x) & 15) || (((uintptr_t)y) & 3)) return x2i_set_error(X2I_ERR_ALIGN, "proj_conv5x5: x must be 16-byte aligned");
  if ((long long)C * S * H * 2 >= 0x7f000000LL) return x2i_set_error(X2I_ERR_SHAPE, "proj_conv5x5: one sample's [C,S,H] slab must stay below 2 GB");
  static_assert((TS + 4) % 5 == 0 && TS >= 6, "row groups of five");
  const int G = (C + CL - 1) / CL;
  const long long blocks1 = (long long)((H + TH - 1) / TH) * ((S + TS - 1) / TS) * B;
  if (blocks1 >= 512) {  // two stacked tiles per block still leave >= 256 blocks
    const int shm = 2 * Geo<2>::STAGE_BYTES + G * CL * 5 * 8 * 4;
    const int rc = x2i_ensure_dynamic_smem((const void*)conv5x5_kernel<2>, shm);
    if (rc) return rc;
    dim3 grid((H + TH - 1) / TH, (S + TS * 2 - 1) / (TS * 2), B);
    hipLaunchKernelGGL(conv5x5_kernel<2>, grid, dim3(256), shm, stream, (const bf16_t*)x, w, bias, (bf16_t*)y, C, S, H);
  } else {
    const int shm = 2 * Geo<1>::STAGE_BYTES + G * CL * 5 * 8 * 4;
    const int rc = x2i_ensure_dynamic_smem((const void*)conv5x5_kernel<1>, shm);
    if (rc) return rc;
    dim3 grid((H + TH - 1) / TH, (S + TS - 1) / TS, B);
    hipLaunchKernelGGL(conv5x5_kernel<1>, grid, dim3(256), shm, stream, (const bf16_t*)x, w, bias, (bf16_t*)y, C, S, H);
  }
  return x2i_check_launch("proj_conv5x5");
}

int x2i_launch_layer_mean(const void* x, const float* scale, void* y, int B, int C, long long plane, hipStream_t stream) {
  if (!x || !y) return x2i_set_error(X2I_ERR_ARG, "layer_mean: null pointer");
  if (B <= 0 || C <= 0 || plane <= 0 || plane % 8) return x2i_set_error(X2I_ERR_SHAPE, "layer_mean: S*H must be a multiple of 8");
  const long long plane8 = plane / 8;
  const long long blocks = (plane8 + 255) / 256;
  hipLaunchKernelGGL(layer_mean_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048), B), dim3(256), 0, stream, (const bf16_t*)x,
                     scale, (bf16_t*)y, C, plane8);
  return x2i_check_launch("layer_mean");
}
