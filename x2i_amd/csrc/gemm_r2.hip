// "Two residents" bf16 MFMA GEMM: 256 x 128 output tile per workgroup, 4 waves stacked along M (64 x 128 wave tiles, 128 accumulator
// registers), <= 256 registers per wave and 64 KiB of LDS per workgroup, so that TWO workgroups share a CU: while one is in its epilogue
// the other one's K-loop owns the matrix pipe.  The answer to "an epilogue that runs beside the next K-loop"  with
// the hardware's wave scheduler as the interleaver instead of a generated epilogue stream.  A fragments come straight from global memory
// (a wave's rows are its own), W through a 4-stage LDS ring; K-loop = one generated asm statement (gen_gemm_r2.py -> gemm_r2_loop.inc).
// Same MFMA, same k order per output as every other bf16 GEMM kernel here: bit-identical results (tested).  Price: the 256 x 128 tile
// fetches 1.5 x the operand bytes per FLOP of the 256^2 tile from L2.  Launcher: gemm.hip (option gemm_r2).
#include "gemm_device.h"
#include "gemm_r2_loop.inc"

namespace x2i_gemm {
namespace {

constexpr int R2_BM = 256, R2_BN = 128;

// VAR (measurement library only): 1 = no A loads, 2 = no W DMA, 3 = neither, 7 = MFMA only (wrong results by design)
template <int ACT, bool RES, bool HASC2, int VAR = 0>
__global__ __launch_bounds__(256, 2) void gemm_r2_bf16_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [4 stages][W image 16 KiB]; epilogue staging afterwards
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int z = blockIdx.y;

  const int T = p.tilesM * p.tilesN;
  int bid = blockIdx.x;
  {
    const int q = T >> 3, r = T & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int GM = p.gm;
  const int per_group = GM * p.tilesN;
  const int group = bid / per_group;
  const int first_m = group * GM;
  const int gsize = min(p.tilesM - first_m, GM);
  const int tm = first_m + (bid % per_group) % gsize;
  const int tn = (bid % per_group) / gsize;
  const int m0 = tm * R2_BM, n0 = tn * R2_BN;
  const int m_wave = m0 + wave * 64;

  const bf16_t* Az = p.A + (long long)z * p.a_bs;
  const uint32_t a_bytes = (uint32_t)(((long long)(p.M - 1) * p.lda + p.K) * 2);
  const uint32_t w_bytes = (uint32_t)(((long long)(p.N - 1) * p.ldw + p.K) * 2);
  __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Az, 0, a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (long long)z * p.w_bs), 0, w_bytes, 0x00020000);

  // A fragment i of this wave: lane -> row 16 i + (lane & 15), 16 bytes at k element 8 (lane >> 4) of the k-half
  uint32_t va[4], vw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = m_wave + i * 16 + (lane & 15);
    va[i] = (row < p.M) ? (uint32_t)(((long long)row * p.lda + (lane >> 4) * 8) * 2) : 0x80000000u;
  }
  // W piece jj of this wave = row group g = jj*4 + wave (rows 8g .. 8g+7 of the tile); lane -> (k-half, row in group, physical chunk)
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int g = jj * 4 + wave;
    const int khl = lane >> 5, r = (lane >> 2) & 7, cphys = lane & 3;
    const int row = g * 8 + r;
    const int kel = khl * 32 + ((cphys ^ (3 * (g & 1))) << 3);
    vw[jj] = (n0 + row < p.N) ? (uint32_t)(((long long)(n0 + row) * p.ldw + kel) * 2) : 0x80000000u;
  }
  const int frow = lane & 15;
  const uint32_t frag = (frow >> 3) * 1024 + (frow & 7) * 64 + (((lane >> 4) ^ (3 * ((frow >> 3) & 1))) << 4);
  const uint32_t lw = (uint32_t)(uintptr_t)smem + frag;   // + stage * 16384 + j * 2048 + 512 for k-half 1 (immediates of the loop)
  const uint32_t dma = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem + wave * 1024);
  const int nk = p.K / BK;

  f32x4_t acc[2][4][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[h][i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  {
    bf16x8_t wr[8], af[2][2][4];
    uint32_t s_koffa, s_koffw, s_it;
#define X2I_R2_LOOP_STMT(TEXT)                                                                                            \
    asm volatile(TEXT                                                                                                     \
                 : X2I_GEMM_R2_OPS_ACC(acc), X2I_GEMM_R2_OPS_FRAG(wr, af), [koffa] "=&s"(s_koffa), [koffw] "=&s"(s_koffw),      \
                   [it] "=&s"(s_it)                                                                                          \
                 : X2I_GEMM_R2_OPS_VOFF(va, vw), [lw] "v"(lw), [dma] "s"(dma), [ra] "s"(a_rsrc), [rw] "s"(w_rsrc), [nk] "s"(nk) \
                 : "memory", "scc", "m0")
#ifdef X2I_ABLATION
    if constexpr (VAR == 1) X2I_R2_LOOP_STMT(X2I_GEMM_R2_LOOP_NOA);
    else if constexpr (VAR == 2) X2I_R2_LOOP_STMT(X2I_GEMM_R2_LOOP_NOW);
    else if constexpr (VAR == 3) X2I_R2_LOOP_STMT(X2I_GEMM_R2_LOOP_NOMEM);
    else if constexpr (VAR == 7) X2I_R2_LOOP_STMT(X2I_GEMM_R2_LOOP_MFMA);
    else
#endif
    X2I_R2_LOOP_STMT(X2I_GEMM_R2_LOOP);
#undef X2I_R2_LOOP_STMT
  }
  // the statement ends with vmcnt(0) + s_barrier: every wave is done with the ring, which now serves as epilogue staging
#ifdef X2I_ABLATION
  if (p.act2 == 77) {  // measurement only: no epilogue at all -- what the K-loops alone take
    asm volatile("" ::X2I_GEMM_R2_OPS_ACC_IN(acc));
    return;
  }
#endif
  if (((p.N | p.ldc) & 7) == 0 && (!RES || (p.ldr & 3) == 0) && ((((uintptr_t)p.C) | ((uintptr_t)p.C2)) & 15) == 0 && (p.c_bs & 7) == 0) {
    char* wl = smem + wave * (64 * EPI_ROW_BYTES);   // 9 KiB per wave, reused by the second half (the region is private: LDS operations of one wave complete in order)
    epilogue_store_lds<ACT, RES, HASC2, 4>(p, acc[0], z, m_wave, n0, lane, wl);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    epilogue_store_lds<ACT, RES, HASC2, 4>(p, acc[1], z, m_wave, n0 + 64, lane, wl);
    return;
  }
  epilogue_store<ACT, RES, false, HASC2, 4, 4>(p, acc[0], z, m_wave + (lane & 15), n0 + (lane >> 4) * 4);
  epilogue_store<ACT, RES, false, HASC2, 4, 4>(p, acc[1], z, m_wave + (lane & 15), n0 + 64 + (lane >> 4) * 4);
}

}  // namespace

kern_t pick_gemm_r2(int act, bool res, bool f32, bool c2, int var) {
  if (f32) return nullptr;
#ifdef X2I_ABLATION
  if (var && act == X2I_ACT_NONE && !res && !c2) {
    switch (var) {
      case 1: return gemm_r2_bf16_kernel<X2I_ACT_NONE, false, false, 1>;
      case 2: return gemm_r2_bf16_kernel<X2I_ACT_NONE, false, false, 2>;
      case 3: return gemm_r2_bf16_kernel<X2I_ACT_NONE, false, false, 3>;
      case 7: return gemm_r2_bf16_kernel<X2I_ACT_NONE, false, false, 7>;
    }
    return nullptr;
  }
#endif
  (void)var;
  if (res) return (act == X2I_ACT_NONE && !c2) ? (kern_t)gemm_r2_bf16_kernel<X2I_ACT_NONE, true, false> : nullptr;
  if (c2) return act == X2I_ACT_NONE ? (kern_t)gemm_r2_bf16_kernel<X2I_ACT_NONE, false, true> : nullptr;
  switch (act) {
    case X2I_ACT_NONE: return gemm_r2_bf16_kernel<X2I_ACT_NONE, false, false>;
    case X2I_ACT_GELU_TANH: return gemm_r2_bf16_kernel<X2I_ACT_GELU_TANH, false, false>;
    case X2I_ACT_SILU: return gemm_r2_bf16_kernel<X2I_ACT_SILU, false, false>;
  }
  return nullptr;
}

}  // namespace x2i_gemm
