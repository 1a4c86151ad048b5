// Measurement-only GEMM kernels: the k-half-unit predecessor of the product 256^2 kernel and its ablation variants ("wrong
// results by design" for ABL != 0; tools/gemm_ablate.py).  Compiled only with -DX2I_ABLATION into libx2i_hip_ablate.so;
// the product library libx2i_hip.so contains none of this.
#ifdef X2I_ABLATION
#include "gemm_device.h"

namespace x2i_gemm {
namespace {

// ------------------------------------------------------------------------------------------------------------
// 256x256x64 pipelined kernel (8 waves, 1 workgroup per CU, 128 KiB LDS) for the large DiT GEMMs.
//
// A K-tile (64 deep) is staged as FOUR 16 KiB units -- A[256 rows][k 0..31], W[256][0..31], A[256][32..63],
// W[256][32..63] -- and consumed in four phases of 16 MFMAs per wave: (k-half 0, m-half 0), (0,1), (1,0), (1,1).
// Phase p of tile t also issues the LDS-DMA of unit p of tile t+1 into the other LDS buffer, so a unit is needed
// >= 3 phases after it was issued: the main loop only ever waits with a COUNTED `s_waitcnt vmcnt(4)` (two younger
// units stay in flight across the barrier) and never drains the load queue.  Two barriers per K-tile (phases 0 and 2:
// the points where freshly landed units are first read).  Wave (wm, wn) owns rows wm*128.., cols wn*64..: 8x4 MFMA
// tiles = 128 accumulator registers; per K-tile it issues 24 ds_read_b128 for 64 MFMAs.
// Unit image: [256 rows][4 chunks of 16 B]; 4 rows share a 256-byte bank row, so the conflict-free swizzle is
// chunk ^ (3 * ((row >> 3) & 1)) (derived for the ds_read_b128 lane groups {0-3,12-15,20-27}, ...).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void stage_unit(__amdgpu_buffer_rsrc_t rsrc, char* lds_unit, const uint32_t (&voff)[2],
                                           uint32_t koff_bytes, int wave) {
#pragma unroll
  for (int j = 0; j < 2; ++j)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds_unit + j * 8192 + wave * 1024),
                                             16, voff[j], koff_bytes, 0, 0);
}

// ABL: ablation bits for tools/gemm_ablate.py (wrong results by design): 1 = no ds_reads after the first K-tile,
// 2 = no barriers / load waits, 4 = no global->LDS DMA after the prologue.  ABL = 0 is the product kernel.
template <int ACT, bool RES, bool OUTF32, bool HASC2, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm256_bf16_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 tiles][A.k0 | W.k0 | A.k1 | W.k1]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int z = blockIdx.y;

  const int T = p.tilesM * p.tilesN;
  int bid = blockIdx.x;
  {
    const int q = T >> 3, r = T & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  constexpr int GM = 4;  // 4 x 8 tile patch per XCD (32 CUs)
  const int per_group = GM * p.tilesN;
  const int group = bid / per_group;
  const int first_m = group * GM;
  const int gsize = min(p.tilesM - first_m, GM);
  const int tm = first_m + (bid % per_group) % gsize;
  const int tn = (bid % per_group) / gsize;
  const int m0 = tm * BM2, n0 = tn * BN2;

  const bf16_t* Az = p.A + (long long)z * p.a_bs;
  const uint32_t a_bytes = (uint32_t)(((long long)(p.M - 1) * p.lda + p.K) * 2);
  const uint32_t w_bytes = (uint32_t)(((long long)(p.N - 1) * p.ldw + p.K) * 2);
  __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Az, 0, a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (long long)z * p.w_bs), 0, w_bytes, 0x00020000);

  uint32_t a_voff[2], w_voff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int pch = j * 512 + tid;
    const int row = pch >> 2, cphys = pch & 3;
    const int clog = cphys ^ (3 * ((row >> 3) & 1));
    a_voff[j] = (uint32_t)(((long long)(m0 + row) * p.lda + clog * 8) * 2);
    w_voff[j] = (uint32_t)(((long long)(n0 + row) * p.ldw + clog * 8) * 2);
    if (m0 + row >= p.M) a_voff[j] = 0x80000000u;
    if (n0 + row >= p.N) w_voff[j] = 0x80000000u;
  }

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // fragment address inside a unit: row r -> r*64 bytes, logical chunk (lane>>4) -> physical chunk ^ (3*((r>>3)&1));
  // all fragment rows of a lane are (lane&15) + multiple of 16, so the swizzle term is lane-constant
  const int frow = lane & 15;
  const uint32_t frag = frow * 64 + (((lane >> 4) ^ (3 * ((frow >> 3) & 1))) << 4);
  const uint32_t a_base = wm * 128 * 64 + frag;  // + i*1024 per m-tile
  const uint32_t b_base = wn * 64 * 64 + frag;   // + j*1024 per n-tile

  const int nk = p.K / BK;
  // ---- software pipeline (see header comment): unit u = 4*tile + {0:A.k0, 1:W.k0, 2:A.k1, 3:W.k1} lives in LDS slot u % 8
  // and is DMA-issued LEAD = 5 phases before the phase with the same number; fragments of phase G+1 are read from LDS
  // while the MFMAs of phase G run (two register sets); barriers only at odd phases, where freshly landed units are
  // first read.  In flight across a barrier: (LEAD - 3) = 2 units = 4 loads per thread (counted vmcnt, never 0).
  constexpr int LEAD = 5;
  const int total_units = 4 * nk;
  auto issue_unit = [&](int u) {
    if ((ABL & 4) && u >= LEAD) return;
    const int t = u >> 2, pu = u & 3;
    char* dst = smem + (u & 7) * UNIT_BYTES;
    // ABL 256: every unit re-reads k = 0 (cache-hot source) -- separates "data arrives late" from "issue / LDS-write cost"
    const uint32_t koff = (ABL & 256) ? 0u : (uint32_t)(t * BK + (pu >> 1) * 32) * 2;
    if (pu & 1) stage_unit(w_rsrc, dst, w_voff, koff, wave);
    else stage_unit(a_rsrc, dst, a_voff, koff, wave);
  };
  auto wait_units_in_flight = [&](int units) {  // wave-uniform small integer -> immediate vmcnt
    if (units >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (units == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
#pragma unroll
  for (int u = 0; u < LEAD; ++u)
    if (u < total_units) issue_unit(u);

  bf16x8_t wf[2][4], af[2][4];
  // first fragments: units 0 (A.k0) and 1 (W.k0) of tile 0
  wait_units_in_flight(min(LEAD, total_units) - 2);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int j = 0; j < 4; ++j) wf[0][j] = *(const bf16x8_t*)(smem + 1 * UNIT_BYTES + b_base + j * 1024);
#pragma unroll
  for (int i = 0; i < 4; ++i) af[0][i] = *(const bf16x8_t*)(smem + 0 * UNIT_BYTES + a_base + i * 1024);

  // One K-tile (4 phases).  STEADY = not one of the last two tiles: every unit issue and every wait is unconditional, so the
  // whole tile is ONE basic block and the compiler is free to place the DMA pieces and LDS reads among the MFMAs.
  auto ktile = [&](int kt, auto steady_c) {
    constexpr bool STEADY = decltype(steady_c)::value;
    const char* cur = smem + (kt & 1) * TILE2_BYTES;
    const char* nxt = smem + ((kt + 1) & 1) * TILE2_BYTES;
    const bool more = STEADY || (kt + 1 < nk);
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      const int G = 4 * kt + ph;
      const int kh = ph >> 1, mh = ph & 1;
      if (ph & 1) {
        // odd phase: the units read below ((A.k1,W.k1) of this tile at ph 1, (A.k0,W.k0) of the next at ph 3) must have
        // landed for every wave; units issued so far = G-1+LEAD, needed = G+2
        const bool need = (ph == 1) || more;
        if (need && !((ABL & 2) && kt > 0)) {
          if constexpr (STEADY) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          } else {
            const int last_issued = min(G - 1 + LEAD, total_units - 1);
            wait_units_in_flight(last_issued - (G + 2));
          }
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
      }
      if (STEADY || G + LEAD < total_units) issue_unit(G + LEAD);
      // ---- LDS -> registers for phase G+1
      if (!((ABL & 1) && kt > 0)) {
        if (ph == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) af[1][i] = *(const bf16x8_t*)(cur + 0 * UNIT_BYTES + a_base + (4 + i) * 1024);
        } else if (ph == 1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) wf[1][j] = *(const bf16x8_t*)(cur + 3 * UNIT_BYTES + b_base + j * 1024);
#pragma unroll
          for (int i = 0; i < 4; ++i) af[0][i] = *(const bf16x8_t*)(cur + 2 * UNIT_BYTES + a_base + i * 1024);
        } else if (ph == 2) {
#pragma unroll
          for (int i = 0; i < 4; ++i) af[1][i] = *(const bf16x8_t*)(cur + 2 * UNIT_BYTES + a_base + (4 + i) * 1024);
        } else if (more) {
#pragma unroll
          for (int j = 0; j < 4; ++j) wf[0][j] = *(const bf16x8_t*)(nxt + 1 * UNIT_BYTES + b_base + j * 1024);
#pragma unroll
          for (int i = 0; i < 4; ++i) af[0][i] = *(const bf16x8_t*)(nxt + 0 * UNIT_BYTES + a_base + i * 1024);
        }
      }
      // ---- 16 MFMAs of phase G on the register set loaded during phase G-1
      constexpr int VAR = ABL >> 4;  // scheduling experiments (ABL >= 16): 1 = setprio, 2 = sched_group interleave, 3 = both
      if (VAR & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[mh * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kh][j], af[mh][i], acc[mh * 4 + i][j], 0, 0, 0);
      if (VAR & 1) __builtin_amdgcn_s_setprio(0);
      if (VAR & 2) {
        // interleave: 2 MFMA, 1 DS read, ... ; the two DMA pieces after the 4th and 10th MFMA
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);  // MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
          if (k == 1 || k == 4) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read (LDS-DMA)
        }
      }
    }
  };
  {
    int kt = 0;
    if (!(ABL & (7 | 128))) {  // ABL 128: A/B switch, run every tile through the general (branchy) form
      for (; kt < nk - 2; ++kt) ktile(kt, std::true_type{});
    }
    for (; kt < nk; ++kt) ktile(kt, std::false_type{});
  }
  if (ABL & 8) {  // ablation: no epilogue (keep the accumulators alive with one predicated store)
    float sacc = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (sacc == 12345.678f) ((float*)p.C)[tid] = sacc;
    return;
  }
  if constexpr (ACT == X2I_ACT_NONE && !RES && !OUTF32 && !HASC2 && ABL == 0) {
    if (p.q_on) {
      __syncthreads();  // every wave is done reading the operand ring before it is reused as staging space
      epilogue_qkv<8, 4, 512>(p, acc, z, m0, n0, wm, wn, lane, tid, smem);
      return;
    }
  }
  if constexpr (!OUTF32) {
    // whole-line stores through LDS need 16-byte aligned rows and N % 8 == 0 (wave-uniform test)
    if (((p.N | p.ldc) & 7) == 0 && (!RES || (p.ldr & 3) == 0) && ((((uintptr_t)p.C) | ((uintptr_t)p.C2)) & 15) == 0 && (p.c_bs & 7) == 0) {
      __syncthreads();  // every wave is done reading the operand ring before it is reused as staging space
      epilogue_store_lds<ACT, RES, HASC2, 8>(p, acc, z, m0 + wm * 128, n0 + wn * 64, lane, smem + wave * EPI_WAVE_BYTES);
      return;
    }
  }
  epilogue_store<ACT, RES, OUTF32, HASC2, 8, 4>(p, acc, z, m0 + wm * 128 + (lane & 15), n0 + wn * 64 + (lane >> 4) * 4);
}

}  // namespace

kern_t pick_gemm256u(int act, bool res, bool f32, bool c2, int abl) {
  kern_t k = nullptr;
  if (abl) {
    switch (abl) {
#define X2I_ABL(N_) case N_: return gemm256_bf16_kernel<X2I_ACT_NONE, false, false, false, N_>;
      X2I_ABL(1) X2I_ABL(2) X2I_ABL(3) X2I_ABL(4) X2I_ABL(7) X2I_ABL(8) X2I_ABL(16) X2I_ABL(32) X2I_ABL(48) X2I_ABL(128) X2I_ABL(256)
#undef X2I_ABL
      default: return nullptr;
    }
  }
#define X2I_PICK(A_, R_, F_, C_) k = gemm256_bf16_kernel<A_, R_, F_, C_>;
  X2I_GEMM_PICK_TABLE(X2I_PICK)
#undef X2I_PICK
  return k;
}

}  // namespace x2i_gemm
#endif  // X2I_ABLATION
