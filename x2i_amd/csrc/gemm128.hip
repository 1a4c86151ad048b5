// 128x128x64 bf16 MFMA GEMM kernel (4 waves, 64 KiB LDS, 2 workgroups per CU): small / text-stream linears, the peeled
// tails of the 256^2 launches and the implicit-GEMM NHWC convolutions with < 256 output channels.  Launcher: gemm.hip.
#include "gemm_device.h"

namespace x2i_gemm {
namespace {

// Epilogue variants are compile-time (ACT, RES, OUTF32, HASC2) so that the accumulator array is only ever indexed
// with constants (a runtime-indexed ext_vector array is demoted to scratch memory by hipcc).
template <int ACT, bool RES, bool OUTF32, bool HASC2, bool CONV>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 bufs][A 16K | B 16K]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int z = blockIdx.y;

  // ---- XCD-aware tile id: block b runs on XCD b%8; give each XCD a contiguous run of logical tile ids
  const int T = p.tilesM * p.tilesN;
  int bid = blockIdx.x;
  {
    const int q = T >> 3, r = T & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // group-of-8 M bands, N fastest across the band
  constexpr int GM = 8;
  const int per_group = GM * p.tilesN;
  const int group = bid / per_group;
  const int first_m = group * GM;
  const int gsize = min(p.tilesM - first_m, GM);
  const int tm = first_m + (bid % per_group) % gsize;
  const int tn = (bid % per_group) / gsize;
  const int m0 = tm * BM, n0 = tn * BN;

  const bf16_t* Az = p.A + (long long)z * p.a_bs;
  // buffer descriptors: num_records = bytes from base to the end of the last valid row
  const uint32_t a_bytes = CONV ? (uint32_t)((long long)p.cH * p.cW * p.cCin * 2)
                                : (uint32_t)(((long long)(p.M - 1) * p.lda + p.K) * 2);
  const uint32_t w_bytes = (uint32_t)(((long long)(p.N - 1) * p.ldw + p.K) * 2);
  __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Az, 0, a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (long long)(z / p.wdiv) * p.w_bs), 0, w_bytes, 0x00020000);

  // per-thread source offsets of its 4 chunks per operand tile (row r, physical chunk c holds logical chunk c^swz)
  uint32_t a_voff[4], w_voff[4];
  int c_oy[4], c_ox[4], c_cl[4];  // CONV: output pixel of each chunk row (times stride, minus pad), logical chunk
  int c_base[4];                  // CONV: byte offset of tap (0,0), channel c_cl (may be negative: masked by c_mask)
  uint32_t c_mask[4];             // CONV: bit (ky*KW + kx) = tap lies inside the image
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int pch = j * 256 + tid;
    const int row = pch >> 3, cphys = pch & 7;
    const int clog = cphys ^ ((row >> 1) & 7);
    // rows past M/N: offset lands beyond num_records -> hardware returns 0
    a_voff[j] = (uint32_t)(((long long)(m0 + row) * p.lda + clog * 8) * 2);
    w_voff[j] = (uint32_t)(((long long)(n0 + row) * p.ldw + clog * 8) * 2);
    if (m0 + row >= p.M) a_voff[j] = 0x80000000u;
    if (n0 + row >= p.N) w_voff[j] = 0x80000000u;
    if (CONV) {
      const int m = m0 + row;
      const int oy = m / p.cOW, ox = m - oy * p.cOW;
      c_oy[j] = oy * p.cStride - p.cPad;
      c_ox[j] = ox * p.cStride - p.cPadW;
      c_cl[j] = clog * 8;
      c_base[j] = ((c_oy[j] * p.cW + c_ox[j]) * p.cCin + c_cl[j]) * 2;
      uint32_t mask = 0;
      if (m < p.M) {
        const int KH = p.K / (p.cKW * p.cCin);
        for (int ky = 0; ky < KH; ++ky)
          for (int kx = 0; kx < p.cKW; ++kx) {
            const int iy = c_oy[j] + ky, ix = c_ox[j] + kx;
            if (iy >= 0 && iy < (p.cH << (p.cUp & 1)) && ix >= 0 && ix < (p.cW << (p.cUp >> 1))) mask |= 1u << (ky * p.cKW + kx);
          }
      }
      c_mask[j] = mask;
    }
  }
  // CONV: gather addresses for one K-tile = one filter tap (ky,kx) and a 64-channel slice of the NHWC input; the tap
  // state advances incrementally (wave-uniform scalars, no divisions in the loop); out-of-image taps (zero padding) are
  // mapped beyond num_records so the DMA writes zeros
  int s_ky = 0, s_kx = 0, s_c0 = 0;
  auto conv_offsets = [&]() {
    const int tap = s_ky * p.cKW + s_kx;
    if (p.cUp) {
      // x2 nearest upsampling fused into the gather: source pixel = coordinate >> 1 on the upsampled grid
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int iy = c_oy[j] + s_ky, ix = c_ox[j] + s_kx;
        a_voff[j] = ((c_mask[j] >> tap) & 1) ? (uint32_t)((((iy >> (p.cUp & 1)) * p.cW + (ix >> (p.cUp >> 1))) * p.cCin + s_c0 + c_cl[j]) * 2) : 0x80000000u;
      }
    } else {
      const int toff = ((s_ky * p.cW + s_kx) * p.cCin + s_c0) * 2;
#pragma unroll
      for (int j = 0; j < 4; ++j) a_voff[j] = ((c_mask[j] >> tap) & 1) ? (uint32_t)(c_base[j] + toff) : 0x80000000u;
    }
    s_c0 += BK;
    if (s_c0 >= p.cCin) {
      s_c0 = 0;
      if (++s_kx == p.cKW) {
        s_kx = 0;
        ++s_ky;
      }
    }
  };
  if (CONV) conv_offsets();

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: lane reads row (lane&15) (+16*i), logical chunk kk*4 + (lane>>4)
  const int frow = lane & 15;
  const int fswz = (frow >> 1) & 7;
  uint32_t frag_off[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) frag_off[kk] = frow * 128 + (((kk * 4 + (lane >> 4)) ^ fswz) << 4);
  const uint32_t a_frag_base = wm * 64 * 128;  // bytes: wave's first A row
  const uint32_t b_frag_base = wn * 64 * 128;

  const int nk = p.K / BK;
  stage_tile(a_rsrc, smem, a_voff, 0, wave);
  stage_tile(w_rsrc, smem + TILE_BYTES, w_voff, 0, wave);
  // hipcc does not count LDS-DMA (buffer_load ... lds) as pending LDS writes at a barrier: wait explicitly
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // all tiles but the last stage their successor unconditionally (one basic block per K-step); the last one only computes.
  // Fragment pipeline (pinned with sched_barrier -- hipcc otherwise sinks every ds_read to just in front of its MFMAs and waits
  // lgkmcnt(0) for it): the k-half-0 fragments of a tile are read right after the barrier that publishes it, under the issue of the
  // next tile's eight DMA pieces; the k-half-1 fragments are read before the 16 MFMAs of k-half 0 start.
  bf16x8_t af[2][4], wf[2][4];
  auto load_frags = [&](const char* tile, int kk) {
#pragma unroll
    for (int i = 0; i < 4; ++i) af[kk][i] = *(const bf16x8_t*)(tile + a_frag_base + i * 2048 + frag_off[kk]);
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[kk][j] = *(const bf16x8_t*)(tile + TILE_BYTES + b_frag_base + j * 2048 + frag_off[kk]);
  };
  load_frags(smem, 0);
  auto ktile = [&](int kt, auto stage_next) {
    char* cur = smem + (kt & 1) * 2 * TILE_BYTES;
    char* nxt = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
    if constexpr (decltype(stage_next)::value) {
      const uint32_t koff = (uint32_t)(kt + 1) * BK * 2;
      if (CONV) conv_offsets();
      stage_tile(a_rsrc, nxt, a_voff, CONV ? 0u : koff, wave);
      stage_tile(w_rsrc, nxt + TILE_BYTES, w_voff, koff, wave);
    }
    load_frags(cur, 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the next tile has landed
    __syncthreads();                                  // ... everyone's has, and everyone is done reading `cur`
    if constexpr (decltype(stage_next)::value) load_frags(nxt, 0);
  };
  for (int kt = 0; kt < nk - 1; ++kt) ktile(kt, std::true_type{});
  ktile(nk - 1, std::false_type{});

  if constexpr (ACT == X2I_ACT_NONE && !RES && !OUTF32 && !HASC2 && !CONV) {
    if (p.q_on) {
      epilogue_qkv<4, 2, 256>(p, acc, z, m0, n0, wm, wn, lane, tid, smem);
      return;
    }
  }
  if constexpr (!OUTF32) {
    // whole-line stores through LDS (see epilogue_store_lds); needs 16-byte aligned rows and N % 8 == 0
    if (((p.N | p.ldc) & 7) == 0 && (!RES || (p.ldr & 3) == 0) && ((((uintptr_t)p.C) | ((uintptr_t)p.C2)) & 15) == 0 && (p.c_bs & 7) == 0) {
      epilogue_store_lds<ACT, RES, HASC2, 4, CONV>(p, acc, z, m0 + wm * 64, n0 + wn * 64, lane, smem + wave * (EPI_WAVE_BYTES / 2));
      return;
    }
  }
  epilogue_store<ACT, RES, OUTF32, HASC2, 4, 4>(p, acc, z, m0 + wm * 64 + (lane & 15), n0 + wn * 64 + (lane >> 4) * 4);
}

}  // namespace

kern_t pick_gemm128(int act, bool res, bool f32, bool c2, bool conv) {
  kern_t k = nullptr;
#define X2I_PICK(A_, R_, F_, C_) k = conv ? gemm_bf16_kernel<A_, R_, F_, C_, true> : gemm_bf16_kernel<A_, R_, F_, C_, false>;
  X2I_GEMM_PICK_TABLE(X2I_PICK)
#undef X2I_PICK
  return k;
}

}  // namespace x2i_gemm
