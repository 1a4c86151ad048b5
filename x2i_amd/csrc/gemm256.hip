// 256x256x64 bf16 MFMA GEMM kernel, full-line LDS-DMA staging (8 waves, 144 KiB LDS, 1 workgroup per CU): the large DiT
// linears and the implicit-GEMM convolutions with >= 256 output channels.  Launcher: gemm.hip.
#include "gemm_device.h"

namespace x2i_gemm {
namespace {

// ------------------------------------------------------------------------------------------------------------
// 256x256x64, full-line staging ("L" form).  Same tile, waves, fragment pipeline and epilogues as gemm256_bf16_kernel;
// what changes is the shape of an LDS-DMA piece: a wave instruction fetches 8 rows x 128 B (whole cache lines: both
// k-halves of a row) instead of 16 rows x 64 B, halving the number of lines the texture path looks up per byte staged.
// The LDS image is lane-linear, so the two k-halves of those 8 rows land in the two 512-byte halves of the piece:
//   operand image (32 KiB) = [32 row groups][k-half][8 rows][4 chunks of 16 B]   (chunk swizzle ^ 3*(group & 1) as before:
//   group stride 1 KiB and k-half stride 512 B are both multiples of the 256-byte bank period, so the fragment reads hit
//   the same banks as in the k-half-major image).
// With both k-halves of a row arriving together there are no k-half units to consume progressively: a K-tile is a plain
// double buffer -- the whole next tile (4 A + 4 W pieces per wave) is issued during phases 0 and 1 and must have landed
// by the barrier at phase 3, where its first fragments are read; one barrier per K-tile.
// ------------------------------------------------------------------------------------------------------------
// CONV = true: A is the implicit-GEMM gather of an NHWC image (one filter tap x 64 channels per K-tile, exactly one 128-byte
// line per output pixel and piece row), as in the 128^2 kernel; used for the convolutions with >= 256 output channels.
template <int ACT, bool RES, bool OUTF32, bool HASC2, bool CONV = false>
__global__ __launch_bounds__(512, 2) void gemm256l_bf16_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 tiles][A image 32 KiB | W image 32 KiB]
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
  const int GM = p.gm;  // tile-rows per group: the XCD's 32 concurrent tiles form a GM x 32/GM patch (chosen by the launcher)
  const int per_group = GM * p.tilesN;
  const int group = bid / per_group;
  const int first_m = group * GM;
  const int gsize = min(p.tilesM - first_m, GM);
  const int tm = first_m + (bid % per_group) % gsize;
  const int tn = (bid % per_group) / gsize;
  const int m0 = tm * BM2, n0 = tn * BN2;

  const bf16_t* Az = p.A + (long long)z * p.a_bs;
  const uint32_t a_bytes = CONV ? (uint32_t)((long long)p.cH * p.cW * p.cCin * 2) : (uint32_t)(((long long)(p.M - 1) * p.lda + p.K) * 2);
  const uint32_t w_bytes = (uint32_t)(((long long)(p.N - 1) * p.ldw + p.K) * 2);
  __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Az, 0, a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (long long)(z / p.wdiv) * p.w_bs), 0, w_bytes, 0x00020000);

  // piece q = jj*8 + wave (jj = 0..3) covers row group q (rows 8q..8q+7); lane -> (k-half, row in group, physical chunk)
  uint32_t a_voff[4], w_voff[4];
  int c_base[4], c_oy[4], c_ox[4];  // CONV: byte offset of tap (0,0) for this lane's (pixel, channel chunk); pixel origin
  uint32_t c_mask[4];               // CONV: bit (ky*KW + kx) = tap lies inside the image
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int g = jj * 8 + wave;
    const int khl = lane >> 5, r = (lane >> 2) & 7, cphys = lane & 3;
    const int row = g * 8 + r;
    const int kel = khl * 32 + ((cphys ^ (3 * (g & 1))) << 3);
    a_voff[jj] = (m0 + row < p.M) ? (uint32_t)(((long long)(m0 + row) * p.lda + kel) * 2) : 0x80000000u;
    w_voff[jj] = (n0 + row < p.N) ? (uint32_t)(((long long)(n0 + row) * p.ldw + kel) * 2) : 0x80000000u;
    if constexpr (CONV) {
      const int m = m0 + row;
      const int oy = m / p.cOW, ox = m - oy * p.cOW;
      c_oy[jj] = oy * p.cStride - p.cPad;
      c_ox[jj] = ox * p.cStride - p.cPadW;
      c_base[jj] = ((c_oy[jj] * p.cW + c_ox[jj]) * p.cCin + kel) * 2;
      uint32_t mask = 0;
      if (m < p.M) {
        const int KH = p.K / (p.cKW * p.cCin);
        for (int ky = 0; ky < KH; ++ky)
          for (int kx = 0; kx < p.cKW; ++kx) {
            const int iy = c_oy[jj] + ky, ix = c_ox[jj] + kx;
            if (iy >= 0 && iy < (p.cH << (p.cUp & 1)) && ix >= 0 && ix < (p.cW << (p.cUp >> 1))) mask |= 1u << (ky * p.cKW + kx);
          }
      }
      c_mask[jj] = mask;
      c_base[jj] -= kel * 2;  // keep the chunk offset separate: the x2-upsample form rebuilds the pixel part
    }
  }
  const int c_kel = ((lane >> 5) * 32 + (((lane & 3) ^ (3 * (wave & 1))) << 3)) * 2;  // CONV: this lane's channel-chunk bytes
  int s_ky = 0, s_kx = 0, s_c0 = 0;  // CONV: tap / channel slice of the next K-tile to stage (tiles are staged in order)

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // fragment (16-row block b of the operand image, k-half kh): rows 16b + frow -> group 2b + (frow >> 3)
  const int frow = lane & 15;
  const uint32_t frag = (frow >> 3) * 1024 + (frow & 7) * 64 + (((lane >> 4) ^ (3 * ((frow >> 3) & 1))) << 4);
  const uint32_t a_base = wm * 8 * 2048 + frag;          // + i*2048 per m-tile, + kh*512
  const uint32_t b_base = 32768 + wn * 4 * 2048 + frag;  // + j*2048 per n-tile, + kh*512

  const int nk = p.K / BK;
  auto issue_a = [&](int t) {
    char* dst = smem + (t & 1) * TILE2_BYTES;
    if constexpr (CONV) {
      const int tap = s_ky * p.cKW + s_kx;
      if (p.cUp) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int iy = c_oy[jj] + s_ky, ix = c_ox[jj] + s_kx;
          a_voff[jj] = ((c_mask[jj] >> tap) & 1) ? (uint32_t)((((iy >> (p.cUp & 1)) * p.cW + (ix >> (p.cUp >> 1))) * p.cCin + s_c0) * 2 + c_kel) : 0x80000000u;
        }
      } else {
        const int toff = ((s_ky * p.cW + s_kx) * p.cCin + s_c0) * 2 + c_kel;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) a_voff[jj] = ((c_mask[jj] >> tap) & 1) ? (uint32_t)(c_base[jj] + toff) : 0x80000000u;
      }
      s_c0 += BK;
      if (s_c0 >= p.cCin) {
        s_c0 = 0;
        if (++s_kx == p.cKW) {
          s_kx = 0;
          ++s_ky;
        }
      }
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(dst + (jj * 8 + wave) * 1024), 16,
                                               a_voff[jj], CONV ? 0u : (uint32_t)(t * BK) * 2, 0, 0);
  };
  auto issue_w = [&](int t) {
    char* dst = smem + (t & 1) * TILE2_BYTES + 32768;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(dst + (jj * 8 + wave) * 1024), 16,
                                               w_voff[jj], (uint32_t)(t * BK) * 2, 0, 0);
  };
  issue_a(0);
  issue_w(0);
  bf16x8_t wf[2][4], af[2][4];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int j = 0; j < 4; ++j) wf[0][j] = *(const bf16x8_t*)(smem + b_base + j * 2048);
#pragma unroll
  for (int i = 0; i < 4; ++i) af[0][i] = *(const bf16x8_t*)(smem + a_base + i * 2048);

  auto ktile = [&](int kt, auto more_c) {
    constexpr bool MORE = decltype(more_c)::value;  // a successor tile exists (everything below is then unconditional)
    const char* cur = smem + (kt & 1) * TILE2_BYTES;
    const char* nxt = smem + ((kt + 1) & 1) * TILE2_BYTES;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      const int kh = ph >> 1, mh = ph & 1;
      if (ph == 3 && MORE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next tile was issued two phases ago; nothing younger in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if (MORE && ph == 0) issue_a(kt + 1);  // spread over two phases: all eight pieces in phase 0 measured 3-4 % slower
      if (MORE && ph == 1) issue_w(kt + 1);
      // ---- LDS -> registers for the next phase
      if (ph == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) af[1][i] = *(const bf16x8_t*)(cur + a_base + (4 + i) * 2048);
      } else if (ph == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[1][j] = *(const bf16x8_t*)(cur + b_base + j * 2048 + 512);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[0][i] = *(const bf16x8_t*)(cur + a_base + i * 2048 + 512);
      } else if (ph == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) af[1][i] = *(const bf16x8_t*)(cur + a_base + (4 + i) * 2048 + 512);
      } else if (MORE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[0][j] = *(const bf16x8_t*)(nxt + b_base + j * 2048);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[0][i] = *(const bf16x8_t*)(nxt + a_base + i * 2048);
      }
      // pin "this phase's DMA issue and the next phase's fragment reads first, then the 16 MFMAs": hipcc otherwise sinks each
      // ds_read to just in front of its first MFMA and waits lgkmcnt(0/1) for it (+1.3 % over the DiT shapes, +2.5 % at K = 3072)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[mh * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kh][j], af[mh][i], acc[mh * 4 + i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (int kt = 0; kt < nk - 1; ++kt) ktile(kt, std::true_type{});
  ktile(nk - 1, std::false_type{});

  if constexpr (ACT == X2I_ACT_NONE && !RES && !OUTF32 && !HASC2 && !CONV) {
    if (p.q_on) {
      __syncthreads();
      epilogue_qkv<8, 4, 512>(p, acc, z, m0, n0, wm, wn, lane, tid, smem);
      return;
    }
  }
  if constexpr (!OUTF32) {
    if (((p.N | p.ldc) & 7) == 0 && (!RES || (p.ldr & 3) == 0) && ((((uintptr_t)p.C) | ((uintptr_t)p.C2)) & 15) == 0 && (p.c_bs & 7) == 0) {
      __syncthreads();
      epilogue_store_lds<ACT, RES, HASC2, 8, CONV>(p, acc, z, m0 + wm * 128, n0 + wn * 64, lane, smem + wave * EPI_WAVE_BYTES);
      return;
    }
  }
  epilogue_store<ACT, RES, OUTF32, HASC2, 8, 4>(p, acc, z, m0 + wm * 128 + (lane & 15), n0 + wn * 64 + (lane >> 4) * 4);
}

}  // namespace

kern_t pick_gemm256l(int act, bool res, bool f32, bool c2, bool conv) {
  kern_t k = nullptr;
#define X2I_PICK(A_, R_, F_, C_) k = conv ? gemm256l_bf16_kernel<A_, R_, F_, C_, true> : gemm256l_bf16_kernel<A_, R_, F_, C_, false>;
  X2I_GEMM_PICK_TABLE(X2I_PICK)
#undef X2I_PICK
  return k;
}

}  // namespace x2i_gemm
