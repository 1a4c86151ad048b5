// 256x256x128 fp8 (OCP e4m3fn) MFMA GEMM: the fp8 variant of the full-line 256^2 kernel (gemm256.hip) for the large DiT linears
// (BASELINE north_star: "MFMA bf16/fp8 for the QKV/out-proj and MLP GEMMs").  Launcher: gemm.hip (x2i_gemm_fp8).
//
//   C[z][m][n] = epi( a_scale[z][m] * w_scale[n] * alpha * sum_k A8[z][m][k] * W8[n][k] )      A8, W8: e4m3 bytes, K contiguous
//
// What carries over unchanged from the bf16 kernel: a K-tile is 128 BYTES of every row (= 128 e4m3 elements instead of 64 bf16),
// so the LDS-DMA pieces (8 rows x one whole 128-byte line), the operand images [32 row groups][k-half][8 rows][4 chunks], the XOR
// swizzle, the double buffer with one barrier per K-tile and the tile order are byte for byte the same.  What changes:
//   * v_mfma_scale_f32_16x16x128_f8f6f4 (unit E8M0 scales) consumes BOTH 64-byte k-halves of a fragment row in one instruction:
//     lane group g = lane >> 4 feeds bytes [16g, 16g+16) of k-half 0 and of k-half 1 -- a permutation of k that is the same for
//     A and W, so the dot products are unchanged -- i.e. exactly the two ds_read_b128 the bf16 kernel issues per fragment.
//     32 MFMAs of 32 cycles per K-tile and wave instead of 64 of 16: the same matrix-pipe time for twice the K depth.
//   * phases run over row-tile pairs (4 phases x 2 row tiles x 4 column tiles = 8 MFMAs); the wave's four W operands (both k-halves,
//     32 VGPRs) stay resident for the whole K-tile and are replaced in place, column by column, behind the last phase's MFMAs;
//     A operands are double-buffered per phase (2 x 16 VGPRs): 64 fragment registers, as in the bf16 kernel.
//   * the dequantisation scales multiply the accumulators once, in front of the shared epilogues (bias / GELU / gated residual,
//     LDS-staged whole-line stores); optional e4m3 output (GELU -> next GEMM's A operand) with saturation at +-448.
#include "gemm_device.h"

namespace x2i_gemm {
namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
union frag8 {
  i32x8_t v;
  i32x4_t h[2];  // k-half 0 / k-half 1
};

constexpr int UNIT_SCALE = 0x7f7f7f7f;  // E8M0 127 = 2^0 in every byte

// e4m3 output epilogue: act(acc + bias) * oinv -> saturate -> 4 packed bytes per lane (4 consecutive n), parked per wave in LDS
// (row stride 80 B) and written as 64-byte row segments with 16-byte stores.
template <int ACT>
__device__ __forceinline__ void epilogue_store_fp8(const GemmP& p, f32x4_t (&acc)[8][4], int z, int m_wave, int n_wave, int lane,
                                                   char* wave_lds) {
  constexpr int ROWB = 80;
  const int mlane = lane & 15, ng = lane >> 4;
  const float* b2 = p.bias2 ? p.bias2 + (long long)z * p.bias2_bs : nullptr;
  uint8_t* Cz = (uint8_t*)p.C + (long long)z * p.c_bs;
  static_for<4>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int n = n_wave + j * 16 + ng * 4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (n + 3 < p.N) {
      if (p.bias) {
        const uint2 bb = *(const uint2*)(p.bias + n);
        bv[0] = __uint_as_float(bb.x << 16); bv[1] = __uint_as_float(bb.x & 0xffff0000u);
        bv[2] = __uint_as_float(bb.y << 16); bv[3] = __uint_as_float(bb.y & 0xffff0000u);
      }
      if (b2) {
        const f32x4_t t4 = *(const f32x4_t*)(b2 + n);
        bv[0] += t4[0]; bv[1] += t4[1]; bv[2] += t4[2]; bv[3] += t4[3];
      }
    }
    static_for<8>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = __builtin_amdgcn_fmed3f(apply_act(acc[i][j][r] + bv[r], ACT) * p.f_oinv, -448.f, 448.f);
      int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
      pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
      *(int*)(wave_lds + (i * 16 + mlane) * ROWB + j * 16 + ng * 4) = pk;
    });
  });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the region is private to this wave
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 16 + (lane >> 2), c = lane & 3;
    const i32x4_t d = *(const i32x4_t*)(wave_lds + row * ROWB + c * 16);
    const int m = m_wave + row, n = n_wave + c * 16;
    if (m < p.M && n + 15 < p.N) *(i32x4_t*)(Cz + (long long)m * p.ldc + n) = d;
  }
}

template <int ACT, bool RES, bool OUT8>
__global__ __launch_bounds__(512, 2) void gemm256_fp8_kernel(GemmP p) {
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
  const int GM = p.gm;
  const int per_group = GM * p.tilesN;
  const int group = bid / per_group;
  const int first_m = group * GM;
  const int gsize = min(p.tilesM - first_m, GM);
  const int tm = first_m + (bid % per_group) % gsize;
  const int tn = (bid % per_group) / gsize;
  const int m0 = tm * BM2, n0 = tn * BN2;

  const uint8_t* Az = (const uint8_t*)p.A + (long long)z * p.a_bs;
  const uint32_t a_bytes = (uint32_t)((long long)(p.M - 1) * p.lda + p.K);
  const uint32_t w_bytes = (uint32_t)((long long)(p.N - 1) * p.ldw + p.K);
  __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Az, 0, a_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const uint8_t*)p.W + (long long)z * p.w_bs), 0, w_bytes, 0x00020000);

  // piece q = jj*8 + wave covers row group q (rows 8q..8q+7); lane -> (k-half, row in group, physical 16-byte chunk)
  uint32_t a_voff[4], w_voff[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) {
    const int g = jj * 8 + wave;
    const int khl = lane >> 5, r = (lane >> 2) & 7, cphys = lane & 3;
    const int row = g * 8 + r;
    const int kb = khl * 64 + ((cphys ^ (3 * (g & 1))) << 4);  // byte offset inside the 128-byte K-tile line
    a_voff[jj] = (m0 + row < p.M) ? (uint32_t)((long long)(m0 + row) * p.lda + kb) : 0x80000000u;
    w_voff[jj] = (n0 + row < p.N) ? (uint32_t)((long long)(n0 + row) * p.ldw + kb) : 0x80000000u;
  }

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15;
  const uint32_t frag = (frow >> 3) * 1024 + (frow & 7) * 64 + (((lane >> 4) ^ (3 * ((frow >> 3) & 1))) << 4);
  const uint32_t a_base = wm * 8 * 2048 + frag;          // + i*2048 per m-tile, + kh*512
  const uint32_t b_base = 32768 + wn * 4 * 2048 + frag;  // + j*2048 per n-tile, + kh*512

  const int nk = p.K / 128;
  auto issue_a = [&](int t) {
    char* dst = smem + (t & 1) * TILE2_BYTES;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(dst + (jj * 8 + wave) * 1024), 16,
                                               a_voff[jj], (uint32_t)t * 128u, 0, 0);
  };
  auto issue_w = [&](int t) {
    char* dst = smem + (t & 1) * TILE2_BYTES + 32768;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(dst + (jj * 8 + wave) * 1024), 16,
                                               w_voff[jj], (uint32_t)t * 128u, 0, 0);
  };
  auto load_a = [&](const char* tile, int i, frag8& f) {
    f.h[0] = *(const i32x4_t*)(tile + a_base + i * 2048);
    f.h[1] = *(const i32x4_t*)(tile + a_base + i * 2048 + 512);
  };
  auto load_w = [&](const char* tile, int j, frag8& f) {
    f.h[0] = *(const i32x4_t*)(tile + b_base + j * 2048);
    f.h[1] = *(const i32x4_t*)(tile + b_base + j * 2048 + 512);
  };
  issue_a(0);
  issue_w(0);
  frag8 wf[4], af[2][2];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int j = 0; j < 4; ++j) load_w(smem, j, wf[j]);
  load_a(smem, 0, af[0][0]);
  load_a(smem, 1, af[0][1]);

  auto ktile = [&](int kt, auto more_c) {
    constexpr bool MORE = decltype(more_c)::value;
    const char* cur = smem + (kt & 1) * TILE2_BYTES;
    const char* nxt = smem + ((kt + 1) & 1) * TILE2_BYTES;
#pragma unroll
    for (int ph = 0; ph < 4; ++ph) {
      if (ph == 3 && MORE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next tile was issued in phases 0 / 1; nothing younger in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      if (MORE && ph == 0) issue_a(kt + 1);
      if (MORE && ph == 1) issue_w(kt + 1);
      // A operands of the next phase (next tile's first pair at the last phase)
      if (ph < 3) {
        load_a(cur, 2 * ph + 2, af[(ph + 1) & 1][0]);
        load_a(cur, 2 * ph + 3, af[(ph + 1) & 1][1]);
      } else if (MORE) {
        load_a(nxt, 0, af[0][0]);
        load_a(nxt, 1, af[0][1]);
      }
      __builtin_amdgcn_sched_barrier(0);  // DMA issue + next phase's A operands first, then the MFMAs (see gemm256.hip)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
          acc[2 * ph + ii][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[j].v, af[ph & 1][ii].v, acc[2 * ph + ii][j], 0, 0, 0,
                                                                               UNIT_SCALE, 0, UNIT_SCALE);
        if (ph == 3 && MORE) load_w(nxt, j, wf[j]);  // this column's W operand is dead for the current tile: replace it in place
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (int kt = 0; kt < nk - 1; ++kt) ktile(kt, std::true_type{});
  ktile(nk - 1, std::false_type{});

  // ---- dequantisation + bias: acc <- fma(acc * (w_scale[n] * alpha), a_scale[m], bias[n] + bias2[z][n]) -- a multiply and an explicit fma,
  // exactly as the persistent form (gemm256p.hip) spells it (bit-identical, tested); the shared epilogues below then add a zero bias
  // (lane owns rows m_wave + i*16 + (lane & 15), columns n + 0..3)
  GemmP pe = p;
  pe.bias = nullptr;
  pe.bias2 = nullptr;
  {
    const int mrow = m0 + wm * 128 + (lane & 15), ncol = n0 + wn * 64 + (lane >> 4) * 4;
    const float* sa = p.f_sa ? p.f_sa + (long long)z * p.f_sa_bs : nullptr;
    const float* b2 = p.bias2 ? p.bias2 + (long long)z * p.bias2_bs : nullptr;
    float sw[4][4], bb[4][4];
    static_for<4>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const int n = ncol + j * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool in = n + r < p.N;
        sw[j][r] = (p.f_sw && in) ? p.f_sw[n + r] * p.f_alpha : p.f_alpha;
        bb[j][r] = (p.bias && in) ? bf16_to_f32(p.bias[n + r]) : 0.f;
        if (b2 && in) bb[j][r] += b2[n + r];
      }
    });
    static_for<8>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int m = mrow + i * 16;
      const float s = (sa && m < p.M) ? sa[m] : 1.f;
      static_for<4>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = fmaf(acc[i][j][r] * sw[j][r], s, bb[j][r]);
      });
    });
  }
  __syncthreads();  // every wave is done reading the operand images before they are reused as staging space
  if constexpr (ACT == X2I_ACT_NONE && !RES && !OUT8) {
    if (p.q_on) {  // fused per-head RMSNorm + RoPE + head-major / transposed stores (x2i_gemm_qkv_fp8), same code as the bf16 kernel
      epilogue_qkv<8, 4, 512>(pe, acc, z, m0, n0, wm, wn, lane, tid, smem);
      return;
    }
  }
  if constexpr (OUT8) {
    epilogue_store_fp8<ACT>(pe, acc, z, m0 + wm * 128, n0 + wn * 64, lane, smem + wave * EPI_WAVE_BYTES);
  } else {
    epilogue_store_lds<ACT, RES, false, 8>(pe, acc, z, m0 + wm * 128, n0 + wn * 64, lane, smem + wave * EPI_WAVE_BYTES);
  }
}

}  // namespace

kern_t pick_gemm256_fp8(int act, bool res, bool out8) {
  if (out8) {
    if (res) return nullptr;
    switch (act) {
      case X2I_ACT_NONE: return gemm256_fp8_kernel<X2I_ACT_NONE, false, true>;
      case X2I_ACT_GELU_TANH: return gemm256_fp8_kernel<X2I_ACT_GELU_TANH, false, true>;
      default: return nullptr;
    }
  }
  if (res) return act == X2I_ACT_NONE ? gemm256_fp8_kernel<X2I_ACT_NONE, true, false> : nullptr;
  switch (act) {
    case X2I_ACT_NONE: return gemm256_fp8_kernel<X2I_ACT_NONE, false, false>;
    case X2I_ACT_GELU_TANH: return gemm256_fp8_kernel<X2I_ACT_GELU_TANH, false, false>;
    default: return nullptr;
  }
}

}  // namespace x2i_gemm
