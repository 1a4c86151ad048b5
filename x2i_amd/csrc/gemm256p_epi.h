// Epilogue pipelines of the persistent four-wave GEMM kernels (gemm256p.hip: linears; gemm256c.hip: implicit-GEMM convolutions): a wave's
// 128 x 128 accumulators leave in eight 32-row x 64-column chunks through 8 KiB of private LDS staging.  Moved here from gemm256p.hip in
// round 6 so that the convolution kernel shares them (CONV adds the output row pitch, the channel moments and drops the gate).
#pragma once
#include <type_traits>
#include "gemm_device.h"

namespace x2i_gemm {
namespace {

constexpr int P_STAGE_OFF = 2 * TILE2_BYTES;  // 128 KiB: behind the two operand buffers
constexpr int P_STAGE_WAVE = 8192;

// Epilogue of one wave (128 x 128 outputs) in eight 32-row x 64-column chunks through a double-buffered 2 x 4 KiB staging area.
// Image of a chunk: [32 rows][128 B], 16-byte chunk ch of row r at physical chunk ch ^ ((r >> 1) & 7); a lane parks its four
// consecutive columns with one ds_write_b64, rows leave as whole 128-byte lines (16-byte stores, 8 lines per wave instruction).
// Same arithmetic (explicit fmaf, same rounding points) as epilogue_store_lds of the one-tile kernels: bit-identical results.
// The chunks run as a three-stage pipeline (what bounded the first, chunk-after-chunk form of this epilogue was not store
// bandwidth but LATENCY: per chunk one LDS round trip for the parked values plus four more, each `ds_read_b128 -> s_waitcnt -> store`
// behind its own exec-mask branch -- 7.5 us per tile against 1.3 us of K-loop hand-over, tools/gemm_unit_timeline.py):
//   A(q): accumulators of chunk q -> bias / activation -> bf16 -> staging buffer q & 1      (VALU + 8-byte LDS writes)
//   B(q): the chunk's four 16-byte row pieces back from LDS                                  (issued together, ONE wait)
//   C(q): four buffer_store_dwordx4                                                          (no branches: rows >= M fall behind the
//         descriptor's num_records, columns >= N get the out-of-range offset bit)
// order  A(0) | B(0) A(1) C(0) | B(1) A(2) C(1) | ...: B(q)'s LDS latency hides behind A(q+1)'s VALU work, C(q) never waits for LDS.
// RES: out = bf16(gate * act(acc + bias) + residual), one rounding as everywhere.  The residual comes STRAIGHT INTO REGISTERS in the
// accumulator layout (a lane's four consecutive columns = one 8-byte load; the four column groups of a row share a 128-byte line, so
// L2 sees every line once), two chunks ahead of its use.  (The first form of this epilogue fetched residual rows by LDS-DMA into the
// staging buffer, one chunk ahead -- all the look-ahead 8 KiB allow -- and waited ~2 us per chunk for it: 17 us per tile.)
// e4m3 operands (F8): the accumulators hold sum_k A8 W8; the value every epilogue starts from is
//     deq(acc) + bias = fma(acc * (w_scale[n] * alpha), a_scale[z][m], bias[n])
// -- a multiply and an explicitly spelled fma (two instructions per element; nothing left for the compiler to contract one way
// here and another way there): the one-tile kernel (gemm256_fp8.hip) spells it the same way, and the two are bit-identical (tested).
// The scales are fetched BEFORE the unit's K-loop statement (deq_load: branch-free vector loads, rows / columns beyond the
// problem read a clamped address -- their results are dropped by the stores), so their latency hides behind the K-loop.
template <bool F8> struct Deq {};
template <> struct Deq<true> {
  float sw[8][4];   // w_scale[n] * alpha of this lane's 4 columns per 16-column block
  float sr[4][2];   // a_scale of rows m_wave + c * 32 + i * 16 + (lane & 15)
};
template <bool F8>
__device__ __forceinline__ void deq_load(const GemmP& p, int z, int m_wave, int n_wave, int lane, Deq<F8>& d) {
  if constexpr (F8) {
    const int ng = lane >> 4, mlane = lane & 15;
    static_for<8>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const int n = n_wave + j * 16 + ng * 4;
      f32x4_t v = {1.f, 1.f, 1.f, 1.f};
      if (p.f_sw) v = *(const f32x4_t*)(p.f_sw + (n < p.N ? n : 0));   // (N % 8 == 0 and 16-byte aligned scales: launcher)
#pragma unroll
      for (int r = 0; r < 4; ++r) d.sw[j][r] = v[r] * p.f_alpha;
    });
    const float* sa = p.f_sa ? p.f_sa + (long long)z * p.f_sa_bs : nullptr;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < 2; ++i) d.sr[c][i] = sa ? sa[min(m_wave + c * 32 + i * 16 + mlane, p.M - 1)] : 1.f;
  }
}

// e4m3 OUTPUT epilogue of one wave (x2i_gemm_fp8 with out_fp8: GELU(ff.net.0 / proj_mlp) written as the next GEMM's A operand), same
// three-stage pipeline over eight 32-row x 64-column chunks as epilogue_chunked_pipe below, on bytes: a chunk's staging image is
// [32 rows][64 B] (16-byte piece pc of row r at pc ^ ((r >> 1) & 3): conflict-free 4-byte parks), a lane parks its four consecutive
// columns as one packed dword, rows leave as 64-byte runs (16-byte stores, 16 rows per wave instruction).  Arithmetic of
// epilogue_store_fp8 (gemm256_fp8.hip): sat(act(deq(acc) + bias) * out_inv_scale) -> v_cvt_pk_fp8_f32.
template <int ACT, bool UNIT_OUT>
__device__ __forceinline__ void epilogue_chunked_pipe_e4m3(const GemmP& p, f32x4_t (&acc)[2][4][2][4], int z, int m_wave, int n_wave, int lane,
                                                           char* stage, const Deq<true>& dq) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const int mlane = lane & 15, ng = lane >> 4;
  const float* b2 = p.bias2 ? p.bias2 + (long long)z * p.bias2_bs : nullptr;
#ifdef X2I_ABLATION
  if (p.act2 >= 80) b2 = nullptr;  // (measurement: bias2 carries the timestamp buffer)
#endif
  float bv[8][4];
  static_for<8>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int n = n_wave + j * 16 + ng * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[j][r] = 0.f;
    if (n + 3 < p.N) {
      if (p.bias) {
        const uint2 bb = *(const uint2*)(p.bias + n);
        bv[j][0] = __uint_as_float(bb.x << 16); bv[j][1] = __uint_as_float(bb.x & 0xffff0000u);
        bv[j][2] = __uint_as_float(bb.y << 16); bv[j][3] = __uint_as_float(bb.y & 0xffff0000u);
      }
      if (b2) {
        const f32x4_t t4 = *(const f32x4_t*)(b2 + n);
        bv[j][0] += t4[0]; bv[j][1] += t4[1]; bv[j][2] += t4[2]; bv[j][3] += t4[3];
      }
    }
  });
  const uint32_t c_bytes = (uint32_t)((long long)(p.M - 1) * p.ldc + p.N);
  __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((uint8_t*)p.C + (long long)z * p.c_bs), 0, c_bytes, 0x00020000);
  // store piece `it` (0 / 1) of a chunk: row it * 16 + (lane >> 2), 16-byte piece (lane & 3) ^ swizzle(row)
  const int srow = lane >> 2, spc = lane & 3;
  uint32_t voff[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = it * 16 + srow;
      const int n = n_wave + h * 64 + ((spc ^ ((row >> 1) & 3)) << 4);
      voff[h][it] = (n + 15 < p.N) ? (uint32_t)((long long)(m_wave + row) * p.ldc + n) : 0x80000000u;
    }
  asm volatile("" ::: "memory");
  auto stage_a = [&](auto qc, auto jc, int which) {
    constexpr int q = decltype(qc)::value;
    constexpr int j = decltype(jc)::value;
    constexpr int h = q >> 2, c = q & 3;
    char* buf = stage + which * 2048;
    // all eight accumulators of the quarter first (the reads are volatile asm -- see the bf16 form -- and a volatile statement between
    // two elements pins their order: with the read inside the element loop the eight dependent chains -- multiply, fma, GELU's exp and
    // rcp, clamp -- ran strictly one after the other, ~11 cycles per instruction; read up front, the compiler interleaves them)
    float x[8];
    static_for<2>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t;   // (through a local: a variable that appears ONLY as an asm operand inside a lambda is not captured by clang)
        const float a = acc[h][c][i][j][r];
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(a));
        x[i * 4 + r] = t;
      }
    });
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = fmaf(x[e] * dq.sw[h * 4 + j][e & 3], dq.sr[c][e >> 2], bv[h * 4 + j][e & 3]);
    apply_act8(x, ACT);   // (GELU: eight at once, x2i_common.h; a breadth-first GELU with scheduling barriers between its steps was SLOWER: r04q)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if constexpr (!UNIT_OUT) x[e] *= p.f_oinv;   // (out_inv_scale == 1, the model's setting: the multiply is skipped -- x * 1 is x)
      x[e] = __builtin_amdgcn_fmed3f(x[e], -448.f, 448.f);
    }
    static_for<2>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      int pk = __builtin_amdgcn_cvt_pk_fp8_f32(x[i * 4], x[i * 4 + 1], 0, false);
      pk = __builtin_amdgcn_cvt_pk_fp8_f32(x[i * 4 + 2], x[i * 4 + 3], pk, true);
      const int row = i * 16 + mlane;
      *(int*)(buf + row * 64 + ((j ^ ((row >> 1) & 3)) << 4) + (ng << 2)) = pk;
    });
  };
  u32x4 d[2];
  static_for<4>([&](auto jc) { stage_a(std::integral_constant<int, 0>{}, jc, 0); });
  static_for<8>([&](auto qc) {
    constexpr int q = decltype(qc)::value;
    constexpr int h = q >> 2, c = q & 3;
    char* buf = stage + (q & 1) * 2048;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 2; ++it) d[it] = *(const u32x4*)(buf + it * 1024 + lane * 16);
    const uint32_t soff = (uint32_t)((long long)c * 32 * p.ldc);
    static_for<2>([&](auto hc) {
      constexpr int hf = decltype(hc)::value;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (q + 1 < 8) {
        stage_a(std::integral_constant<int, q + 1>{}, std::integral_constant<int, 2 * hf>{}, (q + 1) & 1);
        stage_a(std::integral_constant<int, q + 1>{}, std::integral_constant<int, 2 * hf + 1>{}, (q + 1) & 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_raw_buffer_store_b128(d[hf], c_rsrc, voff[h][hf], soff, 0);
    });
  });
}

// FXADD: the accumulators hold only the LAST K range of the tile (parallel split with fix-up); `npre` other workgroups have parked the
// sums of the earlier ranges in slabs fx_slab0, fx_slab0 + 8, ... (accumulator layout, [64 tiles][256 threads][16 B]); they are fetched a
// chunk ahead, summed in slab order and added to the accumulators in front of the bias: out = epi((p_0 + p_1 + ...) + acc).
// CONV (gemm256c.hip): no gate (the launcher refuses one; gate = 1 is exactly `+`), an optional output ROW PITCH (x2i_conv_desc.out_row_pitch: a chunk's
// rows are then placed one by one -- m -> (m / OW) * pitch + (m % OW) * ldc, the division by a float reciprocal + fix-up) and the channel-quad
// moments of the rounded outputs (x2i_conv_desc.moments: mom_add in the order of the one-tile kernels' 128-row wave tiles -- bit-identical sums).
// m / d for 0 <= m < 2^24, d > 0 (rd = 1.0f / d): the float quotient is off by at most one
__device__ __forceinline__ int fast_div(int m, int d, float rd) {
  int q = (int)((float)m * rd);
  int r = m - q * d;
  q += (r >= d) ? 1 : 0;
  q -= (r < 0) ? 1 : 0;
  return q;
}
// Invalid-tap mask of one output pixel (bit ky * KW + kx = that tap lies outside the H x W image) in closed form: the out-of-image taps of a row / a
// column are a prefix and a suffix of it.  rep = sum over ky of 1 << (ky * KW) replicates the column bits into every filter row; filter rows that are
// outside themselves (top / bottom image rows only: a rarely taken loop) become all-ones.
__device__ __forceinline__ uint32_t conv_tap_mask(int iy0, int ix0, int H, int W, int KH, int KW, uint32_t rep, uint32_t fullrow) {
  auto edge = [](int i0, int n, int k) -> uint32_t {   // bits t < k with (unsigned)(i0 + t) >= n
    const uint32_t all = k >= 32 ? 0xffffffffu : ((1u << k) - 1u);
    const int lo = min(max(-i0, 0), k);            // taps below 0
    const int hi = min(max(n - i0, 0), k);         // first tap at or behind n
    const uint32_t below = lo >= 32 ? 0xffffffffu : ((1u << lo) - 1u);
    const uint32_t upto = hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u);
    return (below | ~upto) & all;
  };
  uint32_t mask = edge(ix0, W, KW) * rep;
  uint32_t rowbad = edge(iy0, H, KH);
  while (rowbad) {
    const int ky = __builtin_ctz(rowbad);
    rowbad &= rowbad - 1;
    mask |= fullrow << (ky * KW);
  }
  return mask;
}
template <int ACT, bool HASC2, bool RES, bool F8 = false, bool FXADD = false, bool CONV = false>
__device__ __forceinline__ void epilogue_chunked_pipe(const GemmP& p, f32x4_t (&acc)[2][4][2][4], int z, int m_wave, int n_wave, int lane,
                                                      char* stage, const Deq<F8>& dq, int fx_slab0 = 0, int npre = 0, int tid = 0) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  const int mlane = lane & 15, ng = lane >> 4;
  const int srow = lane >> 3, sch = lane & 7;
  const float* b2 = p.bias2 ? p.bias2 + (long long)z * p.bias2_bs : nullptr;
#ifdef X2I_ABLATION
  if (p.act2 >= 80) b2 = nullptr;  // (measurement: bias2 carries the timestamp buffer, tools/gemm_unit_timeline.py)
#endif
  constexpr bool GATE = RES && !CONV;
  const float* gz = (GATE && p.gate) ? p.gate + (long long)z * p.gate_bs : nullptr;
  float bv[8][4], gv[GATE ? 8 : 1][4];
  float mom_s[CONV ? 8 : 1], mom_q[CONV ? 8 : 1];   // CONV: sum / sum of squares of this lane's channel quad per 16-column block
  if constexpr (CONV) {
#pragma unroll
    for (int j = 0; j < 8; ++j) mom_s[j] = mom_q[j] = 0.f;
  }
  static_for<8>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int n = n_wave + j * 16 + ng * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bv[j][r] = 0.f;
      if constexpr (GATE) gv[j][r] = 1.f;
    }
    if (n + 3 < p.N) {
      if (p.bias) {
        const bf16_t* biasz = p.bias;
        if constexpr (CONV) biasz += (long long)(z / p.wdiv) * p.bias_gs;   // (grouped weights: x2i_gemm_args.w_group)
        const uint2 bb = *(const uint2*)(biasz + n);
        bv[j][0] = __uint_as_float(bb.x << 16); bv[j][1] = __uint_as_float(bb.x & 0xffff0000u);
        bv[j][2] = __uint_as_float(bb.y << 16); bv[j][3] = __uint_as_float(bb.y & 0xffff0000u);
      }
      if constexpr (GATE) {
        if (gz) {
          const f32x4_t g4 = *(const f32x4_t*)(gz + n);
          gv[j][0] = g4[0]; gv[j][1] = g4[1]; gv[j][2] = g4[2]; gv[j][3] = g4[3];
        }
      }
      if (b2) {
        const f32x4_t t4 = *(const f32x4_t*)(b2 + n);
        bv[j][0] += t4[0]; bv[j][1] += t4[1]; bv[j][2] += t4[2]; bv[j][3] += t4[3];
      }
    }
  });
  // residual: descriptor of this batch item, per-lane offsets of the (i = 0 / 1) row blocks of the chunk at (c = 0, h = 0).  Round 6: the residual
  // comes as FOUR 16-byte loads per chunk instead of eight 8-byte ones -- a lane fetches eight consecutive columns (of column block 2 jp + (ng & 1),
  // half (ng >> 1): a row's four lanes cover 64 contiguous bytes) and v_permlane16_swap turns two such registers into the accumulator layout of the
  // blocks 2 jp and 2 jp + 1 (the exchange of gemm512c.hip's direct epilogue, run backwards; it is its own inverse).  Half the memory instructions
  // of what was the most expensive epilogue of the step (14.5 us per tile against 5.5 us for the plain one, tools/gemm_unit_timeline.py)
  __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0, 0x00020000);
  uint32_t roff[2] = {0, 0};
  if constexpr (RES) {
    r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res + (long long)z * p.r_bs), 0, (uint32_t)(((long long)(p.M - 1) * p.ldr + p.N) * 2), 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i) roff[i] = (uint32_t)(((long long)(m_wave + i * 16 + mlane) * p.ldr + n_wave + 16 * (ng & 1) + 8 * (ng >> 1)) * 2);
  }
  // fix-up partials: window of two chunks, rp[q & 1][i][j] = sum over the predecessors' slabs of accumulator tile (2c + i, 4h + j)
  f32x4_t rp[FXADD ? 2 : 1][2][4];
  auto load_part = [&](auto qc) {
    if constexpr (FXADD) {
      constexpr int q = decltype(qc)::value;
      constexpr int h = q >> 2, c = q & 3;
      const float* base = p.sk_slabs + (long long)fx_slab0 * (SK_SLAB_BYTES / 4) + tid * 4;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float* src = base + ((2 * c + i) * 8 + 4 * h + j) * 1024;
          f32x4_t sum = *(const f32x4_t*)src;
          for (int u = 1; u < npre; ++u) sum += *(const f32x4_t*)(src + (long long)u * 8 * (SK_SLAB_BYTES / 4));   // (the XCD's workgroups, hence their slabs, are 8 apart)
          rp[q & 1][i][j] = sum;
        }
    }
  };
  u32x2 rres[RES ? 3 : 1][2][4];   // residual window: chunk q lives in rres[q % 3]
  auto load_res = [&](auto qc) {
    if constexpr (RES) {
      constexpr int q = decltype(qc)::value;
      constexpr int h = q >> 2, c = q & 3;
      const uint32_t soff = (uint32_t)(((long long)c * 32 * p.ldr + h * 64) * 2);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          const u32x4 l = __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, roff[i] + jp * 64, soff, 0);
          rres[q % 3][i][2 * jp] = (u32x2){l[0], l[1]};          // (exchanged into the accumulator layout where the pair is first used: stage_a)
          rres[q % 3][i][2 * jp + 1] = (u32x2){l[2], l[3]};
        }
    }
  };
  // output descriptors of this batch item: rows at or behind M are out of range (dropped by the hardware)
  const bool pitched = CONV && p.cRowPitch != 0;   // (workgroup-uniform)
  const uint32_t c_bytes = pitched ? (uint32_t)(((long long)(p.M / p.cOW - 1) * p.cRowPitch + (long long)(p.cOW - 1) * p.ldc + p.N) * 2)
                                   : (uint32_t)(((long long)(p.M - 1) * p.ldc + p.N) * 2);
  __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((bf16_t*)p.C + (long long)z * p.c_bs), 0, c_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t c2_rsrc = c_rsrc;
  if constexpr (HASC2) c2_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C2 + (long long)z * p.c_bs), 0, c_bytes, 0x00020000);
  const float r_ow = CONV ? 1.0f / (float)(p.cOW > 0 ? p.cOW : 1) : 0.f;
  // per-lane store offsets of the chunk at (c = 0, h): piece `it` is row it*8 + srow, 16-byte column group sch ^ swizzle(row)
  uint32_t voff[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + srow;
      const int n = n_wave + h * 64 + ((sch ^ ((row >> 1) & 7)) << 3);
      voff[h][it] = (n + 7 < p.N) ? (uint32_t)(((long long)(m_wave + row) * p.ldc + n) * 2) : 0x80000000u;
    }
  asm volatile("" ::: "memory");  // the bias loads stay in front of everything below
  // A(q), column group j of the chunk (a quarter of the stage: 8 accumulator reads, bias / activation, two 8-byte parks)
  auto stage_a = [&](auto qc, auto jc, int pass, int which) {
    constexpr int q = decltype(qc)::value;
    constexpr int j = decltype(jc)::value;
    constexpr int h = q >> 2, c = q & 3;
    char* buf = stage + which * 4096;
    if constexpr (RES && (j & 1) == 0) {   // the residual of column blocks j, j + 1: loaded layout -> accumulator layout (both 16-row blocks)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const auto a = __builtin_amdgcn_permlane16_swap(rres[q % 3][i][j][0], rres[q % 3][i][j + 1][0], false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(rres[q % 3][i][j][1], rres[q % 3][i][j + 1][1], false, false);
        rres[q % 3][i][j] = (u32x2){a[0], b[0]};
        rres[q % 3][i][j + 1] = (u32x2){a[1], b[1]};
      }
    }
    float x[8];   // (the quarter's eight accumulators first, then eight independent chains; GELU breadth first: see the e4m3 form)
    static_for<2>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t;   // (through a local: a variable that appears ONLY as an asm operand inside a lambda is not captured by clang)
        const float a = acc[h][c][i][j][r];
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(a));
        x[i * 4 + r] = t;
      }
    });
    static_for<2>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = x[i * 4 + r];
        if constexpr (FXADD) t = rp[q & 1][i][j][r] + t;
        if constexpr (F8) t = fmaf(t * dq.sw[h * 4 + j][r], dq.sr[c][i], bv[h * 4 + j][r]);
        else t = t + bv[h * 4 + j][r];
        x[i * 4 + r] = t;
      }
    });
    apply_act8(x, ACT);   // (GELU: eight at once, x2i_common.h; a breadth-first GELU with scheduling barriers between its steps was SLOWER: r04q)
    static_for<2>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = x[i * 4 + r];
      if constexpr (RES) {
        const u32x2 r2 = rres[q % 3][i][j];
        if constexpr (CONV) {   // (fmaf(1, v, r) == v + r: one rounding either way)
          v[0] += __uint_as_float(r2[0] << 16); v[1] += __uint_as_float(r2[0] & 0xffff0000u);
          v[2] += __uint_as_float(r2[1] << 16); v[3] += __uint_as_float(r2[1] & 0xffff0000u);
        } else {
        v[0] = fmaf(gv[h * 4 + j][0], v[0], __uint_as_float(r2[0] << 16));
        v[1] = fmaf(gv[h * 4 + j][1], v[1], __uint_as_float(r2[0] & 0xffff0000u));
        v[2] = fmaf(gv[h * 4 + j][2], v[2], __uint_as_float(r2[1] << 16));
        v[3] = fmaf(gv[h * 4 + j][3], v[3], __uint_as_float(r2[1] & 0xffff0000u));
        }
      }
      if (pass == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.act2);
      }
      const int row = i * 16 + mlane;
      char* slot = buf + row * 128 + ((((j << 1) | (ng >> 1)) ^ ((row >> 1) & 7)) << 4) + ((ng & 1) << 3);
      const uint2 pk = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
      *(uint2*)slot = pk;
      if constexpr (CONV) {
        if (p.cMom) mom_add(mom_s[h * 4 + j], mom_q[h * 4 + j], pk.x, pk.y, m_wave + c * 32 + row < p.M);
      }
    });
  };
  constexpr int NPASS = HASC2 ? 2 : 1;
  constexpr int NST = 8 * NPASS;  // pipeline steps: (chunk, pass), pass-minor; step s uses staging buffer s & 1
  u32x4 d[4];
  auto run_a = [&](auto sc, auto jc) {
    constexpr int s_ = decltype(sc)::value;
    stage_a(std::integral_constant<int, s_ / NPASS>{}, jc, s_ % NPASS, s_ & 1);
  };
  load_part(std::integral_constant<int, 0>{});
  load_part(std::integral_constant<int, 1>{});
  load_res(std::integral_constant<int, 0>{});
  load_res(std::integral_constant<int, 1>{});
  static_for<4>([&](auto jc) { run_a(std::integral_constant<int, 0>{}, jc); });
  static_for<NST>([&](auto sc) {
    constexpr int s_ = decltype(sc)::value;
    constexpr int q = s_ / NPASS, pass = s_ % NPASS;
    constexpr int h = q >> 2, c = q & 3;
    char* buf = stage + (s_ & 1) * 4096;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (RES && s_ + 2 < NST) load_res(std::integral_constant<int, s_ + 2>{});   // (window slot of chunk s-1, consumed by A(s-1))
    if constexpr (FXADD && s_ + 2 < NST) load_part(std::integral_constant<int, s_ + 2>{});   // (slot of chunk s, consumed by A(s) in step s - 1; HASC2 is never combined with FXADD)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // A(s)'s writes have landed (and B(s-1)'s reads returned long ago)
#pragma unroll
    for (int it = 0; it < 4; ++it) d[it] = *(const u32x4*)(buf + it * 1024 + lane * 16);   // B(s)
    const uint32_t soff = pitched ? 0u : (uint32_t)((long long)c * 32 * p.ldc * 2);
    // A(s+1) into the other buffer, a quarter at a time, one store of C(s) behind each quarter: a wave's stores issue at roughly one per
    // 150 cycles whatever sits between them (tools/ubench/store_issue.hip), so the VALU work between two stores is free
    static_for<4>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (s_ + 1 < NST) run_a(std::integral_constant<int, s_ + 1>{}, jc);
      __builtin_amdgcn_sched_barrier(0);
      uint32_t vo = voff[h][j];
      if constexpr (CONV) {
        if (pitched) {   // piece j of the chunk = row j * 8 + srow: placed by its own (output row, output column)
          const int row = j * 8 + srow;
          const int m = m_wave + c * 32 + row;
          const int n = n_wave + h * 64 + ((sch ^ ((row >> 1) & 7)) << 3);
          const int oy = fast_div(m, p.cOW, r_ow);
          vo = (m < p.M && n + 7 < p.N) ? (uint32_t)(((long long)oy * p.cRowPitch + (long long)(m - oy * p.cOW) * p.ldc + n) * 2) : 0x80000000u;
        }
      }
      __builtin_amdgcn_raw_buffer_store_b128(d[j], pass == 0 ? c_rsrc : c2_rsrc, vo, soff, 0);
    });
  });
  if constexpr (CONV) {
    if (p.cMom) {
      static_for<8>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        mom_flush(p, mom_s[j], mom_q[j], z, m_wave >> 7, n_wave + j * 16 + ng * 4, lane);
      });
    }
  }
}

}  // namespace
}  // namespace x2i_gemm
