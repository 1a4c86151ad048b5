// Persistent form of the 4-wave 256x256x64 bf16 GEMM (gemm256w.hip): one workgroup per CU walks a list of output tiles and the
// K-loop never stops at a tile boundary -- the last two K-iterations of a tile fetch K-tiles 0 / 1 of the workgroup's NEXT output
// tile (gen_gemm256w.py, X2I_GEMM256P_MAIN), so the epilogue of tile n (accumulators -> LDS staging -> whole-line stores) runs with
// the operands of tile n+1 already landing, and tile n+1 starts multiplying straight out of registers.  LDS: 128 KiB operand ring +
// 4 x 8 KiB per-wave staging = all 160 KiB; epilogues leave in 32-row chunks.  Same image / MFMA / k order as every other bf16
// GEMM kernel here: bit-identical results (tested).  Launcher: gemm.hip (plain and batched launches with whole-line bf16 epilogues; the batch items' tiles form one list).
#include <type_traits>
#include "gemm_device.h"
#include "gemm256p_epi.h"
#include "gemm256w_loop.inc"
#include "gemm256f8_loop.inc"

namespace x2i_gemm {
namespace {

// FX with every part parked ("A": a tile cut into P equal K ranges, one workgroup each, all at work at the same time): instead of ONE
// finisher reading P - 1 slabs and running the whole epilogue while the others idle (measured on the single-block proj_out of a 512^2
// batch-1 step: K-loops done at 84 us, launch done at 181 us -- tools/fx_timeline.py, profiles/r05f_*), EVERY part finishes the chunks
// q = j, j + P, ... of each wave's 128 x 128 quadrant: it adds the P parked ranges of those chunks in slab order (p_0 + p_1 + ... : the
// association the one-finisher form uses, bit-identical to it) and runs the gated-residual epilogue of epilogue_chunked_pipe on them
// (same arithmetic, same staging image, same whole-line stores).  Chunk indices are run-time values here -- nothing is read from
// the accumulator registers.
__device__ __forceinline__ void epilogue_fx_slice(const GemmP& p, int z, int m_wave, int n_wave, int lane, char* stage, int slab0, int P, int j,
                                                  int tid) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  const int mlane = lane & 15, ng = lane >> 4;
  const int srow = lane >> 3, sch = lane & 7;
  const float* b2 = p.bias2 ? p.bias2 + (long long)z * p.bias2_bs : nullptr;
#ifdef X2I_ABLATION
  if (p.act2 >= 80) b2 = nullptr;
#endif
  const float* gz = p.gate ? p.gate + (long long)z * p.gate_bs : nullptr;
  __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res + (long long)z * p.r_bs), 0, (uint32_t)(((long long)(p.M - 1) * p.ldr + p.N) * 2), 0x00020000);
  const uint32_t c_bytes = (uint32_t)(((long long)(p.M - 1) * p.ldc + p.N) * 2);
  __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((bf16_t*)p.C + (long long)z * p.c_bs), 0, c_bytes, 0x00020000);
  const float* sbase = p.sk_slabs + (long long)slab0 * (SK_SLAB_BYTES / 4) + tid * 4;
  for (int q = j; q < 8; q += P) {
    const int h = q >> 2, c = q & 3;
    // bias / gate of this lane's 4 columns per 16-column block of the chunk's column half
    float bv[4][4], gv[4][4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int n = n_wave + h * 64 + jj * 16 + ng * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[jj][r] = 0.f, gv[jj][r] = 1.f;
      if (n + 3 < p.N) {
        if (p.bias) {
          const uint2 bb = *(const uint2*)(p.bias + n);
          bv[jj][0] = __uint_as_float(bb.x << 16); bv[jj][1] = __uint_as_float(bb.x & 0xffff0000u);
          bv[jj][2] = __uint_as_float(bb.y << 16); bv[jj][3] = __uint_as_float(bb.y & 0xffff0000u);
        }
        if (gz) {
          const f32x4_t g4 = *(const f32x4_t*)(gz + n);
          gv[jj][0] = g4[0]; gv[jj][1] = g4[1]; gv[jj][2] = g4[2]; gv[jj][3] = g4[3];
        }
        if (b2) {
          const f32x4_t t4 = *(const f32x4_t*)(b2 + n);
          bv[jj][0] += t4[0]; bv[jj][1] += t4[1]; bv[jj][2] += t4[2]; bv[jj][3] += t4[3];
        }
      }
    }
    u32x2 rres[2][4];
    const uint32_t rsoff = (uint32_t)(((long long)c * 32 * p.ldr + h * 64) * 2);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
        rres[i][jj] = __builtin_amdgcn_raw_buffer_load_b64(r_rsrc, (uint32_t)(((long long)(m_wave + i * 16 + mlane) * p.ldr + n_wave + ng * 4) * 2) + jj * 32, rsoff, 0);
    // the P parked ranges of the chunk's eight accumulator tiles, summed in slab order (two slabs in flight)
    f32x4_t sum[2][4];
    for (int u = 0; u < P; ++u) {
      const float* su = sbase + (long long)u * 8 * (SK_SLAB_BYTES / 4);
      f32x4_t v[2][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) v[i][jj] = *(const f32x4_t*)(su + ((2 * c + i) * 8 + 4 * h + jj) * 1024);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) sum[i][jj] = u ? sum[i][jj] + v[i][jj] : v[i][jj];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = i * 16 + mlane;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = sum[i][jj][r] + bv[jj][r];
        const u32x2 r2 = rres[i][jj];
        v[0] = fmaf(gv[jj][0], v[0], __uint_as_float(r2[0] << 16));
        v[1] = fmaf(gv[jj][1], v[1], __uint_as_float(r2[0] & 0xffff0000u));
        v[2] = fmaf(gv[jj][2], v[2], __uint_as_float(r2[1] << 16));
        v[3] = fmaf(gv[jj][3], v[3], __uint_as_float(r2[1] & 0xffff0000u));
        char* slot = stage + row * 128 + ((((jj << 1) | (ng >> 1)) ^ ((row >> 1) & 7)) << 4) + ((ng & 1) << 3);
        *(uint2*)slot = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    u32x4 d[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) d[it] = *(const u32x4*)(stage + it * 1024 + lane * 16);
    const uint32_t soff = (uint32_t)((long long)c * 32 * p.ldc * 2);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + srow;
      const int n = n_wave + h * 64 + ((sch ^ ((row >> 1) & 7)) << 3);
      const uint32_t vo = (n + 7 < p.N) ? (uint32_t)(((long long)(m_wave + row) * p.ldc + n) * 2) : 0x80000000u;
      __builtin_amdgcn_raw_buffer_store_b128(d[it], c_rsrc, vo, soff, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the staging buffer is rewritten by the next chunk)
  }
}

// Fused QKV epilogue of one wave (x2i_gemm_qkv_bf16: q / k RMSNorm + RoPE + head split, V transposed) on the persistent kernel's
// 8 KiB of private staging.  A wave's 128 x 128 outputs are 128 tokens of exactly ONE head of the q, k or v section (tile columns
// never straddle a section; checked by the launcher).  Same arithmetic and rounding points as qkv_park / qkv_finish (gemm_device.h):
// the tile is parked as bf16(acc + bias) and read back, so the results equal the one-tile kernels' bit for bit.
//   q / k: eight 16-token chunks ([16 rows][256 B], 16-byte chunk ch of row r at ch ^ r; double-buffered); a pass = 4 tokens x 16
//          lanes x 8 dims.  The fp32 cos / sin rows of chunk n+1 are requested BEFORE chunk n's stores are issued, so the wait for
//          them never includes a store (a load queued behind a store is only known to have returned once the store has).
//   v:     four 64-token x 64-dim chunks ([64 rows][128 B], chunk ch of row r at ch ^ ((r >> 3) ^ (r >> 1)) & 7: conflict-free for the
//          row-wise 8-byte parks AND the column-wise 4-byte gathers); a lane gathers two adjacent dims of 8 consecutive tokens and
//          writes two 16-byte token runs of V^T, 8 lanes cover a 128-byte line.
template <bool F8 = false>
__device__ __forceinline__ void epilogue_qkv_chunked(const GemmP& p, f32x4_t (&acc)[2][4][2][4], int z, int m_wave, int n_wave, int lane,
                                                     char* stage, const Deq<F8>& dq) {
  // every per-lane offset below is cheap to recompute; hidden from loop-invariant code motion, or hipcc computes the lot once in front
  // of the persistent loop and then spills it (73 registers, reloaded from scratch between this epilogue's stores)
  asm volatile("" : "+v"(lane));
  const int mlane = lane & 15, ng = lane >> 4;
  const int Dm = p.q_H * 128;
  const int sec = n_wave / Dm;  // 0 = q, 1 = k, 2 = v
  const int head = (n_wave - sec * Dm) >> 7;
  uint2 bvp[8];  // bias of this lane's 4 columns per 16-column block, kept as packed bf16 pairs (registers are scarce here)
  static_for<8>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int n = n_wave + j * 16 + ng * 4;
    bvp[j] = (p.bias && n + 3 < p.N) ? *(const uint2*)(p.bias + n) : make_uint2(0u, 0u);
  });
  auto bias_of = [&](int j, int r) {
    const uint32_t u = (r & 2) ? bvp[j].y : bvp[j].x;
    return __uint_as_float((r & 1) ? (u & 0xffff0000u) : (u << 16));
  };
  const TokMap tmap = tok_map(p, z, m_wave);
  auto token = [&](int m, int& b, int& st) { tok_of(tmap, m - m_wave, b, st); };  // rows m_wave <= m < m_wave + 128
  // measurement only (tools/qkv_parts_build.sh builds one library per value with -DX2I_QKV_ABL=<n>; wrong results by design; never defined in
  // the product or the ablate library -- a RUN-TIME switch here costs the instantiation registers and distorts what it measures):
  // 86 = no cos / sin loads, 87 = no 16-lane RMS reduction, 88 = no Q / K stores, 89 = no V^T stores, 90 = q / k tiles parked only (no
  // read-back, arithmetic or stores), 91 = v tiles parked only, 92 = no epilogue at all
#ifdef X2I_QKV_ABL
  constexpr int qabl = X2I_QKV_ABL;
#else
  constexpr int qabl = 0;
#endif
  if constexpr (qabl == 92) {
    asm volatile("" ::X2I_GEMM256P_OPS_ACC_IN(acc));
    return;
  }
  if (sec < 2) {
    const int c = lane & 15, rsub = lane >> 4;  // 8-dim chunk of the head; row of the pass
    float w[8];
    {
      const bf16x8_t wv = *(const bf16x8_t*)((sec ? p.q_nk : p.q_nq) + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = bf16_to_f32((bf16_t)wv[j]) * (sec ? 1.f : p.q_qs);
    }
    // Q / K rows and the RoPE tables through buffer descriptors with 32-bit offsets (the launcher keeps Q / K below 2 GB): no 64-bit
    // address arithmetic per row, and rows at or behind M get the out-of-range offset instead of a branch around their store
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    __amdgpu_buffer_rsrc_t q_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(sec ? p.q_K : p.q_Q), 0, 0x7ffffff0u, 0x00020000);
    __amdgpu_buffer_rsrc_t cos_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.q_cos, 0, 0x7ffffff0u, 0x00020000);
    __amdgpu_buffer_rsrc_t sin_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.q_sin, 0, 0x7ffffff0u, 0x00020000);
    // pair-form RoPE table (x2i_qkv_desc: sin == NULL): f32 [S][64][2] = (cos, sin) per dim pair.  The two table forms are two instantiations of
    // the body below (a register array indexed by a run-time set would live in scratch memory)
    auto qk_body = [&](auto pairs_c) {
    constexpr bool pairs = decltype(pairs_c)::value;
    // Table rows of a half chunk (2 passes = 8 tokens), one register SET per half.  Pair form: 2 x 16 bytes per pass (a lane's four dim pairs),
    // two sets, and the loads of half n + 2 are requested as soon as half n has been computed -- TWO halves ahead: one half (~0.4 us) did not
    // cover a table load's latency behind the epilogue's own stores, and the q / k tiles spent most of their ~10 us waiting for it
    // (tools/qkv_parts.py, profiles/r06b_*).  The separate cos / sin tables need twice the registers per half: one set, one half ahead.
    f32x4_t cs[pairs ? 2 : 1][2][4];   // [set][pass][cos 0..3 | cos 4..7 | sin 0..3 | sin 4..7]  (pair form: [pairs 0, 1 | pairs 2, 3], entries 2, 3 unused)
    uint32_t qoff[pairs ? 2 : 1][2];   // ... and where the pass's 16 bytes of Q / K go
    auto load_cs = [&](int half, auto set_c) {
      constexpr int set = decltype(set_c)::value;  // half = 2 passes (8 tokens) of the 16-token chunks
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int dm = half * 8 + ps * 4 + rsub;
        const bool valid = m_wave + dm < p.M;
        int b, st;
        tok_of(tmap, valid ? dm : 0, b, st);
        const uint32_t co = (uint32_t)(st * 128 + c * 8) * 4u;
        if constexpr (qabl == 86) {
          cs[set][ps][0] = cs[set][ps][1] = (f32x4_t){1.f, 1.f, 1.f, 1.f};
          cs[set][ps][2] = cs[set][ps][3] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        } else if constexpr (pairs) {
          cs[set][ps][0] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(cos_rsrc, co, 0, 0));
          cs[set][ps][1] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(cos_rsrc, co + 16, 0, 0));
        } else {
          cs[set][ps][0] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(cos_rsrc, co, 0, 0));
          cs[set][ps][1] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(cos_rsrc, co + 16, 0, 0));
          cs[set][ps][2] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(sin_rsrc, co, 0, 0));
          cs[set][ps][3] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(sin_rsrc, co + 16, 0, 0));
        }
        qoff[set][ps] = valid ? (uint32_t)(((b * p.q_H + head) * p.q_Spad + st) * 128 + c * 8) * 2u : 0x80000000u;
      }
    };
    auto park = [&](auto qc) {  // chunk q16 (16 tokens x 128 dims) as bf16(acc + bias) into staging buffer q16 & 1
      constexpr int q16 = decltype(qc)::value;
      constexpr int c2 = q16 >> 1, rr = q16 & 1;
      char* buf = stage + (q16 & 1) * 4096;
      static_for<2>([&](auto hc) {
        constexpr int h = decltype(hc)::value;
        static_for<4>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          float a[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float t;
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(acc[h][c2][rr][j][r]));
            if constexpr (F8) a[r] = fmaf(t * dq.sw[h * 4 + j][r], dq.sr[c2][rr], bias_of(h * 4 + j, r));
            else a[r] = t + bias_of(h * 4 + j, r);
          }
          *(uint2*)(buf + mlane * 256 + (((h * 8 + j * 2 + (ng >> 1)) ^ mlane) << 4) + ((ng & 1) << 3)) =
              make_uint2(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]));
        });
      });
    };
    asm volatile("" ::: "memory");
    load_cs(0, std::integral_constant<int, 0>{});
    if constexpr (pairs) load_cs(1, std::integral_constant<int, 1>{});
    park(std::integral_constant<int, 0>{});
    static_for<8>([&](auto qc) {
      constexpr int q16 = decltype(qc)::value;
      char* buf = stage + (q16 & 1) * 4096;
      if constexpr (qabl == 90) {
        if constexpr (q16 + 1 < 8) park(std::integral_constant<int, q16 + 1>{});
        return;
      } else {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the chunk is parked
      bf16x8_t xv[4];                                       // its four passes' rows, requested together: one LDS round trip per chunk
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = k * 4 + rsub;
        xv[k] = *(const bf16x8_t*)(buf + row * 256 + ((c ^ row) << 4));
      }
      __builtin_amdgcn_sched_barrier(0);
      static_for<2>([&](auto hfc) {
        constexpr int hf = decltype(hfc)::value;  // passes 2hf, 2hf + 1 of this chunk = half-chunk index 2 q16 + hf
        constexpr int half = 2 * q16 + hf;
        constexpr int set = pairs ? (half & 1) : 0;
        u32x4 outv[2];
        uint32_t so[2];
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          float x[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] = bf16_to_f32((bf16_t)xv[hf * 2 + ps][j]);
          float ss = sumsq8(x);
          if constexpr (qabl != 87) {
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);  // the 16 lanes of this token
          }
          const float r = rms_rsqrt128(ss, p.q_eps);
          const f32x4_t t0 = cs[set][ps][0], t1 = cs[set][ps][1];
          float csv[8], snv[8];
          if constexpr (pairs) {
            csv[0] = csv[1] = t0[0]; snv[0] = snv[1] = t0[1]; csv[2] = csv[3] = t0[2]; snv[2] = snv[3] = t0[3];
            csv[4] = csv[5] = t1[0]; snv[4] = snv[5] = t1[1]; csv[6] = csv[7] = t1[2]; snv[6] = snv[7] = t1[3];
          } else {
            const f32x4_t t2 = cs[0][ps][2], t3 = cs[0][ps][3];
#pragma unroll
            for (int j = 0; j < 4; ++j) csv[j] = t0[j], csv[4 + j] = t1[j], snv[j] = t2[j], snv[4 + j] = t3[j];
          }
          float o8[8];
          norm_rope8(x, r, w, csv, snv, o8);
#pragma unroll
          for (int j = 0; j < 4; ++j) outv[ps][j] = pack_bf16x2(o8[2 * j], o8[2 * j + 1]);
          so[ps] = qoff[set][ps];
        }
        __builtin_amdgcn_sched_barrier(0);
        // the next table rows in front of this half's stores (a load queued behind a store is only known to have returned once the store has)
        if constexpr (pairs) {
          if constexpr (half + 2 < 16) load_cs(half + 2, std::integral_constant<int, set>{});
        } else {
          if constexpr (half + 1 < 16) load_cs(half + 1, std::integral_constant<int, 0>{});
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (qabl != 88) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) __builtin_amdgcn_raw_buffer_store_b128(outv[ps], q_rsrc, so[ps], 0, 0);
        } else {
          asm volatile("" ::"v"(outv[0]), "v"(outv[1]));
        }
      });
      // (parking the next chunk HERE, behind this chunk's stores.  In front of the compute it would hide one more LDS round trip, but
      // with that order the kernel's Q, K AND V^T all come out wrong -- deterministically, also with a full LDS wait behind the park and
      // with M0 declared clobbered -- for a reason not found; tools/qkv_persistent_vs_onetile.py is the check)
      if constexpr (q16 + 1 < 8) park(std::integral_constant<int, q16 + 1>{});
      }
    });
    };
    if (p.q_sin == nullptr) qk_body(std::true_type{});
    else qk_body(std::false_type{});
  } else {
    const int ch_lo = lane & 7, dp_lo = lane >> 3;
    const bool aligned = ((p.q_tok_off | p.q_rpb | p.q_row0 | p.M | p.q_Spad) & 7) == 0;
    auto fsw = [](int row) { return ((row >> 3) ^ (row >> 1)) & 7; };
    static_for<4>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      constexpr int h = q >> 1, cp = q & 1;  // dims 64h .. 64h+63 of the head, tokens 64cp .. 64cp+63 of the wave
      __builtin_amdgcn_sched_barrier(0);
      static_for<2>([&](auto cc) {
        constexpr int c2 = 2 * cp + decltype(cc)::value;
        static_for<2>([&](auto rc) {
          constexpr int rr = decltype(rc)::value;
          const int row = decltype(cc)::value * 32 + rr * 16 + mlane;
          static_for<4>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            float a[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float t;
              asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(acc[h][c2][rr][j][r]));
              if constexpr (F8) a[r] = fmaf(t * dq.sw[h * 4 + j][r], dq.sr[c2][rr], bias_of(h * 4 + j, r));
              else a[r] = t + bias_of(h * 4 + j, r);
            }
            *(uint2*)(stage + row * 128 + (((j * 2 + (ng >> 1)) ^ fsw(row)) << 4) + ((ng & 1) << 3)) =
                make_uint2(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]));
          });
        });
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (qabl == 91) return;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int dp = it * 8 + dp_lo;  // dim pair of this 64-dim half
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int row = ch_lo * 8 + k;
          v[k] = *(const uint32_t*)(stage + row * 128 + (((dp >> 2) ^ fsw(row)) << 4) + ((dp & 3) << 2));
        }
        const int m = m_wave + cp * 64 + ch_lo * 8;
        if constexpr (qabl == 89) {
          asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
        } else if (m < p.M) {
          const int d = h * 64 + dp * 2;
          int b, st;
          token(m, b, st);
          bf16_t* row0 = p.q_VT + (((long long)b * p.q_H + head) * 128 + d) * p.q_Spad;
          if (aligned) {
            union { bf16x8_t v8; uint32_t uu[4]; uint2 h2[2]; } lo, hi;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              lo.uu[k] = (v[2 * k] & 0xffffu) | (v[2 * k + 1] << 16);
              hi.uu[k] = (v[2 * k] >> 16) | (v[2 * k + 1] & 0xffff0000u);
            }
            if (p.q_vperm) {   // span-permuted V^T (x2i_vt_pos): the run of eight tokens is two runs of four, eight positions apart
              const int ps = x2i_vt_pos(st, 1);
              *(uint2*)(row0 + ps) = lo.h2[0]; *(uint2*)(row0 + ps + 8) = lo.h2[1];
              *(uint2*)(row0 + p.q_Spad + ps) = hi.h2[0]; *(uint2*)(row0 + p.q_Spad + ps + 8) = hi.h2[1];
            } else {
              *(bf16x8_t*)(row0 + st) = lo.v8;
              *(bf16x8_t*)(row0 + p.q_Spad + st) = hi.v8;
            }
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              if (m + k < p.M) {
                int bk, sk;
                token(m + k, bk, sk);
                bf16_t* rk = p.q_VT + (((long long)bk * p.q_H + head) * 128 + d) * p.q_Spad + x2i_vt_pos(sk, p.q_vperm);
                rk[0] = (bf16_t)(v[k] & 0xffffu);
                rk[p.q_Spad] = (bf16_t)(v[k] >> 16);
              }
            }
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // gathered before the next chunk is parked
    });
  }
}

// One unit of a workgroup's list: K-tiles [k0, k0 + len) of the output tile with virtual block id vb.  Whole tiles have k0 = 0,
// len = nk; the tiles of the last, partly filled round are cut along K into segments handed from workgroup to workgroup ("stream-K",
// chained: the next segment CONTINUES the accumulators of the previous one, so every output is summed in exactly the order of an
// undivided tile -- bit-identical results, no reduction of partial sums).
struct Unit {
  int vb, k0, len, slab;  // slab = index of the split tile (partial-accumulator slab and progress flag); -1: whole tile
  int kind = 0, tag = 0;  // FX units: kind 1 = park this range's sums in slab[workgroup] (flag <- tag), 2 = finish the tile: add the `slab`
                          // parked ranges of the workgroups in front (flags == tag), epilogue; tag = batch item + 1
};

template <bool PAIR> using GemmArg = std::conditional_t<PAIR, GemmP2, GemmP>;
__device__ __forceinline__ const GemmP& prob(const GemmP& a, int) { return a; }
__device__ __forceinline__ const GemmP& first(const GemmP& a) { return a; }
__device__ __forceinline__ const GemmP& first(const GemmP2& a) { return a.p[0]; }
__device__ __forceinline__ const GemmP& second(const GemmP& a) { return a; }
__device__ __forceinline__ const GemmP& second(const GemmP2& a) { return a.p[1]; }
// problem `sel` of a grouped launch, field by field (a dynamic index into the kernel argument would move it to scratch memory and
// its fields into VGPRs; a scalar select per field keeps every one of them an SGPR)
__device__ __forceinline__ GemmP prob(const GemmP2& a, int sel) {
  const GemmP &x = a.p[0], &y = a.p[1];
  GemmP q;
#define X2I_F(f) q.f = sel ? y.f : x.f;
  X2I_F(A) X2I_F(a_bs) X2I_F(lda) X2I_F(W) X2I_F(ldw) X2I_F(w_bs) X2I_F(bias) X2I_F(C) X2I_F(c_bs) X2I_F(ldc) X2I_F(C2) X2I_F(act2)
  X2I_F(gate) X2I_F(gate_bs) X2I_F(res) X2I_F(r_bs) X2I_F(ldr) X2I_F(bias2) X2I_F(bias2_bs) X2I_F(M) X2I_F(N) X2I_F(K) X2I_F(act)
  X2I_F(out_f32) X2I_F(tilesM) X2I_F(tilesN) X2I_F(cH) X2I_F(cW) X2I_F(cCin) X2I_F(cOW) X2I_F(cKW) X2I_F(cStride) X2I_F(cPad) X2I_F(cUp) X2I_F(cPadW) X2I_F(cRowPitch) X2I_F(cKorder) X2I_F(cMom) X2I_F(cMomBlocks)
  X2I_F(q_on) X2I_F(q_H) X2I_F(q_Spad) X2I_F(q_tok_off) X2I_F(q_rpb) X2I_F(q_row0) X2I_F(q_vperm) X2I_F(gm) X2I_F(q_eps) X2I_F(q_qs) X2I_F(q_nq) X2I_F(q_nk)
  X2I_F(q_cos) X2I_F(q_sin) X2I_F(q_Q) X2I_F(q_K) X2I_F(q_VT) X2I_F(f_sa) X2I_F(f_sa_bs) X2I_F(f_sw) X2I_F(f_alpha) X2I_F(f_oinv) X2I_F(f_out8)
  X2I_F(nbatch) X2I_F(sk_on) X2I_F(sk_slabs) X2I_F(sk_flags) X2I_F(fx_v0)
#undef X2I_F
  return q;
}

// PAIR: a GROUPED launch of two problems with the same K and the same epilogue kind (the image-stream and the text-stream linear of
// a double block: same layer type, different weights, 8 : 1 in rows): one tile list, problem 1's tiles behind problem 0's, so the
// small problem's tiles ride in the rounds of the large one instead of under-filling a launch of their own.
// F8: e4m3 operands (x2i_gemm_fp8 / x2i_gemm_qkv_fp8) -- the same unit list, stream-K chain, LDS images and epilogue pipelines around the
// K-loop of gen_gemm256f8.py (one K = 128 MFMA per accumulator and K-tile, fragments in v128..v255); every byte offset below is
// computed with the element size ES.  OUT8: e4m3 output (epilogue_chunked_pipe_e4m3).
// FX: parallel split with fix-up for launches whose tiles cannot fill the chip (fewer tiles per batch item than CUs, deep K): the tiles of
// ONE batch item are cut along K so that every workgroup has (about) the same number of K-tiles (see the per-XCD rule below), every part
// starts its accumulators from zero -- no workgroup waits for another one's K-loop -- and the workgroup that finishes a tile adds the parked
// sums of its other parts in its epilogue (FXADD).  The cuts depend on the item's tile count, K and G only and every item is cut the same
// way: a sample's result does not depend on the batch it rides in.  NOT bit-identical to the one-tile kernels (the K sum is associated
// differently); deterministic.
template <int ACT, bool RES, bool HASC2, bool QKV = false, bool PAIR = false, bool F8 = false, bool OUT8 = false, bool FX = false>
__global__ __launch_bounds__(256) void gemm256p_kernel(GemmArg<PAIR> pp) {
  constexpr int ES = F8 ? 1 : 2;            // bytes per operand element
  constexpr int KT = F8 ? 128 : BK;         // elements per K-tile (128 bytes of a row either way)
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 tiles][A image 32 KiB | W image 32 KiB][4 waves x 8 KiB staging]
  const GemmP& p = first(pp);  // launch-wide fields (K, nbatch of problem 0, stream-K workspace) live in problem 0's descriptor
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int TT0 = p.tilesM * p.tilesN * p.nbatch;  // virtual block vb < TT0: problem 0, item z = vb / T, tile vb % T of that item
  int TT = TT0;
  if constexpr (PAIR) TT += second(pp).tilesM * second(pp).tilesN * second(pp).nbatch;
  const int G = gridDim.x, w = blockIdx.x;
  const int nk = p.K / KT;

  auto tile_of = [&](int vb, int& sel, int& z, int& m0, int& n0) {  // the XCD-aware patch order of gemm256.hip, applied to the virtual block id
    sel = PAIR ? min(max(vb - TT0 + 1, 0), 1) : 0;
    const GemmP& q = prob(pp, sel);
    const int T = q.tilesM * q.tilesN;
    const int v = vb - sel * TT0;
    z = v / T;
    int bid = v - z * T;
    const int qq = T >> 3, r = T & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
    const int GM = q.gm;
    const int per_group = GM * q.tilesN;
    const int group = bid / per_group;
    const int first_m = group * GM;
    const int gsize = min(q.tilesM - first_m, GM);
    m0 = (first_m + (bid % per_group) % gsize) * BM2;
    n0 = ((bid % per_group) / gsize) * BN2;
  };
  const int khl = lane >> 5, r8 = (lane >> 2) & 7, cphys = lane & 3;
  const int kby = khl * 64 + ((cphys ^ (3 * (wave & 1))) << 4);  // byte within the K-tile line; group parity = wave parity (4 pieces per row-group step)
  auto offsets = [&](int sel, int z, int m0, int n0, uint32_t (&va)[8], uint32_t (&vw)[8]) {  // (W is shared by the batch items: w_bs == 0)
    const GemmP& q = prob(pp, sel);
#ifdef X2I_ABLATION
    // measurement only (tools/gemm_fabric_price.py; wrong results by design): every workgroup READS the operand panels of a folded tile
    // coordinate -- 75: tile (0, 0) for everybody (the panels stay in every XCD's L2: no fabric traffic), 76: a 4 x 8 tile patch (18.9 MB at
    // K = 3072: out of L2, inside the Infinity Cache: fabric traffic as in the product, no HBM traffic) -- while the epilogue writes the real tile
    if (p.act2 == 75) m0 = 0, n0 = 0;
    if (p.act2 == 76) m0 &= 1023, n0 &= 2047;
#endif
    const long long zoff = (long long)z * q.a_bs;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int row = (jj * 4 + wave) * 8 + r8;
      va[jj] = (m0 + row < q.M) ? (uint32_t)((zoff + (long long)(m0 + row) * q.lda) * ES + kby) : 0x80000000u;
      vw[jj] = (n0 + row < q.N) ? (uint32_t)(((long long)(n0 + row) * q.ldw) * ES + kby) : 0x80000000u;
    }
  };
  auto uni = [](auto v) { return (decltype(v))__builtin_amdgcn_readfirstlane((int)v); };  // (workgroup-uniform by construction; makes it provable)
  auto mk_rsrc = [&](const void* ptr, uint32_t bytes) {
    const unsigned long long a = (unsigned long long)(uintptr_t)ptr;
    const unsigned long long au = ((unsigned long long)(unsigned)uni((int)(a >> 32)) << 32) | (unsigned)uni((int)(a & 0xffffffffu));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)au, 0, (uint32_t)uni((int)bytes), 0x00020000);
  };
  auto a_rsrc_of = [&](int sel) {
    const GemmP& q = prob(pp, sel);
    return mk_rsrc(q.A, (uint32_t)(((long long)(q.nbatch - 1) * q.a_bs + (long long)(q.M - 1) * q.lda + q.K) * ES));  // < 2 GB (launcher)
  };
  auto w_rsrc_of = [&](int sel) {
    const GemmP& q = prob(pp, sel);
    return mk_rsrc(q.W, (uint32_t)(((long long)(q.N - 1) * q.ldw + q.K) * ES));
  };
  // ---- FX: what this workgroup does for ONE batch item (the same for every item): fx_cnt units, unit u = K-tiles [fx_k0, fx_k0 + fx_len) of
  // the XCD's tile fx_t0 + u * fx_tstep of problem fx_sel; fx_kind 1 = park the sums (P), 2 = finish the tile (F: add fx_npre parked parts)
  int fx_sel = 0, fx_Tp = 0, fx_nbz = 0, fx_kind = 0, fx_cnt = 0, fx_t0 = 0, fx_tstep = 1, fx_k0 = 0, fx_len = 0, fx_npre = 0, fx_modeB = 0, fx_slab_base = 0, fx_part = 0;
  const int fx_x = w & 7;
  if constexpr (FX) {
    // Per XCD: workgroup w runs on XCD w & 7 (round-robin dispatch) and the tile order gives that XCD the tiles bid = 8 t + (w & 7) of an
    // item (tile_of: its L2 then holds their operand panels).  The XCD's G / 8 workgroups share THOSE Tx tiles, v of them per problem:
    //   v < 2 Tx  ("B"): workgroup t < Tx multiplies K-tiles [0, c) of tile t and finishes it; the other nt = v - Tx workgroups multiply
    //                    the tails [c, nk) of tiles j, j + nt, ... one after the other and park them; c = nk tw / (tw + 1), tw = ceil(Tx / nt):
    //                    a tail workgroup's tw tails take as long as a head, and every head's tail is parked before the head needs it
    //   v >= 2 Tx ("A"): tile t is cut into v / Tx (the first v % Tx tiles: one more) equal parts, one workgroup each; the last part finishes
    // Every workgroup of an XCD sweeps K in step with the others on the same tile row (the heads all start at K-tile 0), so the A panel
    // of a tile row is fetched once per XCD -- a tile-major stream-K cut (every workgroup at another K offset) ran 5-40 % SLOWER than
    // whole tiles: profiles/r04g_fx_bench_streamk_cut.log.
    const int wi = w >> 3, Gx = G >> 3;
    const int v0 = p.fx_v0;                     // of the XCD's Gx workgroups, v0 take problem 0's tiles
    fx_sel = PAIR ? min(max(wi - v0 + 1, 0), 1) : 0;
    const GemmP& q = prob(pp, fx_sel);
    const int li = wi - fx_sel * v0, v = fx_sel ? Gx - v0 : v0;
    fx_Tp = q.tilesM * q.tilesN;
    fx_nbz = q.nbatch;
    fx_slab_base = fx_sel * (first(pp).tilesM * first(pp).tilesN);
    const int Tx = fx_Tp >> 3;                  // (tiles per item: a multiple of 8, launcher)
    if (v < 2 * Tx) {
      fx_modeB = 1;
      const int nt = v - Tx;                    // (0: every workgroup takes a whole tile)
      const int tw = nt > 0 ? (Tx + nt - 1) / nt : 0;
      const int c = nt > 0 ? min(max(nk * tw / (tw + 1), 3), nk - 3) : nk;
      if (li < Tx) {
        fx_kind = 2; fx_cnt = 1; fx_t0 = li; fx_k0 = 0; fx_len = c; fx_npre = nt > 0 ? 1 : 0;
      } else {
        const int j = li - Tx;
        fx_kind = 1; fx_t0 = j; fx_tstep = nt; fx_cnt = (Tx - j + nt - 1) / nt; fx_k0 = c; fx_len = nk - c;
      }
    } else {
      const int base = v / Tx, extra = v - base * Tx;     // tiles t < extra: base + 1 parts
      const int cutw = extra * (base + 1);
      const int t = li < cutw ? li / (base + 1) : extra + (li - cutw) / base;
      const int P = t < extra ? base + 1 : base;
      const int j = li < cutw ? li - t * (base + 1) : (li - cutw) - (t - extra) * base;
      const int k_lo = (int)((long long)nk * j / P), k_hi = (int)((long long)nk * (j + 1) / P);
      fx_cnt = 1; fx_t0 = t; fx_k0 = k_lo; fx_len = k_hi - k_lo;
      fx_kind = P > 1 ? 3 : 2;      // 3: every part parks its range and finishes the chunks j, j + P, ... (epilogue_fx_slice); one part: a whole tile
      fx_npre = P - 1;
      fx_part = j;
    }
  }
  // ---- this workgroup's unit list: S whole tiles (vb = w + s*G) and, with stream-K, up to two segments of the last round's tiles
  int S = TT / G;
  auto hasXlen = [](const Unit& u) { return u.len > 0; };
  Unit segH = {0, 0, 0, -1}, segX = {0, 0, 0, -1};  // head segment (k0 = 0, runs first) / continuing segment (k0 > 0)
  int posX = 0;                                     // whole tiles in front of segX
  if (p.sk_on) {
    const int r = TT - S * G;                       // tiles of the last round: K-tile range [0, r*nk) is dealt out over Gp workgroups
    const long long R = (long long)r * nk;
    const int Gp = (int)min((long long)G, R / 6);    // segments of at least 6 K-tiles
    if (w < Gp) {
      auto snap = [&](long long b) {                // no segment shorter than 3 K-tiles: move a cut that close to a tile edge onto it
        const int m = (int)(b % nk);
        return (m != 0 && m < 3) ? b - m : (m > nk - 3 ? b + (nk - m) : b);
      };
      const long long lo = snap(R * w / Gp), hi = snap(R * (w + 1) / Gp);
      if (hi > lo) {
        const int e0 = (int)(lo / nk), e1 = (int)((hi - 1) / nk);
        const int k0 = (int)(lo - (long long)e0 * nk);
        const Unit first = {S * G + e0, k0, (e1 > e0 ? nk : (int)(hi - (long long)e0 * nk)) - k0, e0};
        if (k0 == 0) segH = first; else segX = first;
        if (e1 > e0) segH = Unit{S * G + e1, 0, (int)(hi - (long long)e1 * nk), e1};  // (the launcher keeps shares below one tile)
        if (segH.len == nk) segH.slab = -1;        // an undivided tile after all
        // a continuing segment runs behind a share of the whole tiles that grows with its place in the tile's chain (the closing
        // segment last), so that its predecessor has normally published before it is reached
        posX = hasXlen(segX) ? min(S, (int)((long long)S * segX.k0 / max(1, nk - segX.len))) : 0;
      }
    }
  } else if (w < TT - S * G) {
    S += 1;  // no splitting: the last round's tiles are whole tiles of the first workgroups
  }
  const int hasH = segH.len > 0, hasX = segX.len > 0;
  const int n_units = FX ? fx_cnt * fx_nbz : S + hasH + hasX;
  auto unit = [&](int i) -> Unit {
    if constexpr (FX) {
      if (i >= n_units) return Unit{-1, 0, 0, -1};
      const int zi = i / fx_cnt, uu = i - zi * fx_cnt;
      const int tile = (fx_t0 + uu * fx_tstep) * 8 + fx_x;
      Unit u;
      u.vb = fx_sel * TT0 + zi * fx_Tp + tile;
      u.k0 = fx_k0;
      u.len = fx_len;
      // P: the slab this range is parked in -- "B": the tile's (a tail workgroup parks several per item), "A": the workgroup's;
      // F: the first of the fx_npre slabs to add (they are 8 apart in "A": the XCD's workgroups)
      u.slab = fx_modeB ? fx_slab_base + tile : (fx_kind == 1 ? w : (fx_kind == 3 ? w - 8 * fx_part : w - 8 * fx_npre));   // (kind 3: the tile's FIRST slab; this part's own is w)
      u.kind = fx_kind;
      u.tag = zi + 1;
      return Unit{uni(u.vb), uni(u.k0), uni(u.len), uni(u.slab), uni(u.kind), uni(u.tag)};
    }
    Unit u{w + (i - hasH) * G, 0, nk, -1};
    if (i >= n_units) u = Unit{-1, 0, 0, -1};
    else if (hasH && i == 0) u = segH;
    else if (hasX) {
      const int j = i - hasH;
      if (j == posX) u = segX;
      else if (j > posX) u.vb -= G;
    }
    return Unit{uni(u.vb), uni(u.k0), uni(u.len), uni(u.slab)};
  };


  const int frow = lane & 15;
  const uint32_t frag = (frow >> 3) * 1024 + (frow & 7) * 64 + (((lane >> 4) ^ (3 * ((frow >> 3) & 1))) << 4);
  uint32_t la = (uint32_t)(uintptr_t)smem + wm * 8 * 2048 + frag;
  uint32_t lw = (uint32_t)(uintptr_t)smem + 32768 + wn * 8 * 2048 + frag;
  uint32_t dma = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem + wave * 1024);
  char* stage = smem + P_STAGE_OFF + wave * P_STAGE_WAVE;
  const uint32_t slab_vo = (uint32_t)tid * 16;  // slab layout [64 accumulator tiles][256 threads][16 B]

  auto slab_rsrc = [&](int slab) {
    const unsigned long long a = (unsigned long long)(uintptr_t)p.sk_slabs + (unsigned long long)slab * SK_SLAB_BYTES;
    const unsigned long long au = ((unsigned long long)(unsigned)uni((int)(a >> 32)) << 32) | (unsigned)uni((int)(a & 0xffffffffu));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)au, 0, (uint32_t)SK_SLAB_BYTES, 0x00020000);
  };
  if (n_units == 0) return;  // (workgroup-uniform)
#ifdef X2I_ABLATION
  if (p.act2 >= 81 && p.act2 <= 85) {  // measurement only: start offsets -- 82: by workgroup, spread over 80 us; 81 / 83 / 84 / 85: by XCD (w & 7) x 10 / 5 / 2.5 / 20 us
    int n = (w & 7) * 10;
    if (p.act2 == 82) n = (((w * 167) & 255) * 80) >> 8;
    if (p.act2 == 83) n = (w & 7) * 5;
    if (p.act2 == 85) n = (w & 7) * 20;
    if (p.act2 == 84) {
      for (int i = 0; i < (w & 7) * 5; ++i) __builtin_amdgcn_s_sleep(13);  // ~ 0.5 us
      n = 0;
    }
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(27);
  }
#endif
  // ---- FX, every part parks (kind 3): flags and the deferred slice (see the unit loop)
  auto fx_flag = [&](int buf, int slab) { return p.sk_flags + buf * 512 + slab; };
  auto fx_spin = [&](unsigned* f, unsigned want, unsigned shift, unsigned mask) {   // (thread 0) bounded: a lost partner leaves a marker, not a hung GPU
    int spins = 0;
    while (((__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> shift) & mask) != want) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > (1 << 22)) {
        __hip_atomic_store(p.sk_flags + SK_ERR_SLOT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  };
  auto fx_reclaim = [&](int buf) {   // this workgroup's slab of buffer `buf` is free once all P parts have read it
    if (tid == 0) {
      fx_spin(fx_flag(buf, w), (unsigned)(fx_npre + 1), 0u, 0xffu);
      __hip_atomic_store(fx_flag(buf, w), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  };
  auto fx_slice_of = [&](int usel, int uz, int um0, int un0, int uslab0, int utag, int buf) {
    const int P = fx_npre + 1;
    if (tid == 0) {
      for (int q = 0; q < P; ++q) fx_spin(fx_flag(buf, uslab0 + 8 * q), (unsigned)utag, 8u, 0xffffffu);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    epilogue_fx_slice(prob(pp, usel), uz, um0 + wm * 128, un0 + wn * 128, lane, stage, uslab0 + 256 * buf, P, fx_part, tid);
    __syncthreads();   // every wave has read the slabs
    if (tid == 0)
      for (int q = 0; q < P; ++q) __hip_atomic_fetch_add(fx_flag(buf, uslab0 + 8 * q), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  int pend_valid = 0, pend_sel = 0, pend_z = 0, pend_m0 = 0, pend_n0 = 0, pend_slab0 = 0, pend_tag = 0;
  Unit cur = unit(0);
  int sel, z, m0, n0;
  tile_of(cur.vb, sel, z, m0, n0);
  uint32_t va[8], vw[8], na[8], nw[8];
  offsets(sel, z, m0, n0, va, vw);
  __amdgpu_buffer_rsrc_t a_rsrc = a_rsrc_of(sel), w_rsrc = w_rsrc_of(sel);
  bf16x8_t fr[F8 ? 1 : 32];  // bf16: wa 0..7 | wb 8..15 | aa 16..23 | ab 24..31 (e4m3: fragments are v128..v255 inside the statements)
  uint32_t s_koff, s_it, s_so;
  int k0b = cur.k0 * 128;       // bytes per K-tile and row
  if constexpr (F8) {
    asm volatile(X2I_GEMM256F8_PRO
                 : [koff] "=&s"(s_koff)
                 : X2I_GEMM256W_OPS_VOFF(va, vw), [dma] "s"(dma), [ra] "s"(a_rsrc), [rw] "s"(w_rsrc), [k0b] "s"(k0b)
                 : "memory", "scc", "m0");
  } else {
    asm volatile(X2I_GEMM256P_PRO
                 : X2I_GEMM256P_OPS_FRAG0_OUT(fr), [koff] "=&s"(s_koff)
                 : X2I_GEMM256W_OPS_VOFF(va, vw), [la] "v"(la), [lw] "v"(lw), [dma] "s"(dma), [ra] "s"(a_rsrc), [rw] "s"(w_rsrc), [k0b] "s"(k0b)
                 : "memory", "scc", "m0");
  }
  for (int ui = 0;; ++ui) {
    const Unit nxt = unit(ui + 1);
    const bool has_next = nxt.vb >= 0;
    // (integer arithmetic, not a comparison, and no resource descriptor behind a branch: both would end up in VGPRs)
    int nsel = PAIR ? min(max(nxt.vb - TT0 + 1, 0), 1) : 0, nz = 0, nm0 = 0, nn0 = 0;
    if constexpr (PAIR) a_rsrc = a_rsrc_of(sel), w_rsrc = w_rsrc_of(sel);   // (not carried around the loop: recomputed, provably scalar)
    const __amdgpu_buffer_rsrc_t na_rsrc = PAIR ? a_rsrc_of(nsel) : a_rsrc, nw_rsrc = PAIR ? w_rsrc_of(nsel) : w_rsrc;
    if (has_next) {
      tile_of(nxt.vb, nsel, nz, nm0, nn0);
      offsets(nsel, nz, nm0, nn0, na, nw);
    } else {
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) na[jj] = nw[jj] = 0x80000000u;  // behind the last unit: every piece out of range (zero fill, no fetch)
    }
    const int nk0b = nxt.k0 * 128, len = cur.len;
    f32x4_t acc[2][4][2][4];
    const int fromp = FX ? 0 : min(cur.k0, 1);  // (integer arithmetic, not a comparison: hipcc materialises an i1 in a VGPR, which cannot feed "s"); FX ranges always start from zero
    if (fromp) {
      // continue a tile: wait until the predecessor has published K-tiles [0, k0) (flag == k0) -- bounded, so that a lost
      // predecessor leaves a marker instead of a hung GPU
      unsigned* flag = p.sk_flags + cur.slab;
      if (tid == 0) {
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)cur.k0) {
          __builtin_amdgcn_s_sleep(32);
          if (++spins > (1 << 22)) {
            __hip_atomic_store(p.sk_flags + SK_ERR_SLOT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      __syncthreads();
    }
    // ONE statement pair for both kinds of unit (see gen_gemm256w.py): LOAD_PARTIAL fetches the predecessor's accumulators or does
    // nothing; MAIN continues from them or starts from zero (first K-tile with C = 0, no clearing pass)
    __amdgpu_buffer_rsrc_t s_rsrc = slab_rsrc(cur.slab < 0 ? 0 : cur.slab);
    asm volatile(X2I_GEMM256P_LOAD_PARTIAL
                 : X2I_GEMM256P_OPS_ACC_OUT(acc), [so] "=&s"(s_so)
                 : [vo] "v"(slab_vo), [rs] "s"(s_rsrc), [fromp] "s"(fromp)
                 : "memory", "scc");
    const int zs = 1 - fromp;
    Deq<F8> dq;   // e4m3: the dequantisation scales of this unit's rows / columns, requested in front of its K-loop
    deq_load<F8>(prob(pp, sel), z, m0 + wm * 128, n0 + wn * 128, lane, dq);
#ifdef X2I_ABLATION
    // measurement only: 100 MHz timestamps of every unit's K-loop (start, end) of the first 4 workgroups and one per XCD, into p.bias2
    unsigned long long* tdbg = (p.act2 >= 79 && tid == 0 && w < 16) ? (unsigned long long*)p.bias2 + w * 64 : nullptr;
    if (tdbg && ui < 31) tdbg[2 * ui] = __builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (F8) {
      asm volatile(X2I_GEMM256F8_MAIN
                   : X2I_GEMM256P_OPS_ACC_IO(acc), [la] "+v"(la), [lw] "+v"(lw), [dma] "+s"(dma), [koff] "=&s"(s_koff), [it] "=&s"(s_it)
                   : X2I_GEMM256W_OPS_VOFF(va, vw), X2I_GEMM256P_OPS_NEXT(na, nw), [ra] "s"(a_rsrc), [rw] "s"(w_rsrc), [nra] "s"(na_rsrc),
                     [nrw] "s"(nw_rsrc), [nk] "s"(len), [k0b] "s"(k0b), [nk0b] "s"(nk0b), [zs] "s"(zs)
                   : "memory", "scc", "m0", X2I_GEMM256F8_FRAG_CLOBBERS);
    } else {
      asm volatile(X2I_GEMM256P_MAIN
                   : X2I_GEMM256P_OPS_ACC_IO(acc), X2I_GEMM256P_OPS_FRAG0_IO(fr), X2I_GEMM256P_OPS_FRAG1(fr), [la] "+v"(la), [lw] "+v"(lw),
                     [dma] "+s"(dma), [koff] "=&s"(s_koff), [it] "=&s"(s_it)
                   : X2I_GEMM256W_OPS_VOFF(va, vw), X2I_GEMM256P_OPS_NEXT(na, nw), [ra] "s"(a_rsrc), [rw] "s"(w_rsrc), [nra] "s"(na_rsrc),
                     [nrw] "s"(nw_rsrc), [nk] "s"(len),
                     [k0b] "s"(k0b), [nk0b] "s"(nk0b), [zs] "s"(zs)
                   : "memory", "scc", "m0");
    }
#ifdef X2I_ABLATION
    if (tdbg && ui < 31) tdbg[2 * ui + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    auto spin_until = [&](unsigned* flag, unsigned want) {   // (thread 0) bounded, so that a lost partner leaves a marker instead of a hung GPU
      int spins = 0;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
        __builtin_amdgcn_s_sleep(32);
        if (++spins > (1 << 22)) {
          __hip_atomic_store(p.sk_flags + SK_ERR_SLOT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    };
    if constexpr (FX) {
      if (cur.kind == 3) {
        // every part of the tile parks its range and finishes a slice (epilogue_fx_slice).  The slabs are DOUBLE-BUFFERED by the unit's parity
        // (slab w + 256 b, flags 512 b + slab), so a workgroup parks item u, goes straight into item u + 1's K-loop (already prefetched by
        // the seamless loop) and finishes its slice of item u behind it, when every other part has long been parked: only the last item of
        // a launch pays the park -> wait -> slice tail (round 5: batches of 2 / 4 512^2 samples lost 3-20 % per launch to that tail)
        const int buf = ui & 1;
        if (ui >= 2) fx_reclaim(buf);          // the slab's last use was unit ui - 2: all P parts have read it by now
        __amdgpu_buffer_rsrc_t s_rsrc = slab_rsrc(w + 256 * buf);
        asm volatile(X2I_GEMM256P_STORE_PARTIAL
                     : [so] "=&s"(s_so)
                     : X2I_GEMM256P_OPS_ACC_IN(acc), [vo] "v"(slab_vo), [rs] "s"(s_rsrc)
                     : "memory", "scc");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(fx_flag(buf, w), (unsigned)cur.tag << 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (pend_valid) fx_slice_of(pend_sel, pend_z, pend_m0, pend_n0, pend_slab0, pend_tag, buf ^ 1);
        pend_valid = 1; pend_sel = sel; pend_z = z; pend_m0 = m0; pend_n0 = n0; pend_slab0 = cur.slab; pend_tag = cur.tag;
      } else if (cur.kind == 1) {
        // park this range's sums; the slab is free once the finisher of the previous item's tile has reset its flag
        if (tid == 0) spin_until(p.sk_flags + cur.slab, 0u);
        __syncthreads();
        __amdgpu_buffer_rsrc_t s_rsrc = slab_rsrc(cur.slab);
        asm volatile(X2I_GEMM256P_STORE_PARTIAL
                     : [so] "=&s"(s_so)
                     : X2I_GEMM256P_OPS_ACC_IN(acc), [vo] "v"(slab_vo), [rs] "s"(s_rsrc)
                     : "memory", "scc");
        __syncthreads();
        // (relaxed: the slab went out as write-through stores drained inside the statement; a RELEASE here would write back the XCD's whole L2)
        if (tid == 0) __hip_atomic_store(p.sk_flags + cur.slab, (unsigned)cur.tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        // finish the tile: fx_npre other workgroups have parked (or are about to park) the other K ranges
        const int npre = fx_npre;
        if (npre > 0) {
          if (tid == 0) {
            for (int q = 0; q < npre; ++q) spin_until(p.sk_flags + (cur.slab + 8 * q), (unsigned)cur.tag);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          }
          __syncthreads();
          epilogue_chunked_pipe<ACT, HASC2, RES, F8, true>(prob(pp, sel), acc, z, m0 + wm * 128, n0 + wn * 128, lane, stage, dq, cur.slab, npre, tid);
          __syncthreads();   // every wave has read the slabs: hand them back
          if (tid == 0)
            for (int q = 0; q < npre; ++q) __hip_atomic_store(p.sk_flags + (cur.slab + 8 * q), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          epilogue_chunked_pipe<ACT, HASC2, RES, F8>(prob(pp, sel), acc, z, m0 + wm * 128, n0 + wn * 128, lane, stage, dq);
        }
      }
    } else if (cur.k0 + cur.len < nk) {
      // hand the accumulators to the next segment of this tile: write-through stores, drained inside the statement; the flag
      // (K-tiles accumulated so far) goes out behind a workgroup barrier
      __amdgpu_buffer_rsrc_t s_rsrc = slab_rsrc(cur.slab);
      asm volatile(X2I_GEMM256P_STORE_PARTIAL
                   : [so] "=&s"(s_so)
                   : X2I_GEMM256P_OPS_ACC_IN(acc), [vo] "v"(slab_vo), [rs] "s"(s_rsrc)
                   : "memory", "scc");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(p.sk_flags + cur.slab, (unsigned)(cur.k0 + cur.len), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      // ---- epilogue of (z, m0, n0): per-wave private staging, no workgroup barrier; the next unit's first two K-tiles are in flight
      if (cur.k0 > 0 && tid == 0) __hip_atomic_store(p.sk_flags + cur.slab, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // tile closed: flag back to 0
#ifdef X2I_ABLATION
      if (p.act2 == 77 || p.act2 == 79) {  // measurement only (tools/gemm_epilogue_cost.py): no epilogue at all -- what the K-loops alone take
        asm volatile("" ::X2I_GEMM256P_OPS_ACC_IN(acc));
      } else
#endif
      if constexpr (QKV) epilogue_qkv_chunked<F8>(prob(pp, sel), acc, z, m0 + wm * 128, n0 + wn * 128, lane, stage, dq);
      else if constexpr (OUT8) {
        if (p.f_oinv == 1.f) epilogue_chunked_pipe_e4m3<ACT, true>(prob(pp, sel), acc, z, m0 + wm * 128, n0 + wn * 128, lane, stage, dq);
        else epilogue_chunked_pipe_e4m3<ACT, false>(prob(pp, sel), acc, z, m0 + wm * 128, n0 + wn * 128, lane, stage, dq);
      } else epilogue_chunked_pipe<ACT, HASC2, RES, F8>(prob(pp, sel), acc, z, m0 + wm * 128, n0 + wn * 128, lane, stage, dq);
    }
    if (!has_next) break;
    cur = nxt; sel = nsel; z = nz; m0 = nm0; n0 = nn0; k0b = nk0b;

#pragma unroll
    for (int jj = 0; jj < 8; ++jj) va[jj] = na[jj], vw[jj] = nw[jj];
  }
  if constexpr (FX) {
    if (pend_valid) {   // the last item's slice, then both buffers' slabs back (flags return to zero: the workspace invariant between launches)
      const int last = (n_units - 1) & 1;
      fx_slice_of(pend_sel, pend_z, pend_m0, pend_n0, pend_slab0, pend_tag, last);
      if (n_units >= 2) fx_reclaim(last ^ 1);
      fx_reclaim(last);
    }
  }
  asm volatile(X2I_GEMM256P_DRAIN ::: "memory");   // (the e4m3 loop's drain is the same two instructions)
}

}  // namespace

kern_t pick_gemm256p_qkv() { return gemm256p_kernel<X2I_ACT_NONE, false, false, true>; }

// grouped (two-problem) forms: the layer types of a double-stream block
kern2_t pick_gemm256p_pair(int act, bool res, bool qkv, bool c2) {
  if (qkv) return gemm256p_kernel<X2I_ACT_NONE, false, false, true, true>;
  if (c2) return (act == X2I_ACT_NONE && !res) ? (kern2_t)gemm256p_kernel<X2I_ACT_NONE, false, true, false, true> : nullptr;
  if (res) return act == X2I_ACT_NONE ? (kern2_t)gemm256p_kernel<X2I_ACT_NONE, true, false, false, true> : nullptr;
  if (act == X2I_ACT_GELU_TANH) return gemm256p_kernel<X2I_ACT_GELU_TANH, false, false, false, true>;
  if (act == X2I_ACT_NONE) return gemm256p_kernel<X2I_ACT_NONE, false, false, false, true>;
  return nullptr;
}

// e4m3 operands: the epilogue set of gemm256_fp8.hip
kern_t pick_gemm256p_fp8(int act, bool res, bool out8, bool qkv) {
  if (qkv) return (!res && !out8 && act == X2I_ACT_NONE) ? (kern_t)gemm256p_kernel<X2I_ACT_NONE, false, false, true, false, true> : nullptr;
  if (out8) {
    if (res) return nullptr;
    if (act == X2I_ACT_NONE) return gemm256p_kernel<X2I_ACT_NONE, false, false, false, false, true, true>;
    if (act == X2I_ACT_GELU_TANH) return gemm256p_kernel<X2I_ACT_GELU_TANH, false, false, false, false, true, true>;
    return nullptr;
  }
  if (res) return act == X2I_ACT_NONE ? (kern_t)gemm256p_kernel<X2I_ACT_NONE, true, false, false, false, true> : nullptr;
  if (act == X2I_ACT_NONE) return gemm256p_kernel<X2I_ACT_NONE, false, false, false, false, true>;
  if (act == X2I_ACT_GELU_TANH) return gemm256p_kernel<X2I_ACT_GELU_TANH, false, false, false, false, true>;
  return nullptr;
}

kern_t pick_gemm256p_fx() { return gemm256p_kernel<X2I_ACT_NONE, true, false, false, false, false, false, true>; }
kern2_t pick_gemm256p_pair_fx() { return gemm256p_kernel<X2I_ACT_NONE, true, false, false, true, false, false, true>; }

kern_t pick_gemm256p(int act, bool res, bool f32, bool c2) {
  if (f32) return nullptr;
  if (res) return (act == X2I_ACT_NONE && !c2) ? (kern_t)gemm256p_kernel<X2I_ACT_NONE, true, false> : nullptr;
  if (c2) return act == X2I_ACT_NONE ? (kern_t)gemm256p_kernel<X2I_ACT_NONE, false, true> : nullptr;
  switch (act) {
    case X2I_ACT_NONE: return gemm256p_kernel<X2I_ACT_NONE, false, false>;
    case X2I_ACT_GELU_TANH: return gemm256p_kernel<X2I_ACT_GELU_TANH, false, false>;
    case X2I_ACT_GELU_ERF: return gemm256p_kernel<X2I_ACT_GELU_ERF, false, false>;
    case X2I_ACT_SILU: return gemm256p_kernel<X2I_ACT_SILU, false, false>;
    case X2I_ACT_RELU: return gemm256p_kernel<X2I_ACT_RELU, false, false>;
  }
  return nullptr;
}

}  // namespace x2i_gemm
