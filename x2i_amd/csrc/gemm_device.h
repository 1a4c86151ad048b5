// Device-side building blocks shared by the bf16 MFMA GEMM kernels (gemm128.hip, gemm256.hip, gemm_ablate.hip): the kernel
// parameter block, LDS-DMA staging helpers and the fused epilogues.  See gemm.hip for the launcher and the operator contract.
#pragma once
#include "x2i_common.h"
#include "x2i_kernels.h"
#include <type_traits>

namespace x2i_gemm {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

struct GemmP {
  const bf16_t* A; long long a_bs; int lda;
  const bf16_t* W; int ldw; long long w_bs;
  int wdiv, bias_gs;                       // grouped weights (x2i_gemm_args.w_group): batch item z reads W + (z / wdiv) * w_bs and bias + (z / wdiv) * bias_gs; off: 1, 0
  const bf16_t* bias;
  void* C; long long c_bs; int ldc;
  bf16_t* C2; int act2;
  const float* gate; long long gate_bs;
  const bf16_t* res; long long r_bs; int ldr;
  const float* bias2; long long bias2_bs;  // optional f32 per-batch additive vector [batch][N]
  int M, N, K;
  int act, out_f32;
  int tilesM, tilesN;
  // implicit-GEMM convolution view of A (NHWC input [batch][cH][cW][cCin], K ordered [ky][kx][ci]); cCin % 64 == 0
  int cH, cW, cCin, cOW, cKW, cStride, cPad, cUp;  // cUp: nearest-neighbour x2 upsampling fused into the gather, bit 0 = along H, bit 1 = along W
  int cPadW;                                       // padding along W (cPad: along H)
  int cRowPitch;                                   // conv: elements between two output ROWS of C (0 = cOW * ldc, dense): x2i_conv_desc.out_row_pitch
  int cKorder;                                     // conv, persistent four-wave kernels: 1 = K order (ky, channel slice, kx), 0 = (ky, kx, channel slice) (option conv_korder)
  float* cMom; int cMomBlocks;                     // conv: per-row-block channel-quad moments of the bf16 outputs, f32 [batch][cMomBlocks][N / 4][2] (nullptr: off)
  // fused q/k-norm + RoPE + head split + V transpose epilogue (x2i_gemm_qkv_bf16); q_on = 0: plain epilogue
  int q_on, q_H, q_Spad, q_tok_off, q_rpb, q_row0, q_vperm;   // q_vperm: V^T span-permuted (x2i_vt_pos)
  int gm;
  float q_eps, q_qs;
  const bf16_t *q_nq, *q_nk;
  const float *q_cos, *q_sin;
  bf16_t *q_Q, *q_K, *q_VT;
  // fp8 (e4m3) operands (x2i_gemm_fp8): acc * f_sa[z][m] * f_sw[n] * f_alpha replaces acc in every epilogue; f_out8 = 1: C is
  // e4m3, value = sat(epi * f_oinv)
  const float* f_sa; long long f_sa_bs;
  const float* f_sw;
  float f_alpha, f_oinv;
  int f_out8;
  // persistent kernel (gemm256p.hip): nbatch = batch items in the tile list; sk_on = 1: the tiles of the last, partly filled round
  // are split along K over the workgroups (chained partial accumulators: sk_slabs = [tile][64][256][16 B] f32, sk_flags[tile] = K-tiles
  // of that tile accumulated so far, sk_flags[SK_ERR_SLOT] = give-up marker)
  int nbatch, sk_on;
  float* sk_slabs;
  unsigned* sk_flags;
  // parallel split with fix-up ("FX", gemm256p.hip): fx_v0 = workgroups per XCD (of G / 8) that share the K-tile space of problem 0's
  // tiles of one batch item on that XCD; the others share problem 1's (grouped launches).  0: off
  int fx_v0;
};
struct GemmP2 { GemmP p[2]; };                  // grouped launch of the persistent kernel (gemm256p.hip)
constexpr int SK_MAX_TILES = 256;               // split tiles per launch (< number of CUs)
constexpr int SK_ERR_SLOT = SK_MAX_TILES;       // sk_flags[SK_ERR_SLOT] != 0: a segment gave up waiting for its predecessor
constexpr long long SK_SLAB_BYTES = 256LL * 256 * 4;
constexpr int SK_SLABS = 2 * SK_MAX_TILES;      // slabs in a workspace: the parallel split (FX) double-buffers its parked partial sums (flags of buffer b at 512 b)

// Row m of a QKV GEMM -> (sample b, position st in the joint sequence) without a division per row: the division happens ONCE for a
// uniform base row; rows behind it wrap by subtraction (at most once when a sample has at least as many rows as the span, 256).  (The per-row `mg / rows_per_sample` of the first form was most of the fused epilogue's instruction count.)
struct TokMap {
  int b0, r0, rpb, tok_off;
};
__device__ __forceinline__ TokMap tok_map(const GemmP& p, int z, int m_base_uniform) {
  const int mg0 = __builtin_amdgcn_readfirstlane(p.q_row0 + m_base_uniform);
  const int q = mg0 / p.q_rpb;
  return TokMap{z + q, mg0 - q * p.q_rpb, p.q_rpb, p.q_tok_off};
}
__device__ __forceinline__ void tok_of(const TokMap& t, int dm, int& b, int& st) {  // dm = m - m_base, 0 <= dm < 256
  int r = t.r0 + dm;
  int q = r >= t.rpb ? 1 : 0;  // one wrap, branch-free: all there is when a sample has >= 256 rows (every shape of the model)
  r -= q ? t.rpb : 0;
  if (__builtin_expect(r >= t.rpb, 0)) {  // shorter samples: keep wrapping
    do {
      r -= t.rpb;
      ++q;
    } while (r >= t.rpb);
  }
  b = t.b0 + q;
  st = t.tok_off + r;
}

__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t rsrc, char* lds_tile, const uint32_t (&voff)[4],
                                           uint32_t koff_bytes, int wave) {
  // 1024 16-byte chunks per tile; instruction j covers chunks [j*256 + wave*64, +64): LDS dest is wave-uniform
  // base + lane*16 (added by hardware)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds_tile + j * 4096 + wave * 1024),
                                             16, voff[j], koff_bytes, 0, 0);
  }
}

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// Shared epilogue: the wave owns MT x NT 16x16 accumulator tiles; lane owns row m = mrow + i*16 and the four
// consecutive columns n = ncol + j*16 + 0..3 of each tile (operands were swapped in the MFMA).
// element offset of output row m inside one batch item of C: m * ldc, or -- convolutions with an output row pitch (x2i_conv_desc.out_row_pitch:
// the phases of an upsampling convolution interleave in rows AND columns) -- (m / OW) * pitch + (m % OW) * ldc
__device__ __forceinline__ long long c_row_off(const GemmP& p, int m) {
  if (p.cRowPitch) {
    const int oy = m / p.cOW;
    return (long long)oy * p.cRowPitch + (long long)(m - oy * p.cOW) * p.ldc;
  }
  return (long long)m * p.ldc;
}

template <int ACT, bool RES, bool OUTF32, bool HASC2, int MT, int NT>
__device__ __forceinline__ void epilogue_store(const GemmP& p, f32x4_t (&acc)[MT][NT], int z, int mrow, int ncol) {
  // ---- epilogue: lane owns m = m_base + i*16 + (lane&15), n = n_base + j*16 + (lane>>4)*4 + 0..3
  const float* gz = (RES && p.gate) ? p.gate + (long long)z * p.gate_bs : nullptr;
  const bf16_t* rz = RES ? p.res + (long long)z * p.r_bs : nullptr;
  const bf16_t* biasz = p.bias ? p.bias + (long long)(z / p.wdiv) * p.bias_gs : nullptr;   // (grouped weights: x2i_gemm_args.w_group)
  const bool vec_ok = ((p.N & 3) == 0) && ((p.ldc & 3) == 0) && (!RES || (p.ldr & 3) == 0);
  static_for<NT>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int n = ncol + j * 16;
    if (n < p.N) {
      const bool full = vec_ok && (n + 3 < p.N);
      float bv[4] = {0.f, 0.f, 0.f, 0.f}, gv[4] = {1.f, 1.f, 1.f, 1.f};
      const float* b2 = p.bias2 ? p.bias2 + (long long)z * p.bias2_bs : nullptr;
      if (full) {
        if (biasz) {
          const uint2 b2 = *(const uint2*)(biasz + n);
          bv[0] = __uint_as_float(b2.x << 16); bv[1] = __uint_as_float(b2.x & 0xffff0000u);
          bv[2] = __uint_as_float(b2.y << 16); bv[3] = __uint_as_float(b2.y & 0xffff0000u);
        }
        if (gz) {
          const f32x4_t g4 = *(const f32x4_t*)(gz + n);
          gv[0] = g4[0]; gv[1] = g4[1]; gv[2] = g4[2]; gv[3] = g4[3];
        }
        if (b2) {
          const f32x4_t t4 = *(const f32x4_t*)(b2 + n);
          bv[0] += t4[0]; bv[1] += t4[1]; bv[2] += t4[2]; bv[3] += t4[3];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r < p.N) {
            if (biasz) bv[r] = bf16_to_f32(biasz[n + r]);
            if (gz) gv[r] = gz[n + r];
            if (b2) bv[r] += b2[n + r];
          }
        }
      }
      static_for<MT>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int m = mrow + i * 16;
        if (m < p.M) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = apply_act(acc[i][j][r] + bv[r], ACT);
          const long long coff = (long long)z * p.c_bs + c_row_off(p, m) + n;
          if (full) {
            if constexpr (RES) {
              const uint2 r2 = *(const uint2*)(rz + (long long)m * p.ldr + n);
              v[0] = fmaf(gv[0], v[0], __uint_as_float(r2.x << 16));
              v[1] = fmaf(gv[1], v[1], __uint_as_float(r2.x & 0xffff0000u));
              v[2] = fmaf(gv[2], v[2], __uint_as_float(r2.y << 16));
              v[3] = fmaf(gv[3], v[3], __uint_as_float(r2.y & 0xffff0000u));
            }
            if constexpr (OUTF32) {
              *(f32x4_t*)((float*)p.C + coff) = (f32x4_t){v[0], v[1], v[2], v[3]};
            } else {
              *(uint2*)((bf16_t*)p.C + coff) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
            if constexpr (HASC2) {
              *(uint2*)(p.C2 + coff) = make_uint2(pack_bf16x2(apply_act(v[0], p.act2), apply_act(v[1], p.act2)),
                                                  pack_bf16x2(apply_act(v[2], p.act2), apply_act(v[3], p.act2)));
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (n + r < p.N) {
                float x = v[r];
                if constexpr (RES) x = fmaf(gv[r], x, bf16_to_f32(rz[(long long)m * p.ldr + n + r]));
                if constexpr (OUTF32) ((float*)p.C)[coff + r] = x;
                else ((bf16_t*)p.C)[coff + r] = f32_to_bf16(x);
                if constexpr (HASC2) p.C2[coff + r] = f32_to_bf16(apply_act(x, p.act2));
              }
            }
          }
        }
      });
    }
  });
}

// LDS-staged epilogue (both tile kernels): every wave parks its finished (MT*16)x64 bf16 sub-tile in a private LDS region
// (row stride 144 B: 16-byte aligned, spreads the 16 rows of a ds_write_b64 over the banks) and writes it out as whole
// 128-byte row segments with 16-byte stores -- a wave store instruction covers 8 full cache lines instead of sixteen
// 32-byte fragments (the direct accumulator layout), which is what the HBM-bound epilogue of the large-N GEMMs needs.
constexpr int EPI_ROW_BYTES = 144;
constexpr int EPI_WAVE_BYTES = 128 * EPI_ROW_BYTES;  // 18 KiB per wave, 144 KiB per workgroup

// Channel moments from the epilogue (x2i_conv_desc.moments): a lane holds rows i*16 + (lane & 15), columns n .. n+3 (one channel QUAD) of column
// block j; `lo`, `hi` are the bf16 pairs it is about to store (the ROUNDED outputs -- what the GroupNorm behind the conv will read).  mom_add
// accumulates the quad's sum and sum of squares over the lane's rows with four v_dot2c_f32_bf16 (pair . ones, pair . pair: exact products, f32
// sums), mom_flush sums the 16 lanes of a DPP row (row_ror 8, 4, 2, 1: fixed order) and lets lane (lane & 15) == 0 write the quad's (sum, sum of squares) of
// this wave tile's row block.
typedef __attribute__((ext_vector_type(2))) __bf16 mom_bf2_t;
__device__ __forceinline__ void mom_add(float& ms, float& mq, uint32_t lo, uint32_t hi, bool valid) {
  if (!valid) return;
  const mom_bf2_t a = __builtin_bit_cast(mom_bf2_t, lo), b = __builtin_bit_cast(mom_bf2_t, hi), one = __builtin_bit_cast(mom_bf2_t, 0x3f803f80u);
  ms = __builtin_amdgcn_fdot2_f32_bf16(a, one, ms, false);
  ms = __builtin_amdgcn_fdot2_f32_bf16(b, one, ms, false);
  mq = __builtin_amdgcn_fdot2_f32_bf16(a, a, mq, false);
  mq = __builtin_amdgcn_fdot2_f32_bf16(b, b, mq, false);
}
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));  // row_ror:8
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));  // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));  // row_ror:2
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));  // row_ror:1
  return v;
}
__device__ __forceinline__ void mom_flush(const GemmP& p, float ms, float mq, int z, int row_block, int n, int lane) {
  const float s = row16_sum(ms), q = row16_sum(mq);
  if ((lane & 15) == 0 && n + 3 < p.N) {   // partial sums: f32 [batch][row blocks][N / 4][2]
    float* dst = p.cMom + (((long long)z * p.cMomBlocks + row_block) * (p.N >> 2) + (n >> 2)) * 2;
    *(float2*)dst = make_float2(s, q);
  }
}

template <int ACT, bool RES, bool HASC2, int MT, bool MOM = false>
__device__ __forceinline__ void epilogue_store_lds(const GemmP& p, f32x4_t (&acc)[MT][4], int z, int m_wave, int n_wave, int lane,
                                                   char* wave_lds) {
  const int mlane = lane & 15, ng = lane >> 4;
  const float* gz = (RES && p.gate) ? p.gate + (long long)z * p.gate_bs : nullptr;
  const bf16_t* rz = RES ? p.res + (long long)z * p.r_bs : nullptr;
  const bf16_t* biasz = p.bias ? p.bias + (long long)(z / p.wdiv) * p.bias_gs : nullptr;   // (grouped weights: x2i_gemm_args.w_group)
  const float* b2 = p.bias2 ? p.bias2 + (long long)z * p.bias2_bs : nullptr;
  bf16_t* Cz = (bf16_t*)p.C + (long long)z * p.c_bs;
  bf16_t* C2z = HASC2 ? p.C2 + (long long)z * p.c_bs : nullptr;
  if constexpr (RES && !HASC2) {
    // Gated-residual form with the residual tile fetched by LDS-DMA as whole 128-byte row segments (the direct form reads
    // it as 32-byte accumulator-layout fragments, which is what makes short-K launches -- the ControlNeXt residual convs --
    // epilogue-bound).  Staging image here: [MT*16 rows][128 B], 16-byte chunk c of row r holds logical chunk
    // c ^ ((r>>1)&7) (the DMA image is lane-linear, so the swizzle is applied on the source address); results overwrite
    // the residual in place and leave with 16-byte stores.
    const long long res_bytes = ((long long)(p.M - 1) * p.ldr + p.N) * 2;
    if ((p.ldr & 7) == 0 && (p.r_bs & 7) == 0 && (((uintptr_t)p.res) & 15) == 0 && res_bytes < 0x7f000000LL) {
      __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)rz, 0, (uint32_t)res_bytes, 0x00020000);
      const int srow = lane >> 3, sch = lane & 7;
#pragma unroll
      for (int it = 0; it < MT * 2; ++it) {
        const int row = it * 8 + srow;
        const int m = m_wave + row, n = n_wave + ((sch ^ ((row >> 1) & 7)) << 3);
        const uint32_t off = (m < p.M && n < p.N) ? (uint32_t)(((long long)m * p.ldr + n) * 2) : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rsrc, (__attribute__((address_space(3))) void*)(wave_lds + it * 1024), 16, off, 0, 0, 0);
      }
      // (MOM = the convolution kernels: no gate there -- the launcher refuses one -- and gate = 1 is exactly `+`: sixteen registers less, which
      // keeps the 128^2 kernel's residual + moments form at two workgroups per CU)
      float bvv[4][4], gvv[MOM ? 1 : 4][4];
      static_for<4>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int n = n_wave + j * 16 + ng * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) bvv[j][r] = 0.f, gvv[MOM ? 0 : j][r] = 1.f;
        if (n + 3 < p.N) {
          if (biasz) {
            const uint2 bb = *(const uint2*)(biasz + n);
            bvv[j][0] = __uint_as_float(bb.x << 16); bvv[j][1] = __uint_as_float(bb.x & 0xffff0000u);
            bvv[j][2] = __uint_as_float(bb.y << 16); bvv[j][3] = __uint_as_float(bb.y & 0xffff0000u);
          }
          if (!MOM && gz) {
            const f32x4_t g4 = *(const f32x4_t*)(gz + n);
            gvv[MOM ? 0 : j][0] = g4[0]; gvv[MOM ? 0 : j][1] = g4[1]; gvv[MOM ? 0 : j][2] = g4[2]; gvv[MOM ? 0 : j][3] = g4[3];
          }
          if (b2) {
            const f32x4_t t4 = *(const f32x4_t*)(b2 + n);
            bvv[j][0] += t4[0]; bvv[j][1] += t4[1]; bvv[j][2] += t4[2]; bvv[j][3] += t4[3];
          }
        }
      });
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the region is private to this wave: no barrier needed
      static_for<4>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        float ms = 0.f, mq = 0.f;
        static_for<MT>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          const int row = i * 16 + mlane;
          char* slot = wave_lds + row * 128 + ((((j << 1) | (ng >> 1)) ^ ((row >> 1) & 7)) << 4) + ((ng & 1) << 3);
          const uint2 r2 = *(const uint2*)slot;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = apply_act(acc[i][j][r] + bvv[j][r], ACT);
          if constexpr (MOM) {   // (fmaf(1, v, r) == v + r: one rounding either way)
            v[0] += __uint_as_float(r2.x << 16); v[1] += __uint_as_float(r2.x & 0xffff0000u);
            v[2] += __uint_as_float(r2.y << 16); v[3] += __uint_as_float(r2.y & 0xffff0000u);
          } else {
            v[0] = fmaf(gvv[j][0], v[0], __uint_as_float(r2.x << 16));
            v[1] = fmaf(gvv[j][1], v[1], __uint_as_float(r2.x & 0xffff0000u));
            v[2] = fmaf(gvv[j][2], v[2], __uint_as_float(r2.y << 16));
            v[3] = fmaf(gvv[j][3], v[3], __uint_as_float(r2.y & 0xffff0000u));
          }
          const uint2 pk = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
          *(uint2*)slot = pk;
          if constexpr (MOM) {
            if (p.cMom) mom_add(ms, mq, pk.x, pk.y, m_wave + row < p.M);
          }
        });
        if constexpr (MOM) {
          if (p.cMom) mom_flush(p, ms, mq, z, m_wave / (MT * 16), n_wave + j * 16 + ng * 4, lane);
        }
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < MT * 2; ++it) {
        const int row = it * 8 + srow;
        const bf16x8_t d = *(const bf16x8_t*)(wave_lds + it * 1024 + lane * 16);
        const int m = m_wave + row, n = n_wave + ((sch ^ ((row >> 1) & 7)) << 3);
        if (m < p.M && n + 7 < p.N) *(bf16x8_t*)(Cz + c_row_off(p, m) + n) = d;
      }
      return;
    }
  }
  constexpr int NPASS = HASC2 ? 2 : 1;
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    static_for<4>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const int n = n_wave + j * 16 + ng * 4;
      float bv[4] = {0.f, 0.f, 0.f, 0.f}, gv[4] = {1.f, 1.f, 1.f, 1.f};
      if (n + 3 < p.N) {
        if (biasz) {
          const uint2 bb = *(const uint2*)(biasz + n);
          bv[0] = __uint_as_float(bb.x << 16); bv[1] = __uint_as_float(bb.x & 0xffff0000u);
          bv[2] = __uint_as_float(bb.y << 16); bv[3] = __uint_as_float(bb.y & 0xffff0000u);
        }
        if (gz) {
          const f32x4_t g4 = *(const f32x4_t*)(gz + n);
          gv[0] = g4[0]; gv[1] = g4[1]; gv[2] = g4[2]; gv[3] = g4[3];
        }
        if (b2) {
          const f32x4_t t4 = *(const f32x4_t*)(b2 + n);
          bv[0] += t4[0]; bv[1] += t4[1]; bv[2] += t4[2]; bv[3] += t4[3];
        }
      }
      float ms = 0.f, mq = 0.f;
      static_for<MT>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int mrel = i * 16 + mlane;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act(acc[i][j][r] + bv[r], ACT);
        if constexpr (RES) {
          const int m = m_wave + mrel;
          if (m < p.M && n + 3 < p.N) {
            const uint2 r2 = *(const uint2*)(rz + (long long)m * p.ldr + n);
            v[0] = fmaf(gv[0], v[0], __uint_as_float(r2.x << 16));
            v[1] = fmaf(gv[1], v[1], __uint_as_float(r2.x & 0xffff0000u));
            v[2] = fmaf(gv[2], v[2], __uint_as_float(r2.y << 16));
            v[3] = fmaf(gv[3], v[3], __uint_as_float(r2.y & 0xffff0000u));
          }
        }
        if (pass == 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.act2);
        }
        const uint2 pk = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        *(uint2*)(wave_lds + mrel * EPI_ROW_BYTES + (j * 16 + ng * 4) * 2) = pk;
        if constexpr (MOM) {
          if (p.cMom && pass == 0) mom_add(ms, mq, pk.x, pk.y, m_wave + mrel < p.M);
        }
      });
      if constexpr (MOM) {
        if (p.cMom && pass == 0) mom_flush(p, ms, mq, z, m_wave / (MT * 16), n, lane);
      }
    });
    // the region is private to this wave: LDS operations of one wave complete in order, only the data hazard matters
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bf16_t* dst = (pass == 0) ? Cz : C2z;
#pragma unroll
    for (int it = 0; it < MT * 2; ++it) {
      const int row = it * 8 + (lane >> 3), c = lane & 7;
      const bf16x8_t d = *(const bf16x8_t*)(wave_lds + row * EPI_ROW_BYTES + c * 16);
      const int m = m_wave + row, n = n_wave + c * 8;
      if (m < p.M && n + 7 < p.N) *(bf16x8_t*)(dst + c_row_off(p, m) + n) = d;
    }
    if (NPASS == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// ------------------------------------------------------------------------------------------------------------
// Fused QKV epilogue (x2i_gemm_qkv_bf16).  The workgroup's finished tile -- TR tokens x TC columns, i.e. TC/128 whole
// heads of the q, k or v section -- is parked in LDS as bf16(acc + bias) (per-wave regions of the staged epilogue, row
// stride 144 B) and leaves in attention layout:
//   q / k tile: 16 lanes x 8 dims per (token, head): RMSNorm over the 128 dims (fp32), * norm weight, RoPE on adjacent
//               pairs with the fp32 cos/sin row of the token's joint position, 16-byte stores into Q/K [B,H,Spad,128]
//   v tile:     transposed through LDS: a lane gathers two adjacent dims of 8 consecutive tokens (8 ds_read_b32) and
//               writes two 16-byte token runs of VT [B,H,128,Spad]; 8 lanes cover a 128-byte line
// Same arithmetic as qk_norm_rope_kernel / v_transpose_kernel (elementwise.hip), which remain the unfused form.
// ------------------------------------------------------------------------------------------------------------
// step 1: a wave parks one MT*16 x 64 bf16(acc + bias) sub-tile (tile columns n_col0 .. n_col0 + 63) in its LDS region
template <int MT>
__device__ __forceinline__ void qkv_park(const GemmP& p, f32x4_t (&acc)[MT][4], int n_col0, int lane, char* wave_lds) {
  const int mlane = lane & 15, ng = lane >> 4;
  static_for<4>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int n = n_col0 + j * 16 + ng * 4;
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && n + 3 < p.N) {
      const uint2 bb = *(const uint2*)(p.bias + n);
      bv[0] = __uint_as_float(bb.x << 16); bv[1] = __uint_as_float(bb.x & 0xffff0000u);
      bv[2] = __uint_as_float(bb.y << 16); bv[3] = __uint_as_float(bb.y & 0xffff0000u);
    }
    static_for<MT>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      *(uint2*)(wave_lds + (i * 16 + mlane) * EPI_ROW_BYTES + (j * 16 + ng * 4) * 2) =
          make_uint2(pack_bf16x2(acc[i][j][0] + bv[0], acc[i][j][1] + bv[1]), pack_bf16x2(acc[i][j][2] + bv[2], acc[i][j][3] + bv[3]));
    });
  });
}

// step 2 (after a workgroup barrier): the parked tile -- regions indexed [(row / WR) * WN + (col >> 6)], WR rows x 64 columns each --
// leaves in attention layout, NT threads
template <int WR, int WN, int NT>
__device__ __forceinline__ void qkv_finish(const GemmP& p, int z, int m0, int n0, int lane, int tid, char* smem);

template <int MT, int WN, int NT>
__device__ __forceinline__ void epilogue_qkv(const GemmP& p, f32x4_t (&acc)[MT][4], int z, int m0, int n0, int wm, int wn, int lane,
                                             int tid, char* smem) {
  qkv_park<MT>(p, acc, n0 + wn * 64, lane, smem + (wm * WN + wn) * (MT * 16 * EPI_ROW_BYTES));
  __syncthreads();
  qkv_finish<MT * 16, WN, NT>(p, z, m0, n0, lane, tid, smem);
}

// 4-wave 128 x 128 wave tiles (gemm256w.hip): a wave parks its two 64-column halves as the regions two 64-column waves would use
__device__ __forceinline__ void epilogue_qkv_w4(const GemmP& p, f32x4_t (&acc)[2][8][4], int z, int m0, int n0, int wm, int wn, int lane,
                                                int tid, char* smem) {
  qkv_park<8>(p, acc[0], n0 + wn * 128, lane, smem + (wm * 4 + wn * 2) * (128 * EPI_ROW_BYTES));
  qkv_park<8>(p, acc[1], n0 + wn * 128 + 64, lane, smem + (wm * 4 + wn * 2 + 1) * (128 * EPI_ROW_BYTES));
  __syncthreads();
  qkv_finish<128, 4, 256>(p, z, m0, n0, lane, tid, smem);
}

template <int WR, int WN, int NT>
__device__ __forceinline__ void qkv_finish(const GemmP& p, int z, int m0, int n0, int lane, int tid, char* smem) {
  constexpr int TR = 2 * WR;   // tile rows (two waves / wave rows along M in every kernel)
  constexpr int TC = WN * 64;  // tile columns
  constexpr int REGION = WR * EPI_ROW_BYTES;
  constexpr int HEADS = TC / 128;
  const int Dm = p.q_H * 128;
  const int sec = n0 / Dm;  // 0 = q, 1 = k, 2 = v (a tile never straddles sections: Dm % TC == 0, checked by the launcher)
  const int head0 = (n0 - sec * Dm) >> 7;
  const TokMap tmap = tok_map(p, z, m0);  // (tile rows m0 .. m0 + TR - 1, TR <= 256)
  auto lds_at = [&](int row, int col) -> const char* {  // bf16 element (row, col) of the tile
    return smem + ((row / WR) * WN + (col >> 6)) * REGION + (row % WR) * EPI_ROW_BYTES + (col & 63) * 2;
  };
  if (sec < 2) {
    const int c = tid & 15;  // 8-dim chunk of the head; the same for every iteration (NT % 16 == 0)
    float w[8];
    {
      const bf16x8_t wv = *(const bf16x8_t*)((sec ? p.q_nk : p.q_nq) + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = bf16_to_f32((bf16_t)wv[j]) * (sec ? 1.f : p.q_qs);
    }
    bf16_t* dstbase = sec ? p.q_K : p.q_Q;
#pragma unroll 2
    for (int u = tid >> 4; u < TR * HEADS; u += NT / 16) {
      const int hh = u % HEADS, row = u / HEADS;
      const int m = m0 + row;
      const bf16x8_t xv = *(const bf16x8_t*)lds_at(row, hh * 128 + c * 8);
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = bf16_to_f32((bf16_t)xv[j]);
      float ss = sumsq8(x);
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);  // the 16 lanes of this (token, head)
      if (m < p.M) {
        const float r = rms_rsqrt128(ss, p.q_eps);
        int b, st;
        tok_of(tmap, row, b, st);
        float cs[8], sn[8];
        if (p.q_sin) {
          const float* cp = p.q_cos + (long long)st * 128 + c * 8;
          const float* sp = p.q_sin + (long long)st * 128 + c * 8;
          const f32x4_t c0 = *(const f32x4_t*)cp, c1 = *(const f32x4_t*)(cp + 4);
          const f32x4_t s0 = *(const f32x4_t*)sp, s1 = *(const f32x4_t*)(sp + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) cs[j] = c0[j], cs[4 + j] = c1[j], sn[j] = s0[j], sn[4 + j] = s1[j];
        } else {   // pair-form table (x2i_qkv_desc: sin == NULL): f32 [S][64][2] = (cos, sin) of dim pair k -- the same values, half the bytes
          const float* pp = p.q_cos + (long long)st * 128 + c * 8;
          const f32x4_t p0 = *(const f32x4_t*)pp, p1 = *(const f32x4_t*)(pp + 4);
          cs[0] = cs[1] = p0[0]; sn[0] = sn[1] = p0[1]; cs[2] = cs[3] = p0[2]; sn[2] = sn[3] = p0[3];
          cs[4] = cs[5] = p1[0]; sn[4] = sn[5] = p1[1]; cs[6] = cs[7] = p1[2]; sn[6] = sn[7] = p1[3];
        }
        float o[8];
        norm_rope8(x, r, w, cs, sn, o);
        union { bf16x8_t v8; uint32_t uu[4]; } pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) pk.uu[j] = pack_bf16x2(o[2 * j], o[2 * j + 1]);
        *(bf16x8_t*)(dstbase + (((long long)b * p.q_H + head0 + hh) * p.q_Spad + st) * 128 + c * 8) = pk.v8;
      }
    }
  } else {
    const int wave = tid >> 6;
    const int ch_lo = lane & 7, dp_lo = lane >> 3;
    constexpr int CG = TR / 64;                  // groups of 8 token-chunks (64 tokens)
    constexpr int WITS = CG * (TC / 16);         // wave-iterations: x groups of 8 dim-pairs (16 dims)
    // 8-token runs are whole and 16-byte aligned in VT when every row offset is a multiple of 8
    const bool aligned = ((p.q_tok_off | p.q_rpb | p.q_row0 | p.M | p.q_Spad) & 7) == 0;
    for (int wi = wave; wi < WITS; wi += NT / 64) {
      const int ch = (wi % CG) * 8 + ch_lo, dp = (wi / CG) * 8 + dp_lo;
      const int d0 = dp * 2;  // tile column of the first of the two dims
      uint32_t v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = *(const uint32_t*)lds_at(ch * 8 + k, d0);
      const int m = m0 + ch * 8;
      if (m >= p.M) continue;
      const int h = head0 + (d0 >> 7), d = d0 & 127;
      int b, st;
      tok_of(tmap, m - m0, b, st);
      bf16_t* row0 = p.q_VT + (((long long)b * p.q_H + h) * 128 + d) * p.q_Spad;
      if (aligned) {
        union { bf16x8_t v8; uint32_t uu[4]; uint2 h2[2]; } lo, hi;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          lo.uu[k] = (v[2 * k] & 0xffffu) | (v[2 * k + 1] << 16);
          hi.uu[k] = (v[2 * k] >> 16) | (v[2 * k + 1] & 0xffff0000u);
        }
        if (p.q_vperm) {   // span-permuted V^T: the run of eight tokens is two runs of four, eight positions apart
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
            tok_of(tmap, m + k - m0, bk, sk);
            bf16_t* rk = p.q_VT + (((long long)bk * p.q_H + h) * 128 + d) * p.q_Spad + x2i_vt_pos(sk, p.q_vperm);
            rk[0] = (bf16_t)(v[k] & 0xffffu);
            rk[p.q_Spad] = (bf16_t)(v[k] >> 16);
          }
        }
      }
    }
  }
}


constexpr int BM2 = 256, BN2 = 256;
constexpr int UNIT_BYTES = 256 * 32 * 2;        // 16 KiB
constexpr int TILE2_BYTES = 4 * UNIT_BYTES;     // 64 KiB per K-tile
constexpr int SMEM2_BYTES = 8 * 18432;          // 144 KiB: 128 KiB operand ring, reused as 8 x 18 KiB epilogue staging

typedef void (*kern_t)(GemmP);
typedef void (*kern2_t)(GemmP2);
// kernel pickers (one per translation unit so the kernel families compile in parallel); nullptr = no MFMA instantiation
// for this epilogue combination
kern_t pick_gemm128(int act, bool res, bool f32, bool c2, bool conv);
kern_t pick_gemm256l(int act, bool res, bool f32, bool c2, bool conv);
kern_t pick_gemm256w(int act, bool res, bool f32, bool c2);  // 4 waves, hand-scheduled K-loop (gemm256w.hip)
kern_t pick_gemm256p(int act, bool res, bool f32, bool c2);  // the same K-loop, persistent over output tiles (gemm256p.hip)
kern_t pick_gemm256c(int act, bool res);                     // implicit-GEMM convolutions on the same core (gemm256c.hip)
kern_t pick_gemm512c(int act, bool res);                     // ... with <= 128 output channels: 512 x 128 tiles (gemm512c.hip)
constexpr int SMEM5C_BYTES = 2 * 65536 + 2 * 16384;          // A buffers + W buffers = all 160 KiB
kern_t pick_gemm256p_qkv();                                 // ... with the fused QKV epilogue (x2i_gemm_qkv_bf16)
kern2_t pick_gemm256p_pair(int act, bool res, bool qkv, bool c2 = false);
kern_t pick_gemm256p_fx();        // gated-residual epilogue, parallel split with fix-up (launches that cannot fill the chip with whole tiles)
kern2_t pick_gemm256p_pair_fx();  // ... grouped form     // ... over the tiles of two problems (x2i_gemm_pair_bf16 / x2i_gemm_qkv_pair_bf16)
constexpr int SMEM2P_BYTES = 2 * TILE2_BYTES + 4 * 8192;     // 128 KiB operand ring + 4 x 8 KiB staging = all 160 KiB
kern_t pick_gemm_r2(int act, bool res, bool f32, bool c2, int var);  // (measurement library only)  // 256 x 128 tiles, two resident workgroups per CU (gemm_r2.hip); var: measurement builds
constexpr int SMEM_R2_BYTES = 4 * 16384;                     // 4-stage W ring (reused as 4 x 9 KiB epilogue staging)
kern_t pick_gemm256_fp8(int act, bool res, bool out8);
kern_t pick_gemm256p_fp8(int act, bool res, bool out8, bool qkv);  // ... in the persistent four-wave form (gemm256p.hip, gen_gemm256f8.py)  // e4m3 operands, MX-scaled K = 128 MFMA (gemm256_fp8.hip)
#ifdef X2I_ABLATION
kern_t pick_gemm256w_var(int var);  // A/B schedules of the 4-wave K-loop (option gemm_w4 = 1 + var), plain epilogue only
kern_t pick_gemm256u(int act, bool res, bool f32, bool c2, int abl);  // k-half-unit form + measurement-only variants
#endif

// One table of the epilogue combinations that have MFMA instantiations (anything else takes the generic kernel).
#define X2I_GEMM_PICK_TABLE(PICK)                                                   \
  if (!res && !f32 && !c2) {                                                        \
    switch (act) {                                                                  \
      case X2I_ACT_NONE: PICK(X2I_ACT_NONE, false, false, false) break;             \
      case X2I_ACT_GELU_TANH: PICK(X2I_ACT_GELU_TANH, false, false, false) break;   \
      case X2I_ACT_GELU_ERF: PICK(X2I_ACT_GELU_ERF, false, false, false) break;     \
      case X2I_ACT_SILU: PICK(X2I_ACT_SILU, false, false, false) break;             \
      case X2I_ACT_RELU: PICK(X2I_ACT_RELU, false, false, false) break;             \
    }                                                                               \
  } else if (act == X2I_ACT_NONE) {                                                 \
    if (res && !f32 && !c2) PICK(X2I_ACT_NONE, true, false, false)                  \
    else if (!res && f32 && !c2) PICK(X2I_ACT_NONE, false, true, false)             \
    else if (!res && !f32 && c2) PICK(X2I_ACT_NONE, false, false, true)             \
  }

}  // namespace x2i_gemm
