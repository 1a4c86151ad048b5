// bf16 MFMA GEMM with fused epilogues for the FLUX DiT / projector linears.
//
//   C[z][m][n] = epi( sum_k A[z][m][k] * W[n][k] )         (nn.Linear layout: W is [N,K], K contiguous)
//   v = acc + bias[n];  v = act(v);  if (res) v = res[z][m][n] + (gate ? gate[z][n] : 1) * v
//
// Replaces every nn.Linear on the hot path (reference: lightcontrol/lightcontrol_flux.py:64,66,256,257,282
// and the diffusers Attention/FeedForward linears built at :69-80,:135-153; utils/proj.py:18-25).
//
// CDNA4 mapping (v1 "step-3" structure of the CDNA guide):
//   * 128x128x64 block tile, 256 threads = 4 waves in 2(M) x 2(N), each wave 64x64 = 4x4 MFMA 16x16x32 tiles
//   * operands staged HBM -> LDS with buffer_load_dwordx4 ... lds (LDS-DMA, no VGPR round trip); the buffer
//     descriptor's num_records gives free zero-fill for ragged M / N edges
//   * LDS image is lane-linear (DMA constraint), so the bank-conflict XOR swizzle is applied on the SOURCE
//     chunk index and again on the ds_read_b128 address (same involution both sides)
//   * double-buffered LDS, one barrier per K-step; next tile's DMA overlaps this tile's MFMAs
//   * operands swapped (D = W_frag x A_frag) so each lane ends up with 4 consecutive n of one row m
//     -> 8-byte bf16x4 stores and contiguous bias / gate / residual reads
//   * grid is XCD-aware: consecutive tiles of a group-of-8 M band land on the same XCD's L2
#include "x2i_common.h"
#include "x2i_kernels.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

struct GemmP {
  const bf16_t* A; long long a_bs; int lda;
  const bf16_t* W; int ldw; long long w_bs;
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
  int cH, cW, cCin, cOW, cKW, cStride, cPad, cUp;  // cUp = 1: nearest-neighbour x2 upsampling fused into the gather
  // fused q/k-norm + RoPE + head split + V transpose epilogue (x2i_gemm_qkv_bf16); q_on = 0: plain epilogue
  int q_on, q_H, q_Spad, q_tok_off, q_rpb, q_row0;
  int gm;
  float q_eps;
  const bf16_t *q_nq, *q_nk;
  const float *q_cos, *q_sin;
  bf16_t *q_Q, *q_K, *q_VT;
};

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
template <int ACT, bool RES, bool OUTF32, bool HASC2, int MT, int NT>
__device__ __forceinline__ void epilogue_store(const GemmP& p, f32x4_t (&acc)[MT][NT], int z, int mrow, int ncol) {
  // ---- epilogue: lane owns m = m_base + i*16 + (lane&15), n = n_base + j*16 + (lane>>4)*4 + 0..3
  const float* gz = (RES && p.gate) ? p.gate + (long long)z * p.gate_bs : nullptr;
  const bf16_t* rz = RES ? p.res + (long long)z * p.r_bs : nullptr;
  const bool vec_ok = ((p.N & 3) == 0) && ((p.ldc & 3) == 0) && (!RES || (p.ldr & 3) == 0);
  static_for<NT>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int n = ncol + j * 16;
    if (n < p.N) {
      const bool full = vec_ok && (n + 3 < p.N);
      float bv[4] = {0.f, 0.f, 0.f, 0.f}, gv[4] = {1.f, 1.f, 1.f, 1.f};
      const float* b2 = p.bias2 ? p.bias2 + (long long)z * p.bias2_bs : nullptr;
      if (full) {
        if (p.bias) {
          const uint2 b2 = *(const uint2*)(p.bias + n);
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
            if (p.bias) bv[r] = bf16_to_f32(p.bias[n + r]);
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
          const long long coff = (long long)z * p.c_bs + (long long)m * p.ldc + n;
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

template <int ACT, bool RES, bool HASC2, int MT>
__device__ __forceinline__ void epilogue_store_lds(const GemmP& p, f32x4_t (&acc)[MT][4], int z, int m_wave, int n_wave, int lane,
                                                   char* wave_lds) {
  const int mlane = lane & 15, ng = lane >> 4;
  const float* gz = (RES && p.gate) ? p.gate + (long long)z * p.gate_bs : nullptr;
  const bf16_t* rz = RES ? p.res + (long long)z * p.r_bs : nullptr;
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
      float bvv[4][4], gvv[4][4];
      static_for<4>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int n = n_wave + j * 16 + ng * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) bvv[j][r] = 0.f, gvv[j][r] = 1.f;
        if (n + 3 < p.N) {
          if (p.bias) {
            const uint2 bb = *(const uint2*)(p.bias + n);
            bvv[j][0] = __uint_as_float(bb.x << 16); bvv[j][1] = __uint_as_float(bb.x & 0xffff0000u);
            bvv[j][2] = __uint_as_float(bb.y << 16); bvv[j][3] = __uint_as_float(bb.y & 0xffff0000u);
          }
          if (gz) {
            const f32x4_t g4 = *(const f32x4_t*)(gz + n);
            gvv[j][0] = g4[0]; gvv[j][1] = g4[1]; gvv[j][2] = g4[2]; gvv[j][3] = g4[3];
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
        static_for<MT>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          const int row = i * 16 + mlane;
          char* slot = wave_lds + row * 128 + ((((j << 1) | (ng >> 1)) ^ ((row >> 1) & 7)) << 4) + ((ng & 1) << 3);
          const uint2 r2 = *(const uint2*)slot;
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = apply_act(acc[i][j][r] + bvv[j][r], ACT);
          v[0] = fmaf(gvv[j][0], v[0], __uint_as_float(r2.x << 16));
          v[1] = fmaf(gvv[j][1], v[1], __uint_as_float(r2.x & 0xffff0000u));
          v[2] = fmaf(gvv[j][2], v[2], __uint_as_float(r2.y << 16));
          v[3] = fmaf(gvv[j][3], v[3], __uint_as_float(r2.y & 0xffff0000u));
          *(uint2*)slot = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        });
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < MT * 2; ++it) {
        const int row = it * 8 + srow;
        const bf16x8_t d = *(const bf16x8_t*)(wave_lds + it * 1024 + lane * 16);
        const int m = m_wave + row, n = n_wave + ((sch ^ ((row >> 1) & 7)) << 3);
        if (m < p.M && n + 7 < p.N) *(bf16x8_t*)(Cz + (long long)m * p.ldc + n) = d;
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
        if (p.bias) {
          const uint2 bb = *(const uint2*)(p.bias + n);
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
        *(uint2*)(wave_lds + mrel * EPI_ROW_BYTES + (j * 16 + ng * 4) * 2) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
      });
    });
    // the region is private to this wave: LDS operations of one wave complete in order, only the data hazard matters
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bf16_t* dst = (pass == 0) ? Cz : C2z;
#pragma unroll
    for (int it = 0; it < MT * 2; ++it) {
      const int row = it * 8 + (lane >> 3), c = lane & 7;
      const bf16x8_t d = *(const bf16x8_t*)(wave_lds + row * EPI_ROW_BYTES + c * 16);
      const int m = m_wave + row, n = n_wave + c * 8;
      if (m < p.M && n + 7 < p.N) *(bf16x8_t*)(dst + (long long)m * p.ldc + n) = d;
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
template <int MT, int WN, int NT>
__device__ __forceinline__ void epilogue_qkv(const GemmP& p, f32x4_t (&acc)[MT][4], int z, int m0, int n0, int wm, int wn, int lane,
                                             int tid, char* smem) {
  constexpr int WR = MT * 16;  // rows per wave
  constexpr int TR = 2 * WR;   // tile rows (two waves along M in both kernels)
  constexpr int TC = WN * 64;  // tile columns
  constexpr int REGION = WR * EPI_ROW_BYTES;
  constexpr int HEADS = TC / 128;
  {
    char* wave_lds = smem + (wm * WN + wn) * REGION;
    const int mlane = lane & 15, ng = lane >> 4;
    static_for<4>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const int n = n0 + wn * 64 + j * 16 + ng * 4;
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
  __syncthreads();
  const int Dm = p.q_H * 128;
  const int sec = n0 / Dm;  // 0 = q, 1 = k, 2 = v (a tile never straddles sections: Dm % TC == 0, checked by the launcher)
  const int head0 = (n0 - sec * Dm) >> 7;
  auto lds_at = [&](int row, int col) -> const char* {  // bf16 element (row, col) of the tile
    return smem + ((row / WR) * WN + (col >> 6)) * REGION + (row % WR) * EPI_ROW_BYTES + (col & 63) * 2;
  };
  if (sec < 2) {
    const int c = tid & 15;  // 8-dim chunk of the head; the same for every iteration (NT % 16 == 0)
    float w[8];
    {
      const bf16x8_t wv = *(const bf16x8_t*)((sec ? p.q_nk : p.q_nq) + c * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = bf16_to_f32((bf16_t)wv[j]);
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
      float ss = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += x[j] * x[j];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);  // the 16 lanes of this (token, head)
      if (m < p.M) {
        const float r = rsqrtf(ss * (1.f / 128.f) + p.q_eps);
        const int mg = p.q_row0 + m;
        const int b = z + mg / p.q_rpb, st = p.q_tok_off + mg % p.q_rpb;
        const float* cp = p.q_cos + (long long)st * 128 + c * 8;
        const float* sp = p.q_sin + (long long)st * 128 + c * 8;
        const f32x4_t c0 = *(const f32x4_t*)cp, c1 = *(const f32x4_t*)(cp + 4);
        const f32x4_t s0 = *(const f32x4_t*)sp, s1 = *(const f32x4_t*)(sp + 4);
        const float cs[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
        const float sn[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const float a = x[j] * r * w[j], bb = x[j + 1] * r * w[j + 1];
          o[j] = a * cs[j] - bb * sn[j];
          o[j + 1] = bb * cs[j + 1] + a * sn[j + 1];
        }
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
      const int mg = p.q_row0 + m;
      const int b = z + mg / p.q_rpb, st = p.q_tok_off + mg % p.q_rpb;
      bf16_t* row0 = p.q_VT + (((long long)b * p.q_H + h) * 128 + d) * p.q_Spad;
      if (aligned) {
        union { bf16x8_t v8; uint32_t uu[4]; } lo, hi;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          lo.uu[k] = (v[2 * k] & 0xffffu) | (v[2 * k + 1] << 16);
          hi.uu[k] = (v[2 * k] >> 16) | (v[2 * k + 1] & 0xffff0000u);
        }
        *(bf16x8_t*)(row0 + st) = lo.v8;
        *(bf16x8_t*)(row0 + p.q_Spad + st) = hi.v8;
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (m + k < p.M) {
            const int mgk = mg + k;
            const int bk = z + mgk / p.q_rpb, sk = p.q_tok_off + mgk % p.q_rpb;
            bf16_t* rk = p.q_VT + (((long long)bk * p.q_H + h) * 128 + d) * p.q_Spad + sk;
            rk[0] = (bf16_t)(v[k] & 0xffffu);
            rk[p.q_Spad] = (bf16_t)(v[k] >> 16);
          }
        }
      }
    }
  }
}

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
  __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (long long)z * p.w_bs), 0, w_bytes, 0x00020000);

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
      c_ox[j] = ox * p.cStride - p.cPad;
      c_cl[j] = clog * 8;
      c_base[j] = ((c_oy[j] * p.cW + c_ox[j]) * p.cCin + c_cl[j]) * 2;
      uint32_t mask = 0;
      if (m < p.M) {
        const int KH = p.K / (p.cKW * p.cCin);
        for (int ky = 0; ky < KH; ++ky)
          for (int kx = 0; kx < p.cKW; ++kx) {
            const int iy = c_oy[j] + ky, ix = c_ox[j] + kx;
            if (iy >= 0 && iy < (p.cH << p.cUp) && ix >= 0 && ix < (p.cW << p.cUp)) mask |= 1u << (ky * p.cKW + kx);
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
        a_voff[j] = ((c_mask[j] >> tap) & 1) ? (uint32_t)((((iy >> 1) * p.cW + (ix >> 1)) * p.cCin + s_c0 + c_cl[j]) * 2) : 0x80000000u;
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

  // all tiles but the last stage their successor unconditionally (one basic block per K-step); the last one only computes
  auto ktile = [&](int kt, auto stage_next) {
    char* cur = smem + (kt & 1) * 2 * TILE_BYTES;
    if constexpr (decltype(stage_next)::value) {
      char* nxt = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
      const uint32_t koff = (uint32_t)(kt + 1) * BK * 2;
      if (CONV) conv_offsets();
      stage_tile(a_rsrc, nxt, a_voff, CONV ? 0u : koff, wave);
      stage_tile(w_rsrc, nxt + TILE_BYTES, w_voff, koff, wave);
    }
    const char* As = cur + a_frag_base;
    const char* Bs = cur + TILE_BYTES + b_frag_base;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8_t af[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = *(const bf16x8_t*)(As + i * 2048 + frag_off[kk]);
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[j] = *(const bf16x8_t*)(Bs + j * 2048 + frag_off[kk]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the next tile has landed
    __syncthreads();                                  // ... everyone's has, and everyone is done reading `cur`
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
      epilogue_store_lds<ACT, RES, HASC2, 4>(p, acc, z, m0 + wm * 64, n0 + wn * 64, lane, smem + wave * (EPI_WAVE_BYTES / 2));
      return;
    }
  }
  epilogue_store<ACT, RES, OUTF32, HASC2, 4, 4>(p, acc, z, m0 + wm * 64 + (lane & 15), n0 + wn * 64 + (lane >> 4) * 4);
}

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
constexpr int BM2 = 256, BN2 = 256;
constexpr int UNIT_BYTES = 256 * 32 * 2;        // 16 KiB
constexpr int TILE2_BYTES = 4 * UNIT_BYTES;     // 64 KiB per K-tile
constexpr int SMEM2_BYTES = 8 * 18432;          // 144 KiB: 128 KiB operand ring, reused as 8 x 18 KiB epilogue staging

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
  __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (long long)z * p.w_bs), 0, w_bytes, 0x00020000);

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
      c_ox[jj] = ox * p.cStride - p.cPad;
      c_base[jj] = ((c_oy[jj] * p.cW + c_ox[jj]) * p.cCin + kel) * 2;
      uint32_t mask = 0;
      if (m < p.M) {
        const int KH = p.K / (p.cKW * p.cCin);
        for (int ky = 0; ky < KH; ++ky)
          for (int kx = 0; kx < p.cKW; ++kx) {
            const int iy = c_oy[jj] + ky, ix = c_ox[jj] + kx;
            if (iy >= 0 && iy < (p.cH << p.cUp) && ix >= 0 && ix < (p.cW << p.cUp)) mask |= 1u << (ky * p.cKW + kx);
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
          a_voff[jj] = ((c_mask[jj] >> tap) & 1) ? (uint32_t)((((iy >> 1) * p.cW + (ix >> 1)) * p.cCin + s_c0) * 2 + c_kel) : 0x80000000u;
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
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[mh * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kh][j], af[mh][i], acc[mh * 4 + i][j], 0, 0, 0);
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
      epilogue_store_lds<ACT, RES, HASC2, 8>(p, acc, z, m0 + wm * 128, n0 + wn * 64, lane, smem + wave * EPI_WAVE_BYTES);
      return;
    }
  }
  epilogue_store<ACT, RES, OUTF32, HASC2, 8, 4>(p, acc, z, m0 + wm * 128 + (lane & 15), n0 + wn * 64 + (lane >> 4) * 4);
}

// Correct-for-any-shape fallback (K not a multiple of 64, unaligned leading dims): one thread per output.
// Only ever used for tiny problems (e.g. the 3-channel first ControlNeXt conv, reduced-width tests).
__global__ void gemm_naive_kernel(GemmP p) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = blockIdx.y;
  const int z = blockIdx.z;
  if (n >= p.N || m >= p.M) return;
  const bf16_t* a = p.A + (long long)z * p.a_bs + (long long)m * p.lda;
  const bf16_t* w = p.W + (long long)z * p.w_bs + (long long)n * p.ldw;
  float acc = 0.f;
  for (int k = 0; k < p.K; ++k) acc = fmaf(bf16_to_f32(a[k]), bf16_to_f32(w[k]), acc);
  float v = acc + (p.bias ? bf16_to_f32(p.bias[n]) : 0.f) + (p.bias2 ? p.bias2[(long long)z * p.bias2_bs + n] : 0.f);
  v = apply_act(v, p.act);
  if (p.res) {
    const float g = p.gate ? p.gate[(long long)z * p.gate_bs + n] : 1.f;
    v = fmaf(g, v, bf16_to_f32(p.res[(long long)z * p.r_bs + (long long)m * p.ldr + n]));  // one rounding, as in the MFMA kernels
  }
  const long long coff = (long long)z * p.c_bs + (long long)m * p.ldc + n;
  if (p.out_f32) ((float*)p.C)[coff] = v;
  else ((bf16_t*)p.C)[coff] = f32_to_bf16(v);
  if (p.C2) p.C2[coff] = f32_to_bf16(apply_act(v, p.act2));
}

}  // namespace

static int launch_gemm_impl(const x2i_gemm_args* a, const x2i_conv_desc* cd, const x2i_qkv_desc* qd, hipStream_t stream);

int x2i_launch_gemm(const x2i_gemm_args* a, hipStream_t stream) { return launch_gemm_impl(a, nullptr, nullptr, stream); }
int x2i_launch_gemm_conv(const x2i_gemm_args* a, const x2i_conv_desc* cd, hipStream_t stream) {
  return launch_gemm_impl(a, cd, nullptr, stream);
}
int x2i_launch_gemm_qkv(const x2i_gemm_args* a, const x2i_qkv_desc* qd, hipStream_t stream) {
  if (!a || !qd) return x2i_set_error(X2I_ERR_ARG, "gemm_qkv: null pointer");
  if (!qd->norm_q || !qd->norm_k || !qd->cos || !qd->sin || !qd->Q || !qd->K || !qd->VT)
    return x2i_set_error(X2I_ERR_ARG, "gemm_qkv: null pointer in descriptor");
  if (qd->H <= 0 || a->N != 3 * qd->H * 128) return x2i_set_error(X2I_ERR_SHAPE, "gemm_qkv: N=%d must be 3*H*128 (H=%d)", a->N, qd->H);
  if (qd->Spad % 128 || qd->rows_per_sample <= 0 || qd->tok_off < 0 || qd->tok_off + qd->rows_per_sample > qd->Spad)
    return x2i_set_error(X2I_ERR_SHAPE, "gemm_qkv: bad token geometry (tok_off=%d rows_per_sample=%d Spad=%d)", qd->tok_off,
                         qd->rows_per_sample, qd->Spad);
  if (a->act || a->res || a->gate || a->C2 || a->out_f32 || a->bias2)
    return x2i_set_error(X2I_ERR_ARG, "gemm_qkv: only the plain bias epilogue can be fused");
  if ((((uintptr_t)qd->Q | (uintptr_t)qd->K | (uintptr_t)qd->VT | (uintptr_t)qd->norm_q | (uintptr_t)qd->norm_k | (uintptr_t)qd->cos |
        (uintptr_t)qd->sin) & 15) != 0)
    return x2i_set_error(X2I_ERR_ALIGN, "gemm_qkv: descriptor pointers must be 16-byte aligned");
  return launch_gemm_impl(a, nullptr, qd, stream);
}

static int launch_gemm_impl(const x2i_gemm_args* a, const x2i_conv_desc* cd, const x2i_qkv_desc* qd, hipStream_t stream) {
  if (!a || !a->A || !a->W || (!a->C && !qd)) return x2i_set_error(X2I_ERR_ARG, "gemm: null pointer");
  const bool conv = cd != nullptr;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0 || a->batch <= 0) return x2i_set_error(X2I_ERR_SHAPE, "gemm: bad shape M=%d N=%d K=%d batch=%d", a->M, a->N, a->K, a->batch);
  if (a->gate && !a->res) return x2i_set_error(X2I_ERR_ARG, "gemm: gate without residual");
  GemmP p;
  p.A = (const bf16_t*)a->A; p.a_bs = a->a_batch_stride; p.lda = a->lda;
  p.W = (const bf16_t*)a->W; p.ldw = a->ldw; p.w_bs = a->w_batch_stride;
  p.bias = (const bf16_t*)a->bias;
  p.C = a->C; p.c_bs = a->c_batch_stride; p.ldc = a->ldc;
  p.C2 = (bf16_t*)a->C2; p.act2 = a->act2;
  p.gate = a->gate; p.gate_bs = a->gate_batch_stride;
  p.res = (const bf16_t*)a->res; p.r_bs = a->res_batch_stride; p.ldr = a->ldr;
  p.bias2 = a->bias2; p.bias2_bs = a->bias2_batch_stride;
  p.M = a->M; p.N = a->N; p.K = a->K; p.act = a->act; p.out_f32 = a->out_f32;
  p.cH = p.cW = p.cCin = p.cOW = p.cKW = p.cStride = p.cPad = p.cUp = 0;
  p.gm = 4;
  p.q_on = 0; p.q_H = p.q_Spad = p.q_tok_off = p.q_rpb = p.q_row0 = 0; p.q_eps = 0.f;
  p.q_nq = p.q_nk = nullptr; p.q_cos = p.q_sin = nullptr; p.q_Q = p.q_K = p.q_VT = nullptr;
  if (qd) {
    p.q_on = 1; p.q_H = qd->H; p.q_Spad = qd->Spad; p.q_tok_off = qd->tok_off; p.q_rpb = qd->rows_per_sample; p.q_eps = qd->eps;
    p.q_nq = (const bf16_t*)qd->norm_q; p.q_nk = (const bf16_t*)qd->norm_k; p.q_cos = qd->cos; p.q_sin = qd->sin;
    p.q_Q = (bf16_t*)qd->Q; p.q_K = (bf16_t*)qd->K; p.q_VT = (bf16_t*)qd->VT;
    p.ldc = a->N; p.c_bs = 0;  // C is never written
  }
  if (conv) {
    if (cd->Cin % 64 || cd->H <= 0 || cd->W <= 0 || cd->KH <= 0 || cd->KW <= 0 || cd->stride <= 0)
      return x2i_set_error(X2I_ERR_SHAPE, "conv: Cin must be a multiple of 64 (Cin=%d)", cd->Cin);
    const int up = cd->up ? 1 : 0;
    const int OH = ((cd->H << up) + 2 * cd->pad - cd->KH) / cd->stride + 1, OW = ((cd->W << up) + 2 * cd->pad - cd->KW) / cd->stride + 1;
    if (a->M != OH * OW || a->K != cd->KH * cd->KW * cd->Cin)
      return x2i_set_error(X2I_ERR_SHAPE, "conv: M=%d K=%d do not match OH*OW=%d, KH*KW*Cin=%d", a->M, a->K, OH * OW, cd->KH * cd->KW * cd->Cin);
    if ((long long)cd->H * cd->W * cd->Cin * 2 >= 0x7f000000LL) return x2i_set_error(X2I_ERR_SHAPE, "conv: image too large");
    p.cH = cd->H; p.cW = cd->W; p.cCin = cd->Cin; p.cOW = OW; p.cKW = cd->KW; p.cStride = cd->stride; p.cPad = cd->pad; p.cUp = up;
  }
  p.tilesM = (a->M + BM - 1) / BM; p.tilesN = (a->N + BN - 1) / BN;
  const bool fast = (a->K % BK == 0) && (conv || a->lda % 8 == 0) && (a->ldw % 8 == 0) && (((uintptr_t)a->A & 15) == 0) &&
                    (((uintptr_t)a->W & 15) == 0) && ((a->a_batch_stride & 7) == 0) &&
                    (conv || (long long)a->M * a->lda * 2 < 0x7f000000LL) && ((long long)a->N * a->ldw * 2 < 0x7f000000LL);
  typedef void (*kern_t)(GemmP);
  kern_t kern = nullptr, kern2 = nullptr, kern2l = nullptr;
  const bool res = p.res != nullptr, c2 = p.C2 != nullptr, f32 = p.out_f32 != 0;
#define X2I_PICK(A_, R_, F_, C_)                                       \
  {                                                                    \
    kern = conv ? gemm_bf16_kernel<A_, R_, F_, C_, true>               \
                : gemm_bf16_kernel<A_, R_, F_, C_, false>;             \
    kern2 = gemm256_bf16_kernel<A_, R_, F_, C_>;                       \
    kern2l = conv ? gemm256l_bf16_kernel<A_, R_, F_, C_, true>         \
                  : gemm256l_bf16_kernel<A_, R_, F_, C_, false>;       \
  }
  if (!res && !f32 && !c2) {
    switch (p.act) {
      case X2I_ACT_NONE: X2I_PICK(X2I_ACT_NONE, false, false, false) break;
      case X2I_ACT_GELU_TANH: X2I_PICK(X2I_ACT_GELU_TANH, false, false, false) break;
      case X2I_ACT_GELU_ERF: X2I_PICK(X2I_ACT_GELU_ERF, false, false, false) break;
      case X2I_ACT_SILU: X2I_PICK(X2I_ACT_SILU, false, false, false) break;
      case X2I_ACT_RELU: X2I_PICK(X2I_ACT_RELU, false, false, false) break;
    }
  } else if (p.act == X2I_ACT_NONE) {
    if (res && !f32 && !c2) X2I_PICK(X2I_ACT_NONE, true, false, false)
    else if (!res && f32 && !c2) X2I_PICK(X2I_ACT_NONE, false, true, false)
    else if (!res && !f32 && c2) X2I_PICK(X2I_ACT_NONE, false, false, true)
  }
#undef X2I_PICK
  // tile choice: the 256^2 pipelined kernel needs enough tiles to fill 256 CUs (1 workgroup per CU)
  if (const char* ab = getenv("X2I_GEMM_ABLATE")) {  // measurement-only kernels (tools/gemm_ablate.py)
    const int abl = atoi(ab);
    if (abl == 1) kern2 = gemm256_bf16_kernel<X2I_ACT_NONE, false, false, false, 1>;
    if (abl == 2) kern2 = gemm256_bf16_kernel<X2I_ACT_NONE, false, false, false, 2>;
    if (abl == 4) kern2 = gemm256_bf16_kernel<X2I_ACT_NONE, false, false, false, 4>;
    if (abl == 3) kern2 = gemm256_bf16_kernel<X2I_ACT_NONE, false, false, false, 3>;
    if (abl == 7) kern2 = gemm256_bf16_kernel<X2I_ACT_NONE, false, false, false, 7>;
    if (abl == 8) kern2 = gemm256_bf16_kernel<X2I_ACT_NONE, false, false, false, 8>;
    if (abl == 16) kern2 = gemm256_bf16_kernel<X2I_ACT_NONE, false, false, false, 16>;
    if (abl == 32) kern2 = gemm256_bf16_kernel<X2I_ACT_NONE, false, false, false, 32>;
    if (abl == 48) kern2 = gemm256_bf16_kernel<X2I_ACT_NONE, false, false, false, 48>;
    if (abl == 128) kern2 = gemm256_bf16_kernel<X2I_ACT_NONE, false, false, false, 128>;
    if (abl == 256) kern2 = gemm256_bf16_kernel<X2I_ACT_NONE, false, false, false, 256>;
  }
  {  // the full-line staging form is the product kernel; X2I_GEMM_LFORM=0 selects the k-half-unit form (A/B, ablations)
    const char* lf = getenv("X2I_GEMM_LFORM");
    if (!(lf && atoi(lf) == 0) && !getenv("X2I_GEMM_ABLATE") && kern2l) kern2 = kern2l;
    else if (conv) kern2 = nullptr;  // only the full-line kernel has the convolution gather
  }
  const char* force_env = getenv("X2I_GEMM_TILE");  // "128" / "256": debugging and A/B benchmarking override
  const int force = force_env ? atoi(force_env) : 0;
  const long long tiles256 = (long long)((a->M + BM2 - 1) / BM2) * ((a->N + BN2 - 1) / BN2) * a->batch;
  // Tile choice (re-measured with the full-line staging kernel, B = 1, 2, 4): the 256^2 kernel wins from about half a round
  // of tiles upwards (1.0-1.38 PF against 0.8-1.0 PF for 128^2 tiles), also when its last round is partly filled; only
  // launches with very few tiles or few rows per batch item (text stream) fill the GPU better with 128^2 tiles.  The
  // threshold can be moved with X2I_GEMM_MIN256 for A/B runs.
  long long min256 = 128;
  if (const char* me = getenv("X2I_GEMM_MIN256")) min256 = atoll(me);
  // batched launches with few rows per item (text stream, 512 rows per sample) keep 128^2 tiles below three full rounds
  bool use256 = !conv && a->N >= 256 && (tiles256 >= 768 ? a->M >= 256 : (tiles256 >= min256 && a->M >= 1024));
  // convolutions with >= 256 output channels: the full-line kernel's implicit-GEMM form (X2I_CONV256=0: 128^2 tiles, A/B)
  bool conv256 = false;
  if (conv && a->N >= 256 && a->N % 8 == 0 && tiles256 >= min256 && a->M >= 1024 && !(getenv("X2I_CONV256") && atoi(getenv("X2I_CONV256")) == 0))
    conv256 = use256 = true;
  if (force == 128) use256 = false;
  if (force == 256 && !conv) use256 = true;
  if (conv && (!conv256 || !kern2)) use256 = false;
  if (qd && (qd->H * 128) % BN2) use256 = false;  // a 256-column tile must not straddle the q / k / v sections
  if (qd && !(fast && kern)) return x2i_set_error(X2I_ERR_SHAPE, "gemm_qkv: K %% 64 == 0 and 16-byte aligned operands required");
  if (fast && kern && use256) {
    hipError_t e = hipFuncSetAttribute((const void*)kern2, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2_BYTES);
    if (e != hipSuccess) return x2i_set_error(X2I_ERR_HIP, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
    // Tile-quantisation fix: with one 256x256 workgroup per CU the launch runs in rounds of 256 tiles; a last round that
    // is less than ~60% full wastes the machine (e.g. M=4x4608, N=3072: 864 tiles = 3.375 rounds).  Peel the trailing
    // rows of every batch item off into a second launch of the 128x128 kernel (2 workgroups per CU, 4x smaller tiles)
    // that fills in behind the last full round.
    const int tm_all = (a->M + BM2 - 1) / BM2, tn = (a->N + BN2 - 1) / BN2;
    const long long per_row = (long long)tn * a->batch;
    const long long full_rounds = tiles256 / 256, rem = tiles256 % 256;
    int tm_main = tm_all;
    if (force == 0 && !conv && full_rounds >= 1 && rem > 0 && rem <= 160 && !getenv("X2I_GEMM_NOSPLIT")) {
      const long long tm_fit = (full_rounds * 256) / per_row;
      if (tm_fit >= 1 && tm_fit < tm_all) tm_main = (int)tm_fit;
    }
    GemmP pm = p;
    {
      // Patch shape per XCD (measured, profiles/r01g_gm_sweep.log): few tile columns and a deep K -> one tile row at a time, so
      // the XCD's concurrent tiles share each A panel and it leaves HBM once; otherwise near-square patches (6 x 5.3) keep
      // the L2 traffic per K-step lowest; very wide N prefers two rows.
      const char* ge = getenv("X2I_GEMM_GM");
      if (ge && atoi(ge) > 0) pm.gm = atoi(ge);
      else if (tn <= 16) pm.gm = a->K >= 8192 ? 1 : 4;
      else if (tn <= 64) pm.gm = 6;
      else pm.gm = 2;
    }
    pm.M = (tm_main < tm_all) ? tm_main * BM2 : a->M;
    pm.tilesM = tm_main; pm.tilesN = tn;
    hipLaunchKernelGGL(kern2, dim3(pm.tilesM * pm.tilesN, a->batch), dim3(512), SMEM2_BYTES, stream, pm);
    if (tm_main < tm_all) {
      e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
      if (e != hipSuccess) return x2i_set_error(X2I_ERR_HIP, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
      GemmP pt = p;
      const long long r0 = (long long)tm_main * BM2;
      pt.A = p.A + r0 * p.lda;
      pt.C = p.out_f32 ? (void*)((float*)p.C + r0 * p.ldc) : (void*)((bf16_t*)p.C + r0 * p.ldc);
      if (p.C2) pt.C2 = p.C2 + r0 * p.ldc;
      if (p.res) pt.res = p.res + r0 * p.ldr;
      pt.M = a->M - (int)r0;
      pt.q_row0 = (int)r0;
      pt.tilesM = (pt.M + BM - 1) / BM; pt.tilesN = (a->N + BN - 1) / BN;
      hipLaunchKernelGGL(kern, dim3(pt.tilesM * pt.tilesN, a->batch), dim3(256), 4 * TILE_BYTES, stream, pt);
    }
  } else if (fast && kern) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
    if (e != hipSuccess) return x2i_set_error(X2I_ERR_HIP, "gemm: hipFuncSetAttribute: %s", hipGetErrorString(e));
    dim3 grid(p.tilesM * p.tilesN, a->batch);
    hipLaunchKernelGGL(kern, grid, dim3(256), 4 * TILE_BYTES, stream, p);
  } else if (conv) {
    return x2i_set_error(X2I_ERR_SHAPE, "conv: unsupported epilogue/alignment combination");
  } else {
    dim3 grid((a->N + 127) / 128, a->M, a->batch);
    hipLaunchKernelGGL(gemm_naive_kernel, grid, dim3(128), 0, stream, p);
  }
  return x2i_check_launch("gemm");
}
