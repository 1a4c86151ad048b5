// Implicit-GEMM convolution (NHWC bf16) with <= 128 output channels per tile column on the persistent four-wave core: 512 x 128 x 64 tiles.
// The VAE decoder's last up block (256 -> 128 and 128 -> 128 at the image resolution) and ControlNeXt's 128-wide convolutions
// (reference: infer/inference_qwenvl.py:209-217 -> diffusers AutoencoderKL.decode; lightcontrol/lightcontrol_flux.py:593-668,708-749).
//
// Why this tile: with 128 output channels a 256 x 256 tile is half empty and a 256 x 128 tile halves the wave tile (64 x 128 or 128 x 64: 1.5 x
// the LDS reads per MFMA -- LDS-bound at ~0.9 of the matrix pipe before any other cost).  512 x 128 keeps the product kernel's 128 x 128 WAVE
// tile and therefore its whole hand-scheduled MFMA / fragment stream (gen_gemm256w.py, geometry "512"): the four waves are stacked along M, every
// wave reads all of W; per K-tile 16 A pieces (the gather of gemm256c.hip: scalar tap offset + per-row tap mask) and 4 W pieces per wave.  LDS:
// two A buffers of 64 KiB + two W buffers of 16 KiB = all 160 KiB, so there is no room for an epilogue staging area and the epilogue leaves
// STRAIGHT FROM REGISTERS: a lane's four consecutive columns of two neighbouring 16-column blocks are exchanged between the lanes 16 apart
// (v_permlane16_swap: two instructions per 16 x 32 outputs) so that every lane stores 16 bytes and a pixel's four lanes 64 contiguous bytes,
// while the next tile's first two K-tiles are already landing (the seamless hand-over of the persistent kernels).
// Same MFMA, same k order, same epilogue arithmetic as gemm_bf16_kernel<CONV> (gemm128.hip): bit-identical OUTPUTS; the channel moments are
// summed over 128-row blocks instead of 64-row blocks (same values up to the order of an f32 sum; tests/test_conv_w4_gpu.py).
#include "gemm_device.h"
#include "gemm256p_epi.h"
#include "gemm256w_loop.inc"

namespace x2i_gemm {
namespace {

constexpr int BM5 = 512, BN5 = 128;

// epilogue of one wave (128 rows x 128 columns) straight from the accumulators: acc[h][c][r][j] = rows 32c + 16r + (lane & 15), columns 64h + 16j + 4 (lane >> 4) ..
template <int ACT, bool RES>
__device__ __forceinline__ void epilogue_direct(const GemmP& p, f32x4_t (&acc)[2][4][2][4], int z, int m_wave, int n_wave, int lane) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  const int mlane = lane & 15, ng = lane >> 4;
  const float* b2 = p.bias2 ? p.bias2 + (long long)z * p.bias2_bs : nullptr;
  float bv[8][4];
  static_for<8>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    const int n = n_wave + j * 16 + ng * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[j][r] = 0.f;
    if (n + 3 < p.N) {
      if (p.bias) {
        const uint2 bb = *(const uint2*)(p.bias + (long long)(z / p.wdiv) * p.bias_gs + n);   // (grouped weights: x2i_gemm_args.w_group)
        bv[j][0] = __uint_as_float(bb.x << 16); bv[j][1] = __uint_as_float(bb.x & 0xffff0000u);
        bv[j][2] = __uint_as_float(bb.y << 16); bv[j][3] = __uint_as_float(bb.y & 0xffff0000u);
      }
      if (b2) {
        const f32x4_t t4 = *(const f32x4_t*)(b2 + n);
        bv[j][0] += t4[0]; bv[j][1] += t4[1]; bv[j][2] += t4[2]; bv[j][3] += t4[3];
      }
    }
  });
  const uint32_t c_bytes = (uint32_t)(((long long)(p.M - 1) * p.ldc + p.N) * 2);
  __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((bf16_t*)p.C + (long long)z * p.c_bs), 0, c_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t r_rsrc = c_rsrc;
  if constexpr (RES)
    r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res + (long long)z * p.r_bs), 0, (uint32_t)(((long long)(p.M - 1) * p.ldr + p.N) * 2), 0x00020000);
  // after the exchange lane (row, ng) holds 8 consecutive columns of block 2 jp + (ng & 1): columns 16 (2 jp + (ng & 1)) + 8 (ng >> 1) ..
  const int scol = n_wave + 16 * (ng & 1) + 8 * (ng >> 1);
  float mom_s[8], mom_q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) mom_s[j] = mom_q[j] = 0.f;
  asm volatile("" ::: "memory");
  static_for<8>([&](auto ic) {   // 16-row block i = 2c + r of the wave tile
    constexpr int i = decltype(ic)::value;
    constexpr int c = i >> 1, r = i & 1;
    const int m = m_wave + i * 16 + mlane;
    const bool live = m < p.M;
    u32x2 rres[8];
    if constexpr (RES) {   // the residual in accumulator layout: a lane's four columns = one 8-byte load, a row's lanes share its lines
      const uint32_t ro = live ? (uint32_t)(((long long)m * p.ldr + n_wave + ng * 4) * 2) : 0x80000000u;
#pragma unroll
      for (int j = 0; j < 8; ++j) rres[j] = __builtin_amdgcn_raw_buffer_load_b64(r_rsrc, ro + j * 32, 0, 0);
    }
    uint2 pk[8];
    static_for<8>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      constexpr int h = j >> 2, jj = j & 3;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t;
        const float a = acc[h][c][r][jj][e];
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(a));
        v[e] = apply_act(t + bv[j][e], ACT);
      }
      if constexpr (RES) {   // (fmaf(1, v, r) == v + r: one rounding either way)
        v[0] += __uint_as_float(rres[j][0] << 16); v[1] += __uint_as_float(rres[j][0] & 0xffff0000u);
        v[2] += __uint_as_float(rres[j][1] << 16); v[3] += __uint_as_float(rres[j][1] & 0xffff0000u);
      }
      pk[j] = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
      if (p.cMom) mom_add(mom_s[j], mom_q[j], pk[j].x, pk[j].y, live);
    });
    static_for<4>([&](auto jpc) {
      constexpr int jp = decltype(jpc)::value;
      // X = block 2 jp, Y = block 2 jp + 1; swap the odd 16-lane rows of X with the even rows of Y: lane (row 0) gets X of rows 0 and 1 = columns 0..7 of
      // block 2 jp, (row 1) columns 0..7 of block 2 jp + 1, (row 2) columns 8..15 of block 2 jp, (row 3) columns 8..15 of block 2 jp + 1
      const auto s0 = __builtin_amdgcn_permlane16_swap(pk[2 * jp].x, pk[2 * jp + 1].x, false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(pk[2 * jp].y, pk[2 * jp + 1].y, false, false);
      const u32x4 d = {s0[0], s1[0], s0[1], s1[1]};
      const int n = scol + 32 * jp;
      const uint32_t vo = (live && n + 7 < p.N) ? (uint32_t)(((long long)m * p.ldc + n) * 2) : 0x80000000u;
#if defined(X2I_C512_ABL) && X2I_C512_ABL == 2   // measurement only: no global stores
      asm volatile("" ::"v"(d), "v"(vo));
#else
      __builtin_amdgcn_raw_buffer_store_b128(d, c_rsrc, vo, 0, 0);
#endif
    });
  });
  if (p.cMom) {
    static_for<8>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      mom_flush(p, mom_s[j], mom_q[j], z, m_wave >> 7, n_wave + j * 16 + ng * 4, lane);
    });
  }
}

template <int ACT, bool RES>
__global__ __launch_bounds__(256) void gemm512c_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // A buffers [2][64 KiB] | W buffers [2][16 KiB]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = p.tilesM * p.tilesN;
  const int TT = T * p.nbatch;
  const int G = gridDim.x, w = blockIdx.x;
  const int nk = p.K / BK;
  auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };

  auto tile_of = [&](int vb, int& z, int& m0, int& n0) {  // consecutive tiles (= neighbouring image rows: shared halo lines) go round the XCDs' ranges
    z = vb / T;
    int bid = vb - z * T;
    const int qq = T >> 3, r = T & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
    const int tm = bid / p.tilesN;
    m0 = tm * BM5;
    n0 = (bid - tm * p.tilesN) * BN5;
  };
  const int KW = p.cKW, KH = p.K / (p.cKW * p.cCin);
  // K order of the launch (gen_gemm256w.py, ADV_LINES): two nested counters under the filter rows, with the increments of the image offset (A),
  // the weight-row offset (W) and the tap shift (S = 31 - tap index) per level.  p.cKorder 1 (product): (ky, channel slice, kx) -- the taps of a
  // filter row follow each other, so their shifted re-reads of the same image lines hit L2; 0: (ky, kx, channel slice), the weight layout's own
  // order = the order of the other convolution kernels (bit-identical to them)
  const int ccs = p.cCin / BK;
  const int cin2 = p.cCin * 2;
  // (integer arithmetic on ko = 0 / 1, not selects: a select between two constants is an i1 to the compiler, which it keeps in a VGPR -- no "s" operand)
  const int ko = p.cKorder;   // 0 / 1 (launcher)
  const int n0s = ccs + ko * (KW - ccs), n1s = KW + ko * (ccs - KW);
  const int dA0 = 128 + ko * (cin2 - 128), dS0 = -ko;
  const int dA1 = ko * (128 - KW * cin2), dS1 = ko * (KW + 1) - 1;
  const int dA2 = (p.cW - KW) * cin2 + ko * (KW * cin2 - ccs * 128), dW2 = ko * (KW - 1) * cin2, dS2 = -ko * KW;
  const int bias_b = (p.cPad * p.cW + p.cPadW) * p.cCin * 2;
  const float r_ow = 1.0f / (float)p.cOW;
  const uint32_t fullrow = KW >= 32 ? 0xffffffffu : ((1u << KW) - 1u);
  uint32_t rep = 0;
  for (int ky = 0; ky < KH; ++ky) rep |= 1u << (ky * KW);
  const int khl = lane >> 5, r8 = (lane >> 2) & 7, cphys = lane & 3;
  const int kby = khl * 64 + ((cphys ^ (3 * (wave & 1))) << 4);
  auto offsets = [&](int z, int m0, int n0, uint32_t (&va)[16], uint32_t (&vw)[4], uint32_t (&mk)[16]) {
    const long long zoff = (long long)z * p.a_bs * 2 + bias_b + kby;
    const long long wz = (long long)(z / p.wdiv) * p.w_bs * 2;   // grouped weights (x2i_gemm_args.w_group): this item's W inside the one descriptor over all groups
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int row = (jj * 4 + wave) * 8 + r8;
      vw[jj] = (n0 + row < p.N) ? (uint32_t)(wz + (long long)(n0 + row) * p.ldw * 2 + kby) : 0x80000000u;
    }
    int oy = fast_div(m0 + wave * 8 + r8, p.cOW, r_ow), ox = m0 + wave * 8 + r8 - oy * p.cOW;   // piece 0's pixel; the next pieces are 32 pixels apart
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const int m = m0 + (jj * 4 + wave) * 8 + r8;
      if (jj) {
        ox += 32;
        while (ox >= p.cOW) ox -= p.cOW, ++oy;
      }
      const int iy0 = oy * p.cStride - p.cPad, ix0 = ox * p.cStride - p.cPadW;
      const uint32_t mask = conv_tap_mask(iy0, ix0, p.cH, p.cW, KH, KW, rep, fullrow);
      const bool live = m < p.M;
      va[jj] = live ? (uint32_t)(zoff + ((long long)iy0 * p.cW + ix0) * p.cCin * 2) : 0x80000000u;
      mk[jj] = live ? mask : 0xffffffffu;
    }
  };
  auto mk_rsrc = [&](const void* ptr, long long back, uint32_t bytes) {
    const unsigned long long a = (unsigned long long)(uintptr_t)ptr - (unsigned long long)back;
    const unsigned long long au = ((unsigned long long)(unsigned)uni((int)(a >> 32)) << 32) | (unsigned)uni((int)(a & 0xffffffffu));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)au, 0, (uint32_t)uni((int)bytes), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t a_rsrc = mk_rsrc(p.A, bias_b, (uint32_t)(((long long)(p.nbatch - 1) * p.a_bs + (long long)p.cH * p.cW * p.cCin) * 2 + bias_b));
  const __amdgpu_buffer_rsrc_t w_rsrc = mk_rsrc(p.W, 0, (uint32_t)(((long long)((p.nbatch - 1) / p.wdiv) * p.w_bs + (long long)(p.N - 1) * p.ldw + p.K) * 2));

  const int n_units = (TT - w + G - 1) / G;
  if (n_units <= 0) return;  // (workgroup-uniform)

  const int frow = lane & 15;
  const uint32_t frag = (frow >> 3) * 1024 + (frow & 7) * 64 + (((lane >> 4) ^ (3 * ((frow >> 3) & 1))) << 4);
  uint32_t la = (uint32_t)(uintptr_t)smem + wave * 16384 + frag;   // this wave's 128 rows of the A image
  uint32_t lw = (uint32_t)(uintptr_t)smem + 131072 + frag;         // every wave reads all 128 rows of the W image
  uint32_t dma = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem + wave * 1024);
  uint32_t dmaw = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem + 131072 + wave * 1024);

  int z, m0, n0;
  tile_of(w, z, m0, n0);
  uint32_t va[16], vw[4], mk[16], na[16], nw[4], nmk[16], tv[4];
  offsets(z, m0, n0, va, vw, mk);
  bf16x8_t fr[32];
  uint32_t s_koff, s_it, s_tmp, s_koffa, s_sh, s_c0, s_c1, s_msk;
  const uint32_t c8 = 0x80000000u;
#define X2I_CONV_KORDER_OPS [n0s] "s"(n0s), [n1s] "s"(n1s), [dA0] "s"(dA0), [dS0] "s"(dS0), [dA1] "s"(dA1), \
    [dS1] "s"(dS1), [dA2] "s"(dA2), [dW2] "s"(dW2), [dS2] "s"(dS2)
  asm volatile(X2I_GEMM512C_PRO
               : X2I_GEMM256P_OPS_FRAG0_OUT(fr), X2I_GEMM256C_OPS_TMP(tv), [koff] "=&s"(s_koff), [koffa] "=&s"(s_koffa), [sh] "=&s"(s_sh), [kc0] "=&s"(s_c0),
                 [kc1] "=&s"(s_c1), [tmp] "=&s"(s_tmp), [msk] "=&s"(s_msk)
               : X2I_GEMM512C_OPS_VOFF(va, vw), X2I_GEMM512C_OPS_MASK(mk, mk), [la] "v"(la), [lw] "v"(lw), [dma] "s"(dma), [dmaw] "s"(dmaw), [ra] "s"(a_rsrc),
                 [rw] "s"(w_rsrc), [cmsb] "s"(c8), X2I_CONV_KORDER_OPS
               : "memory", "scc", "m0");
  for (int ui = 0;; ++ui) {
    const int nvb = w + (ui + 1) * G;
    const bool has_next = nvb < TT;
    int nz = 0, nm0 = 0, nn0 = 0;
    if (has_next) {
      tile_of(nvb, nz, nm0, nn0);
#if defined(X2I_C512_ABL) && X2I_C512_ABL == 3   // measurement only: the next unit's offsets are not computed (it reads this unit's pixels again)
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) na[jj] = va[jj], nmk[jj] = mk[jj];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) nw[jj] = vw[jj];
#else
      offsets(nz, nm0, nn0, na, nw, nmk);
#endif
    } else {
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) na[jj] = 0x80000000u, nmk[jj] = 0xffffffffu;  // behind the last unit: every piece out of range
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) nw[jj] = 0x80000000u;
    }
    f32x4_t acc[2][4][2][4];
    const int zs = 1;
    asm volatile(X2I_GEMM512C_MAIN
                 : X2I_GEMM256P_OPS_ACC_OUT(acc), X2I_GEMM256P_OPS_FRAG0_IO(fr), X2I_GEMM256P_OPS_FRAG1(fr), X2I_GEMM256C_OPS_TMP(tv), [la] "+v"(la), [lw] "+v"(lw),
                   [dma] "+s"(dma), [dmaw] "+s"(dmaw), [koff] "+s"(s_koff), [it] "=&s"(s_it), [koffa] "+s"(s_koffa), [sh] "+s"(s_sh), [kc0] "+s"(s_c0),
                   [kc1] "+s"(s_c1), [tmp] "=&s"(s_tmp), [msk] "=&s"(s_msk)
                 : X2I_GEMM512C_OPS_VOFF(va, vw), X2I_GEMM512C_OPS_NEXT(na, nw), X2I_GEMM512C_OPS_MASK(mk, nmk), [ra] "s"(a_rsrc), [rw] "s"(w_rsrc),
                   [nk] "s"(nk), [zs] "s"(zs), [cmsb] "s"(c8), X2I_CONV_KORDER_OPS
                 : "memory", "scc", "m0");
#if defined(X2I_C512_ABL) && X2I_C512_ABL == 1   // measurement only (tools/c512_parts_build.sh): no epilogue at all
    asm volatile("" ::X2I_GEMM256P_OPS_ACC_IN(acc));
#else
    epilogue_direct<ACT, RES>(p, acc, z, m0 + wave * 128, n0, lane);
#endif
    if (!has_next) break;
    z = nz; m0 = nm0; n0 = nn0;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) va[jj] = na[jj], mk[jj] = nmk[jj];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) vw[jj] = nw[jj];
  }
  asm volatile(X2I_GEMM256P_DRAIN ::: "memory");
}

}  // namespace

kern_t pick_gemm512c(int act, bool res) {
  if (res) return act == X2I_ACT_NONE ? (kern_t)gemm512c_kernel<X2I_ACT_NONE, true> : nullptr;
  if (act == X2I_ACT_NONE) return gemm512c_kernel<X2I_ACT_NONE, false>;
  if (act == X2I_ACT_RELU) return gemm512c_kernel<X2I_ACT_RELU, false>;
  return nullptr;
}

}  // namespace x2i_gemm
