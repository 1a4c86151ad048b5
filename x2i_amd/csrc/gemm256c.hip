// Implicit-GEMM convolution (NHWC bf16) on the persistent four-wave 256 x 256 x 64 core (gemm256p.hip's organisation, gen_gemm256w.py's
// hand-scheduled K-loop in its convolution form): ControlNeXt's and the VAE decoder's convolutions with >= 256 output channels
// (reference: lightcontrol/lightcontrol_flux.py:593-668,708-749; infer/inference_qwenvl.py:209-217 -> diffusers AutoencoderKL.decode).
//
// A K-tile is ONE filter tap x 64 input channels = exactly one 128-byte line per output pixel and piece row.  What distinguishes the gather
// from a plain GEMM operand is affine in the tap: the line of pixel (oy, ox) for tap (ky, kx), channel slice c sits at
//     base(pixel) + ((ky * W + kx) * Cin + 64 c) * 2,        base(pixel) = ((oy * stride - pad) * W + ox * stride - pad_w) * Cin * 2
// so the K-loop carries the tap part as a SCALAR byte offset that advances by 128 per K-tile (+ a row jump at the end of a filter row) and
// the only per-lane work is the zero padding: a 32-bit mask per piece row (bit t = tap t outside the image) turns the offset out of range
// (two VALU instructions per piece, in the gaps of the MFMA stream).  No physically padded copy of the input is needed.  Everything else --
// LDS images, MFMA / fragment schedule, seamless hand-over to the workgroup's next tile, chunked whole-line epilogue -- is the linear
// kernel's.  Same MFMA, same k order, same epilogue arithmetic as gemm256l_bf16_kernel<CONV> / gemm_bf16_kernel<CONV>: BIT-IDENTICAL outputs and
// channel moments (tests/test_conv_w4_gpu.py).  Launcher: gemm.hip (x2i_conv2d_nhwc_bf16; option "conv_w4").
#include "gemm_device.h"
#include "gemm256p_epi.h"
#include "gemm256w_loop.inc"

namespace x2i_gemm {
namespace {

template <int ACT, bool RES>
__global__ __launch_bounds__(256) void gemm256c_kernel(GemmP p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 tiles][A image 32 KiB | W image 32 KiB][4 waves x 8 KiB staging]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int T = p.tilesM * p.tilesN;
  const int TT = T * p.nbatch;
  const int G = gridDim.x, w = blockIdx.x;
  const int nk = p.K / BK;
  auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };

  auto tile_of = [&](int vb, int& z, int& m0, int& n0) {  // the XCD-aware patch order of the linear kernels
    z = vb / T;
    int bid = vb - z * T;
    const int qq = T >> 3, r = T & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
    const int GM = p.gm;
    const int per_group = GM * p.tilesN;
    const int group = bid / per_group;
    const int first_m = group * GM;
    const int gsize = min(p.tilesM - first_m, GM);
    m0 = (first_m + (bid % per_group) % gsize) * BM2;
    n0 = ((bid % per_group) / gsize) * BN2;
  };
  // ---- the gather's launch constants
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
  const int bias_b = (p.cPad * p.cW + p.cPadW) * p.cCin * 2;   // pixel bases are made non-negative by this many bytes; the descriptor starts as far in front of A
  const float r_ow = 1.0f / (float)p.cOW;
  const uint32_t fullrow = KW >= 32 ? 0xffffffffu : ((1u << KW) - 1u);
  uint32_t rep = 0;
  for (int ky = 0; ky < KH; ++ky) rep |= 1u << (ky * KW);
  const int khl = lane >> 5, r8 = (lane >> 2) & 7, cphys = lane & 3;
  const int kby = khl * 64 + ((cphys ^ (3 * (wave & 1))) << 4);  // byte within the K-tile's 128-byte line (group parity = wave parity)
  auto offsets = [&](int z, int m0, int n0, uint32_t (&va)[8], uint32_t (&vw)[8], uint32_t (&mk)[8]) {
    const long long zoff = (long long)z * p.a_bs * 2 + bias_b + kby;
    const long long wz = (long long)(z / p.wdiv) * p.w_bs * 2;   // grouped weights (x2i_gemm_args.w_group): this item's W inside the one descriptor over all groups
    int oy = fast_div(m0 + wave * 8 + r8, p.cOW, r_ow), ox = m0 + wave * 8 + r8 - oy * p.cOW;   // piece 0's pixel; the next pieces are 32 pixels apart
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      const int row = (jj * 4 + wave) * 8 + r8;
      vw[jj] = (n0 + row < p.N) ? (uint32_t)(wz + (long long)(n0 + row) * p.ldw * 2 + kby) : 0x80000000u;
      const int m = m0 + row;
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
  // one descriptor over all batch items (< 2 GB: launcher), starting bias_b bytes in front of A: every in-image tap of every pixel lies inside it
  const __amdgpu_buffer_rsrc_t a_rsrc = mk_rsrc(p.A, bias_b, (uint32_t)(((long long)(p.nbatch - 1) * p.a_bs + (long long)p.cH * p.cW * p.cCin) * 2 + bias_b));
  const __amdgpu_buffer_rsrc_t w_rsrc = mk_rsrc(p.W, 0, (uint32_t)(((long long)((p.nbatch - 1) / p.wdiv) * p.w_bs + (long long)(p.N - 1) * p.ldw + p.K) * 2));

  // ---- this workgroup's units: whole tiles vb = w, w + G, ...
  const int n_units = (TT - w + G - 1) / G;
  if (n_units <= 0) return;  // (workgroup-uniform)

  const int frow = lane & 15;
  const uint32_t frag = (frow >> 3) * 1024 + (frow & 7) * 64 + (((lane >> 4) ^ (3 * ((frow >> 3) & 1))) << 4);
  uint32_t la = (uint32_t)(uintptr_t)smem + wm * 8 * 2048 + frag;
  uint32_t lw = (uint32_t)(uintptr_t)smem + 32768 + wn * 8 * 2048 + frag;
  uint32_t dma = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)smem + wave * 1024);
  char* stage = smem + P_STAGE_OFF + wave * P_STAGE_WAVE;

  int z, m0, n0;
  tile_of(w, z, m0, n0);
  uint32_t va[8], vw[8], mk[8], na[8], nw[8], nmk[8], tv[4];
  offsets(z, m0, n0, va, vw, mk);
  bf16x8_t fr[32];  // wa 0..7 | wb 8..15 | aa 16..23 | ab 24..31
  uint32_t s_koff, s_it, s_tmp;
  uint32_t s_koffa, s_sh, s_c0, s_c1, s_msk;   // gather state of the K-tile to fetch next (with s_koff); carried from statement to statement
  const uint32_t c8 = 0x80000000u;
#define X2I_CONV_KORDER_OPS [n0s] "s"(n0s), [n1s] "s"(n1s), [dA0] "s"(dA0), [dS0] "s"(dS0), [dA1] "s"(dA1), \
    [dS1] "s"(dS1), [dA2] "s"(dA2), [dW2] "s"(dW2), [dS2] "s"(dS2)
  asm volatile(X2I_GEMM256C_PRO
               : X2I_GEMM256P_OPS_FRAG0_OUT(fr), X2I_GEMM256C_OPS_TMP(tv), [koff] "=&s"(s_koff), [koffa] "=&s"(s_koffa), [sh] "=&s"(s_sh), [kc0] "=&s"(s_c0),
                 [kc1] "=&s"(s_c1), [tmp] "=&s"(s_tmp), [msk] "=&s"(s_msk)
               : X2I_GEMM256W_OPS_VOFF(va, vw), X2I_GEMM256C_OPS_MASK(mk, mk), [la] "v"(la), [lw] "v"(lw), [dma] "s"(dma), [ra] "s"(a_rsrc), [rw] "s"(w_rsrc),
                 [cmsb] "s"(c8), X2I_CONV_KORDER_OPS
               : "memory", "scc", "m0");
  for (int ui = 0;; ++ui) {
    const int nvb = w + (ui + 1) * G;
    const bool has_next = nvb < TT;
    int nz = 0, nm0 = 0, nn0 = 0;
    if (has_next) {
      tile_of(nvb, nz, nm0, nn0);
      offsets(nz, nm0, nn0, na, nw, nmk);
    } else {
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) na[jj] = nw[jj] = 0x80000000u, nmk[jj] = 0xffffffffu;  // behind the last unit: every piece out of range
    }
    f32x4_t acc[2][4][2][4];
    const int zs = 1;   // every unit is a whole tile: the first K-tile takes C = 0
    asm volatile(X2I_GEMM256C_MAIN
                 : X2I_GEMM256P_OPS_ACC_OUT(acc), X2I_GEMM256P_OPS_FRAG0_IO(fr), X2I_GEMM256P_OPS_FRAG1(fr), X2I_GEMM256C_OPS_TMP(tv), [la] "+v"(la), [lw] "+v"(lw),
                   [dma] "+s"(dma), [koff] "+s"(s_koff), [it] "=&s"(s_it), [koffa] "+s"(s_koffa), [sh] "+s"(s_sh), [kc0] "+s"(s_c0), [kc1] "+s"(s_c1),
                   [tmp] "=&s"(s_tmp), [msk] "=&s"(s_msk)
                 : X2I_GEMM256W_OPS_VOFF(va, vw), X2I_GEMM256P_OPS_NEXT(na, nw), X2I_GEMM256C_OPS_MASK(mk, nmk), [ra] "s"(a_rsrc), [rw] "s"(w_rsrc),
                   [nk] "s"(nk), [zs] "s"(zs), [cmsb] "s"(c8), X2I_CONV_KORDER_OPS
                 : "memory", "scc", "m0");
    // ---- epilogue of (z, m0, n0): per-wave private staging, no workgroup barrier; the next unit's first two K-tiles are in flight
    const Deq<false> dq;
    epilogue_chunked_pipe<ACT, false, RES, false, false, true>(p, acc, z, m0 + wm * 128, n0 + wn * 128, lane, stage, dq);
    if (!has_next) break;
    z = nz; m0 = nm0; n0 = nn0;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) va[jj] = na[jj], vw[jj] = nw[jj], mk[jj] = nmk[jj];
  }
  asm volatile(X2I_GEMM256P_DRAIN ::: "memory");
}

}  // namespace

// (act, residual) combinations the convolutions of the path use: plain, ReLU (ControlNeXt mid block), residual add
kern_t pick_gemm256c(int act, bool res) {
  if (res) return act == X2I_ACT_NONE ? (kern_t)gemm256c_kernel<X2I_ACT_NONE, true> : nullptr;
  if (act == X2I_ACT_NONE) return gemm256c_kernel<X2I_ACT_NONE, false>;
  if (act == X2I_ACT_RELU) return gemm256c_kernel<X2I_ACT_RELU, false>;
  if (act == X2I_ACT_SILU) return gemm256c_kernel<X2I_ACT_SILU, false>;
  return nullptr;
}

}  // namespace x2i_gemm
