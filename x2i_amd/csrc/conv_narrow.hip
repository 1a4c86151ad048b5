// Conv2d(Cin -> Cout <= 4, k = 3, stride 1, padding 1) on NHWC bf16: the VAE decoder's conv_out (128 -> 3 channels at the full image
// resolution; diffusers Decoder.conv_out behind conv_norm_out + SiLU, reached from infer/inference_qwenvl.py:213-214).  As an implicit GEMM on
// the 128-column tile kernels three output channels pay for 128 (1.3 ms per four 1024^2 images, profiles/r05zz_vae_b4_kernel_stats.csv);
// here the matrix core's 16 x 16 x 32 shape carries the output channels in its 16 ROWS and 16 neighbouring pixels of an image row in its
// columns:
//   * a wave owns a strip of 16 pixel columns and walks down the rows of its block; the 9 x Cin / 32 weight fragments (cout i = lane & 15,
//     zero rows for i >= Cout) stay in registers for the whole walk;
//   * per INPUT row it fetches 3 (horizontal shift) x Cin / 32 pixel fragments straight from global memory (lane: pixel x0 + (lane & 15) + s,
//     16 bytes of channels) and uses each of them three times -- filter rows ky = 0, 1, 2 add it to the output rows below, at and above it
//     (three rotating accumulators) -- so a fragment load feeds three MFMAs and no LDS is involved;
//   * an output row leaves as 8 bytes per pixel (four bf16 channels, the ones behind Cout are zeros).
// f32 accumulation over all 9 Cin terms as in the GEMM form (another order: same tolerance, not bit-identical).
#include "x2i_common.h"
#include "x2i_kernels.h"

namespace {

template <int KC, int TM>
__device__ __forceinline__ void narrow_row(const bf16_t* __restrict__ xrow, bool row_ok, int x0, int n, int k8, int W, int Cin,
                                           const bf16x8_t (&wf)[3][3][KC], f32x4_t (&acc)[3]) {
  if (!row_ok) return;   // (wave-uniform: a row outside the image adds nothing)
  bf16x8_t xf[3][KC];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int x = x0 + n + s - 1;
    const bool ok = x >= 0 && x < W;
    const bf16_t* p = xrow + (long long)x * Cin + k8;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) xf[s][kc] = ok ? *(const bf16x8_t*)(p + kc * 32) : (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
  }
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int slot = (TM - ky + 3) % 3;   // input row t feeds output row t - ky (relative): accumulator (t - ky) mod 3
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) acc[slot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ky][s][kc], xf[s][kc], acc[slot], 0, 0, 0);
  }
}

template <int KC>
__global__ __launch_bounds__(256, 2) void conv3x3_narrow_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                                const bf16_t* __restrict__ bias, bf16_t* __restrict__ y, int H, int W,
                                                                int Cout, int ldy, int rows_per_block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, g = lane >> 4, k8 = g * 8;
  const int Cin = KC * 32;
  const int x0 = (blockIdx.x * 4 + wave) * 16;
  if (x0 >= W) return;
  const int y0 = blockIdx.y * rows_per_block, nrows = min(rows_per_block, H - y0);
  const long long img = (long long)blockIdx.z * H * W;
  bf16x8_t wf[3][3][KC];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int kc = 0; kc < KC; ++kc)
        wf[ky][s][kc] = (n < Cout) ? *(const bf16x8_t*)(w + ((long long)n * 9 + ky * 3 + s) * Cin + kc * 32 + k8) : (bf16x8_t){0, 0, 0, 0, 0, 0, 0, 0};
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias && g == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = r < Cout ? bf16_to_f32(bias[r]) : 0.f;
  }
  f32x4_t acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  auto finish = [&](int slot_acc_index, int o) {   // output row y0 + o is complete in acc[slot]: lanes 0 .. 15 hold channels 0 .. 3 of pixel x0 + n
    f32x4_t& a = acc[slot_acc_index];
    if (o >= 0 && o < nrows && g == 0 && x0 + n < W) {
      bf16_t* dst = y + (img + (long long)(y0 + o) * W + x0 + n) * ldy;
      *(uint2*)dst = make_uint2(pack_bf16x2(a[0] + bv[0], a[1] + bv[1]), pack_bf16x2(a[2] + bv[2], a[3] + bv[3]));
    }
    a = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  };
  // input rows r = y0 - 1 + t, t = 0 .. nrows + 1, three at a time so that the accumulator rotation is static
  for (int t0 = 0; t0 < nrows + 2; t0 += 3) {
    {
      const int r = y0 - 1 + t0;
      narrow_row<KC, 0>(x + (img + (long long)r * W) * Cin, r >= 0 && r < H && t0 < nrows + 2, x0, n, k8, W, Cin, wf, acc);
      finish(1, t0 - 2);
    }
    {
      const int r = y0 + t0;
      narrow_row<KC, 1>(x + (img + (long long)r * W) * Cin, r >= 0 && r < H && t0 + 1 < nrows + 2, x0, n, k8, W, Cin, wf, acc);
      finish(2, t0 - 1);
    }
    {
      const int r = y0 + 1 + t0;
      narrow_row<KC, 2>(x + (img + (long long)r * W) * Cin, r >= 0 && r < H && t0 + 2 < nrows + 2, x0, n, k8, W, Cin, wf, acc);
      finish(0, t0);
    }
  }
}

}  // namespace

int x2i_launch_conv3x3_narrow(const void* x, const void* w, const void* bias, void* y, int B, int H, int W, int Cin, int Cout, int ldy,
                              hipStream_t stream) {
  if (!x || !w || !y) return x2i_set_error(X2I_ERR_ARG, "conv3x3_narrow: null pointer");
  if (B <= 0 || H <= 0 || W <= 0 || Cout < 1 || Cout > 4 || (Cin != 32 && Cin != 64 && Cin != 96 && Cin != 128) || ldy < 4 || (ldy & 3))
    return x2i_set_error(X2I_ERR_SHAPE, "conv3x3_narrow: serves Cin in {32, 64, 96, 128}, Cout <= 4, ldy a multiple of 4 and >= 4 (Cin=%d Cout=%d ldy=%d)", Cin, Cout, ldy);
  if ((((uintptr_t)x) | ((uintptr_t)w)) & 15 || (((uintptr_t)y) & 7)) return x2i_set_error(X2I_ERR_ALIGN, "conv3x3_narrow: x / w need 16-byte, y 8-byte alignment");
  // rows per block: enough blocks to fill the chip twice over, strips long enough to amortise the weight fragments
  const int strips = (W + 63) / 64;
  int rb = 64;
  while (rb > 8 && (long long)strips * ((H + rb - 1) / rb) * B < 2 * 2 * x2i_num_cus()) rb >>= 1;
  dim3 grid(strips, (H + rb - 1) / rb, B);
#define X2I_NARROW(KC) \
  hipLaunchKernelGGL(conv3x3_narrow_kernel<KC>, grid, dim3(256), 0, stream, (const bf16_t*)x, (const bf16_t*)w, (const bf16_t*)bias, (bf16_t*)y, H, W, Cout, ldy, rb)
  switch (Cin / 32) {
    case 1: X2I_NARROW(1); break;
    case 2: X2I_NARROW(2); break;
    case 3: X2I_NARROW(3); break;
    default: X2I_NARROW(4); break;
  }
#undef X2I_NARROW
  return x2i_check_launch("conv3x3_narrow");
}
