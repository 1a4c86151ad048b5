// Layer-fusion stage of the alignment projector (reference utils/proj.py:62-72): HBM-bound streaming over the
// stacked MLLM hidden states x[B,C,S,H] (106 MB per sample for Qwen2.5-VL-7B).
//
//   conv5x5   : Conv2d(C -> 1, kernel 5, padding 2) over the (S,H) plane   (utils/proj.py:50,68-69)
//   layer_mean: (cha_scale * x).mean(dim=1) / x.mean(dim=1)                (utils/proj.py:66-67,70-71)
//
// conv5x5 reads every input element from HBM once: a block owns a [TS x TH] output tile, walks the C layers, stages
// the (TS+4) x (TH+16) halo tile of each layer in LDS with aligned 16-byte loads (the 2-column halo is rounded out to a
// whole 8-element chunk on each side so that every global load is a full, aligned vector), and each thread slides a
// 5-wide window down its column keeping TS running sums in registers.
#include "x2i_common.h"
#include "x2i_kernels.h"

namespace {

constexpr int TS = 16;           // output rows (tokens) per block
constexpr int TH = 256;          // output columns (features) per block == threads
constexpr int HROWS = TS + 4;    // staged rows
constexpr int HCOLS = TH + 16;   // staged columns: [h0-8, h0+TH+8)
constexpr int CPR = HCOLS / 8;   // 16-byte chunks per staged row (34)

__global__ __launch_bounds__(256) void conv5x5_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, bf16_t* __restrict__ y, int C, int S, int H) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[HROWS][HCOLS];
  __shared__ float wsh[64 * 25];
  const int tid = threadIdx.x;
  const int h0 = blockIdx.x * TH, s0 = blockIdx.y * TS, b = blockIdx.z;
  for (int i = tid; i < C * 25; i += 256) wsh[i] = w[i];
  float acc[TS];
#pragma unroll
  for (int i = 0; i < TS; ++i) acc[i] = 0.f;
  const long long plane = (long long)S * H;
  const bf16_t* xb = x + (long long)b * C * plane;
  for (int c = 0; c < C; ++c) {
    __syncthreads();  // previous layer's tile fully consumed (and wsh visible on the first pass)
    const bf16_t* xc = xb + (long long)c * plane;
    for (int p = tid; p < HROWS * CPR; p += 256) {
      const int r = p / CPR, ch = p - r * CPR;
      const int s = s0 - 2 + r, hh = h0 - 8 + ch * 8;
      bf16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (s >= 0 && s < S && hh >= 0 && hh + 8 <= H) v = *(const bf16x8_t*)(xc + (long long)s * H + hh);
      *(bf16x8_t*)&tile[r][ch * 8] = v;
    }
    __syncthreads();
    const float* wc = wsh + c * 25;
    float wr[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) wr[i] = wc[i];
#pragma unroll
    for (int r = 0; r < HROWS; ++r) {
      float v[5];
#pragma unroll
      for (int j = 0; j < 5; ++j) v[j] = bf16_to_f32(tile[r][tid + 6 + j]);  // column h-2+j  <->  staged col (h-h0)+8-2+j
#pragma unroll
      for (int di = 0; di < 5; ++di) {
        const int o = r - di;  // input row r feeds output row o with kernel row di
        if (o >= 0 && o < TS) {
#pragma unroll
          for (int j = 0; j < 5; ++j) acc[o] += wr[di * 5 + j] * v[j];
        }
      }
    }
  }
  const int h = h0 + tid;
  if (h < H) {
    const float bv = bias ? bias[0] : 0.f;
#pragma unroll
    for (int o = 0; o < TS; ++o) {
      const int s = s0 + o;
      if (s < S) y[((long long)b * S + s) * H + h] = f32_to_bf16(acc[o] + bv);
    }
  }
}

// y[b][i] = (1/C) sum_c scale[c] * x[b][c][i]   (i over the S*H plane), 8 elements per thread
__global__ __launch_bounds__(256) void layer_mean_kernel(const bf16_t* __restrict__ x, const float* __restrict__ scale,
                                                         bf16_t* __restrict__ y, int C, long long plane8) {
  const int b = blockIdx.y;
  const float invC = 1.f / (float)C;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < plane8; i += (long long)gridDim.x * 256) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) {
      const bf16x8_t v = *(const bf16x8_t*)(x + (((long long)b * C + c) * plane8 + i) * 8);
      const float sc = scale ? scale[c] : 1.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += sc * bf16_to_f32((bf16_t)v[j]);
    }
    bf16x8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (short)f32_to_bf16(acc[j] * invC);
    *(bf16x8_t*)(y + ((long long)b * plane8 + i) * 8) = o;
  }
}

}  // namespace

int x2i_launch_proj_conv5x5(const void* x, const float* w, const float* bias, void* y, int B, int C, int S, int H,
                            hipStream_t stream) {
  if (!x || !w || !y) return x2i_set_error(X2I_ERR_ARG, "proj_conv5x5: null pointer");
  if (B <= 0 || C <= 0 || C > 64 || S <= 0 || H <= 0 || H % 8) return x2i_set_error(X2I_ERR_SHAPE, "proj_conv5x5: need C <= 64 and H %% 8 == 0 (C=%d H=%d)", C, H);
  if ((((uintptr_t)x) & 15)) return x2i_set_error(X2I_ERR_ALIGN, "proj_conv5x5: x must be 16-byte aligned");
  dim3 grid((H + TH - 1) / TH, (S + TS - 1) / TS, B);
  hipLaunchKernelGGL(conv5x5_kernel, grid, dim3(256), 0, stream, (const bf16_t*)x, w, bias, (bf16_t*)y, C, S, H);
  return x2i_check_launch("proj_conv5x5");
}

int x2i_launch_layer_mean(const void* x, const float* scale, void* y, int B, int C, long long plane, hipStream_t stream) {
  if (!x || !y) return x2i_set_error(X2I_ERR_ARG, "layer_mean: null pointer");
  if (B <= 0 || C <= 0 || plane <= 0 || plane % 8) return x2i_set_error(X2I_ERR_SHAPE, "layer_mean: S*H must be a multiple of 8");
  const long long plane8 = plane / 8;
  const long long blocks = (plane8 + 255) / 256;
  hipLaunchKernelGGL(layer_mean_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048), B), dim3(256), 0, stream, (const bf16_t*)x,
                     scale, (bf16_t*)y, C, plane8);
  return x2i_check_launch("layer_mean");
}
