// VALU issue-rate probe (gfx950): cycles per wave-instruction of v_dot2c_f32_bf16 / v_fma_f32 / v_pk_fma_f32 with 8 independent
// accumulator chains per lane, 1..2 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
template <int MODE>
__global__ void k(float* out, unsigned a, unsigned b, int iters) {
  float acc[8];
  f32x2 pacc[8];
  for (int i = 0; i < 8; ++i) { acc[i] = threadIdx.x * 0.001f + i; pacc[i] = {acc[i], acc[i] + 1}; }
  unsigned x = a + threadIdx.x, w = b;
  float xf = __uint_as_float(x), wf = __uint_as_float(w);
  f32x2 xp = {xf, xf * 0.5f}, wp = {wf, wf * 0.25f};
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) acc[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, x), __builtin_bit_cast(bf16x2, w), acc[i], false);
        if (MODE == 1) acc[i] = __builtin_fmaf(xf, wf, acc[i]);
        if (MODE == 2) pacc[i] = __builtin_elementwise_fma(xp, wp, pacc[i]);
      }
  }
  long long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i] + pacc[i][0] + pacc[i][1];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}
int main() {
  float* d; hipMalloc(&d, 1 << 24);
  const int iters = 4096;
  const char* names[3] = {"v_dot2c_f32_bf16", "v_fma_f32", "v_pk_fma_f32"};
  for (int waves = 1; waves <= 2; ++waves)
    for (int m = 0; m < 3; ++m) {
      dim3 g(256), bsz(256 * waves);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (m == 0) hipLaunchKernelGGL(k<0>, g, bsz, 0, 0, d, 0x3f803f80u, 0x3f003f00u, iters);
        if (m == 1) hipLaunchKernelGGL(k<1>, g, bsz, 0, 0, d, 0x3f803f80u, 0x3f003f00u, iters);
        if (m == 2) hipLaunchKernelGGL(k<2>, g, bsz, 0, 0, d, 0x3f803f80u, 0x3f003f00u, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      float cyc; hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
      const double winstr = (double)iters * 32;  // wave-instructions per wave
      printf("%-18s %d wave(s)/SIMD: %.2f ms, %.2f clock64 ticks per wave-instruction (one wave), chip rate %.1f G wave-instr/s\n", names[m], waves, ms,
             cyc / winstr, winstr * 256 * 4 * waves / (ms * 1e-3) / 1e9);
    }
  return 0;
}
