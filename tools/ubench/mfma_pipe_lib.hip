// Measurement library for bench.py's LIVE `matrix_pipe_alone` figure: the register-resident MFMA streams of
// mfma_power.hip behind a C entry point, so that the driver's own bench run measures what the power management lets the matrix pipe do on
// the box it runs on -- one wave per SIMD on every CU, operands and accumulators in registers, no memory traffic inside the timed loop.
// Built by __graft_entry__.build() into tools/ubench/libx2i_ubench.so; it is NOT part of libx2i_hip.so and the product package never loads it.
//   variant 1 = v_mfma_f32_16x16x32_bf16, operands rotating        (the figure quoted as `matrix_pipe_alone`)
//   variant 3 = the same with one operand held for eight MFMAs      (the K-loops' pattern)
//   variant 2 = v_mfma_f32_16x16x128_f8f6f4 on e4m3 bytes           (4 x the FLOPs per instruction)
//   variant 0 = v_mfma_f32_32x32x16_bf16
#include "mfma_power_kernel.h"

extern "C" {
// data: device pointer to >= x2i_ubench_data_bytes() bytes of operand bits (the caller fills them: random bf16 in [-2, 2), or zeros);
// iters: loop iterations (64 MFMAs of 16x16x32 per iteration and wave); returns 0 / -1 (unknown variant) / -2 (launch error)
long long x2i_ubench_data_bytes(void) { return 256LL * 256 * 16 * 16; }
double x2i_ubench_flop_per_iter(int variant) {   // whole launch: 1024 waves
  const double fl = 2.0 * 32 * 32 * 16 * 32.0 * 1024;
  return variant == 2 ? 4.0 * fl : fl;
}
int x2i_ubench_mfma_pipe(int variant, int iters, const void* data, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const uint4* d = (const uint4*)data;
  switch (variant) {
    case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, s, d, nullptr, iters); break;
    case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, s, d, nullptr, iters); break;
    case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, s, d, nullptr, iters); break;
    case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, s, d, nullptr, iters); break;
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
}
