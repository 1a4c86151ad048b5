// Tools-only microbenchmark: sustained (power-limited) MFMA rate on RANDOM bf16 operands, by instruction shape --
// v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16 (same operand registers per instruction, half the FLOPs) -- one wave per SIMD,
// operands in VGPRs, accumulators in the accumulator file, launches long enough (hundreds of ms) for the power management to settle.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_power.hip -o tools/ubench/bin/mfma_power
#include <hip/hip_runtime.h>
#include "mfma_power_kernel.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
template <int VAR>
void run(const char* name, const uint4* d, int iters, int reps = 3, double flop_scale = 1.0) {
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  for (int rep = 0; rep < reps; ++rep) {
    hipEventRecord(s);
    hipLaunchKernelGGL(k<VAR>, dim3(256), dim3(256), 0, 0, d, nullptr, iters);
    hipEventRecord(e);
    hipEventSynchronize(e);
    float ms;
    hipEventElapsedTime(&ms, s, e);
    const double fl = 2.0 * 32 * 32 * 16 * 32.0 * iters * 1024;  // FLOPs per loop iteration (32 MFMAs 32x32x16, or 64 MFMAs 16x16x32) x SIMDs
    printf("%-28s run %d %9.3f ms  %8.1f TFLOP/s\n", name, rep, ms, flop_scale * fl / (ms * 1e-3) / 1e12);
  }
}
int main(int argc, char** argv) {
  const int zero = argc > 1 && atoi(argv[1]) == 0;
  const size_t n = (size_t)256 * 256 * 16 * 8;  // bf16 values
  std::vector<unsigned short> h(n);
  srand(1);
  for (auto& x : h) {
    const float f = zero ? 0.f : (rand() / (float)RAND_MAX) * 4.f - 2.f;
    unsigned u;
    memcpy(&u, &f, 4);
    x = (unsigned short)(u >> 16);
  }
  uint4* d;
  hipMalloc(&d, n * 2);
  hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
  const int iters = 400000;  // ~ 0.3 s per launch at 2 PF
  const int reps = argc > 2 ? atoi(argv[2]) : 3;   // (round 5: e.g. 20 launches = ~6 s per shape, long enough for the power management to settle)
  // e4m3: the same random bits read as e4m3 bytes (NaN encodings 0x7f / 0xff occur once in 128 bytes: left in, they are data too)
  run<2>(zero ? "e4m3 16x16x128 zeros" : "e4m3 16x16x128 random", d, iters / 2, reps, 4.0);
  run<0>(zero ? "32x32x16 zeros" : "32x32x16 random", d, iters, reps);
  run<1>(zero ? "16x16x32 zeros" : "16x16x32 random", d, iters, reps);
  run<5>("32x32x16, first operand held x8", d, iters, reps);
  run<6>("32x32x16 held + 4 VALU / MFMA", d, iters, reps);
  run<7>("16x16x32 held + 2 VALU / MFMA", d, iters, reps);
  run<3>("16x16x32, first operand held x8", d, iters, reps);
  run<4>("16x16x32, second operand held x8", d, iters, reps);
  run<0>(zero ? "32x32x16 zeros" : "32x32x16 random", d, iters, reps);
  return 0;
}
