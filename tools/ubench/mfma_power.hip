// Tools-only microbenchmark: sustained (power-limited) MFMA rate on RANDOM bf16 operands, by instruction shape --
// v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16 (same operand registers per instruction, half the FLOPs) -- one wave per SIMD,
// operands in VGPRs, accumulators in the accumulator file, launches long enough (hundreds of ms) for the power management to settle.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_power.hip -o tools/ubench/bin/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int VAR>
__global__ __launch_bounds__(256) void k(const uint4* __restrict__ data, float* out, int iters) {
  // 8 operand pairs v[32:35]..v[95:..]: 16 fragments of 4 registers = v32..v95, loaded from memory (random bf16 in [-2, 2))
  const uint4* p = data + (size_t)(blockIdx.x * 256 + threadIdx.x) * 16;
#define LD(i) asm volatile("global_load_dwordx4 v[%c0:%c1], %2, off offset:%c3" ::"i"(32 + 4 * i), "i"(35 + 4 * i), "v"(p), "i"(16 * i) : "memory");
  REP8(LD)
#undef LD
#define LD(i) asm volatile("global_load_dwordx4 v[%c0:%c1], %2, off offset:%c3" ::"i"(64 + 4 * i), "i"(67 + 4 * i), "v"(p), "i"(128 + 16 * i) : "memory");
  REP8(LD)
#undef LD
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int it = 0; it < iters; ++it) {
    if (VAR == 0) {  // 32x32x16: 8 accumulators of 16 registers, operands rotate over the 8 fragment pairs
#define M(i) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], v[%c2:%c3], v[%c4:%c5], a[%c0:%c1]" ::"i"(16 * i), "i"(16 * i + 15), "i"(32 + 4 * i), "i"(35 + 4 * i), "i"(64 + 4 * i), "i"(67 + 4 * i));
      REP8(M) REP8(M) REP8(M) REP8(M)
#undef M
    } else {  // 16x16x32: 16 accumulators of 4 registers (two per fragment pair), same FLOPs per loop iteration
#define M(i) asm volatile("v_mfma_f32_16x16x32_bf16 a[%c0:%c1], v[%c2:%c3], v[%c4:%c5], a[%c0:%c1]\n v_mfma_f32_16x16x32_bf16 a[%c6:%c7], v[%c4:%c5], v[%c2:%c3], a[%c6:%c7]" ::"i"(8 * i), "i"(8 * i + 3), "i"(32 + 4 * i), "i"(35 + 4 * i), "i"(64 + 4 * i), "i"(67 + 4 * i), "i"(8 * i + 4), "i"(8 * i + 7));
      REP8(M) REP8(M) REP8(M) REP8(M)
#undef M
    }
  }
  if (out && threadIdx.x == 9999) out[0] = 1.f;
}
template <int VAR>
void run(const char* name, const uint4* d, int iters, int reps = 3) {
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
    printf("%-28s run %d %9.3f ms  %8.1f TFLOP/s\n", name, rep, ms, fl / (ms * 1e-3) / 1e12);
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
  run<0>(zero ? "32x32x16 zeros" : "32x32x16 random", d, iters, reps);
  run<1>(zero ? "16x16x32 zeros" : "16x16x32 random", d, iters, reps);
  run<0>(zero ? "32x32x16 zeros" : "32x32x16 random", d, iters, reps);
  return 0;
}
