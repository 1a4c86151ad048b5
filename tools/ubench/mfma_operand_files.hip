// Tools-only microbenchmark: v_mfma_f32_32x32x16_bf16 issue rate by the register file of its operands (A / B / C=D in the architectural
// VGPRs "v" or the accumulator file "a"), 8 independent accumulators, one wave per SIMD (256 threads per workgroup, 1 workgroup per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_operand_files.hip -o /tmp/mfma_of && /tmp/mfma_of
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int VAR>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  // operands: A v[32:35] / a[192:195], B v[36:39] / a[196:199]; accumulators v[64:191] / a[0:127]
  for (int it = 0; it < iters; ++it) {
    if (VAR == 0) {
#define M(i) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], v[32:35], v[36:39], a[%c0:%c1]" ::"i"(16 * i), "i"(16 * i + 15) : "a0");
      REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M)
#undef M
    } else if (VAR == 1) {
#define M(i) asm volatile("v_mfma_f32_32x32x16_bf16 v[%c0:%c1], a[192:195], a[196:199], v[%c0:%c1]" ::"i"(64 + 16 * i), "i"(64 + 16 * i + 15) : "v64");
      REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M)
#undef M
    } else if (VAR == 2) {
#define M(i) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], a[192:195], v[36:39], a[%c0:%c1]" ::"i"(16 * i), "i"(16 * i + 15) : "a0");
      REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M)
#undef M
    } else if (VAR == 3) {
#define M(i) asm volatile("v_mfma_f32_32x32x16_bf16 v[%c0:%c1], v[32:35], v[36:39], v[%c0:%c1]" ::"i"(64 + 16 * i), "i"(64 + 16 * i + 15) : "v64");
      REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M)
#undef M
    } else if (VAR == 4) {
#define M(i) asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], a[192:195], a[196:199], a[%c0:%c1]" ::"i"(16 * i), "i"(16 * i + 15) : "a0");
      REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M) REP8(M)
#undef M
    } else {  // alternate my two kinds: QK form (a, a -> v) and PV form (a, v -> a)
#define M(i) asm volatile("v_mfma_f32_32x32x16_bf16 v[%c0:%c1], a[192:195], a[196:199], v[%c0:%c1]\n v_mfma_f32_32x32x16_bf16 a[%c2:%c3], a[192:195], v[36:39], a[%c2:%c3]" ::"i"(64 + 16 * (i & 3)), "i"(64 + 16 * (i & 3) + 15), "i"(16 * i), "i"(16 * i + 15) : "a0", "v64");
      REP8(M) REP8(M) REP8(M) REP8(M)
#undef M
    }
  }
  if (out && threadIdx.x == 9999) out[0] = 1.f;
}
template <int VAR>
double run(const char* name) {
  const int iters = 4000;
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  hipLaunchKernelGGL(k<VAR>, dim3(256), dim3(256), 0, 0, nullptr, 10);
  hipDeviceSynchronize();
  hipEventRecord(s);
  hipLaunchKernelGGL(k<VAR>, dim3(256), dim3(256), 0, 0, nullptr, iters);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms;
  hipEventElapsedTime(&ms, s, e);
  const double fl = 2.0 * 32 * 32 * 16 * 64.0 * iters * 1024;  // per MFMA x 64 per iteration x iterations x SIMDs
  printf("%-44s %8.3f ms  %8.1f TFLOP/s\n", name, ms, fl / (ms * 1e-3) / 1e12);
  return ms;
}
int main() {
  run<0>("A=v B=v C/D=a (GEMM form)");
  run<1>("A=a B=a C/D=v (attention QK form)");
  run<2>("A=a B=v C/D=a (attention PV form)");
  run<3>("A=v B=v C/D=v");
  run<4>("A=a B=a C/D=a");
  run<5>("QK form / PV form alternating");
  run<0>("A=v B=v C/D=a (GEMM form)");
  return 0;
}
