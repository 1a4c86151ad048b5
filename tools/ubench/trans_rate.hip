// Transcendental vs FMA issue-rate probe (gfx950): does v_exp_f32 / v_rcp_f32 share the VALU issue slot with v_fma_f32 (then an
// exp2 evaluated as a polynomial on the FMA pipe cannot beat it) or does it run beside it?  8 independent chains per lane, one
// wave per SIMD, inline asm so that the instruction mix is exactly what is written.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/trans_rate.hip -o /tmp/trans_rate && /tmp/trans_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int MODE>
__global__ void k(float* out, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b0 = a0, b1 = a1, b2 = a2, b3 = a3, b4 = a4, b5 = a5, b6 = a6, b7 = a7;
  float c = 0.999f;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#define EXP(i) "v_exp_f32 %" #i ", %" #i "\n"
#define RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define FMA(i) "v_fma_f32 %1" #i ", %1" #i ", %16, %16\n"   /* operands 10..17 = b0..b7 */
#define FMB(i) "v_fma_f32 %" #i ", %" #i ", %16, %16\n"
#define PKF(i) "v_pk_fma_f32 %[p" #i "], %[p" #i "], %[pc], %[pc]\n"
    if (MODE == 0) asm volatile(R8(EXP) R8(EXP) R8(EXP) R8(EXP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(c));
    if (MODE == 1) asm volatile(R8(FMB) R8(FMB) R8(FMB) R8(FMB) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(c));
    // 16 exp + 16 fma, interleaved 1 : 1
    if (MODE == 2) asm volatile(EXP(0) FMA(0) EXP(1) FMA(1) EXP(2) FMA(2) EXP(3) FMA(3) EXP(4) FMA(4) EXP(5) FMA(5) EXP(6) FMA(6) EXP(7) FMA(7)
                                EXP(0) FMA(0) EXP(1) FMA(1) EXP(2) FMA(2) EXP(3) FMA(3) EXP(4) FMA(4) EXP(5) FMA(5) EXP(6) FMA(6) EXP(7) FMA(7)
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(c));
    // 8 exp + 24 fma, 1 : 3
    if (MODE == 3) asm volatile(EXP(0) FMA(0) FMA(1) FMA(2) EXP(1) FMA(3) FMA(4) FMA(5) EXP(2) FMA(6) FMA(7) FMA(0) EXP(3) FMA(1) FMA(2) FMA(3)
                                EXP(4) FMA(4) FMA(5) FMA(6) EXP(5) FMA(7) FMA(0) FMA(1) EXP(6) FMA(2) FMA(3) FMA(4) EXP(7) FMA(5) FMA(6) FMA(7)
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(c));
    if (MODE == 4) asm volatile(R8(RCP) R8(RCP) R8(RCP) R8(RCP) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) : "v"(c));
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}
int main() {
  float* d; hipMalloc(&d, 1 << 24);
  const int iters = 4096;
  const char* names[5] = {"32 x v_exp_f32", "32 x v_fma_f32", "16 exp + 16 fma interleaved", "8 exp + 24 fma interleaved", "32 x v_rcp_f32"};
  for (int m = 0; m < 5; ++m) {
    dim3 g(256), bsz(256);
    for (int rep = 0; rep < 2; ++rep) {
      if (m == 0) hipLaunchKernelGGL(k<0>, g, bsz, 0, 0, d, iters);
      if (m == 1) hipLaunchKernelGGL(k<1>, g, bsz, 0, 0, d, iters);
      if (m == 2) hipLaunchKernelGGL(k<2>, g, bsz, 0, 0, d, iters);
      if (m == 3) hipLaunchKernelGGL(k<3>, g, bsz, 0, 0, d, iters);
      if (m == 4) hipLaunchKernelGGL(k<4>, g, bsz, 0, 0, d, iters);
      hipDeviceSynchronize();
    }
    float cyc; hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
    printf("%-30s %.2f clock64 ticks per 32-instruction group = %.2f per instruction\n", names[m], cyc / iters, cyc / iters / 32);
  }
  return 0;
}
