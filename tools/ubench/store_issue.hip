// Tools-only microbenchmark: how long a wave is held by back-to-back 1-KiB global stores (global_store_dwordx4, whole 128-byte lines,
// row stride 24 KiB as in a GEMM epilogue), one wave per SIMD on every CU: time after n stores (s_memtime), i.e. the depth of the
// store path a wave can fill before it blocks, and the steady rate behind it.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/store_issue.hip -o tools/ubench/bin/store_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k(uint4* out, long long stride_rows, unsigned long long* ts, int spacing) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // tile (blockIdx.x): 256 rows x 512 B; wave: 128 rows x 256 B quadrant; one store = 8 rows x 128 B
  char* base = (char*)out + ((long long)(blockIdx.x / 48) * 256 + (wave >> 1) * 128) * stride_rows + (blockIdx.x % 48) * 512 + (wave & 1) * 256;
  const uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
  unsigned long long t[33];
  t[0] = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int row = (i >> 1) * 8 + (lane >> 3), col = (i & 1) * 128 + (lane & 7) * 16;
    *(uint4*)(base + (long long)row * stride_rows + col) = v;
    for (int s = 0; s < spacing; ++s) __builtin_amdgcn_s_sleep(8);
    t[i + 1] = __builtin_readcyclecounter();
  }
  if (blockIdx.x < 4 && lane == 0)
    for (int i = 0; i < 33; ++i) ts[(blockIdx.x * 4 + wave) * 33 + i] = t[i] - t[0];
}
int main() {
  const long long rows = 18432, stride = 12288 * 2;
  uint4* out;
  hipMalloc(&out, rows * stride);
  unsigned long long* ts;
  hipMalloc(&ts, 16 * 33 * 8);
  std::vector<unsigned long long> h(16 * 33);
  for (int spacing : {0, 2, 8}) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, out, stride, ts, spacing);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), ts, h.size() * 8, hipMemcpyDeviceToHost);
    printf("spacing %d (x ~512 cycles of sleep between stores): cycles after n stores, wave 0 of workgroup 0:\n ", spacing);
    for (int i = 1; i <= 32; ++i) printf(" %llu", h[i]);
    printf("\n  wave 3 of workgroup 3:");
    for (int i = 1; i <= 32; ++i) printf(" %llu", h[15 * 33 + i]);
    printf("\n");
  }
  return 0;
}
