// tools/issue_rate.hip -- micro-benchmark: how fast does ONE wavefront issue instructions on
// gfx950?  (dependent fma chain, 4 independent fma chains, and cmp->cndmask pairs.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void dep_chain(float *o, int n, float a, float b) {
  float x = threadIdx.x;
  long long t0 = wall_clock64(); long long c0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < 64; ++j) x = __builtin_fmaf(x, a, b);
  }
  long long c1 = clock64(); long long t1 = wall_clock64();
  o[threadIdx.x] = x;
  if (threadIdx.x == 0) { ((long long *)o)[64] = c1 - c0; ((long long *)o)[65] = t1 - t0; }
}
__global__ void indep_chain(float *o, int n, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  long long c0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) { x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b); }
  }
  long long c1 = clock64();
  o[threadIdx.x] = x0 + x1 + x2 + x3;
  if (threadIdx.x == 0) ((long long *)o)[64] = c1 - c0;
}
__global__ void cmp_sel(float *o, int n, float a, float b) {
  float x = threadIdx.x;
  long long c0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < 32; ++j) { x = (x > a) ? x - b : x + a; }
  }
  long long c1 = clock64();
  o[threadIdx.x] = x;
  if (threadIdx.x == 0) ((long long *)o)[64] = c1 - c0;
}
int main() {
  float *d; hipMalloc(&d, 1024);
  long long h[2];
  const int n = 100000;
  for (int rep = 0; rep < 2; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(dep_chain, dim3(1), dim3(64), 0, 0, d, n, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    auto t1 = std::chrono::steady_clock::now();
    hipMemcpy(h, (char *)d + 64 * 8, 16, hipMemcpyDeviceToHost);
    double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
    printf("dep fma chain : %.2f shader-clk/op, %.3f ns/op wall (%.0f us), wall_clock ticks/op %.3f\n",
           (double)h[0] / (64.0 * n), us * 1e3 / (64.0 * n), us, (double)h[1] / (64.0 * n));
    hipLaunchKernelGGL(indep_chain, dim3(1), dim3(64), 0, 0, d, n, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipMemcpy(h, (char *)d + 64 * 8, 8, hipMemcpyDeviceToHost);
    printf("4 indep chains: %.2f shader-clk/op\n", (double)h[0] / (64.0 * n));
    hipLaunchKernelGGL(cmp_sel, dim3(1), dim3(64), 0, 0, d, n, 100.0f, 0.5f);
    hipDeviceSynchronize();
    hipMemcpy(h, (char *)d + 64 * 8, 8, hipMemcpyDeviceToHost);
    printf("cmp+sel+add   : %.2f shader-clk per (cmp, 2 arith, cndmask) group\n", (double)h[0] / (32.0 * n));
  }
  return 0;
}
