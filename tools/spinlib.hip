// tools/spinlib.hip -- co-runner kernels for tools/corun3.py (one resident wavefront)
#include <hip/hip_runtime.h>
__global__ void spin(long long ticks, long long *o) {
  long long t0 = wall_clock64();
  float x = threadIdx.x;
  while (wall_clock64() - t0 < ticks) {
#pragma unroll
    for (int j = 0; j < 64; ++j) x = __builtin_fmaf(x, 1.0001f, 0.5f);
  }
  if (x == 12345.f) o[0] = 1;
}
__global__ void spin_mem(long long ticks, const float2 *in, float2 *out, int rows) {
  long long t0 = wall_clock64();
  float2 acc{0, 0};
  int r = 0;
  while (wall_clock64() - t0 < ticks) {
    for (int j = 0; j < 64; ++j) {
      float2 v = in[(size_t)r * 64 + threadIdx.x];
      acc.x = __builtin_fmaf(acc.x, 0.5f, v.x); acc.y = __builtin_fmaf(acc.y, 0.5f, v.y);
      out[(size_t)r * 64 + threadIdx.x] = acc;
      r = (r + 1 == rows) ? 0 : r + 1;
    }
  }
}
// scalar-load heavy: walks a table through the scalar cache
__global__ void spin_smem(long long ticks, const float4 *tab, int n, float *o) {
  long long t0 = wall_clock64();
  float acc = 0; int i = 0;
  while (wall_clock64() - t0 < ticks) {
    for (int j = 0; j < 64; ++j) {
      float4 v = tab[__builtin_amdgcn_readfirstlane(i)];
      acc += v.x + v.w;
      i = (i + 4 >= n) ? 0 : i + 4;
    }
  }
  if (acc == 12345.f) o[0] = acc;
}
extern "C" int launch_spin(void *st, int kind, long long ticks, void *a, void *b, int n) {
  hipStream_t s = (hipStream_t)st;
  if (kind == 0) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, ticks, (long long *)b);
  else if (kind == 1) hipLaunchKernelGGL(spin_mem, dim3(1), dim3(64), 0, s, ticks, (const float2 *)a, (float2 *)b, n);
  else hipLaunchKernelGGL(spin_smem, dim3(1), dim3(64), 0, s, ticks, (const float4 *)a, n, (float *)b);
  return (int)hipGetLastError();
}
