// Does a lone wavefront issue faster when only part of its lanes are active?  (gfx950: a wave64 VALU op runs in four
// 16-lane passes; if passes with no active lane were skipped, a serial per-lane recurrence would run faster with its 64
// channels spread over four wavefronts of 16.)  Dependent and independent v_fma_f32 / v_pk_fma_f32 streams with EXEC
// restricted to the first K lanes.   build: hipcc --offload-arch=gfx950 -O3 -o execmask execmask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int KIND>
__global__ __launch_bounds__(64) void probe(unsigned long long *out, float seed, int lanes)
{
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
  f2 p0 = {seed, seed};
  const float k = 1.0001f; const f2 kk = {1.0001f, 0.9999f};
  unsigned long long t0 = 0, t1 = 0;
  if ((int)threadIdx.x < lanes) {
    t0 = __builtin_readcyclecounter();
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < 16; ++it) {
      if (KIND == 0) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1" : "+v"(a0) : "v"(k));) }
      if (KIND == 1) { REP64(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(k));) }
      if (KIND == 2) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(kk));) }
    }
    __builtin_amdgcn_sched_barrier(0);
    t1 = __builtin_readcyclecounter();
  }
  float s = a0 + a1 + a2 + a3 + p0.x + p0.y;
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)(s != 12345.f); }
}
template <int KIND> void run(const char *name, int lanes)
{
  unsigned long long *d, h[2]; hipMalloc(&d, 16);
  hipLaunchKernelGGL(probe<KIND>, dim3(1), dim3(64), 0, 0, d, 1.0f, lanes);
  hipLaunchKernelGGL(probe<KIND>, dim3(1), dim3(64), 0, 0, d, 1.0f, lanes);
  hipDeviceSynchronize();
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("%-28s %2d lanes: %.2f ticks per instruction\n", name, lanes, (double)h[0] / (16.0 * 64 * 4));
  hipFree(d);
}
int main()
{
  for (int lanes : {64, 48, 32, 16, 1}) {
    run<0>("v_fma_f32 dependent", lanes);
    run<1>("v_fma_f32 independent x4", lanes);
    run<2>("v_pk_fma_f32 dependent", lanes);
  }
  return 0;
}
