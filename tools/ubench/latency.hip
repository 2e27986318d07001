// dependent-chain latency probe (gfx950, one wavefront per SIMD): ticks per instruction of a chain in which every
// instruction reads the previous one's result
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int KIND>
__global__ __launch_bounds__(64) void probe(unsigned long long *out, float seed, int iseed)
{
  float a = seed; int n = iseed; f2 p = {seed, seed + 1.0f}; const float k = 1.0001f; const f2 kk = {1.0001f, 0.9999f};
  unsigned u = (unsigned)iseed;
  unsigned long long t0 = __builtin_readcyclecounter();
  __builtin_amdgcn_sched_barrier(0);
  for (int it = 0; it < 16; ++it) {
    if (KIND == 0) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(k));) }
    if (KIND == 1) { REP64(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(k));) }
    if (KIND == 2) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(kk));) }
    if (KIND == 3) { REP64(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(kk));) }
    if (KIND == 4) { REP64(asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel:[1,0] op_sel_hi:[0,1]" : "+v"(p) : "v"(kk));) }   // result halves cross over
    if (KIND == 5) { REP64(asm volatile("v_cvt_i32_f32 %0, %1\n v_cvt_f32_i32 %1, %0" : "+v"(n), "+v"(a));) }            // 2 instructions per rep
    if (KIND == 6) { REP64(asm volatile("v_med3_f32 %0, %0, -1.0, 1.0" : "+v"(a));) }
    if (KIND == 7) { REP64(asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x78" : "+v"(u) : "v"(n), "s"(0x80000000u));) }
    if (KIND == 8) { REP64(asm volatile("v_bfe_i32 %0, %0, 0, 30" : "+v"(n));) }
    if (KIND == 9) { REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(u) : "v"(n));) }
    if (KIND == 12) { REP64(asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(*(unsigned long long *)&p) : "s"(0x200ull));) }
  }
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = (unsigned long long)(a + p.x + p.y + (float)n + (float)u != 12345.f); }
}
template <int KIND> void run(const char *name, int per_rep)
{
  unsigned long long *d; (void)hipMalloc(&d, 64);
  hipLaunchKernelGGL(probe<KIND>, dim3(1), dim3(64), 0, 0, d, 1.0f, 3);
  hipLaunchKernelGGL(probe<KIND>, dim3(1), dim3(64), 0, 0, d, 1.0f, 3);
  (void)hipDeviceSynchronize();
  unsigned long long h[2]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("%-52s %.2f ticks per instruction\n", name, (double)h[0] / (16.0 * 64 * per_rep));
  (void)hipFree(d);
}
int main()
{
  run<0>("v_fma_f32 chain", 1); run<1>("v_mul_f32 chain", 1); run<2>("v_pk_fma_f32 chain", 1); run<3>("v_pk_mul_f32 chain", 1);
  run<4>("v_pk_mul_f32 chain, halves crossing (op_sel)", 1); run<5>("v_cvt_i32_f32 <-> v_cvt_f32_i32 chain", 2);
  run<6>("v_med3_f32 chain", 1); run<7>("v_bitop3_b32 chain", 1); run<8>("v_bfe_i32 chain", 1); run<9>("v_add_u32 chain", 1);
  run<12>("v_lshl_add_u64 chain", 1);
  return 0;
}
