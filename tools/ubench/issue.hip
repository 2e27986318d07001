// lone-wavefront issue-rate probe (gfx950): cycles per instruction for independent / dependent streams of
// v_fma_f32, v_pk_fma_f32, v_pk_add_f32, v_add_f32; one wave per SIMD (256-thread blocks would hide it: 64-thread blocks)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
template <int KIND>
__global__ __launch_bounds__(64) void probe(unsigned long long *out, float seed)
{
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  f2 p0 = {seed, seed}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, p4 = p0 + 4.f, p5 = p0 + 5.f, p6 = p0 + 6.f, p7 = p0 + 7.f;
  const float k = 1.0001f; const f2 kk = {1.0001f, 0.9999f};
  const float sr = __builtin_amdgcn_readfirstlane(__float_as_int(seed)) * 1e-9f + 0.5f, si = sr + 0.25f; const f2 sk = {sr, si}, sk2 = {si, sr}, sk3 = {sr + 1.f, si}, sk4 = {si + 1.f, sr};
  unsigned long long t0 = __builtin_readcyclecounter();
  __builtin_amdgcn_sched_barrier(0);
  for (int it = 0; it < 16; ++it) {
    if (KIND == 0) { REP64(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));) }
    if (KIND == 1) { REP64(asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1" : "+v"(a0) : "v"(k));) }
    if (KIND == 2) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(kk));) }
    if (KIND == 3) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(kk));) }
    if (KIND == 4) { REP64(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(kk));) }
    if (KIND == 5) { REP64(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(k));) }
    // two interleaved dependent chains (distance 2)
    if (KIND == 6) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2\n v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2\n v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2\n v_pk_fma_f32 %0, %0, %2, %2\n v_pk_fma_f32 %1, %1, %2, %2" : "+v"(p0), "+v"(p1) : "v"(kk));) }
    if (KIND == 7) { REP64(asm volatile("v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2\n v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2\n v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2\n v_fma_f32 %0, %0, %2, %2\n v_fma_f32 %1, %1, %2, %2" : "+v"(a0), "+v"(a1) : "v"(k));) }
    // SGPR-pair / SGPR operands and op_sel modifiers (the FIR's tap operand): dependent chains
    if (KIND == 10) { REP64(asm volatile("v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p0) : "s"(sk), "v"(kk));) }
    if (KIND == 11) { REP64(asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(p0) : "s"(sk), "v"(kk));) }
    if (KIND == 12) { REP64(asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(p0) : "v"(kk), "v"(p1));) }
    // the same tap as four v_fma_f32 (two independent chains re / im), taps from SGPRs
    if (KIND == 13) { REP64(asm volatile("v_fma_f32 %0, %2, %4, %0\n v_fma_f32 %1, %2, %5, %1\n v_fma_f32 %0, -%3, %5, %0\n v_fma_f32 %1, %3, %4, %1\n v_fma_f32 %0, %2, %4, %0\n v_fma_f32 %1, %2, %5, %1\n v_fma_f32 %0, -%3, %5, %0\n v_fma_f32 %1, %3, %4, %1" : "+v"(a0), "+v"(a1) : "s"(sr), "s"(si), "v"(a2), "v"(a3));) }
    // as KIND 11 but every instruction reads ANOTHER SGPR pair (the FIR reads a fresh tap per instruction pair)
    if (KIND == 14) { REP64(asm volatile("v_pk_fma_f32 %0, %1, %9, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %0, %1, %9, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %0, %2, %9, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %0, %2, %9, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %0, %3, %9, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %0, %3, %9, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %0, %4, %9, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %0, %4, %9, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(p0) : "s"(sk), "s"(sk2), "s"(sk3), "s"(sk4), "s"(sk), "s"(sk), "s"(sk), "s"(sk), "v"(kk));) }
    // ... and another VGPR sample pair as well
    if (KIND == 15) { REP64(asm volatile("v_pk_fma_f32 %0, %1, %5, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %0, %1, %5, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %0, %2, %6, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %0, %2, %6, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %0, %3, %7, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %0, %3, %7, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n v_pk_fma_f32 %0, %4, %8, %0 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %0, %4, %8, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(p0) : "s"(sk), "s"(sk2), "s"(sk3), "s"(sk4), "v"(p1), "v"(p2), "v"(p3), "v"(p4));) }
    // VALU + SALU mix: does an s_mov take a VALU issue slot of the lone wave?
    if (KIND == 8) { REP64(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n s_mov_b32 s20, 0x3f800000\n v_pk_fma_f32 %1, %1, %8, %8\n s_mov_b32 s21, 0x3f800000\n v_pk_fma_f32 %2, %2, %8, %8\n s_mov_b32 s20, 0x3f800000\n v_pk_fma_f32 %3, %3, %8, %8\n s_mov_b32 s21, 0x3f800000\n v_pk_fma_f32 %4, %4, %8, %8\n s_mov_b32 s20, 0x3f800000\n v_pk_fma_f32 %5, %5, %8, %8\n s_mov_b32 s21, 0x3f800000\n v_pk_fma_f32 %6, %6, %8, %8\n s_mov_b32 s20, 0x3f800000\n v_pk_fma_f32 %7, %7, %8, %8\n s_mov_b32 s21, 0x3f800000" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(kk) : "s20", "s21");) }
    // v_accvgpr_read / write pairs
    if (KIND == 9) { REP64(asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %1\n v_accvgpr_write_b32 a2, %2\n v_accvgpr_write_b32 a3, %3\n v_accvgpr_read_b32 %4, a4\n v_accvgpr_read_b32 %5, a5\n v_accvgpr_read_b32 %6, a6\n v_accvgpr_read_b32 %7, a7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");) }
  }
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.x + p5.x + p6.x + p7.x + p0.y;
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = (unsigned long long)(s != 12345.f); }
}
template <int KIND> void run(const char *name, int blocks)
{
  unsigned long long *d; hipMalloc(&d, blocks * 16);
  hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(64), 0, 0, d, 1.0f);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(64), 0, 0, d, 1.0f);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 2); hipMemcpy(h.data(), d, blocks * 16, hipMemcpyDeviceToHost);
  double acc = 0; for (int b = 0; b < blocks; ++b) acc += (double)h[2 * b];
  const double n = 16.0 * 64 * 8 * (KIND == 8 ? 1 : 1);
  printf("%-44s blocks %5d: %.2f ticks per VALU instruction; kernel %.1f us for %.0f ticks per wave -> %.0f MHz tick\n", name, blocks, acc / blocks / n, ms * 1e3, acc / blocks, acc / blocks / (ms * 1e3));
  hipFree(d);
}
int main()
{
  for (int blocks : {1024, 2048, 4096}) {
    run<0>("v_fma_f32 independent x8", blocks);
    run<1>("v_fma_f32 dependent", blocks);
    run<7>("v_fma_f32 two chains", blocks);
    run<2>("v_pk_fma_f32 independent x8", blocks);
    run<3>("v_pk_fma_f32 dependent", blocks);
    run<6>("v_pk_fma_f32 two chains", blocks);
    run<4>("v_pk_add_f32 independent x8", blocks);
    run<5>("v_add_f32 independent x8", blocks);
    run<8>("v_pk_fma_f32 indep + s_mov_b32 each", blocks);
    run<10>("v_pk_fma_f32 dependent, SGPR-pair src0", blocks);
    run<11>("v_pk_fma_f32 dep, SGPR pair + op_sel/neg", blocks);
    run<12>("v_pk_fma_f32 dep, VGPRs + op_sel/neg", blocks);
    run<13>("4 x v_fma_f32 per tap (SGPR taps, 2 chains)", blocks);
    run<14>("v_pk_fma_f32 dep, 4 SGPR pairs in turn", blocks);
    run<15>("v_pk_fma_f32 dep, 4 SGPR pairs + 4 VGPR pairs", blocks);
    run<9>("v_accvgpr_write x4 + read x4", blocks);
  }
  return 0;
}
