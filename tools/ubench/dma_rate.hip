// LDS-DMA issue rate of ONE wavefront per CU: bytes per tick and GB/s for K requests (1 KiB each) per vmcnt(0) round.
//   mode 0: all 64 lanes, contiguous 16 B per lane;  mode 1: lanes l % 9 == 8 masked off (the FIR ring's pad chunks), source
//   addresses contiguous over the active lanes;  mode 2: as 0 with s_setprio 3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
template <int K, int MODE>
__global__ __launch_bounds__(64) void dma(const char *src, size_t bytes_per_wg, int rounds, unsigned long long *ticks)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (MODE == 2) __builtin_amdgcn_s_setprio(3);
  const int l = threadIdx.x;
  const char *base = src + (size_t)blockIdx.x * bytes_per_wg;
  const int slot = MODE == 1 ? (l / 9) * 8 + (l % 9) : l;
  const bool act = MODE == 1 ? (l % 9) != 8 && l < 63 : true;
  const size_t step = MODE == 1 ? 56 * 16 : 1024;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  size_t off = 0;
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const char *g = base + off + (size_t)slot * 16;
      const unsigned lds0 = (unsigned)(uintptr_t)(smem + (k % 32) * 1024);
      if (act) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(lds0) : "memory", "m0");
      off += step;
      if (off + 1024 > bytes_per_wg) off = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (l == 0) ticks[blockIdx.x] = t1 - t0;
  if (smem[l * 16] == 123 && t1 == 7) ticks[0] = 0;
}
template <int K, int MODE> void run(const char *src, size_t per_wg, unsigned long long *d)
{
  const int wgs = 256, rounds = 4096 / K;
  hipFuncSetAttribute((const void *)dma<K, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((dma<K, MODE>), dim3(wgs), dim3(64), 32768, 0, src, per_wg, rounds, d);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((dma<K, MODE>), dim3(wgs), dim3(64), 32768, 0, src, per_wg, rounds, d);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(wgs); hipMemcpy(h.data(), d, wgs * 8, hipMemcpyDeviceToHost);
  double t = 0; for (auto v : h) t += v; t /= wgs;
  const double inst = (double)rounds * K, bytes = inst * (MODE == 1 ? 56 * 16 : 1024);
  printf("K %2d mode %d: %7.1f ticks per request, %5.2f B/tick per wave; kernel %.1f us -> %.0f GB/s chip (%d single-wave WGs)\n",
         K, MODE, t / inst, bytes / t, ms * 1e3, bytes * wgs / (ms * 1e-3) / 1e9, wgs);
}
int main()
{
  const size_t per_wg = 4 << 20;
  char *src; hipMalloc(&src, per_wg * 256); hipMemset(src, 1, per_wg * 256);
  unsigned long long *d; hipMalloc(&d, 256 * 8);
  run<4, 0>(src, per_wg, d); run<8, 0>(src, per_wg, d); run<16, 0>(src, per_wg, d); run<32, 0>(src, per_wg, d);
  run<16, 1>(src, per_wg, d); run<32, 1>(src, per_wg, d); run<16, 2>(src, per_wg, d);
  return 0;
}
