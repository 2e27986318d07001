// LDS store rate of a 512-thread workgroup per CU: 16 x ds_write_b128 per thread (128 KiB per round), three address patterns.
//   0: linear (tid * 16 + i * 8192);  1: the FIR pair kernel's staging pattern (16 chunks per 272-byte pair-block);  2: as 1 with b64 pairs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long *ticks, float4 seed, int rounds)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  float4 v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = float4{seed.x + i, seed.y, seed.z + tid, seed.w};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = i * 512 + tid;
      const int off = MODE == 0 ? c * 16 : (c >> 4) * 272 + (c & 15) * 16;
      if (MODE == 2) {
        *reinterpret_cast<float2 *>(smem + off) = float2{v[i].x, v[i].y};
        *reinterpret_cast<float2 *>(smem + off + 8) = float2{v[i].z, v[i].w};
      } else *reinterpret_cast<float4 *>(smem + off) = v[i];
    }
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) ticks[blockIdx.x] = t1 - t0;
  if (smem[tid] == 77 && seed.x == 123.f) ticks[0] = 0;
}
template <int MODE> void run(unsigned long long *d)
{
  const int rounds = 64;
  hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 150 * 1024, 0, d, float4{1, 2, 3, 4}, rounds);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256); hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost);
  double t = 0; for (auto v : h) t += v; t /= 256;
  printf("mode %d: %.0f ticks per 128 KiB round = %.1f B/tick per CU\n", MODE, t / rounds, 131072.0 * rounds / t);
}
int main() { unsigned long long *d; hipMalloc(&d, 256 * 8); run<0>(d); run<1>(d); run<2>(d); return 0; }
