// Probe: (1) LDS-DMA (global_load_lds_dwordx4) into LDS offsets beyond 64 KiB (M0 width), builtin vs asm;
//        (2) v_pk_fma_f32 with an SGPR-pair tap (re, im) and op_sel / neg modifiers == the SPEC's two packed fmas.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(64) void dma_probe(const float4 *src, float4 *dst, int lds_off, int use_asm)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int l = threadIdx.x;
  const float4 *g = src + l;
  if (use_asm) {
    unsigned keep;
    const unsigned ldsaddr = (unsigned)(uintptr_t)(smem + lds_off);   // LDS byte address (low 32 bits of the generic pointer offset?)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(__builtin_amdgcn_readfirstlane(ldsaddr)) : "memory");
  } else {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)(smem + lds_off), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  dst[l] = *reinterpret_cast<const float4 *>(smem + lds_off + 16 * l);
}

__global__ __launch_bounds__(64) void fma_probe(const float2 *taps, const float2 *x, float2 *out, int n)
{
  const int l = threadIdx.x;
  v2f acc_ref = {0.f, 0.f}, acc_asm = {0.f, 0.f};
  for (int k = 0; k < n; ++k) {
    const float2 t = taps[k];                       // uniform: scalar load
    const float2 xx = x[k * 64 + l];
    const v2f xv = {xx.x, xx.y};
    // SPEC: acc = fma((re, re), (x.re, x.im), acc); acc = fma((-im, im), (x.im, x.re), acc)
    acc_ref = __builtin_elementwise_fma(v2f{t.x, t.x}, xv, acc_ref);
    acc_ref = __builtin_elementwise_fma(v2f{-t.y, t.y}, xv.yx, acc_ref);
    const v2f tv = {t.x, t.y};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                 "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
                 : "+v"(acc_asm) : "s"(tv), "v"(xv));
  }
  out[l] = float2{acc_ref.x, acc_ref.y};
  out[64 + l] = float2{acc_asm.x, acc_asm.y};
}

int main()
{
  float4 *src, *dst;
  hipMalloc(&src, 64 * 16); hipMalloc(&dst, 64 * 16);
  float h[256], r[256];
  for (int i = 0; i < 256; ++i) h[i] = (float)i + 0.5f;
  hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void *)dma_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int offs[] = {0, 32768, 65536 - 1024, 65536, 100000 / 16 * 16, 150 * 1024, 159 * 1024};
  for (int ua = 0; ua < 2; ++ua)
    for (int off : offs) {
      hipMemset(dst, 0, 1024);
      hipLaunchKernelGGL(dma_probe, dim3(1), dim3(64), 160 * 1024, 0, src, dst, off, ua);
      hipError_t e = hipDeviceSynchronize();
      hipMemcpy(r, dst, sizeof r, hipMemcpyDeviceToHost);
      int bad = 0;
      for (int i = 0; i < 256; ++i) bad += r[i] != h[i];
      printf("%s lds_off %6d: %s (%d bad) %s\n", ua ? "asm    " : "builtin", off, bad ? "WRONG" : "ok", bad, hipGetErrorString(e));
    }
  // fma probe
  const int n = 255;
  float2 *taps, *x, *out;
  hipMalloc(&taps, n * 8); hipMalloc(&x, n * 64 * 8); hipMalloc(&out, 128 * 8);
  float *ht = (float *)malloc(n * 8), *hx = (float *)malloc(n * 64 * 8), ho[256];
  srand(1);
  for (int i = 0; i < 2 * n; ++i) ht[i] = (float)rand() / RAND_MAX - 0.5f;
  for (int i = 0; i < 2 * n * 64; ++i) hx[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(taps, ht, n * 8, hipMemcpyHostToDevice); hipMemcpy(x, hx, n * 64 * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(fma_probe, dim3(1), dim3(64), 0, 0, taps, x, out, n);
  hipDeviceSynchronize();
  hipMemcpy(ho, out, sizeof ho, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 128; ++i) bad += ho[i] != ho[128 + i];
  printf("pk_fma with SGPR tap + op_sel/neg: %s (%d of 128 differ), sample %g %g\n", bad ? "DIFFERENT" : "bit-identical", bad, ho[0], ho[128]);
  return 0;
}
