// readpat.hip -- HBM read rate of a PSD-like workgroup loop under different request shapes.
// Each workgroup (256 threads) walks `nf` consecutive 64 KiB frames; per frame a thread requests 256 B and then
// "computes" for `work` dependent FMAs (so requests come in bursts, as in the PSD kernel).
//   mode 0: 32 x b64 per thread, wave request = 512 B, stride 2 KiB between a thread's requests (the PSD pass-0 shape)
//   mode 1: 16 x b128 per thread, wave request = 1 KiB, stride 4 KiB
//   mode 2: 32 x b64, each wave reads its own contiguous 16 KiB quarter of the frame (512 B requests back to back)
//   mode 3: 16 x b128, contiguous 16 KiB per wave
// build: hipcc --offload-arch=gfx950 -O3 -o readpat readpat.hip ; run: ./readpat
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float __attribute__((ext_vector_type(2))) f2;
typedef float __attribute__((ext_vector_type(4))) f4;

template <int MODE>
__global__ __launch_bounds__(256, 2) void rd(const char *x, int nf, int work, float *out)
{
  const int t = threadIdx.x, w = t >> 6, l = t & 63;
  const char *base = x + (size_t)blockIdx.x * nf * 65536;
  float acc = 0.f;
  f2 a[32];
  auto req = [&](int f) {
    const char *fr = base + (size_t)f * 65536;
    if (MODE == 0) {
#pragma unroll
      for (int q = 0; q < 32; ++q) a[q] = *(const f2 *)(fr + t * 8 + q * 2048);
    } else if (MODE == 1) {
#pragma unroll
      for (int q = 0; q < 16; ++q) { f4 v = *(const f4 *)(fr + t * 16 + q * 4096); a[2 * q] = v.xy; a[2 * q + 1] = v.zw; }
    } else if (MODE == 2) {
#pragma unroll
      for (int q = 0; q < 32; ++q) a[q] = *(const f2 *)(fr + w * 16384 + l * 8 + q * 512);
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) { f4 v = *(const f4 *)(fr + w * 16384 + l * 16 + q * 1024); a[2 * q] = v.xy; a[2 * q + 1] = v.zw; }
    }
  };
  req(0);
  for (int f = 0; f < nf; ++f) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) s += a[q].x * a[q].y;
    if (f + 1 < nf) req(f + 1);
    for (int i = 0; i < work; ++i) s = __builtin_fmaf(s, 1.0001f, 0.5f);
    acc += s;
  }
  out[blockIdx.x * 256 + t] = acc;
}

int main(int argc, char **argv)
{
  const size_t L = (size_t)1 << 31;   // 2 GiB
  char *x; float *out;
  hipMalloc(&x, L); hipMemset(x, 0, L); hipMalloc(&out, (size_t)1 << 26);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int work : {0, 1000, 3000}) for (int nwg : {512, 2048, 32768}) {
    const int nf = (int)(L / 65536 / nwg);
    for (int mode = 0; mode < 4; ++mode) {
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        switch (mode) {
          case 0: rd<0><<<nwg, 256>>>(x, nf, work, out); break;
          case 1: rd<1><<<nwg, 256>>>(x, nf, work, out); break;
          case 2: rd<2><<<nwg, 256>>>(x, nf, work, out); break;
          default: rd<3><<<nwg, 256>>>(x, nf, work, out); break;
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      printf("work %4d wgs %5d frames/wg %3d mode %d: %7.1f us  %6.0f GB/s\n", work, nwg, nf, mode, ms * 1e3, L / ms / 1e6);
    }
  }
  return 0;
}
