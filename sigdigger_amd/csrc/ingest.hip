// ingest.hip -- sample-format converters in front of the path:
// the raw file formats of the file source (Default/SourceConfig/FileSourcePage.cpp:80-104 --
// RAW_UNSIGNED8, RAW_SIGNED8, RAW_SIGNED16, and the PCM payloads of WAV / SigMF) -> SUCOMPLEX.
// The host ships 2 or 4 bytes per sample over PCIe instead of 8 and the GPU expands them:
//   u8 : (v - 128) * 2^-7     s8 : v * 2^-7     s16 : v * 2^-15        (libsndfile's normalisation,
// which is what suscan's file source reads through; exact in binary32, so bit-identical everywhere).
// HBM-bound streaming: 16 bytes in per thread, 16-byte stores out.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.hpp"

namespace {

inline unsigned grid_for(long long n, int block)
{
  long long g = (n + block - 1) / block;
  return (unsigned)(g < 1 ? 1 : (g > 65535 * 16 ? 65535 * 16 : g));
}

template <bool IS_SIGNED> __device__ __forceinline__ float cvt8(uint32_t b)
{
  return IS_SIGNED ? (float)(int)(int8_t)b * 0.0078125f : (float)((int)b - 128) * 0.0078125f;
}

// 16 bytes = 8 complex samples per thread
template <bool IS_SIGNED>
__global__ void ingest8_kernel(const uint8_t *__restrict__ raw, float2 *__restrict__ out, long long nsamp)
{
  const long long nvec = nsamp / 8;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < nvec;
       t += (long long)gridDim.x * blockDim.x) {
    const uint4 v = reinterpret_cast<const uint4 *>(raw)[t];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    float4 *o = reinterpret_cast<float4 *>(out + t * 8);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      o[k] = float4{cvt8<IS_SIGNED>(w[k] & 0xff), cvt8<IS_SIGNED>((w[k] >> 8) & 0xff),
                    cvt8<IS_SIGNED>((w[k] >> 16) & 0xff), cvt8<IS_SIGNED>(w[k] >> 24)};
  }
  // ragged tail (< 8 samples): one thread each
  const long long t = nvec * 8 + blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (t < nsamp) out[t] = float2{cvt8<IS_SIGNED>(raw[2 * t]), cvt8<IS_SIGNED>(raw[2 * t + 1])};
}

// 16 bytes = 4 complex samples per thread
__global__ void ingest16_kernel(const int16_t *__restrict__ raw, float2 *__restrict__ out, long long nsamp)
{
  const long long nvec = nsamp / 4;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < nvec;
       t += (long long)gridDim.x * blockDim.x) {
    const uint4 v = reinterpret_cast<const uint4 *>(raw)[t];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    float4 *o = reinterpret_cast<float4 *>(out + t * 4);
    const float k = 3.0517578125e-05f;
    o[0] = float4{(float)(int16_t)(w[0] & 0xffff) * k, (float)(int16_t)(w[0] >> 16) * k,
                  (float)(int16_t)(w[1] & 0xffff) * k, (float)(int16_t)(w[1] >> 16) * k};
    o[1] = float4{(float)(int16_t)(w[2] & 0xffff) * k, (float)(int16_t)(w[2] >> 16) * k,
                  (float)(int16_t)(w[3] & 0xffff) * k, (float)(int16_t)(w[3] >> 16) * k};
  }
  const long long t = nvec * 4 + blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (t < nsamp) out[t] = float2{(float)raw[2 * t] * 3.0517578125e-05f, (float)raw[2 * t + 1] * 3.0517578125e-05f};
}

// ---- source conditioning (Suscan::Analyzer::setIQReverse / setDCRemove, Suscan/Analyzer.cpp:240-256) ----
// block means in a fixed order (256 workgroups, each a strided slice; lanes tree-reduced): the same bits every run
__global__ __launch_bounds__(256) void block_sum_kernel(const float2 *__restrict__ x, long long nsamp, float *__restrict__ partial)
{
  __shared__ float sr[256], si[256];
  float ar = 0.0f, ai = 0.0f;
  for (long long t = blockIdx.x * 256ll + threadIdx.x; t < nsamp; t += 256ll * gridDim.x) { const float2 v = x[t]; ar += v.x; ai += v.y; }
  sr[threadIdx.x] = ar; si[threadIdx.x] = ai;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { sr[threadIdx.x] += sr[threadIdx.x + o]; si[threadIdx.x] += si[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = sr[0]; partial[2 * blockIdx.x + 1] = si[0]; }
}

__global__ void dc_update_kernel(const float *__restrict__ partial, int nparts, long long nsamp, int iq_reverse, float alpha, int first,
                                 float *__restrict__ dc)
{
  double sr = 0.0, si = 0.0;
  for (int i = 0; i < nparts; ++i) { sr += (double)partial[2 * i]; si += (double)partial[2 * i + 1]; }
  float mr = (float)(sr / (double)nsamp), mi = (float)(si / (double)nsamp);
  if (iq_reverse) { const float t = mr; mr = mi; mi = t; }
  if (first) { dc[0] = mr; dc[1] = mi; }
  else { dc[0] = dc[0] + alpha * (mr - dc[0]); dc[1] = dc[1] + alpha * (mi - dc[1]); }
}

__global__ void source_fix_kernel(float2 *__restrict__ x, long long nsamp, int iq_reverse, const float *__restrict__ dc)
{
  const float dr = dc ? dc[0] : 0.0f, di = dc ? dc[1] : 0.0f;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < nsamp; t += (long long)gridDim.x * blockDim.x) {
    float2 v = x[t];
    if (iq_reverse) v = float2{v.y, v.x};
    if (dc) { v.x = v.x - dr; v.y = v.y - di; }
    x[t] = v;
  }
}

}  // namespace

namespace sdk {

hipError_t source_fix(void *x, long long nsamp, int iq_reverse, float *dc, float alpha, int first, float *partial, hipStream_t st)
{
  if (nsamp <= 0 || (!iq_reverse && !dc)) return hipSuccess;
  float2 *xx = static_cast<float2 *>(x);
  if (dc) {
    hipLaunchKernelGGL(block_sum_kernel, dim3(256), dim3(256), 0, st, xx, nsamp, partial);
    hipLaunchKernelGGL(dc_update_kernel, dim3(1), dim3(1), 0, st, partial, 256, nsamp, iq_reverse, alpha, first, dc);
  }
  long long g = (nsamp + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(source_fix_kernel, dim3((unsigned)g), dim3(256), 0, st, xx, nsamp, iq_reverse, dc);
  return hipGetLastError();
}


hipError_t ingest_iq(int format, const void *raw, long long nsamp, void *out, hipStream_t st)
{
  if (nsamp <= 0) return hipSuccess;
  float2 *o = reinterpret_cast<float2 *>(out);
  switch (format) {
    case 1:   // float32: already SUCOMPLEX
      return raw == out ? hipSuccess : hipMemcpyAsync(out, raw, (size_t)nsamp * 8, hipMemcpyDeviceToDevice, st);
    case 2:
      hipLaunchKernelGGL(ingest8_kernel<false>, dim3(grid_for(nsamp / 8 + 1, 256)), dim3(256), 0, st,
                         reinterpret_cast<const uint8_t *>(raw), o, nsamp);
      break;
    case 3:
      hipLaunchKernelGGL(ingest8_kernel<true>, dim3(grid_for(nsamp / 8 + 1, 256)), dim3(256), 0, st,
                         reinterpret_cast<const uint8_t *>(raw), o, nsamp);
      break;
    case 4:
      hipLaunchKernelGGL(ingest16_kernel, dim3(grid_for(nsamp / 4 + 1, 256)), dim3(256), 0, st,
                         reinterpret_cast<const int16_t *>(raw), o, nsamp);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace sdk
