// stages.hip -- the inspector stages behind the rest of the config vocabulary (SURVEY.md row A7,
// Default/GenericInspector/InspectorCtl/*.cpp): fixed gain (agc.enabled = false), manual carrier
// offset (afc.costas-order = 0, afc.offset), matched filter (mf.type = MANUAL, mf.roll-off) and
// the CMA equalizer (equalizer.type = CMA).  Rows are addressed through sdk::View like the loops.
//
// SPEC.md section I: every output keeps its own operation order (bit-exact across implementations).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.hpp"
#include "sd_math.hpp"

namespace {

using sd::c32;

inline unsigned grid_for(long long n, int block)
{
  long long g = (n + block - 1) / block;
  return (unsigned)(g < 1 ? 1 : (g > 65535 * 16 ? 65535 * 16 : g));
}

// element t of the (channel, time) index space laid out so that consecutive lanes touch consecutive
// memory: time-major views (cs == 1) run lanes over channels, channel-major ones over time
__device__ __forceinline__ void split_index(long long t, int nchan, long long len, bool lanes_are_channels,
                                            int &c, long long &m)
{
  if (lanes_are_channels) { c = (int)(t % nchan); m = t / nchan; }
  else                    { m = t % len;          c = (int)(t / len); }
}

// agc.enabled = false: y = g x   (InspectorCtl/GainControl.cpp:51-60)
__global__ void rows_scale_kernel(const float2 *__restrict__ x, sdk::View xv, float2 *__restrict__ y, sdk::View yv,
                                  int nchan, long long len, float g)
{
  __builtin_amdgcn_s_setprio(3);
  const long long total = len * nchan;
  const bool lac = xv.cs == 1;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    int c; long long m;
    split_index(t, nchan, len, lac, c, m);
    const float2 v = x[c * xv.cs + m * xv.ms];
    y[c * yv.cs + m * yv.ms] = float2{g * v.x, g * v.y};
  }
}

// afc.offset: y_c[m] = x_c[m] * phasor(p0_c + (n0 + m) * dp_c)   (a free-running su_ncqo per channel)
__global__ void rows_xlate_kernel(const float2 *__restrict__ x, sdk::View xv, float2 *__restrict__ y, sdk::View yv,
                                  int nchan, long long len, const uint32_t *__restrict__ dphase,
                                  const uint32_t *__restrict__ phase0, uint64_t n0)
{
  __builtin_amdgcn_s_setprio(3);
  const long long total = len * nchan;
  const bool lac = xv.cs == 1;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    int c; long long m;
    split_index(t, nchan, len, lac, c, m);
    const uint32_t p = phase0[c] + (uint32_t)((n0 + (uint64_t)m) * (uint64_t)dphase[c]);
    float cs, sn;
    sd::phasor_u32(p, cs, sn);
    const float2 v = x[c * xv.cs + m * xv.ms];
    const c32 r = sd::cmul_cs(c32{v.x, v.y}, cs, sn);
    y[c * yv.cs + m * yv.ms] = float2{r.re, r.im};
  }
}

// inspector spectrum sources (section 8f #2): the per-sample transform in front of the inspector PSD
__device__ __forceinline__ c32 csq(c32 a) { return c32{sd::fma_(-a.im, a.im, a.re * a.re), sd::fma_(a.im, a.re, a.re * a.im)}; }

__global__ void spectsrc_kernel(int kind, const float2 *__restrict__ x, long long len, float2 prev0, float2 *__restrict__ y)
{
  __builtin_amdgcn_s_setprio(3);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < len; i += (long long)gridDim.x * blockDim.x) {
    const float2 vv = x[i], pp = i > 0 ? x[i - 1] : prev0;
    const c32 v = {vv.x, vv.y}, prev = {pp.x, pp.y};
    c32 r = {0.0f, 0.0f}, d;
    switch (kind) {
      case 1: r = v; break;
      case 2: r = sd::cmul_conj(v, prev); break;
      case 3: d = sd::cmul_conj(v, prev); r.re = sd::atan2_(d.im, d.re); break;
      case 4: r.re = sd::atan2_(v.im, v.re); break;
      case 5: r = c32{v.re - prev.re, v.im - prev.im}; break;
      case 6: d = c32{v.re - prev.re, v.im - prev.im}; r.re = __builtin_sqrtf(sd::fma_(d.im, d.im, d.re * d.re)); break;
      case 7: r = csq(v); break;
      case 8: r = csq(csq(v)); break;
      case 9: r = csq(csq(csq(v))); break;
      default: break;
    }
    y[i] = float2{r.re, r.im};
  }
}

// mf.type = MANUAL: y_c[m] = sum_k h[k] x_c[m-k], k ascending (one fma chain per component).
// A thread owns R consecutive outputs of one channel and walks the samples they need from the
// newest to the oldest, so every sample is loaded once and each output sees its taps in ascending
// order.  Samples before the block come from hist ([T-1][nchan], time-major).
constexpr int FIR_R = 4;
__global__ void rows_fir_kernel(const float2 *__restrict__ x, sdk::View xv, float2 *__restrict__ y, sdk::View yv,
                                int nchan, long long len, const float *__restrict__ h, int ntaps,
                                const float2 *__restrict__ hist)
{
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ float lh[];
  for (int i = threadIdx.x; i < ntaps; i += blockDim.x) lh[i] = h[i];
  __syncthreads();
  const long long ngroups = (len + FIR_R - 1) / FIR_R;
  const long long total = ngroups * nchan;
  const bool lac = xv.cs == 1;
  const int hl = ntaps - 1;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    int c; long long g;
    split_index(t, nchan, ngroups, lac, c, g);
    const long long m0 = g * FIR_R;
    float yr[FIR_R], yi[FIR_R];
#pragma unroll
    for (int r = 0; r < FIR_R; ++r) { yr[r] = 0.0f; yi[r] = 0.0f; }
    // sample index s relative to the block start, newest first
    for (long long s = m0 + FIR_R - 1; s > m0 - ntaps; --s) {
      float2 v;
      if (s >= len) continue;                               // beyond the block (ragged last group)
      if (s >= 0) v = x[c * xv.cs + s * xv.ms];
      else        v = hist[(long long)(hl + s) * nchan + c];
#pragma unroll
      for (int r = 0; r < FIR_R; ++r) {
        const long long k = m0 + r - s;                     // tap index of this sample for output r
        if (k >= 0 && k < ntaps) {
          const float hk = lh[k];
          yr[r] = sd::fma_(hk, v.x, yr[r]);
          yi[r] = sd::fma_(hk, v.y, yi[r]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < FIR_R; ++r)
      if (m0 + r < len) y[c * yv.cs + (m0 + r) * yv.ms] = float2{yr[r], yi[r]};
  }
}

// history for the next block = last T-1 samples of [hist ; x], per channel (ping-pong buffers)
__global__ void rows_hist_kernel(float2 *__restrict__ hist_next, const float2 *__restrict__ hist,
                                 const float2 *__restrict__ x, sdk::View xv, int nchan, long long len, int hl)
{
  const long long total = (long long)hl * nchan;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(t % nchan);
    const long long k = t / nchan;
    const long long src = k + len;                          // index into [hist ; x]
    hist_next[k * nchan + c] = src < hl ? hist[src * nchan + c] : x[c * xv.cs + (src - hl) * xv.ms];
  }
}

// equalizer.type = CMA: one lane per channel over that channel's own symbol count; N weights and the
// delay line live in registers.  y = sum w[i] d[i];  w[i] -= mu (|y|^2 - 1) y conj(d[i]).
template <int N>
__global__ __launch_bounds__(64) void cma_kernel(float mu, int locked, float2 *__restrict__ w, float2 *__restrict__ dl,
                                                 int nchan, const float2 *__restrict__ x, long long x_stride,
                                                 const uint32_t *__restrict__ count, long long fixed_len,
                                                 float2 *__restrict__ y, long long y_stride)
{
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= nchan) return;
  c32 wr[N], d[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float2 a = w[(long long)i * nchan + c], b = dl[(long long)i * nchan + c];
    wr[i] = c32{a.x, a.y}; d[i] = c32{b.x, b.y};
  }
  const long long n = count ? (long long)count[c] : fixed_len;
  const float2 *xr = x + (long long)c * x_stride;
  float2 *yr_ = y + (long long)c * y_stride;
  for (long long m = 0; m < n; ++m) {
    const float2 v = xr[m];
#pragma unroll
    for (int i = N - 1; i > 0; --i) d[i] = d[i - 1];
    d[0] = c32{v.x, v.y};
    float yr = 0.0f, yi = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      yr = sd::fma_(wr[i].re, d[i].re, yr); yr = sd::fma_(-wr[i].im, d[i].im, yr);
      yi = sd::fma_(wr[i].re, d[i].im, yi); yi = sd::fma_(wr[i].im, d[i].re, yi);
    }
    if (!locked) {
      const float g = sd::fma_(yi, yi, yr * yr) - 1.0f;
      const c32 e = {yr * g, yi * g};
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const c32 t = sd::cmul_conj(e, d[i]);
        wr[i].re = sd::fma_(-mu, t.re, wr[i].re);
        wr[i].im = sd::fma_(-mu, t.im, wr[i].im);
      }
    }
    yr_[m] = float2{yr, yi};
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    w[(long long)i * nchan + c] = float2{wr[i].re, wr[i].im};
    dl[(long long)i * nchan + c] = float2{d[i].re, d[i].im};
  }
}

template <int N>
hipError_t launch_cma(float mu, int locked, void *w, void *dl, int nchan, const void *x, long long xs, const uint32_t *count,
                      long long fixed_len, void *y, long long ys, hipStream_t st)
{
  hipLaunchKernelGGL(cma_kernel<N>, dim3((nchan + 63) / 64), dim3(64), 0, st, mu, locked, reinterpret_cast<float2 *>(w),
                     reinterpret_cast<float2 *>(dl), nchan, reinterpret_cast<const float2 *>(x), xs, count, fixed_len,
                     reinterpret_cast<float2 *>(y), ys);
  return hipGetLastError();
}

}  // namespace

namespace sdk {

hipError_t rows_scale(const void *x, View xv, void *y, View yv, int nchan, long long len, float g, hipStream_t st)
{
  if (len <= 0 || nchan <= 0) return hipSuccess;
  hipLaunchKernelGGL(rows_scale_kernel, dim3(grid_for(len * nchan, 256)), dim3(256), 0, st,
                     reinterpret_cast<const float2 *>(x), xv, reinterpret_cast<float2 *>(y), yv, nchan, len, g);
  return hipGetLastError();
}

hipError_t rows_xlate(const void *x, View xv, void *y, View yv, int nchan, long long len, const uint32_t *dphase,
                      const uint32_t *phase0, uint64_t n0, hipStream_t st)
{
  if (len <= 0 || nchan <= 0) return hipSuccess;
  hipLaunchKernelGGL(rows_xlate_kernel, dim3(grid_for(len * nchan, 256)), dim3(256), 0, st,
                     reinterpret_cast<const float2 *>(x), xv, reinterpret_cast<float2 *>(y), yv, nchan, len, dphase, phase0, n0);
  return hipGetLastError();
}

hipError_t spectsrc_preproc(int kind, const void *x, long long len, float prev_re, float prev_im, void *y, hipStream_t st)
{
  if (len <= 0) return hipSuccess;
  hipLaunchKernelGGL(spectsrc_kernel, dim3(grid_for(len, 256)), dim3(256), 0, st, kind, reinterpret_cast<const float2 *>(x), len,
                     float2{prev_re, prev_im}, reinterpret_cast<float2 *>(y));
  return hipGetLastError();
}

hipError_t rows_fir(const void *x, View xv, void *y, View yv, int nchan, long long len, const float *h, int ntaps,
                    const void *hist, void *hist_next, hipStream_t st)
{
  if (nchan <= 0 || len <= 0) return hipSuccess;
  const long long total = ((len + FIR_R - 1) / FIR_R) * nchan;
  hipLaunchKernelGGL(rows_fir_kernel, dim3(grid_for(total, 256)), dim3(256), sizeof(float) * (size_t)ntaps, st,
                     reinterpret_cast<const float2 *>(x), xv, reinterpret_cast<float2 *>(y), yv, nchan, len, h, ntaps,
                     reinterpret_cast<const float2 *>(hist));
  if (ntaps > 1)
    hipLaunchKernelGGL(rows_hist_kernel, dim3(grid_for((long long)(ntaps - 1) * nchan, 256)), dim3(256), 0, st,
                       reinterpret_cast<float2 *>(hist_next), reinterpret_cast<const float2 *>(hist),
                       reinterpret_cast<const float2 *>(x), xv, nchan, len, ntaps - 1);
  return hipGetLastError();
}

hipError_t cma_feed(int n, float mu, int locked, void *w, void *dl, int nchan, const void *x, long long x_stride,
                    const uint32_t *count, long long fixed_len, void *y, long long y_stride, hipStream_t st)
{
  switch (n) {
#define CMA_CASE(N) case N: return launch_cma<N>(mu, locked, w, dl, nchan, x, x_stride, count, fixed_len, y, y_stride, st);
    CMA_CASE(1) CMA_CASE(2) CMA_CASE(3) CMA_CASE(4) CMA_CASE(5) CMA_CASE(6) CMA_CASE(7) CMA_CASE(8)
    CMA_CASE(9) CMA_CASE(10) CMA_CASE(11) CMA_CASE(12) CMA_CASE(13) CMA_CASE(14) CMA_CASE(15) CMA_CASE(16)
#undef CMA_CASE
    default: return hipErrorInvalidValue;
  }
}

}  // namespace sdk
