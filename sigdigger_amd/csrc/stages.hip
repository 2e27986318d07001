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

__global__ void spectsrc_kernel(int kind, const float2 *__restrict__ x, long long len, float2 prev0, const float2 *__restrict__ prev_dev,
                                float2 *__restrict__ y)
{
  __builtin_amdgcn_s_setprio(3);
  if (prev_dev) prev0 = *prev_dev;                           // the sample before the block, still on the device
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

// ---- section 8f #3: decision space, decider, symbol histogram, SNR estimator ------------------------------
__device__ __forceinline__ float dec_value(float2 x, int mode)
{
  return mode == 0 ? __builtin_sqrtf(sd::fma_(x.y, x.y, x.x * x.x)) : sd::atan2_(x.y, x.x);
}

// InspectorUI::feed, decision-space forwarding (Default/GenericInspector/InspectorUI.cpp:863-873)
__global__ void decision_space_kernel(const float2 *__restrict__ x, long long len, int mode, float *__restrict__ out)
{
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < len; i += (long long)gridDim.x * blockDim.x) {
    const float2 v = x[i];
    out[i] = mode == 0 ? __builtin_sqrtf(sd::fma_(v.y, v.y, v.x * v.x))
                       : (float)((double)sd::atan2_(v.x, -v.y) / 3.14159265358979323846);        // arg(j x) / PI
  }
}

// one byte per symbol instead of eight: sym = clamp(floor((v - vmin) / d), 0, intervals - 1)
__global__ void decide_kernel(const float2 *__restrict__ x, long long len, int mode, int intervals, float vmin, float d,
                              unsigned char *__restrict__ sym)
{
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < len; i += (long long)gridDim.x * blockDim.x) {
    const int s = (int)__builtin_floorf((dec_value(x[i], mode) - vmin) / d);
    sym[i] = (unsigned char)(s < 0 ? 0 : (s > intervals - 1 ? intervals - 1 : s));
  }
}

// integer counts: atomics are exact and order-independent; a block first counts into LDS
__global__ __launch_bounds__(256) void symbol_histogram_kernel(const float2 *__restrict__ x, long long len, int mode,
                                                               float vmin, float d, int nbins, unsigned *__restrict__ hist)
{
  extern __shared__ unsigned lhist[];
  unsigned *lh = lhist;
  for (int i = threadIdx.x; i < nbins; i += blockDim.x) lh[i] = 0;
  __syncthreads();
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < len; i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)__builtin_floorf((dec_value(x[i], mode) - vmin) / d);
    if (b >= 0 && b < nbins) atomicAdd(&lh[b], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nbins; i += blockDim.x) if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

// SNREstimator::feed -> iterate -> recalculateModel (Misc/SNREstimator.cpp:30-169), one workgroup.
// state: [0] sigma, [1] delta, [2] sqerr.  Sums over the bins are fixed-order tree reductions
// (deterministic; the reference adds sequentially -- binary32 rounding differs in the last bits).
constexpr int SNR_T = 256, SNR_MAXLEN = 4096;
__device__ float snr_block_sum(float v, float *red)
{
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = SNR_T / 2; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
  const float r = red[0];
  __syncthreads();
  return r;
}
__device__ float snr_block_max(float v, float *red)
{
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = SNR_T / 2; s > 0; s >>= 1) { if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]); __syncthreads(); }
  const float r = red[0];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(SNR_T) void snr_feed_kernel(const unsigned *__restrict__ history, int length, int intervals,
                                                         float alpha, float *__restrict__ state, float *__restrict__ model)
{
  __shared__ float gaussian[SNR_MAXLEN], red[SNR_T];
  const int tid = threadIdx.x;
  const float sigma = state[0];
  const float hx = 1.f / length;
  // feed(): Htilde = history / max
  float m = 0;
  for (int i = tid; i < length; i += SNR_T) m = fmaxf(m, (float)history[i]);
  unsigned hmax = (unsigned)snr_block_max(m, red);           // exact: counts < 2^24 in practice; guarded below
  for (int i = tid; i < length; i += SNR_T) if (history[i] > hmax) hmax = history[i];
  if (hmax == 0) hmax = 1;
  // recalculateModel(): step 1
  const float sigma2 = sigma * sigma;
  for (int i = tid; i < length; i += SNR_T) {
    float x = i * hx;
    if (x >= .5f) x -= 1.f;
    gaussian[i] = expf(-x * x / sigma2);
  }
  __syncthreads();
  // step 2: every interval adds its shifted gaussian, in interval order
  const float intlen = 1.f / intervals, start = .5f * intlen;
  float himax = 0;
  for (int i = tid; i < length; i += SNR_T) {
    float h = 0.f;
    for (int j = 0; j < intervals; ++j) {
      const float skip = start + j * intlen;
      const float t = 1.f - (skip - floorf(skip));
      const unsigned skipint = (unsigned)floorf(length * skip);
      const unsigned i1 = (unsigned)(length + i - skipint) % (unsigned)length;
      const unsigned i2 = (unsigned)(length + i1 - 1) % (unsigned)length;
      h += t * gaussian[i1];
      h += (1 - t) * gaussian[i2];
    }
    model[i] = h;
    himax = fmaxf(himax, h);
  }
  // step 3: normalise
  const float mx = snr_block_max(himax, red);
  if (mx > 0.f) for (int i = tid; i < length; i += SNR_T) model[i] /= mx;
  // iterate(): gradient step on sigma
  const float sigmainv = 1.f / sigma, sigma3inv = sigmainv * sigmainv * sigmainv;
  float dsum = 0, esum = 0;
  for (int i = tid; i < length; i += SNR_T) {
    float x = i * hx;
    if (x >= .5f) x -= 1.f;
    float term = 0;
    for (int j = 0; j < intervals; ++j) { const float skip = start + j * intlen; term += (x - skip) * (x - skip); }
    const float diff = model[i] - (float)history[i] / hmax;
    term *= diff / sigma3inv;
    dsum += term;
    const float er = diff * diff;
    esum += er * er;
  }
  const float delta = snr_block_sum(dsum, red) / length;
  const float sqerr = snr_block_sum(esum, red);
  if (tid == 0) { state[1] = delta; state[0] = sigma + -alpha * delta; state[2] = sqerr; }
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

// RMSInspector::samplesMessage, raw mode (Default/RMSInspector/RMSInspector.cpp:538-562, :327-338): the power of a sample
// is Re(x conj x), a window's mean its binary64 sum over the count.  One workgroup per window (fixed-order tree);
// the workgroup after the last whole window sums the tail into the carry.
__global__ __launch_bounds__(256) void power_integrate_kernel(const float2 *__restrict__ x, long long len, long long N, long long cnt,
                                                              const double *__restrict__ acc_in, double *__restrict__ acc_out,
                                                              float2 *__restrict__ out, long long K)
{
  __shared__ double sh[256];
  const long long j = blockIdx.x;
  long long lo = j * N - cnt, hi = lo + N;
  if (lo < 0) lo = 0;
  if (hi > len) hi = len;
  double a = 0.0;
  for (long long t = lo + threadIdx.x; t < hi; t += 256) { const float2 v = x[t]; a += (double)(v.x * v.x + v.y * v.y); }   // unfused, as the complex multiply
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) {
    double tot = sh[0];
    if (j == 0) tot += acc_in[0];                           // what earlier feeds left of this window
    if (j < K) out[j] = float2{(float)(tot / (double)N), 0.0f};
    else acc_out[0] = tot;
  }
}

__global__ void baud_nl_kernel(const float2 *__restrict__ x, long long n, float2 *__restrict__ y)
{
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    float v = 0.0f;
    if (t > 0) {
      const float2 a = x[t], b = x[t - 1];
      const float dr = a.x - b.x, di = a.y - b.y;
      v = sd::fma_(dr, dr, di * di);
    }
    y[t] = float2{v, 0.0f};
  }
}

__global__ __launch_bounds__(1024) void baud_line_kernel(const float2 *__restrict__ X, int n, int skip, double *__restrict__ res,
                                                         float *__restrict__ value)
{
  __shared__ float smax[1024];
  __shared__ int sidx[1024];
  const int half = n / 2, tid = threadIdx.x;
  auto P = [&](int k) { const float2 v = X[k]; return sd::fma_(v.x, v.x, v.y * v.y); };
  __shared__ double ssum[1024];
  float mx = 0.0f;
  double sum = 0.0;
  for (int k = skip + tid; k < half; k += 1024) { const float p = P(k); mx = p > mx ? p : mx; sum += (double)p; }
  smax[tid] = mx; ssum[tid] = sum;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) { smax[tid] = smax[tid] > smax[tid + o] ? smax[tid] : smax[tid + o]; ssum[tid] += ssum[tid + o]; }
    __syncthreads();
  }
  // a line, not the tallest noise bin: 20x the mean level (noise alone peaks at ~10x over a few thousand bins)
  const bool line = (double)smax[0] * (double)(half - skip) >= 20.0 * ssum[0];
  const float thr = line ? 0.5f * smax[0] : 0.0f;
  int first = 0x7fffffff;
  for (int k = skip + 1 + tid; k < half - 1; k += 1024) {
    const float p = P(k);
    if (p >= thr && p >= P(k - 1) && p >= P(k + 1)) { first = k; break; }     // strides ascend: a thread's first hit is its lowest
  }
  sidx[tid] = first;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) { if (tid < o) sidx[tid] = sidx[tid] < sidx[tid + o] ? sidx[tid] : sidx[tid + o]; __syncthreads(); }
  if (tid == 0) {
    double c = 0.0;
    if (thr > 0.0f && sidx[0] != 0x7fffffff) {
      double num = 0.0, den = 0.0;
      for (int j = sidx[0] - 4; j <= sidx[0] + 4; ++j) {
        if (j < skip || j >= half) continue;
        const double p = (double)P(j);
        num += p * (double)j; den += p;
      }
      c = den > 0.0 ? num / den : 0.0;
    }
    res[0] = c;
    value[0] = (float)(c / (double)n);                       // normalised baud
  }
}

// sequential scan of at most a few thousand lags: one lane
__global__ void fac_valley_kernel(const float *__restrict__ R, int H, float *__restrict__ out)
{
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float lag = 0.0f;
  const float thr = 0.25f * R[0];
  auto s = [&](int l) { return (R[l - 1] + R[l] + R[l + 1]) * 0.33333334f; };
  for (int l = 2; l + 2 < H; ++l) {
    const float c = s(l);
    if (c < thr && c <= s(l + 1)) { lag = (float)l; break; }   // (a parabola through a V-shaped minimum is biased: integer lag)
  }
  out[0] = lag;
  out[1] = lag > 0.0f ? 1.0f / lag : 0.0f;                   // normalised baud
}

// one workgroup per row: the host-mapped destination is written in 64 B lane pairs, front to back
__global__ __launch_bounds__(1024) void rows_deliver_kernel(const sdk::DeliverItem *__restrict__ items)
{
  const sdk::DeliverItem it = items[blockIdx.x];
  const unsigned n = it.count ? *it.count : it.fixed;
  __syncthreads();                                        // every lane has the count before it is cleared
  if (threadIdx.x == 0) {
    if (it.count) *it.count = 0;
    *it.count_out = n;
  }
  const float2 *src = static_cast<const float2 *>(it.src);
  float2 *dst = static_cast<float2 *>(it.dst);
  if (it.stride == 1) { for (unsigned m = threadIdx.x; m < n; m += 1024) dst[m] = src[m]; }
  else { for (unsigned m = threadIdx.x; m < n; m += 1024) dst[m] = src[(size_t)m * it.stride]; }
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

hipError_t decision_space(const void *x, long long len, int mode, float *out, hipStream_t st)
{
  if (len <= 0) return hipSuccess;
  hipLaunchKernelGGL(decision_space_kernel, dim3(grid_for(len, 256)), dim3(256), 0, st, reinterpret_cast<const float2 *>(x), len, mode, out);
  return hipGetLastError();
}

hipError_t decide(const void *x, long long len, int mode, int intervals, float vmin, float d, unsigned char *sym, hipStream_t st)
{
  if (len <= 0) return hipSuccess;
  hipLaunchKernelGGL(decide_kernel, dim3(grid_for(len, 256)), dim3(256), 0, st, reinterpret_cast<const float2 *>(x), len, mode,
                     intervals, vmin, d, sym);
  return hipGetLastError();
}

hipError_t symbol_histogram(const void *x, long long len, int mode, float vmin, float d, int nbins, unsigned *hist, hipStream_t st)
{
  if (len <= 0) return hipSuccess;
  unsigned g = grid_for(len, 256 * 16);
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(symbol_histogram_kernel, dim3(g), dim3(256), sizeof(unsigned) * (size_t)nbins, st,
                     reinterpret_cast<const float2 *>(x), len, mode, vmin, d, nbins, hist);
  return hipGetLastError();
}

hipError_t snr_feed(const unsigned *history, int length, int intervals, float alpha, float *state, float *model, hipStream_t st)
{
  if (length <= 0 || length > SNR_MAXLEN) return hipErrorInvalidValue;
  hipLaunchKernelGGL(snr_feed_kernel, dim3(1), dim3(SNR_T), 0, st, history, length, intervals, alpha, state, model);
  return hipGetLastError();
}

hipError_t spectsrc_preproc(int kind, const void *x, long long len, float prev_re, float prev_im, const void *prev_dev, void *y, hipStream_t st)
{
  if (len <= 0) return hipSuccess;
  hipLaunchKernelGGL(spectsrc_kernel, dim3(grid_for(len, 256)), dim3(256), 0, st, kind, reinterpret_cast<const float2 *>(x), len,
                     float2{prev_re, prev_im}, reinterpret_cast<const float2 *>(prev_dev), reinterpret_cast<float2 *>(y));
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

hipError_t power_integrate(const void *x, long long len, long long N, long long cnt, const double *acc_in, double *acc_out,
                           void *out, hipStream_t st)
{
  if (len <= 0 || N <= 0) return hipErrorInvalidValue;
  const long long K = (cnt + len) / N;
  if (K + 1 > 0x7fffffffll) return hipErrorInvalidValue;
  hipLaunchKernelGGL(power_integrate_kernel, dim3((unsigned)(K + 1)), dim3(256), 0, st, static_cast<const float2 *>(x), len, N, cnt,
                     acc_in, acc_out, static_cast<float2 *>(out), K);
  return hipGetLastError();
}

hipError_t baud_nl_transform(const void *x, long long n, void *y, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(baud_nl_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, static_cast<const float2 *>(x), n, static_cast<float2 *>(y));
  return hipGetLastError();
}

hipError_t baud_line(const void *X, int n, int skip, double *res, float *value, hipStream_t st)
{
  if (n < 64 || skip < 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(baud_line_kernel, dim3(1), dim3(1024), 0, st, static_cast<const float2 *>(X), n, skip, res, value);
  return hipGetLastError();
}

hipError_t fac_first_valley(const float *fac, int n_half, float *out, hipStream_t st)
{
  if (n_half < 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(fac_valley_kernel, dim3(1), dim3(64), 0, st, fac, n_half, out);
  return hipGetLastError();
}

hipError_t rows_deliver(const DeliverItem *d_items, int n, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(rows_deliver_kernel, dim3((unsigned)n), dim3(1024), 0, st, d_items);
  return hipGetLastError();
}

}  // namespace sdk
