// specview.hip -- SpectrumView of the panoramic scanner (rows P2 / P3, config C5):
// re-binning of PSD frames into the <= 65536-bin accumulate / count arrays and gap
// interpolation, following Panoramic/Scanner.cpp:56-256 operation for operation (double
// precision index geometry, int truncation, binary32 accumulation in source-bin order), so the
// result is bit-identical to the sequential reference loops.
//
//   feed_linear : one thread per destination bin j in [j0, k): independent of each other
//                 (Scanner.cpp:153-184), the inner source-bin sum runs in ascending order.
//   feed_hist   : one frame collapses to its mean and lands in 1-2 bins (Scanner.cpp:187-237);
//                 the mean is a sequential binary32 sum, kept sequential (one lane) for parity.
//   interpolate : the reference walks the bins once, carrying (inGap, left, zero_pos, count)
//                 (Scanner.cpp:56-116).  Here every bin finds its nearest valid neighbours with
//                 a workgroup-wide scan and evaluates the same expressions.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.hpp"

namespace {

constexpr float kDefaultBin = -200.0f;   // SIGDIGGER_SCANNER_DEFAULT_BIN_VALUE
constexpr float kCountMax = 5.0f;        // SIGDIGGER_SCANNER_COUNT_MAX
constexpr float kCountReset = 1.0f;      // SIGDIGGER_SCANNER_COUNT_RESET

__global__ void feed_linear_kernel(sdk::SpecViewLinear g, const float *__restrict__ psdData,
                                   const float *__restrict__ countData, float *__restrict__ psdAccum,
                                   float *__restrict__ psdCount)
{
  const int j = g.j0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= g.k) return;
  const double freqJ = g.viewFreqMin + g.dstBinW * j;
  const double srcBin = (freqJ - g.freqMin) / g.srcBinW;
  int startBin = (int)srcBin;
  int endBin = (int)(srcBin + g.delta);
  const int psdSize = g.psdSize;
  startBin = startBin < 0 ? 0 : (startBin > psdSize - 1 ? psdSize - 1 : startBin);
  endBin = endBin < startBin + 1 ? startBin + 1 : (endBin > psdSize ? psdSize : endBin);
  float acc = 0, cnt = 0;
  for (int i = startBin; i < endBin; i++) {
    acc += psdData[i];
    cnt += countData != nullptr ? countData[i] : 1.0f;
  }
  if (cnt > 0) {
    psdAccum[j] += acc / cnt;
    psdCount[j] += 1;
  }
}

// A whole sweep of linear-mode feeds, one destination bin per thread.  The reference processes the
// frames one after the other -- feedLinearMode then interpolate() -- and the only state interpolate()
// changes is the count cap (count > 5 -> accum = mean, count = 1) of bins that do not end a gap, i.e.
// whose left neighbour is valid.  A linear feed adds exactly 1 to the count of every bin it covers,
// so "left neighbour valid" = it was valid before the sweep or one of the frames so far covered it:
// every bin can replay the sweep on its own, in frame order, with the same binary32 operations.
__global__ __launch_bounds__(256) void sweep_linear_kernel(const sdk::SpecViewLinear *__restrict__ geom, int nframes,
                                                          const float *__restrict__ frames, long long frame_stride,
                                                          const float *__restrict__ cntBefore,
                                                          float *__restrict__ psdAccum, float *__restrict__ psdCount, int n)
{
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  float acc = psdAccum[j], cnt = psdCount[j];
  bool left_valid = j > 0 ? cntBefore[j - 1] > .5f : true;       // snapshot: psdCount[j-1] is being rewritten
  for (int f = 0; f < nframes; ++f) {
    const sdk::SpecViewLinear g = geom[f];                      // wave-uniform: scalar loads
    if (j - 1 >= g.j0 && j - 1 < g.k) left_valid = true;
    if (j >= g.j0 && j < g.k) {
      const float *__restrict__ psdData = frames + (long long)f * frame_stride;
      const double freqJ = g.viewFreqMin + g.dstBinW * j;
      const double srcBin = (freqJ - g.freqMin) / g.srcBinW;
      int startBin = (int)srcBin;
      int endBin = (int)(srcBin + g.delta);
      const int psdSize = g.psdSize;
      startBin = startBin < 0 ? 0 : (startBin > psdSize - 1 ? psdSize - 1 : startBin);
      endBin = endBin < startBin + 1 ? startBin + 1 : (endBin > psdSize ? psdSize : endBin);
      float a = 0, c = 0;
      for (int i = startBin; i < endBin; i++) { a += psdData[i]; c += 1.0f; }
      if (c > 0) { acc += a / c; cnt += 1; }
    }
    if (cnt > kCountMax && left_valid) {                        // interpolate(), Scanner.cpp:87-90
      const float v = acc / cnt;
      cnt = kCountReset;
      acc = v * kCountReset;
    }
  }
  psdAccum[j] = acc;
  psdCount[j] = cnt;
}

__global__ void feed_hist_kernel(sdk::SpecViewHist g, const float *__restrict__ psdData,
                                 float *__restrict__ psdAccum, float *__restrict__ psdCount)
{
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  float accum = 0;
  for (int i = 0; i < g.psdSize; ++i) accum += psdData[i];
  accum *= g.inv;
  const unsigned j = g.j;
  if (g.split) {
    psdCount[j] += 1 - g.t;
    psdAccum[j] += (1 - g.t) * accum;
    if (j + 1 < g.spectrumSize) {
      psdCount[j + 1] += g.t;
      psdAccum[j + 1] += g.t * accum;
    }
  } else {
    psdCount[j] += 1;
    psdAccum[j] += accum;
  }
}

// one workgroup of 1024 threads; thread t owns bins [t*per, (t+1)*per)
__global__ __launch_bounds__(1024) void interpolate_kernel(float *__restrict__ psd, float *__restrict__ psdAccum,
                                                           float *__restrict__ psdCount, int n)
{
  __shared__ int lastValid[1024], firstValid[1024];
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int b0 = t * per, b1 = (b0 + per < n) ? b0 + per : n;
  int lv = -1, fv = n;
  for (int i = b0; i < b1; ++i) {
    if (psdCount[i] > .5f) { lv = i; if (fv == n) fv = i; }
  }
  lastValid[t] = lv;
  firstValid[t] = fv;
  __syncthreads();
  // nearest valid bin strictly before this thread's range / at or after its end
  int L = -1, Rn = n;
  for (int u = t - 1; u >= 0; --u) if (lastValid[u] >= 0) { L = lastValid[u]; break; }
  for (int u = t + 1; u < 1024; ++u) if (firstValid[u] < n) { Rn = firstValid[u]; break; }
  __syncthreads();
  // walk the own range right-to-left once to know each bin's right neighbour
  // (per <= 64: kept in a small per-thread array)
  int rightOf[64];
  {
    int r = Rn;
    for (int i = b1 - 1; i >= b0; --i) {
      rightOf[i - b0] = r;
      if (psdCount[i] > .5f) r = i;
    }
  }
  // the reference resets count / accum of over-counted bins while it walks; those writes must not
  // be seen by other threads before they evaluated accum/count of their neighbours -> compute
  // everything first, write after a barrier
  float outv[64];
  bool  reset[64];
  int left = L;
  for (int i = b0; i < b1; ++i) {
    const int idx = i - b0;
    const float cnt = psdCount[i];
    reset[idx] = false;
    if (cnt > .5f) {
      const float v = psdAccum[i] / cnt;
      outv[idx] = v;
      // the bin that ends a gap is not cap-checked by the reference (Scanner.cpp:92-95)
      const bool ends_gap = (i > 0) && (left != i - 1);
      if (!ends_gap && cnt > kCountMax) reset[idx] = true;
      left = i;
    } else {
      const int R = rightOf[idx];
      const bool first = (left < 0);
      if (R >= n) {
        // trailing zeroes: take the value on the left (default when the whole view is empty)
        outv[idx] = first ? kDefaultBin : psdAccum[left] / psdCount[left];
      } else {
        const float rightv = psdAccum[R] / psdCount[R];
        if (first) {
          outv[idx] = rightv;
        } else {
          const float leftv = psdAccum[left] / psdCount[left];
          const unsigned count = (unsigned)(R - left - 1);
          const unsigned jj = (unsigned)(i - (left + 1));
          const float tt = (float)((float)jj + .5f) / count;
          outv[idx] = (1 - tt) * leftv + tt * rightv;
        }
      }
    }
  }
  __syncthreads();
  for (int i = b0; i < b1; ++i) {
    const int idx = i - b0;
    psd[i] = outv[idx];
    if (reset[idx]) {
      psdCount[i] = kCountReset;
      psdAccum[i] = outv[idx] * kCountReset;
    }
  }
}

}  // namespace

namespace sdk {

hipError_t specview_feed_linear(const SpecViewLinear &g, const float *psd, const float *count, float *accum,
                                float *cnt, hipStream_t st)
{
  const int nb = g.k - g.j0;
  if (nb <= 0) return hipSuccess;
  hipLaunchKernelGGL(feed_linear_kernel, dim3((nb + 255) / 256), dim3(256), 0, st, g, psd, count, accum, cnt);
  return hipGetLastError();
}

hipError_t specview_sweep_linear(const SpecViewLinear *d_geom, int nframes, const float *frames, long long frame_stride,
                                 const float *cnt_before, float *accum, float *cnt, int n, hipStream_t st)
{
  if (n <= 0 || nframes <= 0) return hipSuccess;
  hipLaunchKernelGGL(sweep_linear_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_geom, nframes, frames, frame_stride,
                     cnt_before, accum, cnt, n);
  return hipGetLastError();
}

hipError_t specview_feed_hist(const SpecViewHist &g, const float *psd, float *accum, float *cnt, hipStream_t st)
{
  hipLaunchKernelGGL(feed_hist_kernel, dim3(1), dim3(64), 0, st, g, psd, accum, cnt);
  return hipGetLastError();
}

hipError_t specview_interpolate(float *psd, float *accum, float *cnt, int n, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  if (n > 65536) return hipErrorInvalidValue;
  hipLaunchKernelGGL(interpolate_kernel, dim3(1), dim3(1024), 0, st, psd, accum, cnt, n);
  return hipGetLastError();
}

}  // namespace sdk
