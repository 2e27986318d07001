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
                                   const float *__restrict__ countData, float *__restrict__ acc_sum,
                                   float *__restrict__ acc_cnt)
{
  const int j = g.j0 + blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= g.k) return;
  const double f_bin = g.viewFreqMin + g.dstBinW * j;
  const double src_pos = (f_bin - g.freqMin) / g.srcBinW;
  int b_lo = (int)src_pos;
  int b_hi = (int)(src_pos + g.delta);
  const int psdSize = g.psdSize;
  b_lo = b_lo < 0 ? 0 : (b_lo > psdSize - 1 ? psdSize - 1 : b_lo);
  b_hi = b_hi < b_lo + 1 ? b_lo + 1 : (b_hi > psdSize ? psdSize : b_hi);
  float acc = 0, cnt = 0;
  for (int i = b_lo; i < b_hi; i++) {
    acc += psdData[i];
    cnt += countData != nullptr ? countData[i] : 1.0f;
  }
  if (cnt > 0) {
    acc_sum[j] += acc / cnt;
    acc_cnt[j] += 1;
  }
}

// A whole sweep of linear-mode feeds, one destination bin per thread.  The reference processes the
// frames one after the other -- feedLinearMode then interpolate() -- and the only state interpolate()
// changes is the count cap (count > 5 -> accum = mean, count = 1) of bins that do not end a gap, i.e.
// whose left neighbour is valid.  A linear feed adds exactly 1 to the count of every bin it covers,
// so "left neighbour valid" = it was valid before the sweep or one of the frames so far covered it:
// every bin can replay the sweep on its own, in frame order, with the same binary32 operations.
//
// A frame touches the state of bin j only if it covers j or j-1, and the cap test is a function of that
// state: after a frame that covers neither it gives what it gave before.  So a workgroup (256
// consecutive bins) first lists, in order, the frames that reach its bins at all -- 256 frames per
// round, one per thread, an order-preserving compaction through LDS -- and replays only those; frame 0
// is always listed, because the cap test after the FIRST frame also sees the state left by the
// previous sweep.  (512 dwells over 65536 bins: 3-4 frames per workgroup instead of 512 scalar
// geometry loads per thread.)
__global__ __launch_bounds__(256) void sweep_linear_kernel(const sdk::SpecViewLinear *__restrict__ geom, int nframes,
                                                          const float *__restrict__ frames, long long frame_stride,
                                                          const float *__restrict__ cntBefore,
                                                          float *__restrict__ acc_sum, float *__restrict__ acc_cnt, int n)
{
  __shared__ int list[256];
  __shared__ int wave_cnt[4];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int J = blockIdx.x * blockDim.x;                         // bins [J, J + 255]
  const int j = J + t;
  const bool live = j < n;
  float acc = live ? acc_sum[j] : 0.0f, cnt = live ? acc_cnt[j] : 0.0f;
  bool left_valid = (live && j > 0) ? cntBefore[j - 1] > .5f : true;   // snapshot: acc_cnt[j-1] is being rewritten
  for (int f0 = 0; f0 < nframes; f0 += 256) {
    // which of the frames f0 .. f0+255 reach a bin of this workgroup (or the left neighbour of one)?
    const int f = f0 + t;
    bool hit = false;
    if (f < nframes) {
      const int fj0 = geom[f].j0, fk = geom[f].k;
      hit = f == 0 || (fj0 <= J + 255 && fk >= J && fk > fj0);
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wave_cnt[wv] = __popcll(m);
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) { if (u < wv) base += wave_cnt[u]; total += wave_cnt[u]; }
    if (hit) list[base + __popcll(m & ((1ull << lane) - 1))] = f;
    __syncthreads();
    for (int e = 0; e < total; ++e) {
      const int fe = __builtin_amdgcn_readfirstlane(list[e]);
      const sdk::SpecViewLinear g = geom[fe];                    // wave-uniform: scalar loads
      if (live) {
        if (j - 1 >= g.j0 && j - 1 < g.k) left_valid = true;
        if (j >= g.j0 && j < g.k) {
          const float *__restrict__ psdData = frames + (long long)fe * frame_stride;
          const double f_bin = g.viewFreqMin + g.dstBinW * j;
          const double src_pos = (f_bin - g.freqMin) / g.srcBinW;
          int b_lo = (int)src_pos;
          int b_hi = (int)(src_pos + g.delta);
          const int psdSize = g.psdSize;
          b_lo = b_lo < 0 ? 0 : (b_lo > psdSize - 1 ? psdSize - 1 : b_lo);
          b_hi = b_hi < b_lo + 1 ? b_lo + 1 : (b_hi > psdSize ? psdSize : b_hi);
          // the source bins are summed in ascending order, as the reference does; eight requests in flight at a time
          // (c counts them: a sum of ones, exact)
          float a = 0;
          int i = b_lo;
          for (; i + 8 <= b_hi; i += 8) {
            float s8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) s8[u] = psdData[i + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += s8[u];
          }
          for (; i < b_hi; i++) a += psdData[i];
          const float c = (float)(b_hi - b_lo);
          if (c > 0) { acc += a / c; cnt += 1; }
        }
        if (cnt > kCountMax && left_valid) {                      // interpolate(), Scanner.cpp:87-90
          const float v = acc / cnt;
          cnt = kCountReset;
          acc = v * kCountReset;
        }
      }
    }
    __syncthreads();                                             // the list is rebuilt in the next round
  }
  if (live) {
    acc_sum[j] = acc;
    acc_cnt[j] = cnt;
  }
}

__global__ void feed_hist_kernel(sdk::SpecViewHist g, const float *__restrict__ psdData,
                                 float *__restrict__ acc_sum, float *__restrict__ acc_cnt)
{
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  float accum = 0;
  for (int i = 0; i < g.psdSize; ++i) accum += psdData[i];
  accum *= g.inv;
  const unsigned j = g.j;
  if (g.split) {
    acc_cnt[j] += 1 - g.t;
    acc_sum[j] += (1 - g.t) * accum;
    if (j + 1 < g.spectrumSize) {
      acc_cnt[j + 1] += g.t;
      acc_sum[j + 1] += g.t * accum;
    }
  } else {
    acc_cnt[j] += 1;
    acc_sum[j] += accum;
  }
}

// interpolate(): lane = bin inside a group of 64 consecutive bins.  The valid bins of a group are one ballot, a bin's
// nearest valid neighbours inside its group are two bit scans of that mask, and the neighbours beyond the group come from
// a prefix-max / suffix-min over the (at most 1024) groups' last / first valid bin.  Every global access is a coalesced
// row (the walk of the reference, one thread per 64 consecutive bins, touched 64 cache lines per request: 200 us).
// INTERP_WGS workgroups of 1024 threads share the bins: each recomputes all masks and the group scan for itself (256 KiB
// of counts out of L2, 10 barriers) and then evaluates its own 1/INTERP_WGS of the groups -- one workgroup alone spends
// ~80 instructions per bin on ONE CU (57 us).  The reference resets count / accum of over-counted bins while it walks;
// those writes must not tear under a neighbour's read of (accum, count), so this kernel only records them (one mask per
// group) and interpolate_reset_kernel applies them afterwards.  The expressions are the reference's (Scanner.cpp:56-116).
constexpr int INTERP_WGS = 16;

__global__ __launch_bounds__(1024) void interpolate_kernel(float *__restrict__ psd, const float *__restrict__ acc_sum,
                                                           const float *__restrict__ acc_cnt, int n,
                                                           unsigned long long *__restrict__ resetMask)
{
  __shared__ unsigned long long mask[1024];
  __shared__ int lastv[2][1024], firstv[2][1024];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int G = (n + 63) >> 6;
  // 1. the valid mask of every group: wavefront wv votes on groups 64 wv .. 64 wv + 63, their counts requested 32 rows
  //    at a time (a loop that asks for one row, waits and votes runs at one memory latency per row)
#pragma unroll
  for (int k0 = 0; k0 < 64; k0 += 32) {
    float c[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int i = (wv * 64 + k0 + u) * 64 + lane;
      c[u] = i < n ? acc_cnt[i] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const unsigned long long m = __ballot(c[u] > .5f);
      if (lane == 0) mask[wv * 64 + k0 + u] = m;
    }
  }
  __syncthreads();
  // 2. per group: last valid bin of any earlier group (-1: none), first valid bin of any later group (n: none)
  {
    const unsigned long long m = mask[t];
    lastv[0][t] = m ? t * 64 + 63 - __clzll((long long)m) : -1;
    firstv[0][t] = m ? t * 64 + __ffsll((unsigned long long)m) - 1 : n;
  }
  __syncthreads();
  int cur = 0;
  for (int d = 1; d < 1024; d <<= 1) {
    const int a = lastv[cur][t], b2 = firstv[cur][t];
    const int a2 = t >= d ? lastv[cur][t - d] : -1, b3 = t + d < 1024 ? firstv[cur][t + d] : n;
    lastv[cur ^ 1][t] = a > a2 ? a : a2;
    firstv[cur ^ 1][t] = b2 < b3 ? b2 : b3;
    cur ^= 1;
    __syncthreads();
  }
  // (inclusive scans: group g's outside neighbours are entries g-1 and g+1)
  // 3. the outputs of this workgroup's groups: 1024 / INTERP_WGS of them, 1024 / INTERP_WGS / 16 per wavefront
  constexpr int GPW = 1024 / INTERP_WGS / 16;
  const int g0 = blockIdx.x * (1024 / INTERP_WGS) + wv * GPW;
  float c[GPW], ac[GPW];
#pragma unroll
  for (int u = 0; u < GPW; ++u) {
    const int i = (g0 + u) * 64 + lane;
    c[u] = i < n ? acc_cnt[i] : 0.0f;
    ac[u] = i < n ? acc_sum[i] : 0.0f;
  }
#pragma unroll
  for (int u = 0; u < GPW; ++u) {
    const int g = g0 + u;
    const int i = g * 64 + lane;
    if (g >= G) continue;
    const unsigned long long m = mask[g];
    const unsigned long long below = m & ((1ull << lane) - 1ull);
    const unsigned long long above = lane < 63 ? (m >> (lane + 1)) : 0ull;
    const int left = below ? g * 64 + 63 - __clzll((long long)below) : (g > 0 ? lastv[cur][g - 1] : -1);
    const int R = above ? i + __ffsll(above) : (g + 1 < 1024 ? firstv[cur][g + 1] : n);
    bool reset = false;
    if (i < n) {
      float outv;
      if ((m >> lane) & 1ull) {
        const float cnt = c[u];
        outv = ac[u] / cnt;
        // the bin that ends a gap is not cap-checked by the reference (Scanner.cpp:92-95)
        const bool ends_gap = (i > 0) && (left != i - 1);
        reset = !ends_gap && cnt > kCountMax;
      } else {
        const bool first = (left < 0);
        if (R >= n) {
          // trailing zeroes: take the value on the left (default when the whole view is empty)
          outv = first ? kDefaultBin : acc_sum[left] / acc_cnt[left];
        } else {
          const float rightv = acc_sum[R] / acc_cnt[R];
          if (first) {
            outv = rightv;
          } else {
            const float leftv = acc_sum[left] / acc_cnt[left];
            const unsigned count = (unsigned)(R - left - 1);
            const unsigned jj = (unsigned)(i - (left + 1));
            const float tt = (float)((float)jj + .5f) / count;
            outv = (1 - tt) * leftv + tt * rightv;
          }
        }
      }
      psd[i] = outv;
    }
    const unsigned long long rm = __ballot(reset);
    if (lane == 0) resetMask[g] = rm;
  }
}

// count > 5 -> accum = mean, count = 1 for the bins interpolate_kernel marked (Scanner.cpp:87-90; psd[i] holds the mean)
__global__ __launch_bounds__(256) void interpolate_reset_kernel(const float *__restrict__ psd, float *__restrict__ acc_sum,
                                                                float *__restrict__ acc_cnt, int n,
                                                                const unsigned long long *__restrict__ resetMask)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if ((resetMask[i >> 6] >> (i & 63)) & 1ull) {
    acc_cnt[i] = kCountReset;
    acc_sum[i] = psd[i] * kCountReset;
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

hipError_t specview_interpolate(float *psd, float *accum, float *cnt, int n, unsigned long long *reset_scratch, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  if (n > 65536 || !reset_scratch) return hipErrorInvalidValue;
  // only the workgroups that own a group below ceil(n / 64) have anything to do
  const int groups = (n + 63) / 64, per = 1024 / INTERP_WGS;
  hipLaunchKernelGGL(interpolate_kernel, dim3((groups + per - 1) / per), dim3(1024), 0, st, psd, accum, cnt, n, reset_scratch);
  hipLaunchKernelGGL(interpolate_reset_kernel, dim3((n + 255) / 256), dim3(256), 0, st, psd, accum, cnt, n, reset_scratch);
  return hipGetLastError();
}

}  // namespace sdk
