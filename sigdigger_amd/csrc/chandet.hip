// chandet.hip -- su_channel_detector on the device (row N1; SPEC.md section O): what libsuscan runs on the analyzer's
// spectrum to produce SUSCAN_ANALYZER_MESSAGE_TYPE_CHANNEL lists (Suscan/Analyzer.cpp:75-98; parameters alpha / beta /
// gamma / snr of struct sigutils_channel_detector_params, Suscan/AnalyzerParams.cpp:53-71).
//
//   chandet_update:  S[i] <- first ? P[i] : S[i] + alpha (P[i] - S[i])          smoothed spectrum (linear power)
//   chandet_floor:   N0_inst = median(S) (one workgroup, bitonic sort in LDS); N0 <- first ? N0_inst : N0 + gamma (N0_inst - N0)
//   chandet_find:    in frequency order (fftshift of the natural-order bins) a bin is "up" when S > snr N0; a channel
//                    starts at an up bin with no up bin among the GAP + 1 before it and runs until GAP + 1 consecutive
//                    bins are down; one thread walks one channel: first / last up bin, power sum, power-weighted bin
//                    centroid (binary64), peak.  Channels narrower than MINW up-bins are dropped.  Records land in a
//                    table through an atomic slot counter; the host orders them by first bin.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.hpp"

namespace {

constexpr int GAP = 2, MINW = 2;

__global__ void chandet_update_kernel(float *S, const float *P, int n, float alpha, int first)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float p = P[i];
  S[i] = first ? p : S[i] + alpha * (p - S[i]);
}

// one workgroup of 1024 threads; n = 2^k <= 16384 floats in LDS
__global__ __launch_bounds__(1024) void chandet_floor_kernel(const float *S, int n, float gamma, int first, float *N0)
{
  extern __shared__ float v[];
  for (int i = threadIdx.x; i < n; i += 1024) v[i] = S[i];
  __syncthreads();
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n; i += 1024) {
        const int l = i ^ j;
        if (l > i) {
          const bool up = (i & k) == 0;
          const float a = v[i], b = v[l];
          if ((a > b) == up) { v[i] = b; v[l] = a; }
        }
      }
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) {
    const float med = v[n / 2];                              // upper median of an even count
    N0[0] = first ? med : N0[0] + gamma * (med - N0[0]);
  }
}

__global__ __launch_bounds__(1024) void chandet_find_kernel(const float *S, int n, const float *N0, float snr, sdk::ChanDetRecord *rec,
                                                            unsigned *count, unsigned cap)
{
  const float thr = snr * N0[0];
  const int half = n >> 1;
  auto at = [&](int j) { return S[(j + half) & (n - 1)]; };     // j: frequency order, j = 0 is -fs/2
  for (int j = threadIdx.x; j < n; j += 1024) {
    if (!(at(j) > thr)) continue;
    bool starts = true;
    for (int b = 1; b <= GAP + 1 && j - b >= 0; ++b) if (at(j - b) > thr) { starts = false; break; }
    if (!starts) continue;
    int last = j, down = 0, width = 0;
    double sum = 0, wsum = 0;
    float peak = 0;
    for (int t = j; t < n && down <= GAP; ++t) {
      const float p = at(t);
      if (p > thr) { last = t; down = 0; ++width; sum += (double)p; wsum += (double)p * (double)t; if (p > peak) peak = p; }
      else ++down;
    }
    if (width < MINW) continue;
    const unsigned slot = atomicAdd(count, 1u);
    if (slot < cap) rec[slot] = sdk::ChanDetRecord{j, last, width, peak, sum, wsum};
  }
}

}  // namespace

namespace sdk {

hipError_t chandet_feed(float *S, const float *P, int n, float alpha, float gamma, int first, float *N0, hipStream_t st)
{
  hipLaunchKernelGGL(chandet_update_kernel, dim3((n + 255) / 256), dim3(256), 0, st, S, P, n, alpha, first);
  auto kern = chandet_floor_kernel;
  static bool attr_done_dev[64] = {};                        // a function attribute belongs to a device
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  bool &attr_done = attr_done_dev[dev_ & 63];
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(1), dim3(1024), (size_t)n * sizeof(float), st, S, n, gamma, first, N0);
  return hipGetLastError();
}

hipError_t chandet_find(const float *S, int n, const float *N0, float snr, ChanDetRecord *rec, unsigned *count, unsigned cap,
                        hipStream_t st)
{
  hipError_t e = hipMemsetAsync(count, 0, sizeof(unsigned), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(chandet_find_kernel, dim3(1), dim3(1024), 0, st, S, n, N0, snr, rec, count, cap);
  return hipGetLastError();
}

}  // namespace sdk
