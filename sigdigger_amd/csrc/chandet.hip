// chandet.hip -- su_channel_detector on the device (row N1; SPEC.md section O): what libsuscan runs on the analyzer's
// spectrum to produce SUSCAN_ANALYZER_MESSAGE_TYPE_CHANNEL lists (Suscan/Analyzer.cpp:75-98; parameters alpha / beta /
// gamma / snr of struct sigutils_channel_detector_params, Suscan/AnalyzerParams.cpp:53-71).
//
//   chandet_update:  S[i] <- first ? P[i] : S[i] + alpha (P[i] - S[i])          smoothed spectrum (linear power)
//   chandet_floor:   N0_inst = median(S) (one workgroup, radix select); N0 <- first ? N0_inst : N0 + gamma (N0_inst - N0)
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

// The median of S (the element of rank n / 2 in ascending order: the upper median of an even count), n = 2^k <= 16384, by
// ONE workgroup of 1024 threads: a radix select on the floats' ordered-integer image, four passes of eight bits.  A thread
// keeps its <= 16 values in registers; a pass counts the digits of the values that still carry the prefix found so far
// (LDS histogram), the first wavefront scans the 256 counts and picks the digit that holds the wanted rank.  (Rounds 4 - 5
// sorted the whole spectrum in LDS with a bitonic network -- 91 barrier-separated stages, 103 us per 8192 bins in the live
// analyzer's profile, more than the block's channeliser and PSD together; a selection returns the same element.)
constexpr int FLOOR_THREADS = 1024, FLOOR_EPT = 16;
__device__ __forceinline__ unsigned ordered_key(float f)
{
  const unsigned u = __float_as_uint(f);
  return u ^ ((unsigned)((int)u >> 31) | 0x80000000u);        // negative: every bit flipped; positive: the sign bit set
}
__device__ __forceinline__ float ordered_value(unsigned k)
{
  return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}
__global__ __launch_bounds__(FLOOR_THREADS) void chandet_floor_kernel(const float *S, int n, float gamma, int first, float *N0)
{
  __shared__ unsigned hist[256];
  __shared__ unsigned found[2];                                // the prefix so far, the rank wanted among the values that carry it
  const int tid = threadIdx.x;
  unsigned key[FLOOR_EPT];
#pragma unroll
  for (int e = 0; e < FLOOR_EPT; ++e) {
    const int i = tid + e * FLOOR_THREADS;
    key[e] = i < n ? ordered_key(S[i]) : 0u;
  }
  unsigned prefix = 0, rank = (unsigned)n / 2;
#pragma unroll
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < FLOOR_EPT; ++e) {
      const int i = tid + e * FLOOR_THREADS;
      // (shift = 24: no prefix yet -- the shift by 32 is never evaluated)
      const bool in = i < n && (shift == 24 || (key[e] >> ((shift + 8) & 31)) == prefix);
      if (in) atomicAdd(&hist[(key[e] >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 64) {
      // lane l: digits 4 l .. 4 l + 3; an inclusive scan of the lanes' sums, then the digit inside the one lane whose
      // interval [before, before + sum) holds the rank
      const unsigned c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
      const unsigned sum = c0 + c1 + c2 + c3;
      unsigned incl = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = __shfl_up(incl, o);
        if (tid >= o) incl += up;
      }
      const unsigned before = incl - sum;
      if (rank >= before && rank < incl) {
        unsigned r = rank - before, d = 0;
        if (r >= c0) { r -= c0; d = 1; if (r >= c1) { r -= c1; d = 2; if (r >= c2) { r -= c2; d = 3; } } }
        found[0] = (prefix << 8) | (4u * (unsigned)tid + d);
        found[1] = r;
      }
    }
    __syncthreads();
    prefix = found[0];
    rank = found[1];
  }
  if (tid == 0) {
    const float med = ordered_value(prefix);
    N0[0] = first ? med : N0[0] + gamma * (med - N0[0]);
  }
}

// (the spectrum is staged in LDS in frequency order first: a thread's walk is a chain of dependent reads -- from global
// memory 67 us per 8192 bins with 64 narrow carriers in the live analyzer's profile)
__global__ __launch_bounds__(1024) void chandet_find_kernel(const float *S, int n, const float *N0, float snr, sdk::ChanDetRecord *rec,
                                                            unsigned *count, unsigned cap)
{
  extern __shared__ float sf[];                                 // sf[j]: frequency order, j = 0 is -fs/2 (fftshift of the natural-order bins)
  const float thr = snr * N0[0];
  const int half = n >> 1;
  for (int j = threadIdx.x; j < n; j += 1024) sf[j] = S[(j + half) & (n - 1)];
  __syncthreads();
  auto at = [&](int j) { return sf[j]; };
  for (int j = threadIdx.x; j < n; j += 1024) {
    if (!(at(j) > thr)) continue;
    bool starts = true;
    for (int b = 1; b <= GAP + 1 && j - b >= 0; ++b) if (at(j - b) > thr) { starts = false; break; }
    if (!starts) continue;
    int last = j, down = 0, width = 0;
    double sum = 0, wsum = 0;
    float peak = 0;
    // the walk, eight bins requested at a time (the exit test depends on every bin's value: read one by one, each
    // iteration would wait for its own LDS read); bin for bin the same tests and the same additions in the same order
    bool go = true;
    for (int t0 = j; go; t0 += 8) {
      float pv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) pv[i] = at(t0 + i < n ? t0 + i : n - 1);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int t = t0 + i;
        go = go && t < n && down <= GAP;
        if (go) {
          const float p = pv[i];
          if (p > thr) { last = t; down = 0; ++width; sum += (double)p; wsum += (double)p * (double)t; if (p > peak) peak = p; }
          else ++down;
        }
      }
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
  hipLaunchKernelGGL(chandet_floor_kernel, dim3(1), dim3(FLOOR_THREADS), 0, st, S, n, gamma, first, N0);
  return hipGetLastError();
}

hipError_t chandet_find(const float *S, int n, const float *N0, float snr, ChanDetRecord *rec, unsigned *count, unsigned cap,
                        hipStream_t st)
{
  hipError_t e = hipMemsetAsync(count, 0, sizeof(unsigned), st);
  if (e != hipSuccess) return e;
  auto kern = chandet_find_kernel;
  static bool attr_done_dev[64] = {};                        // a function attribute belongs to a device
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  bool &attr_done = attr_done_dev[dev_ & 63];
  if (!attr_done) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 4);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(1), dim3(1024), (size_t)n * sizeof(float), st, S, n, N0, snr, rec, count, cap);
  return hipGetLastError();
}

}  // namespace sdk
