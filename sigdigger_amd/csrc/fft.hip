// fft.hip -- large one-shot FFTs for the offline Tasks (T9 CarrierDetector, T10 DopplerCalculator:
// Blackman-Harris window, zero-pad to a power of two, one forward FFT of the whole capture,
// |X|^2, arg-max, circular centroid -- Tasks/CarrierDetector.cpp:80-143,
// Tasks/DopplerCalculator.cpp:85-175).
//
// The transform is a Stockham autosort FFT run pass by pass through HBM (radix 16, a radix
// 2/4/8 pass for the remainder, ping-pong buffers).  Every pass reads N*8 and writes N*8 bytes
// with coalesced accesses (reads always; writes in runs of Ns elements), so a 4 Mi-point
// transform moves ~0.4 GB: it is HBM-bound and runs once per Task, unlike the per-block PSD
// (psd.hip) which keeps the whole frame in LDS.  Twiddles are evaluated with sincospif on the
// exact dyadic argument (no table).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.hpp"

namespace {

typedef float cf __attribute__((ext_vector_type(2)));

__device__ __forceinline__ cf cmul(cf a, cf b) { return __builtin_elementwise_fma(a.yy, cf{-b.y, b.x}, a.xx * b); }
__device__ __forceinline__ cf mul_mj(cf a) { return cf{a.y, -a.x}; }

template <int R> __device__ __forceinline__ void dft(cf *v);
template <> __device__ __forceinline__ void dft<2>(cf *v) { cf t = v[0]; v[0] = t + v[1]; v[1] = t - v[1]; }
template <> __device__ __forceinline__ void dft<4>(cf *v)
{
  cf t0 = v[0] + v[2], t1 = v[0] - v[2], t2 = v[1] + v[3], t3 = mul_mj(v[1] - v[3]);
  v[0] = t0 + t2; v[1] = t1 + t3; v[2] = t0 - t2; v[3] = t1 - t3;
}
template <> __device__ __forceinline__ void dft<8>(cf *v)
{
  cf e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
  dft<4>(e); dft<4>(o);
  const float h = 0.70710678118654752440f;
  o[1] = (o[1] + mul_mj(o[1])) * h;
  o[2] = mul_mj(o[2]);
  o[3] = (mul_mj(o[3]) - o[3]) * h;
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = e[i] + o[i]; v[i + 4] = e[i] - o[i]; }
}
template <> __device__ __forceinline__ void dft<16>(cf *v)
{
  cf e[8], o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
  dft<8>(e); dft<8>(o);
  const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
  o[1] = cmul(o[1], cf{ c1, -s1});
  o[2] = (o[2] + mul_mj(o[2])) * h;
  o[3] = cmul(o[3], cf{ s1, -c1});
  o[4] = mul_mj(o[4]);
  o[5] = cmul(o[5], cf{-s1, -c1});
  o[6] = (mul_mj(o[6]) - o[6]) * h;
  o[7] = cmul(o[7], cf{-c1, -s1});
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = e[i] + o[i]; v[i + 8] = e[i] - o[i]; }
}

// one Stockham pass: butterfly j reads in[j + q*N/R], multiplies by W_(Ns*R)^(q*k), k = j mod Ns,
// writes out[(j - k)*R + k + q*Ns]
// (blockIdx.y: the transform of a batch, n elements apart in both buffers -- `in_stride` apart in the input of a WINDOWED
// first pass, which reads the raw samples and applies the window on the way in: one round trip through HBM less)
// POWER (the PSD's last pass): |X|^2 is written instead of X, as floats at the head of the frame's slot of `out`
// FASTTW (the PSD's passes): the powers 1, 2, 4, 8 of the butterfly's twiddle are evaluated, the others are one product each
// (15 sincospif per radix-16 butterfly are four times the arithmetic of the butterfly itself); the Tasks' one-shot
// transforms keep every twiddle exact
// STAGED: see the store at the end (passes with ns < 256 of the batched PSD)
template <int R, bool WINDOWED, bool POWER = false, bool FASTTW = false, bool STAGED = false>
__global__ void fft_pass_kernel(const cf *__restrict__ in, cf *__restrict__ out, long long n, long long ns,
                                long long in_stride, const float *__restrict__ window)
{
  in += (long long)blockIdx.y * (WINDOWED ? in_stride : n);
  out += (long long)blockIdx.y * n;
  const long long nb = n / R;
  for (long long j = blockIdx.x * (long long)blockDim.x + threadIdx.x; j < nb;
       j += (long long)gridDim.x * blockDim.x) {
    const long long k = j & (ns - 1);
    cf v[R];
#pragma unroll
    for (int q = 0; q < R; ++q) v[q] = in[j + q * nb];
    if (WINDOWED) {
#pragma unroll
      for (int q = 0; q < R; ++q) { const float w = window[j + q * nb]; v[q] = cf{v[q].x * w, v[q].y * w}; }
    }
    if (ns > 1) {
      // angle(q) = -2 pi q k / (ns R): q k / (ns R) is a dyadic rational, exact in binary32 for n <= 2^24
      const float base = -2.0f * (float)k / (float)(ns * R);
      if (FASTTW) {
        cf w[R];
#pragma unroll
        for (int q = 1; q < R; q <<= 1) { float sn, cs; sincospif(base * (float)q, &sn, &cs); w[q] = cf{cs, sn}; }
#pragma unroll
        for (int q = 3; q < R; ++q) if (q & (q - 1)) { const int hi = 1 << (31 - __builtin_clz(q)); w[q] = cmul(w[q - hi], w[hi]); }
#pragma unroll
        for (int q = 1; q < R; ++q) v[q] = cmul(v[q], w[q]);
      } else {
#pragma unroll
        for (int q = 1; q < R; ++q) {
          float sn, cs;
          sincospif(base * (float)q, &sn, &cs);
          v[q] = cmul(v[q], cf{cs, sn});
        }
      }
    }
    dft<R>(v);
    const long long j0 = (j - k) * R + k;
    if (POWER) {
      float *po = reinterpret_cast<float *>(out);
#pragma unroll
      for (int q = 0; q < R; ++q) po[j0 + q * ns] = v[q].x * v[q].x + v[q].y * v[q].y;
    } else if (STAGED) {
      // ns < 256: the 256 consecutive butterflies of the workgroup fill one contiguous run of 256 R outputs, but each
      // store of a wavefront would touch 64 lines (stride R for ns = 1, runs of ns otherwise).  Through LDS the run
      // goes out as whole rows.  (the grid-stride loop is uniform: nb is a multiple of the grid's 256-thread blocks)
      __shared__ cf stage[256 * R + 16 * R];                    // one slot of padding every 16 (stride-R writes: see psd.hip)
      auto pad = [](int i) { return i + (i >> 4); };
      const int jl = threadIdx.x, kl = (int)k;                  // ns divides 256: k = j mod ns = jl mod ns
      __syncthreads();                                          // the previous trip's copy-out is done
#pragma unroll
      for (int q = 0; q < R; ++q) stage[pad((jl - kl) * R + kl + q * (int)ns)] = v[q];
      __syncthreads();
      cf *dst = out + (j - jl) * R;
#pragma unroll
      for (int q = 0; q < R; ++q) dst[q * 256 + jl] = stage[pad(q * 256 + jl)];
    } else {
#pragma unroll
      for (int q = 0; q < R; ++q) out[j0 + q * ns] = v[q];
    }
  }
}

// Tasks/CarrierDetector.cpp:80-89: copy, zero-pad, Blackman-Harris over the first len samples
__global__ void window_pad_kernel(const float2 *__restrict__ data, long long len, long long alloc, float2 *__restrict__ buf)
{
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < alloc;
       i += (long long)gridDim.x * blockDim.x) {
    float2 v = float2{0.0f, 0.0f};
    if (i < len) {
      const double t = 2.0 * 3.14159265358979323846 * (double)i / (double)(len - 1);
      const float w = (float)(0.35875 - 0.48829 * cos(t) + 0.14128 * cos(2 * t) - 0.01168 * cos(3 * t));
      const float2 d = data[i];
      v = float2{d.x * w, d.y * w};
    }
    buf[i] = v;
  }
}

// |X|^2 of bins [lo, hi) (others keep Re(X), as the reference's loop leaves them) and the block-wise arg-max (first
// maximum wins, like the sequential scan).  The power is the real part of the reference's complex product
// `x *= conj(x)` (Tasks/DopplerCalculator.cpp:120, CarrierDetector.cpp:113): two rounded squares and their rounded sum,
// not an fma.
__global__ __launch_bounds__(256) void power_argmax_kernel(float2 *__restrict__ buf, long long alloc, long long lo,
                                                           long long hi, float *__restrict__ mirror,
                                                           float *__restrict__ blk_max, long long *__restrict__ blk_idx)
{
  __shared__ float smax[256];
  __shared__ long long sidx[256];
  float best = 0.0f;
  long long bi = 0;
  for (long long i = lo + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < hi;
       i += (long long)gridDim.x * blockDim.x) {
    const float2 x = buf[i];
    const float p = __fadd_rn(__fmul_rn(x.x, x.x), __fmul_rn(x.y, x.y));
    buf[i] = float2{p, 0.0f};
    if (mirror != nullptr) mirror[(alloc - i + alloc / 2) % alloc] = p;      // DopplerCalculator.cpp:128
    if (p > best || (p == best && p > 0.0f && i < bi)) { best = p; bi = i; }
  }
  smax[threadIdx.x] = best; sidx[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const float om = smax[threadIdx.x + s];
      const long long oi = sidx[threadIdx.x + s];
      if (om > smax[threadIdx.x] || (om == smax[threadIdx.x] && om > 0.0f && oi < sidx[threadIdx.x])) {
        smax[threadIdx.x] = om; sidx[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { blk_max[blockIdx.x] = smax[0]; blk_idx[blockIdx.x] = sidx[0]; }
}

// ---- the reference's running sums, in ITS type and order --------------------------------------------------------------
// DopplerCalculator's total energy is a binary32 Kahan sum over the bins in index order (:130-134), its centroid a
// binary32 complex accumulator and its variance a plain binary32 accumulator over the bins in order from `start`
// (:144-158); CarrierDetector's centroid likewise (:120-131).  A binary32 running sum of 2^15 .. 2^24 terms carries a
// rounding error of the order of sqrt(bins) ulp -- 1e-5 relative at 2^19 bins -- so "the reference's number" is the number
// THAT sequence of roundings produces: a pairwise or binary64 sum is more accurate and therefore different (round 3
// compared sigma at 2e-3).  A sequence of dependent roundings is serial by nature: one wavefront per sum walks the bins;
// what is parallel is everything else -- 64 lanes fetch and form the next 256 terms (gather, sincos, products, division)
// while the sum of the previous 256 runs as a chain of uniform adds fed by LDS broadcast reads (one ds_read_b128 per four
// terms).  ~6 cycles per term: 1.3 ms per 2^19 bins for a plain sum, ~5 ms for the Kahan sum -- once per capture.
constexpr int SER_TILE = 256;

// total = Kahan sum of buf[i].x, i = 0 .. bins-1 (bins a multiple of 16)
__global__ __launch_bounds__(64) void serial_kahan_kernel(const float2 *__restrict__ buf, long long bins, float *__restrict__ total_out)
{
#pragma clang fp contract(off)
  __shared__ __attribute__((aligned(16))) float tile[2][SER_TILE];
  const int l = threadIdx.x;
  float total = 0.0f, err = 0.0f;
  float r[SER_TILE / 64];
  auto fetch = [&](long long i0) {
#pragma unroll
    for (int q = 0; q < SER_TILE / 64; ++q) { const long long i = i0 + l + 64 * q; r[q] = i < bins ? buf[i].x : 0.0f; }
  };
  fetch(0);
  int pp = 0;
  for (long long i0 = 0; i0 < bins; i0 += SER_TILE, pp ^= 1) {
#pragma unroll
    for (int q = 0; q < SER_TILE / 64; ++q) tile[pp][l + 64 * q] = r[q];
    __syncthreads();
    if (i0 + SER_TILE < bins) fetch(i0 + SER_TILE);
    const int cnt = bins - i0 < SER_TILE ? (int)(bins - i0) : SER_TILE;      // a multiple of 4
    for (int e = 0; e < cnt; e += 4) {
      const float4 v = *reinterpret_cast<const float4 *>(&tile[pp][e]);
      const float ps[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float y = __fsub_rn(ps[k], err);
        const float t = __fadd_rn(total, y);
        err = __fsub_rn(__fsub_rn(t, total), y);
        total = t;
      }
    }
  }
  if (l == 0) *total_out = total;
}

// blockIdx.x = 0: acc.re, 1: acc.im, 2: the variance accumulator (only with_dispersion); res[0..2] binary32
__global__ __launch_bounds__(64) void serial_centroid_kernel(const float2 *__restrict__ buf, long long alloc, int nblk,
                                                             const float *__restrict__ blk_max,
                                                             const long long *__restrict__ blk_idx,
                                                             const float *__restrict__ total_in, long long bins,
                                                             long long delta, float *__restrict__ res)
{
#pragma clang fp contract(off)
  __shared__ __attribute__((aligned(16))) float tile[2][SER_TILE];
  __shared__ long long smaxidx;
  __shared__ float smaxval;
  const int l = threadIdx.x, which = blockIdx.x;
  if (l == 0) {
    float best = 0.0f; long long bi = 0;
    for (int b = 0; b < nblk; ++b)
      if (blk_max[b] > best || (blk_max[b] == best && best > 0.0f && blk_idx[b] < bi)) { best = blk_max[b]; bi = blk_idx[b]; }
    smaxidx = bi; smaxval = best;
  }
  __syncthreads();
  const long long start = smaxidx - delta;
  const float total = which == 2 ? *total_in : 1.0f;
  const float d2 = __fmul_rn((float)delta, (float)delta);     // delta * delta (:157; an int product there, see SPEC.md T10)
  const float falloc = (float)alloc;
  float r[SER_TILE / 64];
  auto fetch = [&](long long i0) {
#pragma unroll
    for (int q = 0; q < SER_TILE / 64; ++q) {
      const long long i = i0 + l + 64 * q;
      float term = 0.0f;
      if (i < bins) {
        long long j = i + start;
        if (j < 0) j += alloc;
        j %= alloc;
        const float psd = buf[j].x;
        if (which == 2) {
          long long jj = i;
          if (jj >= delta) jj -= bins;
          term = __fdiv_rn(__fdiv_rn(__fmul_rn((float)(jj * jj), psd), total), d2);
        } else {
          const float nFreq = __fdiv_rn(__fmul_rn(2.f, (float)j), falloc);
          float sn, cs;
          sincosf(__fmul_rn(3.14159265358979323846f, nFreq), &sn, &cs);
          term = __fmul_rn(psd, which == 0 ? cs : sn);
        }
      }
      r[q] = term;
    }
  };
  float acc = 0.0f;
  fetch(0);
  int pp = 0;
  for (long long i0 = 0; i0 < bins; i0 += SER_TILE, pp ^= 1) {
#pragma unroll
    for (int q = 0; q < SER_TILE / 64; ++q) tile[pp][l + 64 * q] = r[q];
    __syncthreads();
    if (i0 + SER_TILE < bins) fetch(i0 + SER_TILE);
    const int cnt = bins - i0 < SER_TILE ? (int)(bins - i0) : SER_TILE;
    int e = 0;
    for (; e + 4 <= cnt; e += 4) {
      const float4 v = *reinterpret_cast<const float4 *>(&tile[pp][e]);
      acc = __fadd_rn(acc, v.x); acc = __fadd_rn(acc, v.y); acc = __fadd_rn(acc, v.z); acc = __fadd_rn(acc, v.w);
    }
    for (; e < cnt; ++e) acc = __fadd_rn(acc, tile[pp][e]);
  }
  if (l == 0) { res[which] = acc; if (which == 0) res[3] = smaxval; }
}

// large PSD frames (N > 16384): buf = window .* frame
// The nb frames F0 .. F0 + nb - 1 of a batch fold into their outputs (frame F belongs to output F / navg): per bin the
// powers are summed in frame order -- the sum an output's frames build is the same whatever the batch size --, an output
// whose last frame is in the batch is scaled and written (optional fftshift + dB), one that continues in the next batch
// leaves its partial sum in acc.
__global__ void frame_power_kernel(const float2 *__restrict__ X, long long n, long long F0, int nb, int navg,
                                   float *__restrict__ acc, float sc, int mode, float *__restrict__ out)
{
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float p = 0.0f;
    int r = (int)(F0 % navg);                                  // position of the batch's first frame inside its output
    long long o = F0 / navg;
    for (int f = 0; f < nb; ++f) {
      const float q = reinterpret_cast<const float *>(X + (long long)f * n)[i];     // |X_f[i]|^2, left there by the last pass
      p = r == 0 ? q : q + (f == 0 ? acc[i] : p);
      if (++r == navg) {
        const float ps = p * sc;
        if (mode == 0) out[o * n + i] = ps;
        else out[o * n + ((i + n / 2) & (n - 1))] = 10.0f * log10f(ps + 1e-8f);
        r = 0;
        ++o;
      }
    }
    if (r != 0) acc[i] = p;
  }
}

inline unsigned grid_for(long long n, int block) {
  long long g = (n + block - 1) / block;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

namespace sdk {

// ---- FAC (Default/GenericInspector/FACTab.cpp:181-246) -------------------------------------------
// X -> X conj(X) as a complex with zero imaginary part (:214-215)
__global__ void fac_power_kernel(const float2 *__restrict__ X, float2 *__restrict__ P, long long n)
{
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float2 v = X[i];
    P[i] = float2{__builtin_fmaf(v.y, v.y, v.x * v.x), 0.0f};
  }
}

// |.| of the first half (:220) and the running extrema over the view range (:222-235).  Non-negative
// floats order like their bit patterns, so atomicMax / atomicMin on the bits are exact and
// order-independent.  The inverse transform of a real spectrum is the conjugate of its forward
// transform, so the forward result is used as is (|.| is the same).
__global__ void fac_abs_kernel(const float2 *__restrict__ R, float *__restrict__ a, long long half,
                               long long view_start, long long view_end, unsigned *__restrict__ mx, unsigned *__restrict__ mn)
{
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < half; i += (long long)gridDim.x * blockDim.x) {
    const float2 v = R[i];
    const float m = __builtin_sqrtf(__builtin_fmaf(v.y, v.y, v.x * v.x));
    a[i] = m;
    if (view_start <= i && i < view_end) {
      atomicMax(mx, __float_as_uint(m));
      atomicMin(mn, __float_as_uint(m));
    }
  }
}

// SU_SPLPF_FEED(fac[i], a[i] / max, alpha) (:238-239)
__global__ void fac_ema_kernel(float *__restrict__ fac, const float *__restrict__ a, long long half, float alpha,
                               const unsigned *__restrict__ mx)
{
  const unsigned mb = *mx;
  // the running maximum starts at -inf (bits of +0 here: nothing seen yet); dividing by -inf gives -0
  const float m = mb == 0u ? -__builtin_inff() : __uint_as_float(mb);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < half; i += (long long)gridDim.x * blockDim.x)
    fac[i] += alpha * (a[i] / m - fac[i]);
}

// the passes of `batch` forward FFTs of n = 2^log2n points between the ping-pong buffers a and b; with first_in the first
// pass reads frames `first_stride` apart from there, windowed, instead of from a; with power_last the last pass leaves
// |X|^2 (floats at the head of every frame's slot) instead of X
static hipError_t fft_forward_from(const cf *first_in, long long first_stride, const float *window, void *a, void *b,
                                   int log2n, void **result, hipStream_t st, int batch, bool power_last = false)
{
  const long long n = 1ll << log2n;
  cf *src = reinterpret_cast<cf *>(a), *dst = reinterpret_cast<cf *>(b);
  long long ns = 1;
  int bits = log2n;
  bool first = first_in != nullptr;                            // the first pass reads the raw frames and windows them
  while (bits > 0) {
    const int rb = bits >= 4 ? 4 : bits;
    const long long nb = n >> rb;
    const dim3 grid(grid_for(nb, 256), (unsigned)batch), block(256);
    if (first) {
      switch (rb) {
        case 4: hipLaunchKernelGGL((fft_pass_kernel<16, true, false, false, true>), grid, block, 0, st, first_in, dst, n, ns, first_stride, window); break;
        case 3: hipLaunchKernelGGL((fft_pass_kernel<8, true, false, false, true>), grid, block, 0, st, first_in, dst, n, ns, first_stride, window); break;
        case 2: hipLaunchKernelGGL((fft_pass_kernel<4, true, false, false, true>), grid, block, 0, st, first_in, dst, n, ns, first_stride, window); break;
        default: hipLaunchKernelGGL((fft_pass_kernel<2, true, false, false, true>), grid, block, 0, st, first_in, dst, n, ns, first_stride, window); break;
      }
    } else if (power_last && bits == rb) {
      switch (rb) {
        case 4: hipLaunchKernelGGL((fft_pass_kernel<16, false, true, true>), grid, block, 0, st, src, dst, n, ns, 0ll, nullptr); break;
        case 3: hipLaunchKernelGGL((fft_pass_kernel<8, false, true, true>), grid, block, 0, st, src, dst, n, ns, 0ll, nullptr); break;
        case 2: hipLaunchKernelGGL((fft_pass_kernel<4, false, true, true>), grid, block, 0, st, src, dst, n, ns, 0ll, nullptr); break;
        default: hipLaunchKernelGGL((fft_pass_kernel<2, false, true, true>), grid, block, 0, st, src, dst, n, ns, 0ll, nullptr); break;
      }
    } else if (power_last && ns < 256 && nb % (256ll * grid.x) == 0) {   // a middle pass of the PSD whose stores would scatter
      switch (rb) {
        case 4: hipLaunchKernelGGL((fft_pass_kernel<16, false, false, true, true>), grid, block, 0, st, src, dst, n, ns, 0ll, nullptr); break;
        case 3: hipLaunchKernelGGL((fft_pass_kernel<8, false, false, true, true>), grid, block, 0, st, src, dst, n, ns, 0ll, nullptr); break;
        case 2: hipLaunchKernelGGL((fft_pass_kernel<4, false, false, true, true>), grid, block, 0, st, src, dst, n, ns, 0ll, nullptr); break;
        default: hipLaunchKernelGGL((fft_pass_kernel<2, false, false, true, true>), grid, block, 0, st, src, dst, n, ns, 0ll, nullptr); break;
      }
    } else if (power_last) {                                    // a middle pass of the PSD
      switch (rb) {
        case 4: hipLaunchKernelGGL((fft_pass_kernel<16, false, false, true>), grid, block, 0, st, src, dst, n, ns, 0ll, nullptr); break;
        case 3: hipLaunchKernelGGL((fft_pass_kernel<8, false, false, true>), grid, block, 0, st, src, dst, n, ns, 0ll, nullptr); break;
        case 2: hipLaunchKernelGGL((fft_pass_kernel<4, false, false, true>), grid, block, 0, st, src, dst, n, ns, 0ll, nullptr); break;
        default: hipLaunchKernelGGL((fft_pass_kernel<2, false, false, true>), grid, block, 0, st, src, dst, n, ns, 0ll, nullptr); break;
      }
    } else {
      switch (rb) {
        case 4: hipLaunchKernelGGL((fft_pass_kernel<16, false>), grid, block, 0, st, src, dst, n, ns, 0ll, nullptr); break;
        case 3: hipLaunchKernelGGL((fft_pass_kernel<8, false>), grid, block, 0, st, src, dst, n, ns, 0ll, nullptr); break;
        case 2: hipLaunchKernelGGL((fft_pass_kernel<4, false>), grid, block, 0, st, src, dst, n, ns, 0ll, nullptr); break;
        default: hipLaunchKernelGGL((fft_pass_kernel<2, false>), grid, block, 0, st, src, dst, n, ns, 0ll, nullptr); break;
      }
    }
    first = false;
    ns <<= rb;
    bits -= rb;
    cf *t = src; src = dst; dst = t;
  }
  *result = src;
  return hipGetLastError();
}

// forward FFT of n = 2^log2n points; a and b are ping-pong buffers (input in a); returns the
// buffer holding the result through *result
hipError_t fft_forward(void *a, void *b, int log2n, void **result, hipStream_t st, int batch)
{
  return fft_forward_from(nullptr, 0, nullptr, a, b, log2n, result, st, batch);
}

// PSD of frames too large for the in-LDS kernel: window -> Stockham passes through HBM -> power, navg frames accumulated
// per output.  The frames go through in batches of `batch` consecutive ones (every launch carries the batch in grid.y;
// one frame at a time, a 32768-point frame is six launches of 128 workgroups each -- 20 us per frame, 0.2 % of the HBM
// peak).  a, b: ping-pong buffers of batch * n complex; acc: n floats.
hipError_t psd_frames_large(int log2n, const void *x, long long hop, int navg, const float *window, float scale,
                            int mode, float *out, long long nout, void *a, void *b, float *acc, int batch, hipStream_t st)
{
  const long long n = 1ll << log2n, total = nout * navg;
  const float2 *xx = reinterpret_cast<const float2 *>(x);
  if (batch < 1) batch = 1;
  for (long long F0 = 0; F0 < total; F0 += batch) {
    const int nb = total - F0 < batch ? (int)(total - F0) : batch;
    // (the first pass reads the frames where they lie and writes into b; from there on the buffers alternate)
    void *res = nullptr;
    hipError_t e = fft_forward_from(reinterpret_cast<const cf *>(xx + F0 * hop), hop, window, a, b, log2n, &res, st, nb, true);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(frame_power_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st,
                       reinterpret_cast<const float2 *>(res), n, F0, nb, navg, acc, scale / (float)navg, mode, out);
  }
  return hipGetLastError();
}

// one FAC buffer: a holds the n input samples (destroyed), b is scratch, absbuf n/2 floats
hipError_t fac_feed(void *a, void *b, int log2n, float alpha, long long view_start, long long view_end, float *absbuf,
                    float *fac, unsigned *mx, unsigned *mn, hipStream_t st)
{
  const long long n = 1ll << log2n;
  void *res = nullptr;
  hipError_t e = fft_forward(a, b, log2n, &res, st);
  if (e != hipSuccess) return e;
  void *other = res == a ? b : a;
  hipLaunchKernelGGL(fac_power_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, reinterpret_cast<const float2 *>(res),
                     reinterpret_cast<float2 *>(other), n);
  // second transform: input in `other`, ping-pong with `res`
  void *res2 = nullptr;
  e = fft_forward(other, res, log2n, &res2, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fac_abs_kernel, dim3(grid_for(n / 2, 256)), dim3(256), 0, st, reinterpret_cast<const float2 *>(res2),
                     absbuf, n / 2, view_start, view_end, mx, mn);
  hipLaunchKernelGGL(fac_ema_kernel, dim3(grid_for(n / 2, 256)), dim3(256), 0, st, fac, absbuf, n / 2, alpha, mx);
  return hipGetLastError();
}

hipError_t window_pad(const void *data, long long len, long long alloc, void *buf, hipStream_t st)
{
  hipLaunchKernelGGL(window_pad_kernel, dim3(grid_for(alloc, 256)), dim3(256), 0, st,
                     reinterpret_cast<const float2 *>(data), len, alloc, reinterpret_cast<float2 *>(buf));
  return hipGetLastError();
}

// the carrier estimator's last step on the device: res = {acc.re, acc.im, ...} of spectrum_centroid; value = SU_C_ARG(acc)
// wrapped to (-pi, pi] (Tasks/CarrierDetector.cpp:134-137) as cycles per sample
__global__ void carrier_norm_kernel(const float *__restrict__ res, float *__restrict__ value)
{
  float p = atan2f(res[1], res[0]);
  if (p > 3.14159265358979323846f) p -= 6.28318530717958647692f;
  value[0] = p * 0.15915494309189533577f;
}
hipError_t carrier_norm(const float *res, float *value, hipStream_t st)
{
  hipLaunchKernelGGL(carrier_norm_kernel, dim3(1), dim3(1), 0, st, res, value);
  return hipGetLastError();
}

hipError_t spectrum_centroid(void *buf, long long alloc, long long lo, long long hi, float *mirror, long long bins,
                             long long delta, int with_dispersion, float *blk_max, long long *blk_idx,
                             int nblk, float *res, hipStream_t st)
{
  hipLaunchKernelGGL(power_argmax_kernel, dim3(nblk), dim3(256), 0, st, reinterpret_cast<float2 *>(buf), alloc, lo, hi,
                     mirror, blk_max, blk_idx);
  // res[0] = acc.re, res[1] = acc.im, res[2] = variance accumulator, res[3] = max value, res[4] = total energy
  if (with_dispersion)
    hipLaunchKernelGGL(serial_kahan_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<const float2 *>(buf), bins, res + 4);
  hipLaunchKernelGGL(serial_centroid_kernel, dim3(with_dispersion ? 3 : 2), dim3(64), 0, st, reinterpret_cast<const float2 *>(buf),
                     alloc, nblk, blk_max, blk_idx, res + 4, bins, delta, res);
  return hipGetLastError();
}

}  // namespace sdk
