// loops.hip -- element-wise demodulators (T5/T7/T11) and the per-channel recurrences
// (K6 Costas, K7 PLL, K8 Gardner clock recovery, K9 AGC) for gfx950.  SPEC.md sections E-H.
//
// The recurrences are serial in time and non-linear (sign(), wrap, data-dependent strobes),
// so the only exact parallel axis is the channel: ONE LANE PER CHANNEL, 64 channels per
// wavefront, state in registers for the whole block.  Rows are addressed through a two-stride
// view (kernels.hpp: View); with the time-major layout [time][channel] a wavefront's access
// to one time step is a single contiguous 512-byte transaction, and CHUNK steps are
// prefetched one chunk ahead of the serial arithmetic.
// Compiled with -ffp-contract=off: the arithmetic is the SPEC's fixed binary32 sequence and
// matches the CPU oracle bit for bit.
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.hpp"
#include "tuning.hpp"
#include "sd_math.hpp"
#include "loops_dev.hpp"

namespace {

// ---------------------------------------------------------------------------------------
// T5: QuadDemodTask::work  dest[p] = j/pi * arg(x[p] conj(x[p-1]))
__global__ void quad_demod_kernel(const float2 *__restrict__ x, sdk::View xv, float2 *__restrict__ y, sdk::View yv,
                                  int nchan, long long len, const float2 *__restrict__ prev, int first,
                                  float2 *__restrict__ prev_out)
{
  __builtin_amdgcn_s_setprio(3);   // ahead of the resident recurrence wavefronts (see chan_fir_kernel)
  const float k = 0.318309886183790671538f;
  const long long total = len * nchan;
  const bool time_major = xv.cs < xv.ms;                 // consecutive threads follow the unit stride
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    long long p; int c;
    if (time_major) { p = t / nchan; c = (int)(t - p * nchan); }
    else            { c = (int)(t / len); p = t - (long long)c * len; }
    const float2 *xr = x + (long long)c * xv.cs;
    const float2 v = xr[p * xv.ms];
    float2 out;
    if (p == 0 && first) {
      out = float2{0.0f, 0.0f};
    } else {
      const float2 pv = (p == 0) ? prev[c] : xr[(p - 1) * xv.ms];
      const c32 d = sd::cmul_conj(c32{v.x, v.y}, c32{pv.x, pv.y});
      out = float2{0.0f, k * sd::atan2_(d.im, d.re)};
    }
    y[(long long)c * yv.cs + p * yv.ms] = out;
    if (prev_out != nullptr && p == len - 1) prev_out[c] = v;
  }
}

// T7: DelayedConjTask::work
__global__ void delayed_conj_kernel(const float2 *__restrict__ x, float2 *__restrict__ y, long long len, long long delay)
{
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < len;
       p += (long long)gridDim.x * blockDim.x) {
    float2 out = float2{0.0f, 0.0f};
    if (p >= delay) {
      const float2 v = x[p], pv = x[p - delay];
      const float mag  = __builtin_sqrtf(sd::fma_(pv.x, pv.x, pv.y * pv.y));
      const float kinv = 1.0f / (mag + 1e-3f);
      const c32 d = sd::cmul_conj(c32{v.x, v.y}, c32{pv.x, pv.y});
      out = float2{kinv * d.re, kinv * d.im};
    }
    y[p] = out;
  }
}

// T11: HistogramFeeder::work
__global__ void histogram_kernel(const float2 *__restrict__ x, long long len, int space, float *__restrict__ out)
{
  const long long nout = space == 2 ? len - 1 : len;
  for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < nout;
       q += (long long)gridDim.x * blockDim.x) {
    float r;
    if (space == 0) {
      const float2 v = x[q];
      r = __builtin_sqrtf(sd::fma_(v.x, v.x, v.y * v.y));
    } else if (space == 1) {
      const float2 v = x[q];
      r = sd::atan2_(v.y, v.x);
    } else {
      const float2 v = x[q + 1], pv = x[q];
      const c32 d = sd::cmul_conj(c32{v.x, v.y}, c32{pv.x, pv.y});
      r = sd::atan2_(d.im, d.re);
    }
    out[q] = r;
  }
}

// ---------------------------------------------------------------------------------------
// T8: WaveSampler::sampleManual (Tasks/WaveSampler.cpp:96-175): fractional-boundary boxcar per
// symbol.  Symbols are independent except for `prev`, the last (weighted) sample of the previous
// symbol, which each thread recomputes from the reference's own expressions -> one thread per
// symbol, same double-precision index arithmetic, same binary32 accumulation order.
__device__ __forceinline__ float2 manual_fetch(const float2 *__restrict__ data, long long length, long long i,
                                               long long iStart, long long iEnd, float tStart, float tEnd)
{
  if (i >= 0 && i < length) {
    const float2 d = data[i];
    if (i == iStart) return float2{tStart * d.x, tStart * d.y};
    if (i == iEnd) return float2{tEnd * d.x, tEnd * d.y};
    return d;
  }
  return float2{0.0f, 0.0f};
}

__global__ void sample_manual_kernel(const float2 *__restrict__ data, long long length, double delta,
                                     double sampOffset, double symbolSync, int space, float2 *__restrict__ out,
                                     long long nout)
{
  const float deltaInv = 1.f / (float)delta;
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < nout;
       p += (long long)gridDim.x * blockDim.x) {
    float2 prev = float2{0.0f, 0.0f};
    if (p > 0) {                                            // last sample of symbol p-1
      const double s0 = ((double)(p - 1) - sampOffset) * delta + symbolSync;
      const double e0 = s0 + delta;
      const long long is0 = (long long)floor(s0), ie0 = (long long)ceil(e0);
      const float ts0 = (float)(1 - (s0 - (double)is0)), te0 = (float)(1 - ((double)ie0 - e0));
      prev = manual_fetch(data, length, ie0, is0, ie0, ts0, te0);
    }
    const double start = ((double)p - sampOffset) * delta + symbolSync;
    const double end = start + delta;
    const long long iStart = (long long)floor(start), iEnd = (long long)ceil(end);
    const float tStart = (float)(1 - (start - (double)iStart)), tEnd = (float)(1 - ((double)iEnd - end));
    float ar = 0, ai = 0;
    for (long long i = iStart; i <= iEnd; ++i) {
      const float2 x = manual_fetch(data, length, i, iStart, iEnd, tStart, tEnd);
      if (space == 0) {
        ar = ar + sd::fma_(x.y, x.y, x.x * x.x);
      } else {
        const c32 d = sd::cmul_conj(c32{x.x, x.y}, c32{prev.x, prev.y});
        ar = ar + d.re; ai = ai + d.im;
      }
      prev = x;
    }
    out[p] = space == 0 ? float2{__builtin_sqrtf(deltaInv * ar), 0.0f} : float2{deltaInv * ar, deltaInv * ai};
  }
}

// ---------------------------------------------------------------------------------------
// T8: WaveSampler ZERO_CROSSING (Tasks/WaveSampler.cpp:215-292) over a whole capture.
// The reference works in blocks of 4096 input samples and -- because sampleZeroCrossing() never
// writes prevVar / prevSample back -- every block restarts from prevVar = -1, prevSample = 0; only
// lastZc carries over.  So blocks are independent up to "where was the last crossing before me":
//   zc_var_kernel    var[p] of every sample (parallel),
//   zc_scan_kernel   one thread per block: position of its last crossing,
//   zc_emit_kernel   one thread per block: lastZc from the blocks before it, run lengths ->
//                    round(samples * bnor) symbols into the block's own 4096-slot segment,
//   zc_compact_kernel  segments -> contiguous output (offsets = prefix sum of the counts).
constexpr int ZC_BLOCK = 4096;   // SIGDIGGER_WAVESAMPLER_FEEDER_BLOCK_LENGTH

__global__ void zc_var_kernel(const float2 *__restrict__ data, long long length, int space, int amplitude,
                              float2 thr, float2 ang, float *__restrict__ var)
{
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < length;
       p += (long long)gridDim.x * blockDim.x) {
    const float2 x = data[p];
    float v;
    if (space == 0) {
      float t;
      if (amplitude) { v = sd::fma_(x.y, x.y, x.x * x.x);      t = sd::fma_(thr.y, thr.y, thr.x * thr.x); }
      else           { v = sd::fma_(-x.y, ang.y, x.x * ang.x); t = sd::fma_(-thr.y, ang.y, thr.x * ang.x); }
      v = v - t;
    } else if (space == 1) {
      const float re = sd::fma_(-x.y, ang.y, x.x * ang.x), im = sd::fma_(x.y, ang.x, x.x * ang.y);
      v = sd::atan2_(im, re);
    } else {
      const float2 pv = (p % ZC_BLOCK) == 0 ? float2{0.0f, 0.0f} : data[p - 1];
      const c32 d = sd::cmul_conj(c32{-x.y, x.x}, c32{pv.x, pv.y});      // (SU_I * x) * conj(prev)
      v = sd::atan2_(d.im, d.re);
    }
    var[p] = v;
  }
}

// crossing test of one sample; prevVar is updated by the caller on a hit
__device__ __forceinline__ bool zc_hit(float var, float prevVar, bool last)
{
  return ((var > 0 || var < 0) || last) && (var * prevVar < 0 || last);
}

__global__ void zc_scan_kernel(const float *__restrict__ var, long long length, long long nblocks,
                               long long *__restrict__ last_pos)
{
  const long long b = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  const long long p0 = b * ZC_BLOCK, p1 = (p0 + ZC_BLOCK < length) ? p0 + ZC_BLOCK : length;
  const bool last = (b == nblocks - 1);
  float prevVar = -1.0f;
  long long lp = -1;
  for (long long p = p0; p < p1; ++p) {
    const float v = var[p];
    if (zc_hit(v, prevVar, last)) { lp = p; prevVar = v; }
  }
  last_pos[b] = lp;
}

__global__ void zc_emit_kernel(const float *__restrict__ var, long long length, long long nblocks, float bnor,
                               const long long *__restrict__ last_pos, unsigned char *__restrict__ seg,
                               unsigned *__restrict__ count)
{
  const long long b = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  const long long p0 = b * ZC_BLOCK, p1 = (p0 + ZC_BLOCK < length) ? p0 + ZC_BLOCK : length;
  const bool last = (b == nblocks - 1);
  long long lastZc = 0;
  for (long long q = b - 1; q >= 0; --q) if (last_pos[q] >= 0) { lastZc = last_pos[q]; break; }
  unsigned char *out = seg + b * ZC_BLOCK;
  float prevVar = -1.0f;
  int i = 0;
  for (long long p = p0; p < p1; ++p) {
    const float v = var[p];
    if (zc_hit(v, prevVar, last)) {
      const long long samples = p - lastZc;
      long long symbols = (long long)roundf((float)samples * bnor);
      const unsigned char s = v > 0;
      while (symbols-- > 0 && i < ZC_BLOCK) out[i++] = s;
      lastZc = p;
      prevVar = v;
    }
  }
  count[b] = (unsigned)i;
}

__global__ void zc_compact_kernel(const unsigned char *__restrict__ seg, const unsigned *__restrict__ count,
                                  const unsigned long long *__restrict__ offset, unsigned char *__restrict__ out)
{
  const long long b = blockIdx.x;
  const unsigned n = count[b];
  const unsigned char *src = seg + b * ZC_BLOCK;
  unsigned char *dst = out + offset[b];
  for (unsigned i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

// WaveSampler::sampleGardner, FREQUENCY space (Tasks/WaveSampler.cpp:188-196): x[p] conj(x[p-1])
__global__ void conj_prev_kernel(const float2 *__restrict__ x, float2 *__restrict__ y, long long len, float2 prev0)
{
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < len;
       p += (long long)gridDim.x * blockDim.x) {
    const float2 a = x[p], b = p > 0 ? x[p - 1] : prev0;
    const c32 d = sd::cmul_conj(c32{a.x, a.y}, c32{b.x, b.y});
    y[p] = float2{d.re, d.im};
  }
}


// Where a recurrence kernel's one wavefront sits: workgroups are dealt round the chip's eight XCDs in launch order, so the
// only workgroup of a 64-channel launch always lands on XCD 0 -- and so do the other two recurrence stages', each taking a
// workgroup slot (registers, the clock kernel's 32 KB of LDS) away from the SAME 32 CUs.  A transform launch planned for
// one round of the chip (the channeliser: 745 workgroups, 93-94 per XCD for 96 slots) then finds XCD 0 two or three
// short and runs a second round there: 131 instead of 87 us whenever clock and Costas wavefronts were both resident
// (rocprofv3 kernel trace, profiles/r04_inpipe_penalty.txt).  With `xcd` >= 0 the launch has eight workgroups per block
// of channels and only number `xcd` of each eight does the work: the stages sit on different XCDs.
__device__ __forceinline__ int serial_block(int xcd, bool &mine)
{
  mine = xcd < 0 || (int)(blockIdx.x & 7) == xcd;
  return xcd < 0 ? (int)blockIdx.x : (int)(blockIdx.x >> 3);
}

// A single wavefront issues roughly one instruction every 4-5 cycles, so the cost of a serial
// recurrence is its dynamic instruction count per sample: the loop kind is a template
// parameter (no per-sample scalar branching), full chunks run without bounds checks and only
// the last partial chunk is guarded.
template <int KIND, int ORDER, bool GAIN1>
__global__ __launch_bounds__(64) void costas_kernel(sdk::CostasParams p, sdk::CostasState s, int nchan,
                                                    const float2 *__restrict__ x, sdk::View xv,
                                                    float2 *__restrict__ y, sdk::View yv, long long len, int xcd)
{
  bool mine;
  const int c = serial_block(xcd, mine) * 64 + threadIdx.x;
  if (!mine || c >= nchan) return;
  // (measured both ways, round 6: these parameters in VGPRs instead of the kernel argument's SGPRs -- 21.08 against 20.5 ms per
  // 64 x 262144 samples; the gang's per-lane parameters made wave-uniform -- 828 against 747 us per 64 x 8192.  Each form keeps
  // the operand kind its instruction selection was tuned with.)
  CostasRegs<ORDER> r;
  r.phase = s.phase[c];
  r.omega = s.omega[c];
#pragma unroll
  for (int i = 1; i <= ORDER; ++i) {
    r.xh[i] = c32{s.xh[((i - 1) * 2 + 0) * nchan + c], s.xh[((i - 1) * 2 + 1) * nchan + c]};
    r.yh[i] = c32{s.yh[((i - 1) * 2 + 0) * nchan + c], s.yh[((i - 1) * 2 + 1) * nchan + c]};
  }
  const uint32_t xo = (uint32_t)((long long)c * xv.cs * 8), yo = (uint32_t)((long long)c * yv.cs * 8);
  const long long yms = yv.ms;
  // time-major rows of a 64-channel bank (pitch 64) or a single row (pitch 1): with the pitch a
  // compile-time constant every access of a chunk is scalar base + immediate offset (no 64-bit
  // vector address arithmetic per sample: 6.44 -> 6.03 ms per block)
  if (xv.ms == 64 && yms == 64) {
    stream_row(x, 64, xo, len, [&](long long m, float2 v) {
      st_elem(y, m * 64, yo, costas_step<KIND, ORDER, GAIN1>(p, r, v));
    });
  } else if (xv.ms == 1 && yms == 1) {
    stream_row(x, 1, xo, len, [&](long long m, float2 v) {
      st_elem(y, m, yo, costas_step<KIND, ORDER, GAIN1>(p, r, v));
    });
  } else {
    stream_row(x, xv.ms, xo, len, [&](long long m, float2 v) {
      st_elem(y, m * yms, yo, costas_step<KIND, ORDER, GAIN1>(p, r, v));
    });
  }
  s.phase[c] = r.phase;
  s.omega[c] = r.omega;
#pragma unroll
  for (int i2 = 1; i2 <= ORDER; ++i2) {
    s.xh[((i2 - 1) * 2 + 0) * nchan + c] = r.xh[i2].re; s.xh[((i2 - 1) * 2 + 1) * nchan + c] = r.xh[i2].im;
    s.yh[((i2 - 1) * 2 + 0) * nchan + c] = r.yh[i2].re; s.yh[((i2 - 1) * 2 + 1) * nchan + c] = r.yh[i2].im;
  }
}


__global__ __launch_bounds__(64) void pll_kernel(float alpha, float beta, sdk::PllState s, int nchan,
                                                 const float2 *__restrict__ x, sdk::View xv,
                                                 float2 *__restrict__ y, sdk::View yv, long long len)
{
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= nchan) return;
  uint32_t phase = s.phase[c];
  float omega = s.omega[c];
  const uint32_t xo = (uint32_t)((long long)c * xv.cs * 8), yo = (uint32_t)((long long)c * yv.cs * 8);
  const long long yms = yv.ms;
  if (xv.ms == 64 && yms == 64) {                          // constant pitch: immediate offsets (see costas_kernel)
    stream_row(x, 64, xo, len, [&](long long m, float2 v) { st_elem(y, m * 64, yo, pll_step(alpha, beta, phase, omega, v)); });
  } else {
    stream_row(x, xv.ms, xo, len, [&](long long m, float2 v) {
      st_elem(y, m * yms, yo, pll_step(alpha, beta, phase, omega, v));
    });
  }
  s.phase[c] = phase;
  s.omega[c] = omega;
}

// (clock_kernel: further down, next to the crossing-driven stream it shares with the gangs)

// ---------------------------------------------------------------------------------------
// K9: AGC (SPEC.md section H).  Only the fast / slow level trackers are a recurrence.  The dB
// conversion, the sliding maximum over the magnitude history (a pure function of the last H
// magnitudes: the oracle's "peak", which it maintains by rescanning whenever the old peak
// leaves the history, is always exactly that maximum) and the gain applied to the delayed
// sample are feed-forward, so the bank runs as
//   (1) agc_mag_kernel   : db[m][c]   = 10 log10(|x|^2 + 1e-8)                       parallel
//   (2) agc_peak_kernel  : peak[m][c] = max(db[m-H+1 .. m][c])  (history for m < H-1)  parallel
//   (3) agc_level_kernel : fast / slow levels + hang counter -> lvl[m][c] in place     1 lane/channel
//   (4) agc_apply_kernel : y = x[m - delay] * 10^(lvl*(slope-1)/20) * 0.7              parallel
//   (5) agc_state_kernel : carries the last H-1 magnitudes and `delay` inputs to the next block
__global__ void agc_mag_kernel(const float2 *__restrict__ x, sdk::View xv, int nchan, long long len,
                               float *__restrict__ db)
{
  __builtin_amdgcn_s_setprio(3);   // ahead of the resident recurrence wavefronts (see chan_fir_kernel)
  const long long total = len * nchan;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const long long m = t / nchan;
    const int c = (int)(t - m * nchan);
    const float2 v = x[(long long)c * xv.cs + m * xv.ms];
    const float pw = sd::fma_(v.x, v.x, v.y * v.y) + 1e-8f;
    db[t] = 3.01029995663981195f * sd::log2_(pw);
  }
}

// tile = 64 channels x TM time steps; the TM + H - 1 magnitudes each channel needs are staged once
// in LDS (coalesced rows of 64 floats), then every output takes its maximum over H LDS reads
// (instead of H global loads per output).  max() is exact and order-independent.
constexpr int PEAK_TM = 128;
__global__ __launch_bounds__(256) void agc_peak_kernel(const float *__restrict__ db, const float *__restrict__ hist,
                                                       int nchan, long long len, int H, float *__restrict__ peak)
{
  __builtin_amdgcn_s_setprio(3);   // ahead of the resident recurrence wavefronts (see chan_fir_kernel)
  __shared__ float tile[PEAK_TM + 63][64];
  const long long m0 = (long long)blockIdx.x * PEAK_TM;
  const int c0 = blockIdx.y * 64;
  const int lane = threadIdx.x & 63, rowt = threadIdx.x >> 6;          // 4 rows of 64 channels per pass
  const int c = c0 + lane;
  const int rows = PEAK_TM + H - 1;                                    // tile row r <-> time m0 - (H-1) + r
  for (int r = rowt; r < rows; r += 4) {
    const long long m = m0 - (H - 1) + r;
    float v = -__builtin_inff();
    if (c < nchan && m < len) v = m >= 0 ? db[m * nchan + c] : hist[(m + (H - 1)) * nchan + c];
    tile[r][lane] = v;
  }
  __syncthreads();
  if (c >= nchan) return;
  for (int t = rowt; t < PEAK_TM; t += 4) {
    const long long m = m0 + t;
    if (m >= len) break;
    float pk = tile[t + H - 1][lane];
    for (int i = 1; i < H; ++i) { const float v = tile[t + H - 1 - i][lane]; pk = pk > v ? pk : v; }
    peak[m * nchan + c] = pk;
  }
}

// single-channel banks (the live analyzer's inspectors): threads along time, each output scans its own H
// magnitudes (coalesced across the threads; the maximum does not depend on the order)
__global__ __launch_bounds__(256) void agc_peak1_kernel(const float *__restrict__ db, const float *__restrict__ hist,
                                                        long long len, int H, float *__restrict__ peak)
{
  __builtin_amdgcn_s_setprio(3);
  for (long long m = blockIdx.x * (long long)blockDim.x + threadIdx.x; m < len; m += (long long)gridDim.x * blockDim.x) {
    float pk = db[m];
    for (int i = 1; i < H; ++i) {
      const long long q = m - i;
      const float v = q >= 0 ? db[q] : hist[q + (H - 1)];
      pk = pk > v ? pk : v;
    }
    peak[m] = pk;
  }
}

// (1)+(2) for the items of a gang: a workgroup takes tiles of 256 outputs of its item, stages the 256 + H - 1
// magnitudes they look at in LDS (the history stands in before the row's start) and scans them in the order
// agc_peak1_kernel does.  Same operations per value, so the same bits.
__global__ __launch_bounds__(256) void agc_pre_items_kernel(const sdk::AgcPreItem *__restrict__ items)
{
  __builtin_amdgcn_s_setprio(3);
  __shared__ float t[256 + 63];
  const sdk::AgcPreItem it = items[blockIdx.y];
  const float2 *x = static_cast<const float2 *>(it.x);
  const int hl = it.H - 1;
  for (long long m0 = (long long)blockIdx.x * 256; m0 < it.len; m0 += (long long)gridDim.x * 256) {
    for (int r = threadIdx.x; r < 256 + hl; r += 256) {
      const long long q = m0 - hl + r;
      float v = 0.f;
      if (q < 0) v = it.hist[q + hl];
      else if (q < it.len) {
        const float2 s = x[q];
        v = 3.01029995663981195f * sd::log2_(sd::fma_(s.x, s.x, s.y * s.y) + 1e-8f);
        if (q >= m0) it.db[q] = v;
      }
      t[r] = v;
    }
    __syncthreads();
    const long long m = m0 + threadIdx.x;
    if (m < it.len) {
      float pk = t[threadIdx.x + hl];
      for (int i = 1; i <= hl; ++i) {
        const float v = t[threadIdx.x + hl - i];
        pk = pk > v ? pk : v;
      }
      it.peak[m] = pk;
    }
    __syncthreads();
  }
}

// (5) for the items of a gang: lane i carries delay-line slot i and history slot i (both at most 64 long)
__global__ __launch_bounds__(64) void agc_state_items_kernel(const sdk::AgcStateItem *__restrict__ items)
{
  const sdk::AgcStateItem it = items[blockIdx.x];
  const float2 *x = static_cast<const float2 *>(it.x);
  const int i = threadIdx.x, hl = it.H - 1;
  const long long src = (long long)i + it.len;
  float2 d = {0.f, 0.f};
  float h = 0.f;
  if (i < it.delay) d = src < it.delay ? float2{it.delay_line[src * 2 + 0], it.delay_line[src * 2 + 1]} : x[(src - it.delay) * it.xs];
  if (i < hl) h = src < hl ? it.hist[src] : it.db[(src - hl) * it.dbs];
  __syncthreads();                                        // every slot is read before any is overwritten
  if (i < it.delay) { it.delay_line[i * 2 + 0] = d.x; it.delay_line[i * 2 + 1] = d.y; }
  if (i < hl) it.hist[i] = h;
}

__global__ __launch_bounds__(64) void agc_level_kernel(sdk::AgcParams p, sdk::AgcState s, int nchan,
                                                       long long len, float *__restrict__ peak, int xcd)
{
  bool mine;
  const int c = serial_block(xcd, mine) * 64 + threadIdx.x;
  if (!mine || c >= nchan) return;
  unsigned hang_n = s.hang_n[c];
  float fast = s.fast_level[c], slow = s.slow_level[c];
  const uint32_t lo = (uint32_t)c * 4u;
  // parameters into registers: selecting between two fields of the by-value kernel argument
  // otherwise compiles to an address select + a kernarg load + s_waitcnt vmcnt(0) PER SAMPLE
  const float far = p.fast_alpha_rise, faf = p.fast_alpha_fall, sar = p.slow_alpha_rise, saf = p.slow_alpha_fall;
  const float knee = p.knee;
  const unsigned hang_max = p.hang_max;
  auto level = [&](float pk) {
    float d = pk - fast;
    const float fa = d > 0.0f ? far : faf;
    fast = sd::fma_(fa, d, fast);
    d = pk - slow;
    // hang logic, branch-free: a rise resets the counter, a fall only after hang_max quiet samples
    const bool rise = d > 0.0f;
    const bool fall = !rise && hang_n >= hang_max;
    const float sa = rise ? sar : saf;
    const float upd = sd::fma_(sa, d, slow);
    slow = (rise || fall) ? upd : slow;
    hang_n = rise ? 0u : (fall ? hang_n : hang_n + 1u);
    float lvl = fast > slow ? fast : slow;
    if (lvl < knee) lvl = knee;
    return lvl;
  };
  if (nchan == 64) stream_row(peak, 64, lo, len, [&](long long m, float pk) { st_elem(peak, m * 64, lo, level(pk)); });
  else if (nchan == 1) stream_row(peak, 1, lo, len, [&](long long m, float pk) { st_elem(peak, m, lo, level(pk)); });
  else stream_row(peak, (long long)nchan, lo, len, [&](long long m, float pk) { st_elem(peak, m * (long long)nchan, lo, level(pk)); });
  s.hang_n[c] = hang_n; s.fast_level[c] = fast; s.slow_level[c] = slow;
}

__global__ void agc_apply_kernel(sdk::AgcParams p, const float *__restrict__ delay_line, int nchan,
                                 const float2 *__restrict__ x, sdk::View xv, float2 *__restrict__ y, sdk::View yv,
                                 long long len, const float *__restrict__ lvl)
{
  __builtin_amdgcn_s_setprio(3);   // ahead of the resident recurrence wavefronts (see chan_fir_kernel)
  const long long total = len * nchan;
  const long long delay = p.delay_line_size;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const long long m = t / nchan;
    const int c = (int)(t - m * nchan);
    float2 xd;
    if (m >= delay) {
      xd = x[(long long)c * xv.cs + (m - delay) * xv.ms];
    } else {
      xd = float2{delay_line[(m * 2 + 0) * nchan + c], delay_line[(m * 2 + 1) * nchan + c]};
    }
    const float g_db = lvl[t] * (p.gain_slope - 1.0f);
    const float g = sd::exp2_(g_db * 0.166096404744368117f) * 0.7f;
    y[(long long)c * yv.cs + m * yv.ms] = float2{xd.x * g, xd.y * g};
  }
}

// delay line <- last `delay` samples of [delay line ; x];  magnitude history <- last H-1 values of
// [history ; db]   (one lane per channel; both are tiny)
__global__ void agc_apply_items_kernel(const sdk::AgcApplyItem *__restrict__ items)
{
  __builtin_amdgcn_s_setprio(3);
  const sdk::AgcApplyItem it = items[blockIdx.y];
  const long long delay = it.p.delay_line_size;
  const float2 *x = reinterpret_cast<const float2 *>(it.x);
  float2 *y = reinterpret_cast<float2 *>(it.y);
  for (long long m = it.m0 + blockIdx.x * (long long)blockDim.x + threadIdx.x; m < it.m1; m += (long long)gridDim.x * blockDim.x) {
    const float2 xd = m >= delay ? x[m - delay] : float2{it.delay_line[m * 2 + 0], it.delay_line[m * 2 + 1]};
    const float g_db = it.lvl[m] * (it.p.gain_slope - 1.0f);
    const float g = sd::exp2_(g_db * 0.166096404744368117f) * 0.7f;
    y[m] = float2{xd.x * g, xd.y * g};
  }
}

__global__ __launch_bounds__(64) void agc_state_kernel(float *delay_line, float *hist, int nchan, int delay, int H,
                                                       const float2 *__restrict__ x, sdk::View xv,
                                                       const float *__restrict__ db, long long len)
{
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= nchan) return;
  float2 tmp[64];
#pragma unroll 1
  for (int i = 0; i < delay; ++i) {
    const long long src = (long long)i + len;             // index into [delay line ; x]
    tmp[i] = src < delay ? float2{delay_line[(src * 2 + 0) * nchan + c], delay_line[(src * 2 + 1) * nchan + c]}
                         : x[(long long)c * xv.cs + (src - delay) * xv.ms];
  }
#pragma unroll 1
  for (int i = 0; i < delay; ++i) {
    delay_line[(i * 2 + 0) * nchan + c] = tmp[i].x;
    delay_line[(i * 2 + 1) * nchan + c] = tmp[i].y;
  }
  const int hl = H - 1;
  float th[64];
#pragma unroll 1
  for (int i = 0; i < hl; ++i) {
    const long long src = (long long)i + len;             // index into [history ; db]
    th[i] = src < hl ? hist[src * nchan + c] : db[(src - hl) * nchan + c];
  }
#pragma unroll 1
  for (int i = 0; i < hl; ++i) hist[(long long)i * nchan + c] = th[i];
}

inline unsigned grid_for(long long n, int block) {
  long long g = (n + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (unsigned)g;
}


__global__ __launch_bounds__(64) void clock_kernel(sdk::ClockParams p, sdk::ClockState s, int nchan,
                                                   const float2 *__restrict__ x, sdk::View xv, long long len,
                                                   float2 *__restrict__ sym, long long sym_stride,
                                                   uint32_t *__restrict__ count, int mode, int xcd, int ring_steps)
{
  __shared__ float2 lds[RING * 64];                           // clock_ring's two tiles; clock_stream_tm uses the first
  bool mine;
  const int cc = serial_block(xcd, mine) * 64 + threadIdx.x;
  if (!mine) return;
  const bool live = cc < nchan;                               // idle lanes take part in the wave-wide steps with length 0
  const int c = live ? cc : 0;
  ClockRegs r;
  r.phi = s.phi[c]; r.bnor = s.bnor[c];
  r.halfcycle = s.halfcycle[c];
  r.prev = float2{s.prev[c], s.prev[nchan + c]};
  r.x0 = float2{s.x0[c], s.x0[nchan + c]};
  r.x1 = float2{s.x1[c], s.x1[nchan + c]};
  r.x2 = float2{s.x2[c], s.x2[nchan + c]};
  r.n = count[c];
  const uint32_t xo = (uint32_t)((long long)c * xv.cs * 8);
  float2 *out = sym + (long long)c * sym_stride;
  const long long mylen = live ? len : 0;
  if (mode == 2) {
    if (xv.ms == 64) clock_ring(x, 64, xo, len, live, ring_steps, p, r, out, lds);
    else if (xv.ms == 1) clock_ring(x, 1, xo, len, live, ring_steps, p, r, out, lds);
    else clock_ring(x, xv.ms, xo, len, live, ring_steps, p, r, out, lds);
  } else if (mode == 1) {
    if (xv.ms == 64) clock_stream_tm<false>(x, 64, xo, mylen, p, r, out, lds);
    else if (xv.ms == 1) clock_stream_tm<false>(x, 1, xo, mylen, p, r, out, lds);
    else clock_stream_tm<false>(x, xv.ms, xo, mylen, p, r, out, lds);
  } else {
    if (xv.ms == 64) stream_row(x, 64, xo, mylen, [&](long long, float2 v) { clock_step(p, r, v, out); });
    else if (xv.ms == 1) stream_row(x, 1, xo, mylen, [&](long long, float2 v) { clock_step(p, r, v, out); });
    else stream_row(x, xv.ms, xo, mylen, [&](long long, float2 v) { clock_step(p, r, v, out); });
  }
  if (!live) return;
  s.phi[c] = r.phi; s.bnor[c] = r.bnor; s.halfcycle[c] = r.halfcycle;
  s.prev[c] = r.prev.x; s.prev[nchan + c] = r.prev.y;
  s.x0[c] = r.x0.x; s.x0[nchan + c] = r.x0.y;
  s.x1[c] = r.x1.x; s.x1[nchan + c] = r.x1.y;
  s.x2[c] = r.x2.x; s.x2[nchan + c] = r.x2.y;
  count[c] = r.n;
}

}  // namespace

namespace sdk {

hipError_t quad_demod_batch(const void *x, View xs, void *y, View ys, int nchan, long long len,
                            const void *prev, int first, void *prev_out, hipStream_t st)
{
  if (len <= 0 || nchan <= 0) return hipSuccess;
  hipLaunchKernelGGL(quad_demod_kernel, dim3(grid_for(len * nchan, 256)), dim3(256), 0, st,
                     reinterpret_cast<const float2 *>(x), xs, reinterpret_cast<float2 *>(y), ys, nchan, len,
                     reinterpret_cast<const float2 *>(prev), first, reinterpret_cast<float2 *>(prev_out));
  return hipGetLastError();
}

hipError_t delayed_conj_bulk(const void *x, void *y, long long len, long long delay, hipStream_t st)
{
  if (len <= 0) return hipSuccess;
  hipLaunchKernelGGL(delayed_conj_kernel, dim3(grid_for(len, 256)), dim3(256), 0, st,
                     reinterpret_cast<const float2 *>(x), reinterpret_cast<float2 *>(y), len, delay);
  return hipGetLastError();
}

hipError_t histogram_feed_bulk(const void *x, long long len, int space, float *out, hipStream_t st)
{
  if (len <= 0) return hipSuccess;
  hipLaunchKernelGGL(histogram_kernel, dim3(grid_for(len, 256)), dim3(256), 0, st,
                     reinterpret_cast<const float2 *>(x), len, space, out);
  return hipGetLastError();
}

hipError_t conj_prev_bulk(const void *x, void *y, long long len, float prev_re, float prev_im, hipStream_t st)
{
  if (len <= 0) return hipSuccess;
  hipLaunchKernelGGL(conj_prev_kernel, dim3(grid_for(len, 256)), dim3(256), 0, st, reinterpret_cast<const float2 *>(x),
                     reinterpret_cast<float2 *>(y), len, float2{prev_re, prev_im});
  return hipGetLastError();
}

hipError_t zc_var(const void *data, long long length, int space, int amplitude, float thr_re, float thr_im,
                  float ang_re, float ang_im, float *var, hipStream_t st)
{
  hipLaunchKernelGGL(zc_var_kernel, dim3(grid_for(length, 256)), dim3(256), 0, st, reinterpret_cast<const float2 *>(data),
                     length, space, amplitude, float2{thr_re, thr_im}, float2{ang_re, ang_im}, var);
  return hipGetLastError();
}

hipError_t zc_scan(const float *var, long long length, long long nblocks, long long *last_pos, hipStream_t st)
{
  hipLaunchKernelGGL(zc_scan_kernel, dim3((unsigned)((nblocks + 63) / 64)), dim3(64), 0, st, var, length, nblocks, last_pos);
  return hipGetLastError();
}

hipError_t zc_emit(const float *var, long long length, long long nblocks, float bnor, const long long *last_pos,
                   unsigned char *seg, unsigned *count, hipStream_t st)
{
  hipLaunchKernelGGL(zc_emit_kernel, dim3((unsigned)((nblocks + 63) / 64)), dim3(64), 0, st, var, length, nblocks, bnor,
                     last_pos, seg, count);
  return hipGetLastError();
}

hipError_t zc_compact(const unsigned char *seg, const unsigned *count, const unsigned long long *offset,
                      unsigned char *out, long long nblocks, hipStream_t st)
{
  hipLaunchKernelGGL(zc_compact_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, seg, count, offset, out);
  return hipGetLastError();
}

hipError_t sample_manual_bulk(const void *data, long long length, double symbol_count, double symbol_sync, int space,
                              void *out, long long nout, hipStream_t st)
{
  if (nout <= 0) return hipSuccess;
  const double delta = (double)length / symbol_count;       // Tasks/WaveSampler.cpp:45-46
  const double sampOffset = symbol_sync / delta;
  hipLaunchKernelGGL(sample_manual_kernel, dim3(grid_for(nout, 128)), dim3(128), 0, st,
                     reinterpret_cast<const float2 *>(data), length, delta, sampOffset, symbol_sync, space,
                     reinterpret_cast<float2 *>(out), nout);
  return hipGetLastError();
}

// grid of a recurrence launch and the XCD its wavefronts go to (SUAMD_SERIAL_XCD=0: one workgroup per block of channels on
// whatever XCD the dispatcher's round robin gives it -- XCD 0 for a single block)
static int serial_xcd(int stage, dim3 &grid)
{
  const bool off = sdk::tuning().serial_xcd == 0;
  if (off || grid.x > 64) return -1;
  grid.x *= 8;
  return stage;                                                // AGC 3, Costas 1, clock 2: three different XCDs
}

hipError_t costas_feed(const CostasParams &p, const CostasState &s, int nchan, const void *x, View xs,
                       void *y, View ys, long long len, hipStream_t st)
{
  if (len <= 0 || nchan <= 0) return hipSuccess;
  dim3 grid((nchan + 63) / 64);
  const dim3 block(64);
  const int xcd = serial_xcd(1, grid);
  const float2 *xx = reinterpret_cast<const float2 *>(x);
  float2 *yy = reinterpret_cast<float2 *>(y);
#define SD_COSTAS_CASE(K, O) \
  case (K) * 8 + (O): \
    if (p.gain == 1.0f) hipLaunchKernelGGL((costas_kernel<K, O, true>), grid, block, 0, st, p, s, nchan, xx, xs, yy, ys, len, xcd); \
    else hipLaunchKernelGGL((costas_kernel<K, O, false>), grid, block, 0, st, p, s, nchan, xx, xs, yy, ys, len, xcd); \
    break;
  if (p.order < 0 || p.order > 4 || p.kind < 1 || p.kind > 3) return hipErrorInvalidValue;
  switch (p.kind * 8 + p.order) {
    SD_COSTAS_CASE(1, 0) SD_COSTAS_CASE(1, 1) SD_COSTAS_CASE(1, 2) SD_COSTAS_CASE(1, 3) SD_COSTAS_CASE(1, 4)
    SD_COSTAS_CASE(2, 0) SD_COSTAS_CASE(2, 1) SD_COSTAS_CASE(2, 2) SD_COSTAS_CASE(2, 3) SD_COSTAS_CASE(2, 4)
    SD_COSTAS_CASE(3, 0) SD_COSTAS_CASE(3, 1) SD_COSTAS_CASE(3, 2) SD_COSTAS_CASE(3, 3) SD_COSTAS_CASE(3, 4)
    default: return hipErrorInvalidValue;
  }
#undef SD_COSTAS_CASE
  return hipGetLastError();
}

hipError_t pll_feed(float alpha, float beta, const PllState &s, int nchan, const void *x, View xs,
                    void *y, View ys, long long len, hipStream_t st)
{
  if (len <= 0 || nchan <= 0) return hipSuccess;
  hipLaunchKernelGGL(pll_kernel, dim3((nchan + 63) / 64), dim3(64), 0, st, alpha, beta, s, nchan,
                     reinterpret_cast<const float2 *>(x), xs, reinterpret_cast<float2 *>(y), ys, len);
  return hipGetLastError();
}

hipError_t clock_feed(const ClockParams &p, const ClockState &s, int nchan, const void *x, View xs,
                      long long len, void *sym, long long sym_stride, uint32_t *count, hipStream_t st)
{
  if (len <= 0 || nchan <= 0) return hipSuccess;
  // The schedule.  2 = round by round (clock_ring): a round (~105 instructions + staging) per half cycle whatever the lanes'
  // timing; 0 = lock step: 3 instructions per sample + the ~45-instruction crossing code whenever ANY lane crosses -- with
  // n unrelated clocks that is a fraction 1 - (1 - 1 / k)^n of the samples (k = samples per half cycle).  One or two
  // channels (BASELINE configs[1]: ONE inspector at 5 samples per symbol) cross rarely enough for the lock step to win (125 vs
  // ~100 ns per sample at k = 2.5), 64 staggered ones never (139 vs 56 at k = 7.8); slow symbol rates (k > 24) keep the lock
  // step as well.  1 = clock_stream_tm.  sdk::tuning().clock_mode pins a schedule (A / B).
  const int forced = (int)sdk::tuning().clock_mode;
  const float bhint = 2.0f * p.bmin;
  int steps = (int)ceilf(0.5f / bhint) + 1;
  const double inv_k = fmin(1.0, 2.0 * (double)bhint);
  const double cost_lock = 3.0 + 45.0 * (1.0 - pow(1.0 - inv_k, (double)(nchan < 64 ? nchan : 64)));
  const double cost_ring = 107.0 * inv_k + 2.0;
  const int mode = forced >= 0 ? forced : ((steps <= 25 && cost_ring < cost_lock) ? 2 : 0);
  steps = 3 * ((steps + 2) / 3);                               // (clock_ring advances in groups of three)
  if (steps > 30) steps = 30;
  if (steps < 3) steps = 3;
  dim3 grid((nchan + 63) / 64);
  const int xcd = serial_xcd(2, grid);
  hipLaunchKernelGGL(clock_kernel, grid, dim3(64), 0, st, p, s, nchan,
                     reinterpret_cast<const float2 *>(x), xs, len, reinterpret_cast<float2 *>(sym), sym_stride, count, mode, xcd, steps);
  return hipGetLastError();
}

hipError_t agc_feed_pre(const AgcParams &p, const AgcState &s, int nchan, const void *x, View xv, long long len,
                        float *scratch, hipStream_t st)
{
  if (len <= 0 || nchan <= 0) return hipSuccess;
  const float2 *xx = reinterpret_cast<const float2 *>(x);
  const long long total = len * nchan;
  float *db = scratch, *peak = scratch + total;
  const int H = (int)p.mag_history_size;
  hipLaunchKernelGGL(agc_mag_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, xx, xv, nchan, len, db);
  if (nchan == 1)
    hipLaunchKernelGGL(agc_peak1_kernel, dim3(grid_for(len, 256)), dim3(256), 0, st, db, s.mag_history, len, H, peak);
  else
    hipLaunchKernelGGL(agc_peak_kernel, dim3((unsigned)((len + PEAK_TM - 1) / PEAK_TM), (unsigned)((nchan + 63) / 64)), dim3(256), 0, st,
                       db, s.mag_history, nchan, len, H, peak);
  return hipGetLastError();
}

hipError_t agc_feed_post(const AgcParams &p, const AgcState &s, int nchan, const void *x, View xv, void *y, View yv,
                         long long len, float *scratch, hipStream_t st)
{
  if (len <= 0 || nchan <= 0) return hipSuccess;
  const float2 *xx = reinterpret_cast<const float2 *>(x);
  float2 *yy = reinterpret_cast<float2 *>(y);
  const long long total = len * nchan;
  float *db = scratch, *peak = scratch + total;
  hipLaunchKernelGGL(agc_apply_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, p, s.delay_line, nchan, xx, xv,
                     yy, yv, len, peak);
  hipLaunchKernelGGL(agc_state_kernel, dim3((nchan + 63) / 64), dim3(64), 0, st, s.delay_line, s.mag_history, nchan,
                     (int)p.delay_line_size, (int)p.mag_history_size, xx, xv, db, len);
  return hipGetLastError();
}

hipError_t agc_feed(const AgcParams &p, const AgcState &s, int nchan, const void *x, View xv,
                    void *y, View yv, long long len, float *scratch, hipStream_t st)
{
  if (len <= 0 || nchan <= 0) return hipSuccess;
  hipError_t e = agc_feed_pre(p, s, nchan, x, xv, len, scratch, st);
  if (e != hipSuccess) return e;
  e = agc_feed_level(p, s, nchan, len, scratch, st);
  if (e != hipSuccess) return e;
  return agc_feed_post(p, s, nchan, x, xv, y, yv, len, scratch, st);
}

hipError_t agc_feed_level(const AgcParams &p, const AgcState &s, int nchan, long long len, float *scratch, hipStream_t st)
{
  if (len <= 0 || nchan <= 0) return hipSuccess;
  dim3 grid((nchan + 63) / 64);
  const int xcd = serial_xcd(3, grid);
  hipLaunchKernelGGL(agc_level_kernel, grid, dim3(64), 0, st, p, s, nchan, len, scratch + len * nchan, xcd);
  return hipGetLastError();
}

hipError_t agc_apply_items(const AgcApplyItem *d_items, int n, long long max_span, hipStream_t st)
{
  if (n <= 0 || max_span <= 0) return hipSuccess;
  hipLaunchKernelGGL(agc_apply_items_kernel, dim3(grid_for(max_span, 256), (unsigned)n), dim3(256), 0, st, d_items);
  return hipGetLastError();
}

hipError_t agc_state_update(const AgcParams &p, const AgcState &s, int nchan, const void *x, View xv, long long len,
                            const float *db, hipStream_t st)
{
  if (len <= 0 || nchan <= 0) return hipSuccess;
  hipLaunchKernelGGL(agc_state_kernel, dim3((nchan + 63) / 64), dim3(64), 0, st, s.delay_line, s.mag_history, nchan,
                     (int)p.delay_line_size, (int)p.mag_history_size, reinterpret_cast<const float2 *>(x), xv, db, len);
  return hipGetLastError();
}

hipError_t agc_pre_items(const AgcPreItem *d_items, int n, long long max_len, hipStream_t st)
{
  if (n <= 0 || max_len <= 0) return hipSuccess;
  hipLaunchKernelGGL(agc_pre_items_kernel, dim3(grid_for(max_len, 256), (unsigned)n), dim3(256), 0, st, d_items);
  return hipGetLastError();
}

hipError_t agc_state_items(const AgcStateItem *d_items, int n, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(agc_state_items_kernel, dim3((unsigned)n), dim3(64), 0, st, d_items);
  return hipGetLastError();
}

}  // namespace sdk
