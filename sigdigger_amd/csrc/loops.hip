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

namespace {

using sd::c32;
constexpr int CHUNK = 32;    // time steps prefetched per lane (one chunk ahead of the arithmetic)

// Element (c, m) of a view = base[c*cs + m*ms]: the m*ms part is wave-uniform (scalar base
// address), the c*cs part is a 32-bit per-lane byte offset -> "saddr + voffset" addressing, no
// 64-bit vector address arithmetic per access.
template <typename T>
__device__ __forceinline__ T ld_elem(const T *__restrict__ base, long long uniform_elem, uint32_t lane_off)
{
  const char *b = reinterpret_cast<const char *>(base + uniform_elem);
  return *reinterpret_cast<const T *>(b + lane_off);
}
template <typename T>
__device__ __forceinline__ void st_elem(T *__restrict__ base, long long uniform_elem, uint32_t lane_off, T v)
{
  char *b = reinterpret_cast<char *>(base + uniform_elem);
  *reinterpret_cast<T *>(b + lane_off) = v;
}

// Streams `len` time steps of one lane's row through step(m, value): chunks of CHUNK steps are
// prefetched one chunk ahead; the steady-state loop has no bounds checks (a single wavefront
// issues ~one instruction per 4-5 cycles, so per-sample instruction count is the cost).
template <typename T, typename F>
__device__ __forceinline__ void stream_row(const T *__restrict__ x, long long ms, uint32_t lane_off, long long len,
                                           F step)
{
  long long i = 0;
  if (len >= 2 * CHUNK) {
    T cur[CHUNK], nxt[CHUNK];
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) cur[j] = ld_elem(x, (long long)j * ms, lane_off);
    for (; i + 2 * CHUNK <= len; i += CHUNK) {
#pragma unroll
      for (int j = 0; j < CHUNK; ++j) nxt[j] = ld_elem(x, (i + CHUNK + j) * ms, lane_off);
#pragma unroll
      for (int j = 0; j < CHUNK; ++j) step(i + j, cur[j]);
#pragma unroll
      for (int j = 0; j < CHUNK; ++j) cur[j] = nxt[j];
    }
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) step(i + j, cur[j]);
    i += CHUNK;
  }
  for (; i < len; ++i) step(i, ld_elem(x, i * ms, lane_off));
}

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

// ---------------------------------------------------------------------------------------
// K6: Costas loop
template <int ORDER> struct CostasRegs {
  uint32_t phase; float omega;
  c32 xh[ORDER + 1], yh[ORDER + 1];
};

template <int KIND, int ORDER, bool GAIN1>
__device__ __forceinline__ float2 costas_step(const sdk::CostasParams &p, CostasRegs<ORDER> &r, float2 v)
{
  // history part of the arm filter first: it does not depend on the new sample
  float tr = 0.0f, ti = 0.0f;
#pragma unroll
  for (int q = ORDER; q >= 1; --q) { tr = sd::fma_(p.fb[q], r.xh[q].re, tr); ti = sd::fma_(p.fb[q], r.xh[q].im, ti); }
#pragma unroll
  for (int q = ORDER; q >= 1; --q) { tr = sd::fma_(-p.fa[q], r.yh[q].re, tr); ti = sd::fma_(-p.fa[q], r.yh[q].im, ti); }
  const sd::v2f_ mm = sd::mix_conj(sd::v2f_{v.x, v.y}, sd::phasor_pk(r.phase));   // x * conj(ref)
  const c32 m = {mm.x, mm.y};
  c32 z;
  z.re = sd::fma_(p.fb[0], m.re, tr);
  z.im = sd::fma_(p.fb[0], m.im, ti);
#pragma unroll
  for (int q = ORDER; q >= 2; --q) { r.xh[q] = r.xh[q - 1]; r.yh[q] = r.yh[q - 1]; }
  if (ORDER >= 1) { r.xh[1] = m; r.yh[1] = z; }
  if (!GAIN1) {                                               // gain is 1 upstream; x*1.0f is exact,
    z.re = p.gain * z.re;                                     // so skipping the multiply keeps the bits
    z.im = p.gain * z.im;
  }
  float e;
  if (KIND == 1) {
    e = z.re * z.im;
  } else {
    float sr, si;
    sd::sgn2(z.re, z.im, sr, si);
    if (KIND == 2) {
      e = sr * z.im - si * z.re;
    } else {
      // |re| >= |im|:  e = A - B k,  else  e = A k - B   (A = sgn(re) im, B = sgn(im) re, k = tan(pi/8)).
      // No branch: a lone wavefront pays ~100 cycles for the EXEC bookkeeping of a divergent if / else.  The factor that
      // is not k is 1.0f, and x * 1.0f is exact, so both products are the ones the two-way form computes.
      const bool wide = __builtin_fabsf(z.re) >= __builtin_fabsf(z.im);
      const float ka = wide ? 1.0f : 0.41421356237309504880f, kb = wide ? 0.41421356237309504880f : 1.0f;
      e = (sr * z.im) * ka - (si * z.re) * kb;
    }
  }
  const float dphi = sd::fma_(p.a, e, r.omega);
  r.omega = sd::fma_(p.b, e, r.omega);
  r.phase += (uint32_t)sd::rad_to_dphase(dphi);
  return float2{z.re, z.im};
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

// ---------------------------------------------------------------------------------------
// K7: PLL
__device__ __forceinline__ float2 pll_step(float alpha, float beta, uint32_t &phase, float &omega, float2 v)
{
  const sd::v2f_ mm = sd::mix_conj(sd::v2f_{v.x, v.y}, sd::phasor_pk(phase));
  const float2 m = {mm.x, mm.y};
  float err = sd::atan2_nb_(v.y, v.x) - sd::phase_to_rad(phase);
  if (err >  3.14159265358979323846f) err -= 6.28318530717958647692f;
  if (err < -3.14159265358979323846f) err += 6.28318530717958647692f;
  const float dphi = sd::fma_(beta, err, omega);
  omega = sd::fma_(alpha, err, omega);
  phase += (uint32_t)sd::rad_to_dphase(dphi);
  return m;
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

// ---------------------------------------------------------------------------------------
// K8: Gardner clock recovery; variable-rate output, per-lane append
struct ClockRegs {
  float phi, bnor;
  int halfcycle;
  float2 prev, x0, x1, x2;
  uint32_t n;
};

__device__ __forceinline__ void clock_step(const sdk::ClockParams &p, ClockRegs &r, float2 v, float2 *__restrict__ out)
{
  r.phi = r.phi + r.bnor;
  // The straight path must not touch EXEC: a divergent `if` costs a lone wavefront ~100 cycles per sample in
  // v_cmp -> s_and_saveexec -> VALU hazards even when no lane takes it.  A wave-uniform test (scalar branch on the
  // ballot) keeps the sample that only advances the phase at add + compare + branch; the crossing code sits off the
  // straight path and is entered -- with its lane mask -- only when some lane crosses.
  const bool cross = r.phi >= 0.5f;
  if (__builtin_expect(__any(cross), 0) && cross) {
    const float mu = (r.phi - 0.5f) / r.bnor;
    float2 q;
    q.x = sd::fma_(mu, r.prev.x - v.x, v.x);
    q.y = sd::fma_(mu, r.prev.y - v.y, v.y);
    r.phi = r.phi - 0.5f;
    r.halfcycle = !r.halfcycle;
    if (!r.halfcycle) {
      r.x2 = r.x0;
      r.x0 = q;
      const float dr = r.x0.x - r.x2.x, di = r.x0.y - r.x2.y;
      const float e = p.gain * sd::fma_(r.x1.y, di, r.x1.x * dr);
      r.phi = sd::fma_(p.alpha, e, r.phi);
      float b = sd::fma_(p.beta, e, r.bnor);
      if (b < p.bmin) b = p.bmin;
      if (b > p.bmax) b = p.bmax;
      r.bnor = b;
      out[r.n++] = q;
    } else {
      r.x1 = q;
    }
  }
  r.prev = v;
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

// ---------------------------------------------------------------------------------------
// Gangs: lane j runs item j -- a 1-channel bank with its own parameters, state, rows and length.
// Rows are streamed 16 steps ahead per lane (each lane its own pointer; a chunk is 128 contiguous
// bytes per lane); steps beyond a lane's length are skipped by predication.
__device__ __forceinline__ long long wave_max(long long v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const long long w = __shfl_xor(v, o); v = w > v ? w : v; }
  return v;
}

// Every lane has its own row somewhere in memory.  Reading it lane by lane would touch 64 different
// cache lines per load instruction (measured: the gang ran 4.6x slower per sample than a bank).  Instead the
// wave moves data in tiles of GT samples x 64 items through LDS: row k's tile is loaded by all 64 lanes
// together (one coalesced GT*sizeof(T)-byte access), transposed through LDS (pitch 65: conflict-free both
// ways), and lane k then reads its own samples from LDS; outputs go back the same way.  The loads of tile
// t+1 are issued before the steps of tile t run (they wait in registers), so their latency is hidden.
constexpr int GT = 32;          // samples per tile and row
constexpr int GP = 65;          // LDS pitch (items per sample row + 1)

template <typename T> struct GangLds { T in[GT * GP]; T out[GT * GP]; const T *xp[64]; T *yp[64]; long long ln[64]; };

// HAS_OUT: step() returns a T that is written to row y; otherwise step() returns nothing
template <bool HAS_OUT, typename T, typename F>
__device__ __forceinline__ void gang_stream(GangLds<T> &lds, const T *__restrict__ x, T *__restrict__ y, long long len,
                                            F step)
{
  const int lane = threadIdx.x;
  const long long maxlen = wave_max(len);
  if (maxlen <= 0) return;
  const long long minlen = -wave_max(len > 0 ? -len : -(1ll << 62));      // shortest non-empty row
  // a tile is GT = 32 samples: the 64 lanes cover two rows per access (lane >> 5 picks the row of the pair).
  // The rows a lane touches (2k + half, k < 32) never change: their pointers and lengths are fetched once,
  // through LDS, into registers -- a single wavefront owns the SIMD's whole register file.
  const int half = lane >> 5, sl = lane & 31;
  lds.xp[lane] = x; lds.yp[lane] = y; lds.ln[lane] = len;     // (single wave: no barrier needed)
  const T *rx[32];
  T *ry[32];
  int rl[32];                                                // rows are shorter than 2^31 samples
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    rx[k] = lds.xp[2 * k + half];
    if constexpr (HAS_OUT) ry[k] = lds.yp[2 * k + half];
    rl[k] = (int)lds.ln[2 * k + half];
  }
  T pre[32];
  auto request = [&](long long s0) {                        // rows 2k + half, samples s0 + sl
#pragma unroll
    for (int k = 0; k < 32; ++k) pre[k] = (s0 + sl < rl[k]) ? rx[k][s0 + sl] : T{};
  };
  request(0);
  for (long long s0 = 0; s0 < maxlen; s0 += GT) {
#pragma unroll
    for (int k = 0; k < 32; ++k) lds.in[sl * GP + 2 * k + half] = pre[k];   // sample (s0 + sl) of row 2k + half
    if (s0 + GT < maxlen) request(s0 + GT);
    // steps in groups of CHUNK: the group's inputs are pulled out of LDS first (independent reads), so the
    // recurrence itself never waits on LDS; groups that lie inside every row's length (the common case)
    // run without per-step predication
    for (int g = 0; g < GT; g += CHUNK) {
      T vin[CHUNK], vout[CHUNK];
#pragma unroll
      for (int j = 0; j < CHUNK; ++j) vin[j] = lds.in[(g + j) * GP + lane];
      if (s0 + g + CHUNK <= minlen) {
        if (len > 0) {                                        // empty rows (and lanes without an item) sit the group out
#pragma unroll
          for (int j = 0; j < CHUNK; ++j) {
            if constexpr (HAS_OUT) vout[j] = step(s0 + g + j, vin[j]);
            else step(s0 + g + j, vin[j]);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < CHUNK; ++j) {
          if (s0 + g + j < len) {
            if constexpr (HAS_OUT) vout[j] = step(s0 + g + j, vin[j]);
            else step(s0 + g + j, vin[j]);
          }
        }
      }
      if constexpr (HAS_OUT) {
#pragma unroll
        for (int j = 0; j < CHUNK; ++j) lds.out[(g + j) * GP + lane] = vout[j];
      }
    }
    if constexpr (HAS_OUT) {
#pragma unroll
      for (int k = 0; k < 32; ++k)
        if (s0 + sl < rl[k]) ry[k][s0 + sl] = lds.out[sl * GP + 2 * k + half];
    }
  }
}

// The transposition above costs the lone wavefront ~90 ns per sample (address arithmetic, predicates, LDS round trips:
// a gang ran 2x slower per sample than a bank, tools/gang_bench.py).  It is throughput work, so it moves out of the
// recurrence: a parallel kernel gathers the 64 rows of a group into a time-major slab tm[m][lane] (LDS-transposed
// tiles, coalesced both ways), the recurrence streams the slab exactly like a 64-channel bank (scalar base +
// immediate offsets, one chunk prefetched ahead) -- in place -- and a second parallel kernel scatters the results to
// the rows.  The slab has whole 64-sample tiles plus one tile of slack for the prefetch.
__device__ __forceinline__ long long uniform64(long long v)
{
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(unsigned long long)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}

// (tm / tmo: where the lanes read and write -- the same packed slab for the gather / scatter form; lo / loo: the lane's byte
// offset there.  The slab form of a 64-column pitch runs this loop too, on the producer's slab and with the items' own columns.)
// EVERY lane has work: a lane without an item, or with an empty row, is given a copy of another lane's item by its caller
// (gang_lane) -- it computes and stores exactly what that lane does, and its state is not written back.  That keeps the
// steady-state loop ONE straight path.  With the `if (len > 0)` it used to have around the steps, two paths with different
// numbers of stores in flight met at the back edge, and the compiler's wait there was s_waitcnt vmcnt(0): every chunk waited
// for its own last store to complete (the bank kernels, one path, get counted waits) -- 86 against 77 ns per Costas sample.
template <bool HAS_OUT, typename T, typename F>
__device__ __forceinline__ void gang_stream_tm(const T *tm, T *tmo, const uint32_t lo, const uint32_t loo, long long len, F step)
{
  const long long maxlen = uniform64(wave_max(len));
  if (maxlen <= 0) return;
  const long long minlen = uniform64(-wave_max(-len));
  T cur[CHUNK], nxt[CHUNK];
#pragma unroll
  for (int j = 0; j < CHUNK; ++j) cur[j] = ld_elem(tm, (long long)j * 64, lo);
  long long i = 0;
  for (; i + CHUNK <= minlen; i += CHUNK) {                  // inside every row
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) nxt[j] = ld_elem(tm, (i + CHUNK + j) * 64, lo);
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) {
      if constexpr (HAS_OUT) st_elem(tmo, (i + j) * 64, loo, step(i + j, cur[j]));
      else step(i + j, cur[j]);
    }
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) cur[j] = nxt[j];
  }
  for (; i < maxlen; i += CHUNK) {                           // the rows' ends: step by step
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) nxt[j] = ld_elem(tm, (i + CHUNK + j) * 64, lo);
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) {
      if (i + j < len) {
        if constexpr (HAS_OUT) st_elem(tmo, (i + j) * 64, loo, step(i + j, cur[j]));
        else step(i + j, cur[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) cur[j] = nxt[j];
  }
}

// Which item a lane of a gang works on: its own (lane < count, a row that is not empty), else a copy of the first lane that
// has one.  false: nobody has work.  own: the lane's state goes back to its item.
template <typename Item>
__device__ __forceinline__ bool gang_lane(const Item *__restrict__ items, int count, int j, int *src, bool *own)
{
  const long long mylen = j < count ? items[j].len : 0;
  const unsigned long long has = __ballot(mylen > 0);
  if (has == 0) return false;
  *own = mylen > 0;
  *src = *own ? j : (int)__ffsll((long long)has) - 1;
  return true;
}

// rows of a gang's items <-> slabs: item k of the table (item_bytes apart; its row pointer and length sit at off_ptr /
// off_len) is lane k % 64 of group k / 64.  Tile = 64 samples x 64 rows through LDS (pitch 65).
template <typename T>
__global__ __launch_bounds__(256) void rows_tm_gather_kernel(const char *__restrict__ items, int item_bytes, int off_ptr, int off_len,
                                                             int n, const sdk::GangGroup *__restrict__ groups, T *__restrict__ tm, long long slab)
{
  __builtin_amdgcn_s_setprio(3);
  __shared__ T tile[64][65];
  const int g = blockIdx.y, s = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int first = groups ? groups[g].first : g * 64;
  const int last = groups ? first + groups[g].count : n;
  const long long m0 = (long long)blockIdx.x * 64;
#pragma unroll 4
  for (int p = 0; p < 16; ++p) {
    const int row = p * 4 + q, item = first + row;
    T v = T{};
    if (item < last) {
      const char *it = items + (size_t)item * item_bytes;
      const T *src = *reinterpret_cast<const T *const *>(it + off_ptr);
      const long long len = *reinterpret_cast<const long long *>(it + off_len);
      if (m0 + s < len) v = src[m0 + s];
    }
    tile[row][s] = v;
  }
  __syncthreads();
  T *dst = tm + (size_t)g * slab + m0 * 64;
#pragma unroll 4
  for (int p = 0; p < 16; ++p) { const int ss = p * 4 + q; dst[ss * 64 + s] = tile[s][ss]; }
}

template <typename T>
__global__ __launch_bounds__(256) void rows_tm_scatter_kernel(const char *__restrict__ items, int item_bytes, int off_ptr, int off_len,
                                                              int n, const sdk::GangGroup *__restrict__ groups, const T *__restrict__ tm, long long slab)
{
  __builtin_amdgcn_s_setprio(3);
  __shared__ T tile[64][65];
  const int g = blockIdx.y, s = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int first = groups ? groups[g].first : g * 64;
  const int last = groups ? first + groups[g].count : n;
  const long long m0 = (long long)blockIdx.x * 64;
  const T *src = tm + (size_t)g * slab + m0 * 64;
#pragma unroll 4
  for (int p = 0; p < 16; ++p) { const int ss = p * 4 + q; tile[s][ss] = src[ss * 64 + s]; }
  __syncthreads();
#pragma unroll 4
  for (int p = 0; p < 16; ++p) {
    const int row = p * 4 + q, item = first + row;
    if (item < last) {
      const char *it = items + (size_t)item * item_bytes;
      T *dst = *reinterpret_cast<T *const *>(it + off_ptr);
      const long long len = *reinterpret_cast<const long long *>(it + off_len);
      if (m0 + s < len) dst[m0 + s] = tile[row][s];
    }
  }
}

// (Round 6 tried a UNIFORM form for groups whose items have identical parameters -- 64 inspectors opened alike --, the parameters
// read once and kept in SGPRs like a bank's: SLOWER, 828 against 747 us per 64 x 8192 samples.  A VOP3P instruction reads one
// SGPR operand; thirteen scalar parameters cost more v_mov than the per-lane VGPRs cost anything.  Taken back.)
__device__ __forceinline__ float uniform_f(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }
template <int KIND, int ORDER, bool GAIN1>
__device__ __forceinline__ void costas_gang_body(const sdk::CostasGangItem *__restrict__ items, int count, float2 *tm)
{
  int j; bool live;
  if (!gang_lane(items, count, (int)threadIdx.x, &j, &live)) return;
  const sdk::CostasGangItem it = items[j];
  const sdk::CostasParams p = it.p;                           // per lane: every item its own loop
  const sdk::CostasState s = it.s;
  CostasRegs<ORDER> r;
  r.phase = s.phase[0];
  r.omega = s.omega[0];
#pragma unroll
  for (int i = 1; i <= ORDER; ++i) {
    r.xh[i] = c32{s.xh[(i - 1) * 2 + 0], s.xh[(i - 1) * 2 + 1]};
    r.yh[i] = c32{s.yh[(i - 1) * 2 + 0], s.yh[(i - 1) * 2 + 1]};
  }
  gang_stream_tm<true>(tm, tm, j * 8u, j * 8u, it.len, [&](long long, float2 v) { return costas_step<KIND, ORDER, GAIN1>(p, r, v); });
  if (!live) return;
  s.phase[0] = r.phase;
  s.omega[0] = r.omega;
#pragma unroll
  for (int i2 = 1; i2 <= ORDER; ++i2) {
    s.xh[(i2 - 1) * 2 + 0] = r.xh[i2].re; s.xh[(i2 - 1) * 2 + 1] = r.xh[i2].im;
    s.yh[(i2 - 1) * 2 + 0] = r.yh[i2].re; s.yh[(i2 - 1) * 2 + 1] = r.yh[i2].im;
  }
}

// every loop type of a gang in ONE launch: workgroup g runs the items of groups[g] (all of one kind and arm-filter
// order, which selects the compiled-in loop).  The types used to fork onto side streams and join back: four event
// hops and three sets of gather / scatter launches per call.
__global__ __launch_bounds__(64) void costas_gang_kernel(const sdk::CostasGangItem *__restrict__ items,
                                                         const sdk::GangGroup *__restrict__ groups, float2 *tm, long long slab)
{
  const sdk::GangGroup gd = groups[blockIdx.x];
  const sdk::CostasGangItem *mine = items + gd.first;
  float2 *my = tm + (size_t)blockIdx.x * slab;
  switch (gd.kind * 8 + gd.order) {
#define SD_GANG_CASE(K, O) case (K) * 8 + (O): if (gd.gain1) costas_gang_body<K, O, true>(mine, gd.count, my); \
                                               else costas_gang_body<K, O, false>(mine, gd.count, my); break;
    SD_GANG_CASE(1, 0) SD_GANG_CASE(1, 1) SD_GANG_CASE(1, 2) SD_GANG_CASE(1, 3) SD_GANG_CASE(1, 4)
    SD_GANG_CASE(2, 0) SD_GANG_CASE(2, 1) SD_GANG_CASE(2, 2) SD_GANG_CASE(2, 3) SD_GANG_CASE(2, 4)
    SD_GANG_CASE(3, 0) SD_GANG_CASE(3, 1) SD_GANG_CASE(3, 2) SD_GANG_CASE(3, 3) SD_GANG_CASE(3, 4)
#undef SD_GANG_CASE
    default: break;
  }
}

__global__ __launch_bounds__(64) void pll_gang_kernel(const sdk::PllGangItem *__restrict__ items, int n, float2 *tm, long long slab)
{
  const sdk::PllGangItem *mine = items + (size_t)blockIdx.x * 64;
  int j; bool live;
  if (!gang_lane(mine, n - (int)blockIdx.x * 64, (int)threadIdx.x, &j, &live)) return;
  const sdk::PllGangItem it = mine[j];
  const float alpha = it.alpha, beta = it.beta;
  uint32_t phase = it.s.phase[0];
  float omega = it.s.omega[0];
  float2 *my = tm + (size_t)blockIdx.x * slab;
  gang_stream_tm<true>(my, my, j * 8u, j * 8u, it.len, [&](long long, float2 v) { return pll_step(alpha, beta, phase, omega, v); });
  if (!live) return;
  it.s.phase[0] = phase;
  it.s.omega[0] = omega;
}

// CMA equalizers (SPEC.md section I) of many inspectors: N weights and the delay line per lane
template <int N>
__global__ __launch_bounds__(64) void cma_gang_kernel(const sdk::CmaGangItem *__restrict__ items, int n)
{
  const int j = blockIdx.x * 64 + threadIdx.x;
  const bool live = j < n;
  const sdk::CmaGangItem it = items[live ? j : 0];
  const float mu = it.mu;
  const bool locked = it.locked != 0;
  float2 *w = reinterpret_cast<float2 *>(it.w), *dl = reinterpret_cast<float2 *>(it.dl);
  c32 wr[N], d[N];
#pragma unroll
  for (int i = 0; i < N; ++i) { const float2 a = w[i], b = dl[i]; wr[i] = c32{a.x, a.y}; d[i] = c32{b.x, b.y}; }
  const long long len = live ? (it.count ? (long long)it.count[0] : it.fixed_len) : 0;
  __shared__ GangLds<float2> lds;
  gang_stream<true>(lds, reinterpret_cast<const float2 *>(it.x), reinterpret_cast<float2 *>(it.y), len,
                    [&](long long, float2 v) {
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
    return float2{yr, yi};
  });
  if (!live) return;
#pragma unroll
  for (int i = 0; i < N; ++i) { w[i] = float2{wr[i].re, wr[i].im}; dl[i] = float2{d[i].re, d[i].im}; }
}

// Gardner detectors on a slab, crossing by crossing.  clock_step spends 3 instructions on a sample that only advances
// the phase and ~40 on one where it crosses 0.5 (interpolation, division, loop update) -- and a wavefront pays for the
// crossing whenever ANY lane has one, which with 64 unrelated symbol clocks is every sample (measured: 57 ns per
// sample with aligned clocks, 150 ns with unaligned ones).  Here the tile sits in LDS, each lane keeps its own sample
// index, and the wave alternates between a phase-advance loop (every lane runs to its next crossing or the tile's
// end) and one pass of the crossing code for all lanes that stopped at one: the expensive part runs once per half
// symbol instead of once per sample.  Per lane the operations and their order are clock_step's.
// (Measured alternative: advancing four samples per pass with selects instead of the branchy one-sample loop is
// slower, 7.3 vs 6.3 ms per 64 x 65536 samples -- the loop is mostly scalar instructions, the selects are vector ones.)
constexpr int CT = 64;          // samples per LDS tile
// SLACK: the rows extend (readably) to a whole tile past the longest one -- the gangs' slabs; otherwise the last tile's
// loads are bounded.  Sample m of the lane sits at base[m * pitch] + lo bytes.
template <bool SLACK>
__device__ __forceinline__ void clock_stream_tm(const float2 *base, const long long pitch, const uint32_t lo, long long len,
                                                const sdk::ClockParams &p, ClockRegs &r, float2 *__restrict__ out, float2 *tile)
{
  const int lane = threadIdx.x;
  const long long maxlen = uniform64(wave_max(len));
  if (maxlen <= 0) return;
  float2 pre[CT];
  auto request = [&](long long t0) {
    if (SLACK || t0 + CT <= maxlen) {
#pragma unroll
      for (int j = 0; j < CT; ++j) pre[j] = ld_elem(base, (t0 + j) * pitch, lo);
    } else {
#pragma unroll
      for (int j = 0; j < CT; ++j) pre[j] = t0 + j < maxlen ? ld_elem(base, (t0 + j) * pitch, lo) : float2{0.0f, 0.0f};
    }
  };
  request(0);
  for (long long s0 = 0; s0 < maxlen; s0 += CT) {
#pragma unroll
    for (int j = 0; j < CT; ++j) tile[j * 64 + lane] = pre[j];
    if (s0 + CT < maxlen) request(s0 + CT);                   // the next tile waits in registers while this one is worked on
    const long long left = len - s0;
    const int end = left <= 0 ? 0 : (left < CT ? (int)left : CT);
    int j = 0;
    for (;;) {
      bool crossed = false;
      while (j < end && !crossed) {                           // phase advance: 1 add per sample
        r.phi = r.phi + r.bnor;
        ++j;
        crossed = r.phi >= 0.5f;
      }
      if (!__any(crossed)) break;
      if (crossed) {                                          // sample j-1 of the tile crossed
        const float2 v = tile[(j - 1) * 64 + lane];
        const float2 prev = j >= 2 ? tile[(j - 2) * 64 + lane] : r.prev;
        const float mu = (r.phi - 0.5f) / r.bnor;
        float2 q;
        q.x = sd::fma_(mu, prev.x - v.x, v.x);
        q.y = sd::fma_(mu, prev.y - v.y, v.y);
        r.phi = r.phi - 0.5f;
        r.halfcycle = !r.halfcycle;
        if (!r.halfcycle) {
          r.x2 = r.x0;
          r.x0 = q;
          const float dr = r.x0.x - r.x2.x, di = r.x0.y - r.x2.y;
          const float e = p.gain * sd::fma_(r.x1.y, di, r.x1.x * dr);
          r.phi = sd::fma_(p.alpha, e, r.phi);
          float b = sd::fma_(p.beta, e, r.bnor);
          if (b < p.bmin) b = p.bmin;
          if (b > p.bmax) b = p.bmax;
          r.bnor = b;
          out[r.n++] = q;
        } else {
          r.x1 = q;
        }
      }
    }
    if (end > 0) r.prev = tile[(end - 1) * 64 + lane];
  }
}

// Gardner detectors of a BANK (uniform parameters, one length), round by round: every lane gets to its own next half-cycle
// crossing in the same round, wherever that is -- symbol clocks that are not aligned cost nothing.  What the two schedules
// above pay for: the lock-step form runs the ~45-instruction crossing code whenever ANY lane crosses (every sample, with 64
// unrelated clocks: 150 ns per sample against 57 aligned); clock_stream_tm's divergent advance loop costs a v_cmp -> SALU ->
// branch chain per sample (~95 ns).  Here a round is branch-free:
//   advance   U steps for every lane, no exit test: p += b; t = p - 0.5; the first t >= 0 is the crossing's (phi - 0.5) --
//             picked by an UNSIGNED minimum (a negative float is a large unsigned) --, the steps before it are counted in a
//             bit string (v_alignbit of t's sign).  4 instructions per step, one dependent add.  A lane that has not crossed
//             after U steps (U = ceil(0.5 / bhint) + 1) just carries p_U into the next round.
//   crossing  for all lanes at once, the symbol / half-cycle split as selects (with unrelated clocks both occur in every
//             round anyway), the symbol store under its mask.
// Samples sit in an LDS ring of two 64-row tiles in "y" coordinates (y[0] = the sample before the block, y[i + 1] = x[i]); a
// tile is replaced when every lane has left it, its successor waits in registers meanwhile.  Lanes are never bounded inside
// a round: the rounds stop U samples before the end and a plain per-sample loop (clock_step on per-lane indices) finishes
// the block -- <= 2 U samples per call.  Per lane the operations and their order are clock_step's: same bits.
constexpr int RT = 64, RING = 2 * RT;
// `len` is per lane (a bank passes the same for every live lane; a gang's items have their own): a lane within U samples of its
// end sits the rounds out ("tail") and the rounds go on while any lane is not there yet.  Rows must be readable up to the
// longest lane's length (a bank's are that long; a gang's slab has whole tiles of slack).  `p` may be wave-uniform (a bank:
// SGPRs) or per lane (a gang).
__device__ __forceinline__ void clock_ring(const float2 *base, const long long pitch, const uint32_t lo, const long long len, const bool live,
                                           const int U, const sdk::ClockParams &p, ClockRegs &r, float2 *__restrict__ out, float2 *ring)
{
  const int lane = threadIdx.x;
  const long long mylen = live ? len : 0;
  const long long maxlen = uniform64(wave_max(mylen));
  // y index of the next sample a lane consumes
  long long ny = 1;
  if (maxlen >= 4 * (long long)U + 2 * RT) {
    const int ylen = (int)(mylen < (1ll << 30) ? mylen : (1ll << 30));   // y indices as 32-bit numbers inside a call
    const long long loadable = maxlen < (1ll << 30) ? maxlen : (1ll << 30);
    // tile T holds y[T * RT .. T * RT + RT); request(T) loads it into registers
    float2 pre[RT];
    auto request = [&](int T) {
      const long long y0 = (long long)T * RT;                        // x index: y index - 1
      if (y0 >= 1 && y0 + RT - 1 <= loadable) {
#pragma unroll
        for (int j = 0; j < RT; ++j) pre[j] = ld_elem(base, (y0 + j - 1) * pitch, lo);
      } else {
#pragma unroll
        for (int j = 0; j < RT; ++j) pre[j] = (y0 + j >= 1 && y0 + j <= loadable) ? ld_elem(base, (y0 + j - 1) * pitch, lo) : float2{0.0f, 0.0f};
      }
    };
    auto commit = [&](int T) {
      float2 *dst = ring + (size_t)((T & 1) * RT) * 64 + lane;
#pragma unroll
      for (int j = 0; j < RT; ++j) dst[j * 64] = pre[j];
    };
    request(0); commit(0);
    ring[lane] = r.prev;                                             // y[0]
    request(1); commit(1);
    int wlo = 0;                                                     // the ring holds y[wlo, wlo + RING)
    request(2);
    int n = 1;
    float phi = r.phi, bnor = r.bnor;
    int hc = r.halfcycle;
    float2 x0 = r.x0, x1 = r.x1, x2 = r.x2;
    uint32_t cnt = r.n;
    const int stop = ylen + 1 - U;                                   // a round needs n + U <= ylen + 1 (idle and short lanes: never)
    for (;;) {
      const bool tail = n > stop;                                    // this lane's rounds are over: the rest below, sample by sample
      if (__all(tail)) break;
      // every lane has left the ring's older tile: replace it by the tile that waits in registers, request the next one
      if (__all(tail || n > wlo + RT)) {
        commit(wlo / RT + 2);
        wlo += RT;
        request(wlo / RT + 2);
      }
      const bool act = !tail && n + U <= wlo + RING;                 // (a lane far ahead of the others waits for the ring)
      float pp = phi;
      uint32_t sel = 0xffffffffu, bits = 0;
      for (int g = 0; g < U; g += 3) {                               // (U is a multiple of three: the loop counter is scalar work)
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          pp = pp + bnor;
          const float t = pp - 0.5f;
          const uint32_t tu = __float_as_uint(t);
          sel = tu < sel ? tu : sel;
          bits = __builtin_amdgcn_alignbit(bits, tu, 31);
        }
      }
      const float ts = __uint_as_float(sel);
      const bool crossed = act && ts >= 0.0f;                        // (a NaN phase never crosses, as in clock_step)
      int used = __builtin_popcount(bits) + 1;
      used = used < U ? used : U;
      used = act ? used : 0;
      const int c = n + used - 1;                                    // y index of the crossing sample
      const float2 v = ring[(c & (RING - 1)) * 64 + lane];
      const float2 pv = ring[((c - 1) & (RING - 1)) * 64 + lane];
      const float mu = ts / bnor;
      float2 q;
      q.x = sd::fma_(mu, pv.x - v.x, v.x);
      q.y = sd::fma_(mu, pv.y - v.y, v.y);
      const int hn = hc ^ 1;
      const bool sym = crossed && hn == 0, half = crossed && hn != 0;
      const float dr = q.x - x0.x, di = q.y - x0.y;
      const float e = p.gain * sd::fma_(x1.y, di, x1.x * dr);
      const float phis = sd::fma_(p.alpha, e, ts);
      float b = sd::fma_(p.beta, e, bnor);
      if (b < p.bmin) b = p.bmin;
      if (b > p.bmax) b = p.bmax;
      if (sym) out[cnt] = q;
      cnt += sym ? 1u : 0u;
      x2 = sym ? x0 : x2;
      x0 = sym ? q : x0;
      x1 = half ? q : x1;
      phi = crossed ? (sym ? phis : ts) : (act ? pp : phi);
      bnor = sym ? b : bnor;
      hc = crossed ? hn : hc;
      n += used;
    }
    if (live) {
      r.phi = phi; r.bnor = bnor; r.halfcycle = hc; r.x0 = x0; r.x1 = x1; r.x2 = x2; r.n = cnt;
      ny = n;
      if (n >= 2) r.prev = ld_elem(base, (long long)(n - 2) * pitch, lo);
    }
  }
  // the rest of the block, sample by sample, every lane from where it is
  const long long left = live ? mylen + 1 - ny : 0;
  const long long most = uniform64(wave_max(left));
  for (long long i = 0; i < most; ++i) {
    if (i < left) clock_step(p, r, ld_elem(base, (ny - 1 + i) * pitch, lo), out);
  }
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

__global__ __launch_bounds__(64) void clock_gang_kernel(const sdk::ClockGangItem *__restrict__ items, int n, float2 *tm, long long slab)
{
  __shared__ float2 lds[RING * 64];                             // clock_ring's two tiles; clock_stream_tm uses the first
  const int j = blockIdx.x * 64 + threadIdx.x;
  const bool live = j < n;
  const sdk::ClockGangItem it = items[live ? j : 0];
  sdk::ClockParams p = it.p;
  // (the host writes one `steps` / `uniform` per group of 64 items; lane 0 of a workgroup is always live)
  const int steps = __builtin_amdgcn_readfirstlane(it.steps);
  if (__builtin_amdgcn_readfirstlane(it.uniform)) {
    p.alpha = uniform_f(p.alpha); p.beta = uniform_f(p.beta); p.gain = uniform_f(p.gain); p.bmin = uniform_f(p.bmin); p.bmax = uniform_f(p.bmax);
  }
  const sdk::ClockState s = it.s;
  ClockRegs r;
  r.phi = s.phi[0]; r.bnor = s.bnor[0];
  r.halfcycle = s.halfcycle[0];
  r.prev = float2{s.prev[0], s.prev[1]};
  r.x0 = float2{s.x0[0], s.x0[1]};
  r.x1 = float2{s.x1[0], s.x1[1]};
  r.x2 = float2{s.x2[0], s.x2[1]};
  r.n = it.count[0];
  const long long len = live ? it.len : 0;
  float2 *out = reinterpret_cast<float2 *>(it.sym);
  // round by round when the group's symbol rates allow it (the banks' schedule: staggered symbol clocks cost nothing), else
  // crossing by crossing
  if (steps > 0) clock_ring(tm + (size_t)blockIdx.x * slab, 64, threadIdx.x * 8u, len, live, steps, p, r, out, lds);
  else clock_stream_tm<true>(tm + (size_t)blockIdx.x * slab, 64, threadIdx.x * 8u, len, p, r, out, lds);
  if (!live) return;
  s.phi[0] = r.phi; s.bnor[0] = r.bnor; s.halfcycle[0] = r.halfcycle;
  s.prev[0] = r.prev.x; s.prev[1] = r.prev.y;
  s.x0[0] = r.x0.x; s.x0[1] = r.x0.y;
  s.x1[0] = r.x1.x; s.x1[1] = r.x1.y;
  s.x2[0] = r.x2.x; s.x2[1] = r.x2.y;
  it.count[0] = r.n;
}

__global__ __launch_bounds__(64) void agc_level_gang_kernel(const sdk::AgcGangItem *__restrict__ items, int n, float *tm, long long slab)
{
  const sdk::AgcGangItem *mine = items + (size_t)blockIdx.x * 64;
  int j; bool live;
  if (!gang_lane(mine, n - (int)blockIdx.x * 64, (int)threadIdx.x, &j, &live)) return;
  const sdk::AgcGangItem it = mine[j];
  const sdk::AgcState s = it.s;
  unsigned hang_n = s.hang_n[0];
  float fast = s.fast_level[0], slow = s.slow_level[0];
  const float far = it.p.fast_alpha_rise, faf = it.p.fast_alpha_fall, sar = it.p.slow_alpha_rise, saf = it.p.slow_alpha_fall;
  const float knee = it.p.knee;
  const unsigned hang_max = it.p.hang_max;
  const long long len = it.len;
  float *my = tm + (size_t)blockIdx.x * slab;
  gang_stream_tm<true>(my, my, j * 4u, j * 4u, len, [&](long long, float pk) {
    float d = pk - fast;
    const float fa = d > 0.0f ? far : faf;
    fast = sd::fma_(fa, d, fast);
    d = pk - slow;
    const bool rise = d > 0.0f;
    const bool fall = !rise && hang_n >= hang_max;
    const float sa = rise ? sar : saf;
    const float upd = sd::fma_(sa, d, slow);
    slow = (rise || fall) ? upd : slow;
    hang_n = rise ? 0u : (fall ? hang_n : hang_n + 1u);
    float lvl = fast > slow ? fast : slow;
    if (lvl < knee) lvl = knee;
    return lvl;
  });
  if (!live) return;
  s.hang_n[0] = hang_n; s.fast_level[0] = fast; s.slow_level[0] = slow;
}

// ---------------------------------------------------------------------------------------
// Gangs whose rows are columns of a time-major slab already (kernels.hpp GangSlab): what rows_tm_gather would build is
// what the producer -- the FFT filter bank writing channel c of every time step side by side -- left in memory.  The
// recurrence streams it where it lies: a wave-uniform base, the lane's byte offset from it, a run-time pitch.  The byte
// offsets of a chunk's CHUNK steps are loop invariants and stay in registers (a lone wavefront owns the register file), so
// a load is still scalar base + vector offset and nothing is added per sample.  Per lane the steps are the packed form's.
template <bool HAS_OUT, typename T, typename F>
__device__ __forceinline__ void gang_stream_slab(const T *tin, T *tout, long long pin, long long pout, uint32_t lo_in, uint32_t lo_out,
                                                 long long len, F step)
{
  const long long maxlen = uniform64(wave_max(len));
  if (maxlen <= 0) return;
  const long long minlen = uniform64(-wave_max(-len));       // (every lane has work: gang_lane)
  uint32_t oin[CHUNK], oout[CHUNK];
#pragma unroll
  for (int j = 0; j < CHUNK; ++j) {
    oin[j] = lo_in + (uint32_t)j * (uint32_t)pin * (uint32_t)sizeof(T);
    oout[j] = lo_out + (uint32_t)j * (uint32_t)pout * (uint32_t)sizeof(T);
  }
  T cur[CHUNK], nxt[CHUNK];
#pragma unroll
  for (int j = 0; j < CHUNK; ++j) cur[j] = ld_elem(tin, 0, oin[j]);
  long long i = 0;
  for (; i + CHUNK <= minlen; i += CHUNK) {                  // inside every row: one straight path
    const T *bn = tin + (i + CHUNK) * pin;
    T *bo = tout + i * pout;
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) nxt[j] = ld_elem(bn, 0, oin[j]);
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) {
      if constexpr (HAS_OUT) st_elem(bo, 0, oout[j], step(i + j, cur[j]));
      else step(i + j, cur[j]);
    }
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) cur[j] = nxt[j];
  }
  for (; i < maxlen; i += CHUNK) {
    const T *bn = tin + (i + CHUNK) * pin;
    T *bo = tout + i * pout;
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) nxt[j] = ld_elem(bn, 0, oin[j]);
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) {
      if (i + j < len) {
        if constexpr (HAS_OUT) st_elem(bo, 0, oout[j], step(i + j, cur[j]));
        else step(i + j, cur[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) cur[j] = nxt[j];
  }
}

__device__ __forceinline__ uint32_t slab_offset(const void *p, const void *base)
{
  return (uint32_t)(static_cast<const char *>(p) - static_cast<const char *>(base));
}

template <int KIND, int ORDER, bool GAIN1>
__device__ __forceinline__ void costas_gang_slab_body(const sdk::CostasGangItem *__restrict__ items, int count, const sdk::GangSlab &io)
{
  int j; bool live;
  if (!gang_lane(items, count, (int)threadIdx.x, &j, &live)) return;
  const sdk::CostasGangItem it = items[j];
  const sdk::CostasParams p = it.p;
  const sdk::CostasState s = it.s;
  CostasRegs<ORDER> r;
  r.phase = s.phase[0];
  r.omega = s.omega[0];
#pragma unroll
  for (int i = 1; i <= ORDER; ++i) {
    r.xh[i] = c32{s.xh[(i - 1) * 2 + 0], s.xh[(i - 1) * 2 + 1]};
    r.yh[i] = c32{s.yh[(i - 1) * 2 + 0], s.yh[(i - 1) * 2 + 1]};
  }
  const long long len = it.len;
  auto step = [&](long long, float2 v) { return costas_step<KIND, ORDER, GAIN1>(p, r, v); };
  // a pitch of 64 columns (at most 64 narrow inspectors on the shard -- BASELINE configs[3]'s slice) is the packed slabs' own:
  // the loop with immediate offsets
  if (io.pitch_in == 64 && io.pitch_out == 64)
    gang_stream_tm<true>(static_cast<const float2 *>(io.in), static_cast<float2 *>(io.out), slab_offset(it.x, io.in), slab_offset(it.y, io.out), len, step);
  else
    gang_stream_slab<true>(static_cast<const float2 *>(io.in), static_cast<float2 *>(io.out), io.pitch_in, io.pitch_out,
                           slab_offset(it.x, io.in), slab_offset(it.y, io.out), len, step);
  if (!live) return;
  s.phase[0] = r.phase;
  s.omega[0] = r.omega;
#pragma unroll
  for (int i2 = 1; i2 <= ORDER; ++i2) {
    s.xh[(i2 - 1) * 2 + 0] = r.xh[i2].re; s.xh[(i2 - 1) * 2 + 1] = r.xh[i2].im;
    s.yh[(i2 - 1) * 2 + 0] = r.yh[i2].re; s.yh[(i2 - 1) * 2 + 1] = r.yh[i2].im;
  }
}

__global__ __launch_bounds__(64) void costas_gang_slab_kernel(const sdk::CostasGangItem *__restrict__ items,
                                                              const sdk::GangGroup *__restrict__ groups, sdk::GangSlab io)
{
  const sdk::GangGroup gd = groups[blockIdx.x];
  const sdk::CostasGangItem *mine = items + gd.first;
  switch (gd.kind * 8 + gd.order) {
#define SD_GANG_CASE(K, O) case (K) * 8 + (O): if (gd.gain1) costas_gang_slab_body<K, O, true>(mine, gd.count, io); \
                                               else costas_gang_slab_body<K, O, false>(mine, gd.count, io); break;
    SD_GANG_CASE(1, 0) SD_GANG_CASE(1, 1) SD_GANG_CASE(1, 2) SD_GANG_CASE(1, 3) SD_GANG_CASE(1, 4)
    SD_GANG_CASE(2, 0) SD_GANG_CASE(2, 1) SD_GANG_CASE(2, 2) SD_GANG_CASE(2, 3) SD_GANG_CASE(2, 4)
    SD_GANG_CASE(3, 0) SD_GANG_CASE(3, 1) SD_GANG_CASE(3, 2) SD_GANG_CASE(3, 3) SD_GANG_CASE(3, 4)
#undef SD_GANG_CASE
    default: break;
  }
}

__global__ __launch_bounds__(64) void pll_gang_slab_kernel(const sdk::PllGangItem *__restrict__ items, int n, sdk::GangSlab io)
{
  const sdk::PllGangItem *mine = items + (size_t)blockIdx.x * 64;
  int j; bool live;
  if (!gang_lane(mine, n - (int)blockIdx.x * 64, (int)threadIdx.x, &j, &live)) return;
  const sdk::PllGangItem it = mine[j];
  const float alpha = it.alpha, beta = it.beta;
  uint32_t phase = it.s.phase[0];
  float omega = it.s.omega[0];
  const long long len = it.len;
  auto step = [&](long long, float2 v) { return pll_step(alpha, beta, phase, omega, v); };
  if (io.pitch_in == 64 && io.pitch_out == 64)
    gang_stream_tm<true>(static_cast<const float2 *>(io.in), static_cast<float2 *>(io.out), slab_offset(it.x, io.in), slab_offset(it.y, io.out), len, step);
  else
    gang_stream_slab<true>(static_cast<const float2 *>(io.in), static_cast<float2 *>(io.out), io.pitch_in, io.pitch_out,
                           slab_offset(it.x, io.in), slab_offset(it.y, io.out), len, step);
  if (!live) return;
  it.s.phase[0] = phase;
  it.s.omega[0] = omega;
}

__global__ __launch_bounds__(64) void clock_gang_slab_kernel(const sdk::ClockGangItem *__restrict__ items, int n, sdk::GangSlab io)
{
  __shared__ float2 lds[RING * 64];
  const int j = blockIdx.x * 64 + threadIdx.x;
  const bool live = j < n;
  const sdk::ClockGangItem it = items[live ? j : 0];
  sdk::ClockParams p = it.p;
  const int steps = __builtin_amdgcn_readfirstlane(it.steps);
  if (__builtin_amdgcn_readfirstlane(it.uniform)) {
    p.alpha = uniform_f(p.alpha); p.beta = uniform_f(p.beta); p.gain = uniform_f(p.gain); p.bmin = uniform_f(p.bmin); p.bmax = uniform_f(p.bmax);
  }
  const sdk::ClockState s = it.s;
  ClockRegs r;
  r.phi = s.phi[0]; r.bnor = s.bnor[0];
  r.halfcycle = s.halfcycle[0];
  r.prev = float2{s.prev[0], s.prev[1]};
  r.x0 = float2{s.x0[0], s.x0[1]};
  r.x1 = float2{s.x1[0], s.x1[1]};
  r.x2 = float2{s.x2[0], s.x2[1]};
  r.n = it.count[0];
  const long long len = live ? it.len : 0;
  float2 *out = reinterpret_cast<float2 *>(it.sym);
  const float2 *base = static_cast<const float2 *>(io.in);
  const uint32_t lo = slab_offset(it.x, io.in);
  if (steps > 0) clock_ring(base, io.pitch_in, lo, len, live, steps, p, r, out, lds);
  else clock_stream_tm<true>(base, io.pitch_in, lo, len, p, r, out, lds);
  if (!live) return;
  s.phi[0] = r.phi; s.bnor[0] = r.bnor; s.halfcycle[0] = r.halfcycle;
  s.prev[0] = r.prev.x; s.prev[1] = r.prev.y;
  s.x0[0] = r.x0.x; s.x0[1] = r.x0.y;
  s.x1[0] = r.x1.x; s.x1[1] = r.x1.y;
  s.x2[0] = r.x2.x; s.x2[1] = r.x2.y;
  it.count[0] = r.n;
}

__global__ __launch_bounds__(64) void agc_level_gang_slab_kernel(const sdk::AgcGangItem *__restrict__ items, int n, sdk::GangSlab io)
{
  const sdk::AgcGangItem *mine = items + (size_t)blockIdx.x * 64;
  int j; bool live;
  if (!gang_lane(mine, n - (int)blockIdx.x * 64, (int)threadIdx.x, &j, &live)) return;
  const sdk::AgcGangItem it = mine[j];
  const sdk::AgcState s = it.s;
  unsigned hang_n = s.hang_n[0];
  float fast = s.fast_level[0], slow = s.slow_level[0];
  const float far = it.p.fast_alpha_rise, faf = it.p.fast_alpha_fall, sar = it.p.slow_alpha_rise, saf = it.p.slow_alpha_fall;
  const float knee = it.p.knee;
  const unsigned hang_max = it.p.hang_max;
  const long long len = it.len;
  const uint32_t lo = slab_offset(it.peak, io.in);
  auto step = [&](long long, float pk) {
    float d = pk - fast;
    const float fa = d > 0.0f ? far : faf;
    fast = sd::fma_(fa, d, fast);
    d = pk - slow;
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
  if (io.pitch_in == 64 && io.pitch_out == 64) gang_stream_tm<true>(static_cast<const float *>(io.in), static_cast<float *>(io.out), lo, lo, len, step);
  else gang_stream_slab<true>(static_cast<const float *>(io.in), static_cast<float *>(io.out), io.pitch_in, io.pitch_out, lo, lo, len, step);
  if (!live) return;
  s.hang_n[0] = hang_n; s.fast_level[0] = fast; s.slow_level[0] = slow;
}

// The AGC's feed-forward steps on a slab: item k of the table is lane k % 64 of workgroup row k / 64, its samples sit in
// column it.lane of x, its magnitudes / peaks / levels in the same column of the work slabs (pitch `pw`).  A tile is 64
// items x SLAB_TM time steps, every access a row of adjacent columns.  Per value the operations (and, for the sliding
// maximum, the order of the comparisons: newest first) are agc_pre_items_kernel's / agc_apply_items_kernel's.
// (SLAB_TM = 32 time steps and a halo of the gang's longest history, not of the 63 steps the longest possible one takes: with
// 128 + 63 rows per tile a 2 Mi-sample block was 256 workgroups of 48 KB of LDS and took 52 us -- three times the row form,
// and the next block's channeliser, which wants the same LDS, ran beside it)
constexpr int SLAB_TM = 32;
__global__ __launch_bounds__(256) void agc_pre_slab_kernel(const sdk::AgcSlabItem *__restrict__ items, int n, const float2 *__restrict__ x,
                                                           long long px, float *__restrict__ db, float *__restrict__ peak, long long pw, int halo)
{
  __builtin_amdgcn_s_setprio(3);
  __shared__ float tile[SLAB_TM + 63][64];
  const int lane = threadIdx.x & 63, rowt = threadIdx.x >> 6;
  const int k = blockIdx.y * 64 + lane;
  const bool live = k < n;
  const sdk::AgcSlabItem it = items[live ? k : 0];
  const long long len = live ? it.len : 0;
  const int hl = (int)it.p.mag_history_size - 1;              // <= halo
  const long long col = it.lane;
  const long long m0 = (long long)blockIdx.x * SLAB_TM;
  for (int r = rowt; r < SLAB_TM + halo; r += 4) {            // tile row r <-> time m0 - halo + r
    const long long m = m0 - halo + r;
    float v = 0.f;
    if (m < len) {
      if (m >= 0) {
        const float2 s = x[m * px + col];
        v = 3.01029995663981195f * sd::log2_(sd::fma_(s.x, s.x, s.y * s.y) + 1e-8f);
        if (m >= m0) db[m * pw + col] = v;
      } else if (m + hl >= 0) v = it.s.mag_history[m + hl];
    }
    tile[r][lane] = v;
  }
  __syncthreads();
  for (int t = rowt; t < SLAB_TM; t += 4) {
    const long long m = m0 + t;
    if (m >= len) break;
    float pk = tile[t + halo][lane];
    for (int i = 1; i <= hl; ++i) {
      const float v = tile[t + halo - i][lane];
      pk = pk > v ? pk : v;
    }
    peak[m * pw + col] = pk;
  }
}

constexpr int APPLY_TM = 16;     // time steps per workgroup (64: an 8192-step sub-range was 128 workgroups and took 11 us; the row form 5)
__global__ __launch_bounds__(256) void agc_apply_slab_kernel(const sdk::AgcSlabItem *__restrict__ items, int n, const float2 *__restrict__ x,
                                                             long long px, float2 *__restrict__ y, long long py,
                                                             const float *__restrict__ lvl, long long pw, long long mlo)
{
  __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x & 63, rowt = threadIdx.x >> 6;
  const int k = blockIdx.y * 64 + lane;
  if (k >= n) return;
  const sdk::AgcSlabItem it = items[k];
  const long long delay = it.p.delay_line_size;
  const float slope = it.p.gain_slope - 1.0f;
  const long long mb = mlo + (long long)blockIdx.x * APPLY_TM;
  for (int r = rowt; r < APPLY_TM; r += 4) {
    const long long m = mb + r;
    if (m < it.m0 || m >= it.m1) continue;
    const float2 xd = m >= delay ? x[(m - delay) * px + it.lane] : float2{it.s.delay_line[m * 2 + 0], it.s.delay_line[m * 2 + 1]};
    const float g_db = lvl[m * pw + it.lane] * slope;
    const float g = sd::exp2_(g_db * 0.166096404744368117f) * 0.7f;
    y[m * py + it.lane_y] = float2{xd.x * g, xd.y * g};
  }
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

hipError_t rows_tm_gather(const void *d_items, int item_bytes, int off_ptr, int off_len, int n, const GangGroup *d_groups, int ngroups,
                          int elem_bytes, void *tm, long long slab, long long maxlen, hipStream_t st)
{
  if (n <= 0 || maxlen <= 0) return hipSuccess;
  const dim3 grid((unsigned)((maxlen + 63) / 64), (unsigned)(d_groups ? ngroups : (n + 63) / 64)), block(256);
  if (elem_bytes == 8)
    hipLaunchKernelGGL(rows_tm_gather_kernel<float2>, grid, block, 0, st, static_cast<const char *>(d_items), item_bytes, off_ptr, off_len, n,
                       d_groups, static_cast<float2 *>(tm), slab);
  else if (elem_bytes == 4)
    hipLaunchKernelGGL(rows_tm_gather_kernel<float>, grid, block, 0, st, static_cast<const char *>(d_items), item_bytes, off_ptr, off_len, n,
                       d_groups, static_cast<float *>(tm), slab);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t rows_tm_scatter(const void *d_items, int item_bytes, int off_ptr, int off_len, int n, const GangGroup *d_groups, int ngroups,
                           int elem_bytes, const void *tm, long long slab, long long maxlen, hipStream_t st)
{
  if (n <= 0 || maxlen <= 0) return hipSuccess;
  const dim3 grid((unsigned)((maxlen + 63) / 64), (unsigned)(d_groups ? ngroups : (n + 63) / 64)), block(256);
  if (elem_bytes == 8)
    hipLaunchKernelGGL(rows_tm_scatter_kernel<float2>, grid, block, 0, st, static_cast<const char *>(d_items), item_bytes, off_ptr, off_len, n,
                       d_groups, static_cast<const float2 *>(tm), slab);
  else if (elem_bytes == 4)
    hipLaunchKernelGGL(rows_tm_scatter_kernel<float>, grid, block, 0, st, static_cast<const char *>(d_items), item_bytes, off_ptr, off_len, n,
                       d_groups, static_cast<const float *>(tm), slab);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t agc_level_gang(const AgcGangItem *d_items, int n, void *tm, long long slab, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(agc_level_gang_kernel, dim3((n + 63) / 64), dim3(64), 0, st, d_items, n, static_cast<float *>(tm), slab);
  return hipGetLastError();
}

hipError_t pll_gang(const PllGangItem *d_items, int n, void *tm, long long slab, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(pll_gang_kernel, dim3((n + 63) / 64), dim3(64), 0, st, d_items, n, static_cast<float2 *>(tm), slab);
  return hipGetLastError();
}

hipError_t cma_gang(const CmaGangItem *d_items, int n, int ntaps, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  const dim3 grid((n + 63) / 64), block(64);
  switch (ntaps) {
#define SD_CMA_GANG(N) case N: hipLaunchKernelGGL(cma_gang_kernel<N>, grid, block, 0, st, d_items, n); break;
    SD_CMA_GANG(1) SD_CMA_GANG(2) SD_CMA_GANG(3) SD_CMA_GANG(4) SD_CMA_GANG(5) SD_CMA_GANG(6) SD_CMA_GANG(7) SD_CMA_GANG(8)
    SD_CMA_GANG(9) SD_CMA_GANG(10) SD_CMA_GANG(11) SD_CMA_GANG(12) SD_CMA_GANG(13) SD_CMA_GANG(14) SD_CMA_GANG(15) SD_CMA_GANG(16)
#undef SD_CMA_GANG
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t clock_gang(const ClockGangItem *d_items, int n, void *tm, long long slab, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(clock_gang_kernel, dim3((n + 63) / 64), dim3(64), 0, st, d_items, n, static_cast<float2 *>(tm), slab);
  return hipGetLastError();
}

hipError_t costas_gang(const CostasGangItem *d_items, const GangGroup *d_groups, int ngroups, void *tm, long long slab, hipStream_t st)
{
  if (ngroups <= 0) return hipSuccess;
  hipLaunchKernelGGL(costas_gang_kernel, dim3((unsigned)ngroups), dim3(64), 0, st, d_items, d_groups, static_cast<float2 *>(tm), slab);
  return hipGetLastError();
}

hipError_t costas_gang_slab(const CostasGangItem *d_items, const GangGroup *d_groups, int ngroups, GangSlab io, hipStream_t st)
{
  if (ngroups <= 0) return hipSuccess;
  hipLaunchKernelGGL(costas_gang_slab_kernel, dim3((unsigned)ngroups), dim3(64), 0, st, d_items, d_groups, io);
  return hipGetLastError();
}

hipError_t pll_gang_slab(const PllGangItem *d_items, int n, GangSlab io, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(pll_gang_slab_kernel, dim3((n + 63) / 64), dim3(64), 0, st, d_items, n, io);
  return hipGetLastError();
}

hipError_t clock_gang_slab(const ClockGangItem *d_items, int n, GangSlab io, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(clock_gang_slab_kernel, dim3((n + 63) / 64), dim3(64), 0, st, d_items, n, io);
  return hipGetLastError();
}

hipError_t agc_level_gang_slab(const AgcGangItem *d_items, int n, GangSlab io, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(agc_level_gang_slab_kernel, dim3((n + 63) / 64), dim3(64), 0, st, d_items, n, io);
  return hipGetLastError();
}

hipError_t agc_pre_slab(const AgcSlabItem *d_items, int n, const void *x, long long px, float *db, float *peak, long long pw,
                        long long max_len, int halo, hipStream_t st)
{
  if (n <= 0 || max_len <= 0) return hipSuccess;
  if (halo < 0 || halo > 63) return hipErrorInvalidValue;
  hipLaunchKernelGGL(agc_pre_slab_kernel, dim3((unsigned)((max_len + SLAB_TM - 1) / SLAB_TM), (unsigned)((n + 63) / 64)), dim3(256), 0, st,
                     d_items, n, static_cast<const float2 *>(x), px, db, peak, pw, halo);
  return hipGetLastError();
}

hipError_t agc_apply_slab(const AgcSlabItem *d_items, int n, const void *x, long long px, void *y, long long py, const float *lvl,
                          long long pw, long long mlo, long long mhi, hipStream_t st)
{
  if (n <= 0 || mhi <= mlo) return hipSuccess;
  hipLaunchKernelGGL(agc_apply_slab_kernel, dim3((unsigned)((mhi - mlo + APPLY_TM - 1) / APPLY_TM), (unsigned)((n + 63) / 64)), dim3(256), 0, st,
                     d_items, n, static_cast<const float2 *>(x), px, static_cast<float2 *>(y), py, lvl, pw, mlo);
  return hipGetLastError();
}

}  // namespace sdk
