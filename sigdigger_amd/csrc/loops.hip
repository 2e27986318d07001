// loops.hip -- element-wise demodulators (T5/T7/T11) and the per-channel recurrences
// (K6 Costas, K7 PLL, K8 Gardner clock recovery, K9 AGC) for gfx950.  SPEC.md sections E-H.
//
// The recurrences are serial in time and non-linear (sign(), wrap, data-dependent strobes),
// so the only exact parallel axis is the channel: ONE LANE PER CHANNEL, 64 channels per
// wavefront, state in registers for the whole block, samples streamed from the channel-major
// [channel][time] layout in 64-byte per-lane chunks that are prefetched one chunk ahead.
// Compiled with -ffp-contract=off: the arithmetic is the SPEC's fixed binary32 sequence and
// matches the CPU oracle bit for bit.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.hpp"
#include "sd_math.hpp"

namespace {

using sd::c32;
constexpr int CHUNK = 8;     // samples per lane per prefetch (64 bytes)

__device__ __forceinline__ void load_chunk(const float2 *__restrict__ row, long long i, long long len, float2 *buf)
{
#pragma unroll
  for (int j = 0; j < CHUNK; ++j) buf[j] = (i + j < len) ? row[i + j] : float2{0.0f, 0.0f};
}

// ---------------------------------------------------------------------------------------
// T5: QuadDemodTask::work  dest[p] = j/pi * arg(x[p] conj(x[p-1]))
__global__ void quad_demod_kernel(const float2 *__restrict__ x, long long xs, float2 *__restrict__ y, long long ys,
                                  long long len, const float2 *__restrict__ prev, int first,
                                  float2 *__restrict__ prev_out)
{
  const int c = blockIdx.y;
  const float2 *xr = x + (long long)c * xs;
  float2 *yr = y + (long long)c * ys;
  const float k = 0.318309886183790671538f;
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < len;
       p += (long long)gridDim.x * blockDim.x) {
    const float2 v = xr[p];
    float2 out;
    if (p == 0 && first) {
      out = float2{0.0f, 0.0f};
    } else {
      const float2 pv = (p == 0) ? prev[c] : xr[p - 1];
      const c32 d = sd::cmul_conj(c32{v.x, v.y}, c32{pv.x, pv.y});
      out = float2{0.0f, k * sd::atan2_(d.im, d.re)};
    }
    yr[p] = out;
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
// K6: Costas loop
template <int ORDER> struct CostasRegs {
  uint32_t phase; float omega;
  c32 xh[ORDER + 1], yh[ORDER + 1];
};

template <int ORDER>
__device__ __forceinline__ float2 costas_step(const sdk::CostasParams &p, CostasRegs<ORDER> &r, float2 v)
{
  // history part of the arm filter first: it does not depend on the new sample
  float tr = 0.0f, ti = 0.0f;
#pragma unroll
  for (int q = ORDER; q >= 1; --q) { tr = sd::fma_(p.fb[q], r.xh[q].re, tr); ti = sd::fma_(p.fb[q], r.xh[q].im, ti); }
#pragma unroll
  for (int q = ORDER; q >= 1; --q) { tr = sd::fma_(-p.fa[q], r.yh[q].re, tr); ti = sd::fma_(-p.fa[q], r.yh[q].im, ti); }
  float cs, sn;
  sd::phasor_u32(r.phase, cs, sn);
  c32 m;                                                      // x * conj(ref)
  m.re = sd::fma_(v.y, sn, v.x * cs);
  m.im = sd::fma_(v.y, cs, -(v.x * sn));
  c32 z;
  z.re = sd::fma_(p.fb[0], m.re, tr);
  z.im = sd::fma_(p.fb[0], m.im, ti);
#pragma unroll
  for (int q = ORDER; q >= 2; --q) { r.xh[q] = r.xh[q - 1]; r.yh[q] = r.yh[q - 1]; }
  if (ORDER >= 1) { r.xh[1] = m; r.yh[1] = z; }
  z.re = p.gain * z.re;
  z.im = p.gain * z.im;
  float e;
  if (p.kind == 1) {
    e = z.re * z.im;
  } else if (p.kind == 2) {
    e = sd::sgn(z.re) * z.im - sd::sgn(z.im) * z.re;
  } else {
    if (__builtin_fabsf(z.re) >= __builtin_fabsf(z.im))
      e = sd::sgn(z.re) * z.im - (sd::sgn(z.im) * z.re) * 0.41421356237309504880f;
    else
      e = (sd::sgn(z.re) * z.im) * 0.41421356237309504880f - sd::sgn(z.im) * z.re;
  }
  const float dphi = sd::fma_(p.a, e, r.omega);
  r.omega = sd::fma_(p.b, e, r.omega);
  r.phase += (uint32_t)sd::rad_to_dphase(dphi);
  return float2{z.re, z.im};
}

template <int ORDER>
__global__ __launch_bounds__(64) void costas_kernel(sdk::CostasParams p, sdk::CostasState s, int nchan,
                                                    const float2 *__restrict__ x, long long xs,
                                                    float2 *__restrict__ y, long long ys, long long len)
{
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= nchan) return;
  CostasRegs<ORDER> r;
  r.phase = s.phase[c];
  r.omega = s.omega[c];
#pragma unroll
  for (int i = 1; i <= ORDER; ++i) {
    r.xh[i] = c32{s.xh[((i - 1) * 2 + 0) * nchan + c], s.xh[((i - 1) * 2 + 1) * nchan + c]};
    r.yh[i] = c32{s.yh[((i - 1) * 2 + 0) * nchan + c], s.yh[((i - 1) * 2 + 1) * nchan + c]};
  }
  const float2 *xr = x + (long long)c * xs;
  float2 *yr = y + (long long)c * ys;
  float2 cur[CHUNK], nxt[CHUNK];
  load_chunk(xr, 0, len, cur);
  for (long long i = 0; i < len; i += CHUNK) {
    load_chunk(xr, i + CHUNK, len, nxt);
#pragma unroll
    for (int j = 0; j < CHUNK; ++j)
      if (i + j < len) yr[i + j] = costas_step<ORDER>(p, r, cur[j]);     // len is wave-uniform
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) cur[j] = nxt[j];
  }
  s.phase[c] = r.phase;
  s.omega[c] = r.omega;
#pragma unroll
  for (int i = 1; i <= ORDER; ++i) {
    s.xh[((i - 1) * 2 + 0) * nchan + c] = r.xh[i].re; s.xh[((i - 1) * 2 + 1) * nchan + c] = r.xh[i].im;
    s.yh[((i - 1) * 2 + 0) * nchan + c] = r.yh[i].re; s.yh[((i - 1) * 2 + 1) * nchan + c] = r.yh[i].im;
  }
}

// ---------------------------------------------------------------------------------------
// K7: PLL
__global__ __launch_bounds__(64) void pll_kernel(float alpha, float beta, sdk::PllState s, int nchan,
                                                 const float2 *__restrict__ x, long long xs,
                                                 float2 *__restrict__ y, long long ys, long long len)
{
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= nchan) return;
  uint32_t phase = s.phase[c];
  float omega = s.omega[c];
  const float2 *xr = x + (long long)c * xs;
  float2 *yr = y + (long long)c * ys;
  float2 cur[CHUNK], nxt[CHUNK];
  load_chunk(xr, 0, len, cur);
  for (long long i = 0; i < len; i += CHUNK) {
    load_chunk(xr, i + CHUNK, len, nxt);
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) {
      if (i + j < len) {
        const float2 v = cur[j];
        float cs, sn;
        sd::phasor_u32(phase, cs, sn);
        float2 m;
        m.x = sd::fma_(v.y, sn, v.x * cs);
        m.y = sd::fma_(v.y, cs, -(v.x * sn));
        float err = sd::atan2_(v.y, v.x) - sd::phase_to_rad(phase);
        if (err >  3.14159265358979323846f) err -= 6.28318530717958647692f;
        if (err < -3.14159265358979323846f) err += 6.28318530717958647692f;
        const float dphi = sd::fma_(beta, err, omega);
        omega = sd::fma_(alpha, err, omega);
        phase += (uint32_t)sd::rad_to_dphase(dphi);
        yr[i + j] = m;
      }
    }
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) cur[j] = nxt[j];
  }
  s.phase[c] = phase;
  s.omega[c] = omega;
}

// ---------------------------------------------------------------------------------------
// K8: Gardner clock recovery; variable-rate output, per-lane append
__global__ __launch_bounds__(64) void clock_kernel(sdk::ClockParams p, sdk::ClockState s, int nchan,
                                                   const float2 *__restrict__ x, long long xs, long long len,
                                                   float2 *__restrict__ sym, long long sym_stride,
                                                   uint32_t *__restrict__ count)
{
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= nchan) return;
  float phi = s.phi[c], bnor = s.bnor[c];
  int halfcycle = s.halfcycle[c];
  float2 prev = float2{s.prev[c], s.prev[nchan + c]};
  float2 x0 = float2{s.x0[c], s.x0[nchan + c]};
  float2 x1 = float2{s.x1[c], s.x1[nchan + c]};
  float2 x2 = float2{s.x2[c], s.x2[nchan + c]};
  uint32_t n = count[c];
  const float2 *xr = x + (long long)c * xs;
  float2 *out = sym + (long long)c * sym_stride;
  float2 cur[CHUNK], nxt[CHUNK];
  load_chunk(xr, 0, len, cur);
  for (long long i = 0; i < len; i += CHUNK) {
    load_chunk(xr, i + CHUNK, len, nxt);
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) {
      if (i + j < len) {
        const float2 v = cur[j];
        phi = phi + bnor;
        if (phi >= 0.5f) {
          const float mu = (phi - 0.5f) / bnor;
          float2 q;
          q.x = sd::fma_(mu, prev.x - v.x, v.x);
          q.y = sd::fma_(mu, prev.y - v.y, v.y);
          phi = phi - 0.5f;
          halfcycle = !halfcycle;
          if (!halfcycle) {
            x2 = x0;
            x0 = q;
            const float dr = x0.x - x2.x, di = x0.y - x2.y;
            const float e = p.gain * sd::fma_(x1.y, di, x1.x * dr);
            phi = sd::fma_(p.alpha, e, phi);
            float b = sd::fma_(p.beta, e, bnor);
            if (b < p.bmin) b = p.bmin;
            if (b > p.bmax) b = p.bmax;
            bnor = b;
            out[n++] = q;
          } else {
            x1 = q;
          }
        }
        prev = v;
      }
    }
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) cur[j] = nxt[j];
  }
  s.phi[c] = phi; s.bnor[c] = bnor; s.halfcycle[c] = halfcycle;
  s.prev[c] = prev.x; s.prev[nchan + c] = prev.y;
  s.x0[c] = x0.x; s.x0[nchan + c] = x0.y;
  s.x1[c] = x1.x; s.x1[nchan + c] = x1.y;
  s.x2[c] = x2.x; s.x2[nchan + c] = x2.y;
  count[c] = n;
}

// ---------------------------------------------------------------------------------------
// K9: AGC.  Delay line and magnitude history (<= 64 entries each) live in LDS, laid out
// [entry][lane] so that the common case (all lanes at the same ring position) is conflict-free.
__global__ __launch_bounds__(64) void agc_kernel(sdk::AgcParams p, sdk::AgcState s, int nchan,
                                                 const float2 *__restrict__ x, long long xs,
                                                 float2 *__restrict__ y, long long ys, long long len)
{
  __shared__ float dl_re[64][64], dl_im[64][64], mh[64][64];
  const int lane = threadIdx.x;
  const int c = blockIdx.x * 64 + lane;
  if (c >= nchan) return;
  for (unsigned i = 0; i < p.delay_line_size; ++i) {
    dl_re[i][lane] = s.delay_line[(i * 2 + 0) * nchan + c];
    dl_im[i][lane] = s.delay_line[(i * 2 + 1) * nchan + c];
  }
  for (unsigned i = 0; i < p.mag_history_size; ++i) mh[i][lane] = s.mag_history[i * nchan + c];
  unsigned dptr = s.delay_ptr[c], hptr = s.hist_ptr[c], hang_n = s.hang_n[c];
  float peak = s.peak[c], fast = s.fast_level[c], slow = s.slow_level[c];
  const float2 *xr = x + (long long)c * xs;
  float2 *yr = y + (long long)c * ys;
  float2 cur[CHUNK], nxt[CHUNK];
  load_chunk(xr, 0, len, cur);
  for (long long i = 0; i < len; i += CHUNK) {
    load_chunk(xr, i + CHUNK, len, nxt);
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) {
      if (i + j < len) {
        const float2 v = cur[j];
        const float2 xd = float2{dl_re[dptr][lane], dl_im[dptr][lane]};
        dl_re[dptr][lane] = v.x; dl_im[dptr][lane] = v.y;
        if (++dptr == p.delay_line_size) dptr = 0;
        const float pw = sd::fma_(v.x, v.x, v.y * v.y) + 1e-8f;
        const float x_db = 3.01029995663981195f * sd::log2_(pw);
        const float x_db_old = mh[hptr][lane];
        mh[hptr][lane] = x_db;
        if (++hptr == p.mag_history_size) hptr = 0;
        if (peak < x_db) {
          peak = x_db;
        } else if (peak == x_db_old) {
          float pk = -160.0f;
          for (unsigned q = 0; q < p.mag_history_size; ++q) { const float h = mh[q][lane]; if (pk < h) pk = h; }
          peak = pk;
        }
        float d = peak - fast;
        fast = sd::fma_(d > 0.0f ? p.fast_alpha_rise : p.fast_alpha_fall, d, fast);
        d = peak - slow;
        if (d > 0.0f) {
          slow = sd::fma_(p.slow_alpha_rise, d, slow);
          hang_n = 0;
        } else if (hang_n >= p.hang_max) {
          slow = sd::fma_(p.slow_alpha_fall, d, slow);
        } else {
          ++hang_n;
        }
        float lvl = fast > slow ? fast : slow;
        if (lvl < p.knee) lvl = p.knee;
        const float g_db = lvl * (p.gain_slope - 1.0f);
        const float g = sd::exp2_(g_db * 0.166096404744368117f) * 0.7f;
        yr[i + j] = float2{xd.x * g, xd.y * g};
      }
    }
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) cur[j] = nxt[j];
  }
  for (unsigned i = 0; i < p.delay_line_size; ++i) {
    s.delay_line[(i * 2 + 0) * nchan + c] = dl_re[i][lane];
    s.delay_line[(i * 2 + 1) * nchan + c] = dl_im[i][lane];
  }
  for (unsigned i = 0; i < p.mag_history_size; ++i) s.mag_history[i * nchan + c] = mh[i][lane];
  s.delay_ptr[c] = dptr; s.hist_ptr[c] = hptr; s.hang_n[c] = hang_n;
  s.peak[c] = peak; s.fast_level[c] = fast; s.slow_level[c] = slow;
}

inline unsigned grid_for(long long n, int block) {
  long long g = (n + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

namespace sdk {

hipError_t quad_demod_batch(const void *x, long long xs, void *y, long long ys, int nchan, long long len,
                            const void *prev, int first, void *prev_out, hipStream_t st)
{
  if (len <= 0 || nchan <= 0) return hipSuccess;
  unsigned gx = grid_for(len, 256);
  if ((long long)gx * nchan > 8192) gx = (unsigned)((8192 + nchan - 1) / nchan);
  hipLaunchKernelGGL(quad_demod_kernel, dim3(gx, (unsigned)nchan), dim3(256), 0, st,
                     reinterpret_cast<const float2 *>(x), xs, reinterpret_cast<float2 *>(y), ys, len,
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

hipError_t costas_feed(const CostasParams &p, const CostasState &s, int nchan, const void *x, long long xs,
                       void *y, long long ys, long long len, hipStream_t st)
{
  if (len <= 0 || nchan <= 0) return hipSuccess;
  const dim3 grid((nchan + 63) / 64), block(64);
  const float2 *xx = reinterpret_cast<const float2 *>(x);
  float2 *yy = reinterpret_cast<float2 *>(y);
  switch (p.order) {
    case 0: hipLaunchKernelGGL(costas_kernel<0>, grid, block, 0, st, p, s, nchan, xx, xs, yy, ys, len); break;
    case 1: hipLaunchKernelGGL(costas_kernel<1>, grid, block, 0, st, p, s, nchan, xx, xs, yy, ys, len); break;
    case 2: hipLaunchKernelGGL(costas_kernel<2>, grid, block, 0, st, p, s, nchan, xx, xs, yy, ys, len); break;
    case 3: hipLaunchKernelGGL(costas_kernel<3>, grid, block, 0, st, p, s, nchan, xx, xs, yy, ys, len); break;
    case 4: hipLaunchKernelGGL(costas_kernel<4>, grid, block, 0, st, p, s, nchan, xx, xs, yy, ys, len); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t pll_feed(float alpha, float beta, const PllState &s, int nchan, const void *x, long long xs,
                    void *y, long long ys, long long len, hipStream_t st)
{
  if (len <= 0 || nchan <= 0) return hipSuccess;
  hipLaunchKernelGGL(pll_kernel, dim3((nchan + 63) / 64), dim3(64), 0, st, alpha, beta, s, nchan,
                     reinterpret_cast<const float2 *>(x), xs, reinterpret_cast<float2 *>(y), ys, len);
  return hipGetLastError();
}

hipError_t clock_feed(const ClockParams &p, const ClockState &s, int nchan, const void *x, long long xs,
                      long long len, void *sym, long long sym_stride, uint32_t *count, hipStream_t st)
{
  if (len <= 0 || nchan <= 0) return hipSuccess;
  hipLaunchKernelGGL(clock_kernel, dim3((nchan + 63) / 64), dim3(64), 0, st, p, s, nchan,
                     reinterpret_cast<const float2 *>(x), xs, len, reinterpret_cast<float2 *>(sym), sym_stride, count);
  return hipGetLastError();
}

hipError_t agc_feed(const AgcParams &p, const AgcState &s, int nchan, const void *x, long long xs,
                    void *y, long long ys, long long len, hipStream_t st)
{
  if (len <= 0 || nchan <= 0) return hipSuccess;
  hipLaunchKernelGGL(agc_kernel, dim3((nchan + 63) / 64), dim3(64), 0, st, p, s, nchan,
                     reinterpret_cast<const float2 *>(x), xs, reinterpret_cast<float2 *>(y), ys, len);
  return hipGetLastError();
}

}  // namespace sdk
