// loops_dev.hpp -- device code the recurrence translation units share (loops.hip: the banks, one lane per channel of a
// bank; gangs.hip: the gangs, one lane per 1-channel bank): the streams' element access, the Costas / PLL / Gardner steps
// and the Gardner schedules.  Split out of loops.hip in round 6 so that the two files compile side by side (one file took
// six minutes).  Everything here is __device__ __forceinline__ or a template, in the unnamed namespace of its includer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.hpp"
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

__device__ __forceinline__ long long wave_max(long long v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const long long w = __shfl_xor(v, o); v = w > v ? w : v; }
  return v;
}

__device__ __forceinline__ long long uniform64(long long v)
{
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(unsigned long long)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}

__device__ __forceinline__ float uniform_f(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }


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

}  // namespace
