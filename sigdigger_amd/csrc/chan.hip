// chan.hip -- NCO carrier translate (K4) and the inspector channel bank: translate +
// low-pass + decimate as a complex band-pass polyphase FIR with baseband de-rotation (K4+K5).
// SPEC.md section C.  Compiled with -ffp-contract=off: every output sample is the SPEC's fixed
// chain of binary32 fma operations (taps k ascending), so results are bit-identical to the
// CPU oracle.
//
// chan_fir kernel (variant A, any channel count):
//   * a workgroup owns MT consecutive output instants m of ALL channels of the bank;
//   * the input window those outputs need ((MT-1)*D + ntaps samples) is staged ONCE in LDS,
//     read from HBM with coalesced 8-byte loads, and re-used by every channel -> compulsory
//     HBM traffic 8 B/input sample (+ ntaps/D overlap) + 8*C/D B of output;
//   * LDS layout is polyphase-transposed: sample i of the window sits at
//     row (i mod D), column (i / D).  Lanes of a wave work on consecutive m, so for a given
//     tap every lane reads the same row at consecutive columns: conflict-free ds_read_b64
//     for any D;
//   * taps are wave-uniform (a wave works on NCH channels at a time): they are fetched with
//     scalar loads and used as SGPR operands of v_fma_f32; each LDS read feeds 4*NCH fmas.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "kernels.hpp"
#include "tuning.hpp"
#include "sd_math.hpp"

namespace {

using sd::c32;
typedef float v2f __attribute__((ext_vector_type(2)));

// acc += g * x for one tap, as the two packed fmas of SPEC.md section C:
//   (acc.re, acc.im) = fma((g.re, g.re), (x.re, x.im), acc);  then fma((-g.im, g.im), (x.im, x.re), acc)
// per component: acc.re = fma(g.re, x.re, acc.re); acc.re = fma(-g.im, x.im, acc.re);
//                acc.im = fma(g.re, x.im, acc.im); acc.im = fma( g.im, x.re, acc.im)   (exact fmas)
__device__ __forceinline__ v2f tap_mac(v2f acc, float4 t, v2f x)
{
  const v2f t0 = {t.x, t.y}, t1 = {t.z, t.w};
  acc = __builtin_elementwise_fma(t0, x, acc);
  acc = __builtin_elementwise_fma(t1, x.yx, acc);
  return acc;
}

// ---------------------------------------------------------------------------------------
// T1/K4: y[i] = x[i] * phasor(p0 + (n0+i)*dp)      (Tasks/CarrierXlator.cpp:57-60)
// two samples per thread: 16-byte loads/stores
__global__ void xlate_kernel(const float4 *__restrict__ x, float4 *__restrict__ y, long long npairs,
                             long long len, uint32_t p0, uint32_t dp, uint64_t n0)
{
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < npairs;
       t += (long long)gridDim.x * blockDim.x) {
    const uint64_t n = n0 + 2ull * (uint64_t)t;
    const uint32_t pa = p0 + (uint32_t)(n * (uint64_t)dp);
    const uint32_t pb = pa + dp;
    if (2 * t + 1 < len) {
      const float4 v = x[t];
      float c, s;
      sd::phasor_u32(pa, c, s);
      const c32 a = sd::cmul_cs(c32{v.x, v.y}, c, s);
      sd::phasor_u32(pb, c, s);
      const c32 b = sd::cmul_cs(c32{v.z, v.w}, c, s);
      y[t] = float4{a.re, a.im, b.re, b.im};
    } else {                                            // odd tail
      const float2 v = reinterpret_cast<const float2 *>(x)[2 * t];
      float c, s;
      sd::phasor_u32(pa, c, s);
      const c32 a = sd::cmul_cs(c32{v.x, v.y}, c, s);
      reinterpret_cast<float2 *>(y)[2 * t] = float2{a.re, a.im};
    }
  }
}

// g[c][k] = h[k] * phasor(-(k*dp_c)), stored as (re, re, -im, im): the operand pairs of the two
// packed fmas that accumulate (acc.re, acc.im) -- the SGPR pairs come straight out of s_load,
// no scalar ALU work per tap.
__global__ void modulate_taps_kernel(const float *__restrict__ h, int ntaps, const uint32_t *__restrict__ dphase,
                                     int nchan, float4 *__restrict__ g, float2 *__restrict__ g2)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntaps * nchan) return;
  const int c = t / ntaps, k = t - c * ntaps;
  float cs, sn;
  sd::phasor_u32(0u - (uint32_t)k * dphase[c], cs, sn);
  const float re = h[k] * cs, im = h[k] * sn;
  g[t] = float4{re, re, -im, im};
  if (g2) g2[t] = float2{re, im};                      // chan_stream.hip: (re, im) pairs for scalar loads
}

// ---------------------------------------------------------------------------------------
using sdk::FirGeom;

constexpr int FIR_THREADS = 512;   // 8 waves share one staged window

// One chunk of TC consecutive taps for NCH channels and NOUT outputs per lane: the window samples
// (one ds_read_b64 each, immediate offsets off one base address) and the taps (LDS broadcast reads,
// or scalar loads).  A tap fetched once feeds NOUT*2 packed fmas per lane.
template <int NCH, int NOUT, int TC> struct FirChunk {
  v2f x[NOUT][TC];
  float4 t[NCH][TC];
};

template <int NCH, int NOUT, int TC>
__device__ __forceinline__ void fir_load(FirChunk<NCH, NOUT, TC> &c, const float2 *__restrict__ xlo, int ostride,
                                         const float4 *const (&gp)[NCH], int k)
{
  // taps k .. k+TC-1 read window samples at DEcreasing addresses: xlo points at the sample of tap
  // k+TC-1 of the lane's first output; its other outputs are ostride samples further on
#pragma unroll
  for (int o = 0; o < NOUT; ++o) {
#pragma unroll
    for (int q = 0; q < TC; ++q) { const float2 w = xlo[o * ostride + TC - 1 - q]; c.x[o][q] = v2f{w.x, w.y}; }
  }
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
#pragma unroll
    for (int q = 0; q < TC; ++q) c.t[j][q] = gp[j][k + q];
  }
}

template <int NCH, int NOUT, int TC, bool LDS_TAPS>
__device__ __forceinline__ void fir_touch(const FirChunk<NCH, NOUT, TC> &c)
{
#pragma unroll
  for (int o = 0; o < NOUT; ++o) asm volatile("" :: "v"(c.x[o][0]), "v"(c.x[o][TC - 1]));
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    if (LDS_TAPS) asm volatile("" :: "v"(c.t[j][0].x), "v"(c.t[j][TC - 1].w));
    else          asm volatile("" :: "s"(c.t[j][0].x), "s"(c.t[j][TC - 1].w));
  }
}

template <int NCH, int NOUT, int TC>
__device__ __forceinline__ void fir_mac(v2f (&acc)[NCH][NOUT], const FirChunk<NCH, NOUT, TC> &c)
{
  // tap-major: consecutive packed fmas hit different accumulators; per accumulator the order is
  // the SPEC's (k ascending; (re,re)*x then (-im,im)*x.yx)
#pragma unroll
  for (int q = 0; q < TC; ++q) {
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
      for (int o = 0; o < NOUT; ++o)
        acc[j][o] = __builtin_elementwise_fma(v2f{c.t[j][q].x, c.t[j][q].y}, c.x[o][q], acc[j][o]);
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
      for (int o = 0; o < NOUT; ++o)
        acc[j][o] = __builtin_elementwise_fma(v2f{c.t[j][q].z, c.t[j][q].w}, c.x[o][q].yx, acc[j][o]);
  }
}

// LDS_TAPS: few channels (C*T*16 B fits beside the window): the modulated taps are staged in LDS
// and read with broadcast ds_read_b128 instead of scalar loads.
//
// Window layout in LDS: row-major with PAD samples of padding after every D samples
// (index(i) = i + (i/D)*PAD, D + PAD odd).  Lanes hold consecutive outputs m, i.e. window
// positions D apart -> lane stride D+PAD (odd) -> conflict-free ds_read_b64; consecutive taps of
// one output sit at consecutive addresses, so a chunk of TC taps is TC reads with immediate
// offsets off one base address.  Chunks are double-buffered: the loads of chunk c+1 are issued
// before the fmas of chunk c.
// NOUT: outputs per lane (64 apart); DB: double-buffer the chunks (not when the taps of two chunks
// would not fit the SGPR file).
template <int NCH, int NOUT, int TC, bool LDS_TAPS, bool DB>
__device__ __forceinline__ void chan_fir_body(const float2 *__restrict__ x, const float2 *__restrict__ hist,
                                              float2 *__restrict__ hist_next,
                                              long long len, uint64_t n0,
                                              const float4 *__restrict__ g,
                                              const uint32_t *__restrict__ dphase,
                                              const uint32_t *__restrict__ phase0, const FirGeom &ge,
                                              uint64_t m_first, long long n_out,
                                              float2 *__restrict__ y, sdk::View yv,
                                              const unsigned tile, const unsigned ntiles)   // this workgroup's tile of the feed
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2 *win = reinterpret_cast<float2 *>(smem);
  // The recurrence kernels (Costas / AGC / Gardner) keep one wavefront resident for milliseconds on
  // other streams.  That wave is the oldest on its SIMD and wins issue arbitration, so the two
  // workgroups of this grid that share its CU ran 3.5x longer and, the grid being a single resident
  // round, so did the whole kernel (125 -> 370 us, tools/corun3.py).  Raising the issue priority of
  // these waves puts the straggler behind the throughput kernel instead.
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x;
  const int D = ge.D, T = ge.ntaps, PAD = ge.PAD;
  const long long tile_m0 = (long long)tile * ge.MT;                 // relative to m_first
  const long long nbase = (long long)(m_first + (uint64_t)tile_m0) * D - ge.KD;   // absolute index of window sample 0
  const int span = ge.span;
  const long long hist0 = (long long)n0 - (T - 1);                    // absolute index of hist[0]

  // ---- carry: history for the next block = last T-1 samples of [hist ; x] (ping-pong buffer) ----
  if (tile == ntiles - 1) {
    const int hl = T - 1;
    for (int i = tid; i < hl; i += FIR_THREADS) {
      const long long src = (long long)i + len;                       // index into [hist ; x]
      hist_next[i] = src < hl ? hist[src] : x[src - hl];
    }
  }

  float4 *ltaps = reinterpret_cast<float4 *>(win + (((size_t)ge.lds_samples + 1) & ~(size_t)1));   // 16-B aligned
  if (LDS_TAPS) {
    for (int i = tid; i < ge.nchan * T; i += FIR_THREADS) ltaps[i] = g[i];
  }
  // ---- stage the window: coalesced HBM reads, (nearly) linear LDS writes, 8 loads in flight ----
  const bool interior = nbase >= (long long)n0 && nbase + span <= (long long)n0 + len;   // wave-uniform
  const int dsh = (D & (D - 1)) == 0 ? __builtin_ctz((unsigned)D) : -1;
  const float2 *xin = x + (nbase - (long long)n0);
  for (int i0 = tid; i0 < span; i0 += 8 * FIR_THREADS) {
    float2 v[8];
    if (interior) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * FIR_THREADS;
        v[u] = xin[i < span ? i : span - 1];
      }
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * FIR_THREADS;
        const long long n = nbase + i;
        float2 w = float2{0.0f, 0.0f};
        if (i < span) {
          if (n >= (long long)n0) {
            if (n < (long long)n0 + len) w = x[n - (long long)n0];
          } else if (n >= hist0) {
            w = hist[n - hist0];
          }
        }
        v[u] = w;
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * FIR_THREADS;
      if (i < span) {
        const int blk = dsh >= 0 ? (i >> dsh) : (int)((unsigned)i / (unsigned)D);
        win[i + blk * PAD] = v[u];
      }
    }
  }
  __syncthreads();

  // ---- units of work: (64*NOUT-output sub-tile, group of NCH channels) round-robin over the waves ----
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int nsub = ge.MT / (64 * NOUT);
  const int ngrp = (ge.nchan + NCH - 1) / NCH;
  const int nunits = nsub * ngrp;
  const int S = D + PAD;                                              // lane stride in the window
  const int kd_blk = ge.KD / D;

  for (int u = wave; u < nunits; u += FIR_THREADS / 64) {
    const int sub = u / ngrp;
    const int c0  = (u - sub * ngrp) * NCH;
    const int ml  = sub * 64 * NOUT + lane;                           // first output of this lane within the tile
    v2f acc[NCH][NOUT];
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
      for (int o = 0; o < NOUT; ++o) acc[j][o] = v2f{0.0f, 0.0f};
    const float4 *gp[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = (c0 + j < ge.nchan) ? c0 + j : ge.nchan - 1;      // clamp: duplicates are not stored
      gp[j] = (LDS_TAPS ? (const float4 *)ltaps : g) + (long long)c * T;
    }
    // tap k of output ml reads window sample e = ml*D + KD - k, stored at ml*S + (KD-k) + ((KD-k)/D)*PAD.
    // k = 0 is alone in its D-block (KD is a multiple of D); afterwards every D-block is a run of
    // D taps at consecutive, decreasing addresses.  Output ml + 64*o sits 64*S samples further on.
    const float2 *lbase = win + ml * S;
    const int ostride = 64 * S;
    int k = 0;
    int blk = kd_blk;                                                  // (KD - k) / D of the current run
    int e = ge.KD;                                                     // KD - k
    while (k < T) {
      const int run_full = e - blk * D + 1;                            // taps left in this D-block
      const int run = run_full < T - k ? run_full : T - k;
      const float2 *xr = lbase + e + blk * PAD;                        // sample of tap k+r: xr[-r]
      int r = 0;
      if (run >= TC) {
        FirChunk<NCH, NOUT, TC> ca;
        if (DB && LDS_TAPS) {
          // taps and window both come from LDS (in order): counted waits let chunk c+1 load during c's fmas
          FirChunk<NCH, NOUT, TC> cb;
          fir_load<NCH, NOUT, TC>(ca, xr - (TC - 1), ostride, gp, k);
          for (; r + 2 * TC <= run; r += TC) {
            fir_load<NCH, NOUT, TC>(cb, xr - (r + 2 * TC - 1), ostride, gp, k + r + TC);
            fir_mac<NCH, NOUT, TC>(acc, ca);
            ca = cb;
          }
          fir_mac<NCH, NOUT, TC>(acc, ca);
          r += TC;
        } else if (DB) {
          // scalar taps: ping-pong, no register copies.  Scalar loads return out of order and share lgkmcnt
          // with the LDS reads, so every wait is lgkmcnt(0): fir_touch() (an empty asm reading the chunk's
          // registers) makes the compiler wait for chunk c BEFORE chunk c+1's loads are issued -- then c+1
          // is in flight during c's fmas and is the only thing the next wait covers.
          FirChunk<NCH, NOUT, TC> cb;
          fir_load<NCH, NOUT, TC>(ca, xr - (TC - 1), ostride, gp, k);
          for (; r + 3 * TC <= run; r += 2 * TC) {
            fir_touch<NCH, NOUT, TC, LDS_TAPS>(ca);
            fir_load<NCH, NOUT, TC>(cb, xr - (r + 2 * TC - 1), ostride, gp, k + r + TC);
            __builtin_amdgcn_sched_barrier(0);
            fir_mac<NCH, NOUT, TC>(acc, ca);
            __builtin_amdgcn_sched_barrier(0);
            fir_touch<NCH, NOUT, TC, LDS_TAPS>(cb);
            fir_load<NCH, NOUT, TC>(ca, xr - (r + 3 * TC - 1), ostride, gp, k + r + 2 * TC);
            __builtin_amdgcn_sched_barrier(0);
            fir_mac<NCH, NOUT, TC>(acc, cb);
            __builtin_amdgcn_sched_barrier(0);
          }
          fir_mac<NCH, NOUT, TC>(acc, ca);
          r += TC;
          for (; r + TC <= run; r += TC) {
            fir_load<NCH, NOUT, TC>(ca, xr - (r + TC - 1), ostride, gp, k + r);
            fir_mac<NCH, NOUT, TC>(acc, ca);
          }
        } else {
          for (; r + TC <= run; r += TC) {
            fir_load<NCH, NOUT, TC>(ca, xr - (r + TC - 1), ostride, gp, k + r);
            fir_mac<NCH, NOUT, TC>(acc, ca);
          }
        }
      }
      for (; r < run; ++r) {
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
          const float2 w = xr[o * ostride - r];
          const v2f v = {w.x, w.y};
#pragma unroll
          for (int j = 0; j < NCH; ++j) acc[j][o] = tap_mac(acc[j][o], gp[j][k + r], v);
        }
      }
      k += run;
      e -= run;
      blk -= 1;
    }
    // ---- de-rotate to baseband and store (coalesced over m) ----
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      const long long m_rel = tile_m0 + ml + 64 * o;
      if (m_rel < n_out) {
        const uint64_t n = (m_first + (uint64_t)m_rel) * (uint64_t)D;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
          const int c = c0 + j;
          if (c < ge.nchan) {
            float cs, sn;
            sd::phasor_u32(phase0[c] + (uint32_t)(n * (uint64_t)dphase[c]), cs, sn);
            const c32 r2 = sd::cmul_cs(c32{acc[j][o].x, acc[j][o].y}, cs, sn);
            y[(long long)c * yv.cs + m_rel * yv.ms] = float2{r2.re, r2.im};
          }
        }
      }
    }
  }
}

// Decimations whose 64-output window does not fit LDS (D > ~300 at 255 taps: a narrow channel in a wide capture).
// There the outputs are sparse -- each needs ntaps samples out of every D -- so nothing is staged: a lane owns one
// (output, channel), walks its taps in the SPEC's order straight from memory (8 consecutive taps share a cache
// line) and de-rotates.  Same operations per output as chan_fir_body, so the same bits.
__device__ __forceinline__ void chan_fir_sparse_body(const float2 *__restrict__ x, const float2 *__restrict__ hist,
                                                     float2 *__restrict__ hist_next, long long len, uint64_t n0,
                                                     const float4 *__restrict__ g, const uint32_t *__restrict__ dphase,
                                                     const uint32_t *__restrict__ phase0, int D, int T, int nchan,
                                                     uint64_t m_first, long long n_out, float2 *__restrict__ y, sdk::View yv,
                                                     const unsigned tile, const unsigned ntiles)
{
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x;
  if (tile == ntiles - 1) {
    const int hl = T - 1;
    for (int i = tid; i < hl; i += FIR_THREADS) {
      const long long src = (long long)i + len;                       // index into [hist ; x]
      hist_next[i] = src < hl ? hist[src] : x[src - hl];
    }
  }
  const long long t = (long long)tile * FIR_THREADS + tid;
  if (t >= n_out * nchan) return;
  const int c = (int)(t / n_out);
  const long long m_rel = t - (long long)c * n_out;
  const uint64_t n = (m_first + (uint64_t)m_rel) * (uint64_t)D;
  const long long hist0 = (long long)n0 - (T - 1);
  const float4 *gc = g + (long long)c * T;
  v2f acc = {0.0f, 0.0f};
  if ((long long)n - (T - 1) >= (long long)n0 && (long long)n < (long long)n0 + len) {
    // the whole tap window lies inside this block (every output but the block's first few): a straight walk, eight
    // loads in flight, no bounds tests -- the same taps in the same order
    const float2 *xp = x + ((long long)n - (long long)n0);
    int k = 0;
    for (; k + 8 <= T; k += 8) {
      float2 w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = xp[-(k + u)];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = tap_mac(acc, gc[k + u], v2f{w[u].x, w[u].y});
    }
    for (; k < T; ++k) { const float2 w = xp[-k]; acc = tap_mac(acc, gc[k], v2f{w.x, w.y}); }
  } else {
    for (int k = 0; k < T; ++k) {
      const long long q = (long long)n - k;
      float2 w = float2{0.0f, 0.0f};
      if (q >= (long long)n0) { if (q < (long long)n0 + len) w = x[q - (long long)n0]; }
      else if (q >= hist0) w = hist[q - hist0];
      acc = tap_mac(acc, gc[k], v2f{w.x, w.y});
    }
  }
  float cs, sn;
  sd::phasor_u32(phase0[c] + (uint32_t)(n * (uint64_t)dphase[c]), cs, sn);
  const c32 r2 = sd::cmul_cs(c32{acc.x, acc.y}, cs, sn);
  y[(long long)c * yv.cs + m_rel * yv.ms] = float2{r2.re, r2.im};
}

__global__ __launch_bounds__(FIR_THREADS) void chan_fir_sparse_kernel(const float2 *__restrict__ x, const float2 *__restrict__ hist,
                                                              float2 *__restrict__ hist_next, long long len, uint64_t n0,
                                                              const float4 *__restrict__ g, const uint32_t *__restrict__ dphase,
                                                              const uint32_t *__restrict__ phase0, int D, int T, int nchan,
                                                              uint64_t m_first, long long n_out, float2 *__restrict__ y, sdk::View yv)
{
  chan_fir_sparse_body(x, hist, hist_next, len, n0, g, dphase, phase0, D, T, nchan, m_first, n_out, y, yv, blockIdx.x, gridDim.x);
}

template <int NCH, int NOUT, int TC, bool LDS_TAPS, bool DB>
__global__ __launch_bounds__(FIR_THREADS) void chan_fir_kernel(const float2 *__restrict__ x, const float2 *__restrict__ hist,
                                                       float2 *__restrict__ hist_next,
                                                       long long len, uint64_t n0,
                                                       const float4 *__restrict__ g,
                                                       const uint32_t *__restrict__ dphase,
                                                       const uint32_t *__restrict__ phase0, FirGeom ge,
                                                       uint64_t m_first, long long n_out,
                                                       float2 *__restrict__ y, sdk::View yv)
{
  chan_fir_body<NCH, NOUT, TC, LDS_TAPS, DB>(x, hist, hist_next, len, n0, g, dphase, phase0, ge, m_first, n_out, y, yv,
                                             blockIdx.x, gridDim.x);
}

// the feeds of many 1-channel banks side by side (the live analyzer's inspectors, all reading the same wideband
// block): grid.y = bank, grid.x covers the longest feed; everything a workgroup needs is uniform -> scalar loads
__global__ __launch_bounds__(FIR_THREADS) void chan_fir_gang_kernel(const sdk::ChanGangItem *__restrict__ items)
{
  const sdk::ChanGangItem &it = items[blockIdx.y];
  const sdk::ChanFeedArgs &a = it.a;
  if (it.ntiles == 0 && blockIdx.x == 0) {                            // a feed too short for an output: history only
    const int hl = a.ntaps - 1;
    const float2 *hist = reinterpret_cast<const float2 *>(a.hist), *x = reinterpret_cast<const float2 *>(a.x);
    float2 *hist_next = reinterpret_cast<float2 *>(a.hist_next);
    for (int i = threadIdx.x; i < hl; i += FIR_THREADS) {
      const long long src = (long long)i + a.len;
      hist_next[i] = src < hl ? hist[src] : x[src - hl];
    }
  }
  if (blockIdx.x >= it.ntiles) return;
  if (it.sparse) {
    chan_fir_sparse_body(reinterpret_cast<const float2 *>(a.x), reinterpret_cast<const float2 *>(a.hist),
                         reinterpret_cast<float2 *>(a.hist_next), a.len, a.n0, reinterpret_cast<const float4 *>(a.g),
                         a.dphase, a.phase0, (int)a.D, a.ntaps, 1, a.m_first, a.n_out, reinterpret_cast<float2 *>(a.y), a.yv,
                         blockIdx.x, it.ntiles);
    return;
  }
  chan_fir_body<1, 1, 8, true, true>(reinterpret_cast<const float2 *>(a.x), reinterpret_cast<const float2 *>(a.hist),
                                     reinterpret_cast<float2 *>(a.hist_next), a.len, a.n0, reinterpret_cast<const float4 *>(a.g),
                                     a.dphase, a.phase0, it.ge, a.m_first, a.n_out, reinterpret_cast<float2 *>(a.y), a.yv,
                                     blockIdx.x, it.ntiles);
}

// new history = last (ntaps-1) samples of [old hist ; x]   (only used when a feed produces no output;
// otherwise the last FIR workgroup writes it)
__global__ void update_hist_kernel(float2 *__restrict__ hist_next, const float2 *__restrict__ hist,
                                   const float2 *__restrict__ x, long long len, int hl)
{
  for (int i = threadIdx.x; i < hl; i += blockDim.x) {
    const long long src = (long long)i + len;          // index into [hist ; x]
    hist_next[i] = src < hl ? hist[src] : x[src - hl];
  }
}

inline unsigned grid_for(long long n, int block) {
  long long g = (n + block - 1) / block;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (unsigned)g;
}

template <int NCH, int NOUT, int TC, bool LDS_TAPS, bool DB>
hipError_t launch_fir(const sdk::ChanFeedArgs &a, const FirGeom &ge, size_t lds, unsigned ntiles, hipStream_t st)
{
  auto kern = chan_fir_kernel<NCH, NOUT, TC, LDS_TAPS, DB>;
  static size_t attr_lds_dev[64] = {};                       // a function attribute belongs to a device (sharded analyzer: one per GPU)
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  size_t &attr_lds = attr_lds_dev[dev_ & 63];
  if (lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_lds = lds;
  }
  sdk::launch_timed("chan_fir_kernel", kern, dim3(ntiles), dim3(FIR_THREADS), lds, st,
                     reinterpret_cast<const float2 *>(a.x), reinterpret_cast<const float2 *>(a.hist),
                     reinterpret_cast<float2 *>(a.hist_next), a.len, a.n0,
                     reinterpret_cast<const float4 *>(a.g), a.dphase, a.phase0, ge, a.m_first, a.n_out,
                     reinterpret_cast<float2 *>(a.y), a.yv);
  return hipGetLastError();
}

}  // namespace

namespace sdk {

hipError_t xlate_bulk(const void *x, void *y, long long len, uint32_t p0, uint32_t dp, uint64_t n0, hipStream_t st)
{
  if (len <= 0) return hipSuccess;
  const long long npairs = (len + 1) / 2;
  hipLaunchKernelGGL(xlate_kernel, dim3(grid_for(npairs, 256)), dim3(256), 0, st,
                     reinterpret_cast<const float4 *>(x), reinterpret_cast<float4 *>(y), npairs, len, p0, dp, n0);
  return hipGetLastError();
}

hipError_t chan_modulate_taps(const float *h, int ntaps, const uint32_t *dphase, int nchan, void *g, void *g2, hipStream_t st)
{
  const int total = ntaps * nchan;
  hipLaunchKernelGGL(modulate_taps_kernel, dim3((total + 255) / 256), dim3(256), 0, st, h, ntaps, dphase, nchan,
                     reinterpret_cast<float4 *>(g), reinterpret_cast<float2 *>(g2));
  return hipGetLastError();
}

namespace {
struct FirPlan { FirGeom ge; size_t lds; unsigned ntiles; int nch, nout; bool lds_taps, sparse; };
}

static hipError_t fir_plan(const ChanFeedArgs &a, FirPlan &pl)
{
  FirGeom &ge = pl.ge;
  ge.D = (int)a.D; ge.ntaps = a.ntaps; ge.nchan = a.nchan;
  ge.KD = ((a.ntaps - 1 + ge.D - 1) / ge.D) * ge.D;
  ge.PAD = (ge.D & 1) ? 0 : 1;                                      // D + PAD odd
  const int force_smem = (int)sdk::tuning().fir_smem_taps, force_nout = (int)sdk::tuning().fir_nout;   // tuning knobs
  const bool lds_taps = !force_smem && (size_t)a.nchan * a.ntaps * sizeof(float4) <= 16 * 1024;
  // (measured alternative at C = D = 64: 8 channels per lane 124 us -- 61 SGPRs spill)
  const int nch = a.nchan >= 4 ? 4 : (a.nchan >= 2 ? 2 : 1);
  const int ngrp = (a.nchan + nch - 1) / nch;
  // outputs per lane: taps fetched once are reused for NOUT outputs (halves the scalar-cache tap
  // traffic per fma: 40 -> 48 % of FP32 peak at C = D = 64) when there are plenty of channel groups
  // to keep the waves busy and the window of 128 outputs stays small
  int nout = force_nout ? force_nout : ((ngrp >= 8 && (size_t)(128 * ge.D + ge.KD) * sizeof(float2) <= 72 * 1024) ? 2 : 1);
  if (nout != 1 && nout != 2) nout = 1;
  const int gran = 64 * nout;
  // tile: as many sub-tiles as fit a ~48 KiB window (keeps >= 3 workgroups per CU), but at least
  // enough (sub-tile, channel-group) units to occupy the workgroup's 8 waves
  const int budget = 6144;                                         // samples
  int mt = ((budget - ge.KD) / ge.D) / gran * gran;
  const int want = gran * ((FIR_THREADS / 64 + ngrp - 1) / ngrp);
  if (mt < want && ((size_t)want * ge.D + ge.KD) * sizeof(float2) <= 72 * 1024) mt = want;
  if (mt < gran) mt = gran;
  if (mt > 1024) mt = 1024;
  while (mt > gran && (long long)(mt - gran) >= a.n_out) mt -= gran;     // don't over-tile tiny problems
  ge.MT = mt;
  ge.span = ge.MT * ge.D + ge.KD;
  ge.lds_samples = ge.span + (ge.span / ge.D + 1) * ge.PAD;
  size_t lds = (((size_t)ge.lds_samples + 1) & ~(size_t)1) * sizeof(float2);
  if (lds_taps) lds += (size_t)a.nchan * a.ntaps * sizeof(float4);
  pl.sparse = false;
  if (lds > 160 * 1024) {                                          // the window of 64 outputs does not fit: sparse outputs
    pl.sparse = true; pl.lds = 0; pl.nch = nch; pl.nout = 1; pl.lds_taps = false;
    pl.ntiles = (unsigned)((a.n_out * a.nchan + FIR_THREADS - 1) / FIR_THREADS);
    return hipSuccess;
  }
  pl.lds = lds; pl.ntiles = (unsigned)((a.n_out + ge.MT - 1) / ge.MT);
  pl.nch = nch; pl.nout = nout; pl.lds_taps = lds_taps;
  return hipSuccess;
}

hipError_t chan_feed(const ChanFeedArgs &a, hipStream_t st)
{
  if (a.n_out <= 0) return chan_update_hist(a.hist_next, a.hist, a.x, a.len, a.ntaps, st);
  {
    // one or two channels: the stage is HBM-bound and runs as a stream (chan_stream.hip), same bits
    hipError_t es = hipSuccess;
    if (chan_stream_feed(a, a.g2, st, &es)) return es;
  }
  FirPlan pl;
  hipError_t e = fir_plan(a, pl);
  if (e != hipSuccess) return e;
  const FirGeom &ge = pl.ge;
  const size_t lds = pl.lds;
  const unsigned ntiles = pl.ntiles;
  const int nch = pl.nch, nout = pl.nout;
  const bool lds_taps = pl.lds_taps;
  if (pl.sparse) {
    hipLaunchKernelGGL(chan_fir_sparse_kernel, dim3(ntiles), dim3(FIR_THREADS), 0, st,
                       reinterpret_cast<const float2 *>(a.x), reinterpret_cast<const float2 *>(a.hist),
                       reinterpret_cast<float2 *>(a.hist_next), a.len, a.n0, reinterpret_cast<const float4 *>(a.g), a.dphase,
                       a.phase0, (int)a.D, a.ntaps, a.nchan, a.m_first, a.n_out, reinterpret_cast<float2 *>(a.y), a.yv);
    return hipGetLastError();
  }
#define SD_FIR(NCH_, NOUT_, TC_, LT_, DB_) return launch_fir<NCH_, NOUT_, TC_, LT_, DB_>(a, ge, lds, ntiles, st)
  if (lds_taps) {                                                   // taps from LDS: chunks double-buffered
    if (nout == 2) { if (nch == 4) SD_FIR(4, 2, 2, true, true); if (nch == 2) SD_FIR(2, 2, 4, true, true); SD_FIR(1, 2, 8, true, true); }
    if (nch == 4) SD_FIR(4, 1, 2, true, true); if (nch == 2) SD_FIR(2, 1, 4, true, true); SD_FIR(1, 1, 8, true, true);
  }
  // taps from scalar loads: 4 channels x 4 taps = 64 SGPRs per chunk, so no second chunk in flight
  // 4 channels x 2 taps = 32 SGPRs per chunk, two chunks in flight (ping-pong): 114.7 -> 109.1 us at C = D = 64,
  // 51.7 -> 42.2 us at C = 16 against single chunks of 4 taps
  const int tc4 = (int)sdk::tuning().fir_tc4;
  if (!tc4 && nch == 4) { if (nout == 2) SD_FIR(4, 2, 2, false, true); SD_FIR(4, 1, 2, false, true); }
  if (nout == 2) { if (nch == 4) SD_FIR(4, 2, 4, false, false); if (nch == 2) SD_FIR(2, 2, 4, false, true); SD_FIR(1, 2, 8, false, true); }
  if (nch == 4) SD_FIR(4, 1, 4, false, false); if (nch == 2) SD_FIR(2, 1, 4, false, true); SD_FIR(1, 1, 8, false, true);
#undef SD_FIR
}

hipError_t chan_gang_plan(const ChanFeedArgs &a, ChanGangItem *item)
{
  if (a.nchan != 1) return hipErrorInvalidValue;
  item->a = a;
  item->ntiles = 0;
  item->lds = 0;
  item->sparse = 0;
  item->ge = FirGeom{};
  item->ge.ntaps = a.ntaps;
  if (a.n_out <= 0) return hipSuccess;                               // only the history moves on
  FirPlan pl;
  hipError_t e = fir_plan(a, pl);
  if (e != hipSuccess) return e;
  if (!pl.sparse && (pl.nch != 1 || pl.nout != 1 || !pl.lds_taps)) {          // the gang kernel has the one tiled variant:
    pl.sparse = true; pl.lds = 0;                                             // anything else (> 1024 taps) walks its taps
    pl.ntiles = (unsigned)((a.n_out + FIR_THREADS - 1) / FIR_THREADS);
  }
  item->ge = pl.ge; item->ntiles = pl.ntiles; item->lds = (unsigned)pl.lds; item->sparse = pl.sparse ? 1 : 0;
  return hipSuccess;
}

hipError_t chan_gang_feed(const ChanGangItem *d_items, int n, unsigned max_tiles, unsigned max_lds, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  static size_t attr_lds_dev[64] = {};                       // a function attribute belongs to a device
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  size_t &attr_lds = attr_lds_dev[dev_ & 63];
  if (max_lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(chan_fir_gang_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds);
    if (e != hipSuccess) return e;
    attr_lds = max_lds;
  }
  hipLaunchKernelGGL(chan_fir_gang_kernel, dim3(max_tiles ? max_tiles : 1, (unsigned)n), dim3(FIR_THREADS), max_lds, st, d_items);
  return hipGetLastError();
}

hipError_t chan_update_hist(void *hist_next, const void *hist, const void *x, long long len, int ntaps, hipStream_t st)
{
  const int hl = ntaps - 1;
  if (hl <= 0) return hipSuccess;
  hipLaunchKernelGGL(update_hist_kernel, dim3(1), dim3(256), 0, st, reinterpret_cast<float2 *>(hist_next),
                     reinterpret_cast<const float2 *>(hist), reinterpret_cast<const float2 *>(x), len, hl);
  return hipGetLastError();
}

}  // namespace sdk
