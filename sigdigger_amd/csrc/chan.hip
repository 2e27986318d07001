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
#include "kernels.hpp"
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
                                     int nchan, float4 *__restrict__ g)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntaps * nchan) return;
  const int c = t / ntaps, k = t - c * ntaps;
  float cs, sn;
  sd::phasor_u32(0u - (uint32_t)k * dphase[c], cs, sn);
  const float re = h[k] * cs, im = h[k] * sn;
  g[t] = float4{re, re, -im, im};
}

// ---------------------------------------------------------------------------------------
struct FirGeom {
  int D, ntaps, nchan;
  int MT;            // outputs per workgroup tile (multiple of 64 unless D is huge)
  int KD;            // ceil((ntaps-1)/D)*D : window starts KD samples before the tile's first output
  int COLS;          // columns of the transposed window  (= MT + KD/D)
  int LDW;           // LDS row pitch in samples (odd)
};

constexpr int FIR_THREADS = 512;   // 8 waves share one staged window: 4 workgroups x 8 = 32 waves per CU

template <int NCH>
__global__ __launch_bounds__(FIR_THREADS) void chan_fir_kernel(const float2 *__restrict__ x, const float2 *__restrict__ hist,
                                                       long long len, uint64_t n0,
                                                       const float4 *__restrict__ g,
                                                       const uint32_t *__restrict__ dphase,
                                                       const uint32_t *__restrict__ phase0, FirGeom ge,
                                                       uint64_t m_first, long long n_out,
                                                       float2 *__restrict__ y, sdk::View yv)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2 *win = reinterpret_cast<float2 *>(smem);
  const int tid = threadIdx.x;
  const int D = ge.D, T = ge.ntaps;
  const long long tile_m0 = (long long)blockIdx.x * ge.MT;           // relative to m_first
  // absolute index of window sample i = 0
  const long long nbase = (long long)(m_first + (uint64_t)tile_m0) * D - ge.KD;
  const int span = ge.COLS * D;                                       // samples staged
  const long long hist0 = (long long)n0 - (T - 1);                    // absolute index of hist[0]

  // ---- stage the window: coalesced HBM reads, transposed LDS writes ----
  for (int i = tid; i < span; i += FIR_THREADS) {
    const long long n = nbase + i;
    float2 v = float2{0.0f, 0.0f};
    if (n >= (long long)n0) {
      if (n < (long long)n0 + len) v = x[n - (long long)n0];
    } else if (n >= hist0) {
      v = hist[n - hist0];
    }
    const int row = i % D, col = i / D;
    win[row * ge.LDW + col] = v;
  }
  __syncthreads();

  // ---- units of work: (64-output sub-tile, group of NCH channels) round-robin over waves ----
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int nsub = ge.MT >> 6 ? ge.MT >> 6 : 1;
  const int ngrp = (ge.nchan + NCH - 1) / NCH;
  const int nunits = nsub * ngrp;
  const int kd_cols = ge.KD / D;

  for (int u = wave; u < nunits; u += FIR_THREADS / 64) {
    const int sub = u / ngrp;
    const int c0  = (u - sub * ngrp) * NCH;
    const int ml  = sub * 64 + lane;                                  // output within the tile
    v2f acc[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) acc[j] = v2f{0.0f, 0.0f};
    const float4 *gp[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = (c0 + j < ge.nchan) ? c0 + j : ge.nchan - 1;      // clamp: duplicates are not stored
      gp[j] = g + (long long)c * T;
    }
    // window index of tap k for output ml: i = ml*D + KD - k  -> row (KD-k) mod D, col ml + (KD-k)/D.
    // k = 0 sits at (row 0, column kd_cols); each following tap is one row up (address - LDW)
    // until the row wraps to D-1 of the previous column.  Taps are consumed in runs that end at
    // a wrap, 4 at a time inside a run so that the wave-uniform taps arrive as one
    // s_load_dwordx16 per channel and feed v_pk_fma_f32 directly (the CU's single scalar unit
    // is otherwise the bottleneck: 3 SALU per VALU were measured with the tap-at-a-time form).
    const int mlc = ml < ge.MT ? ml : ge.MT - 1;
    const int ldw = ge.LDW;
    int row = 0, colofs = kd_cols;
    for (int k = 0; k < T;) {
      const int run = (row + 1 < T - k) ? row + 1 : T - k;
      const float2 *wp = win + row * ldw + colofs + mlc;
      int r = 0;
      for (; r + 4 <= run; r += 4) {
        v2f v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float2 w = wp[-(r + q) * ldw]; v[q] = v2f{w.x, w.y}; }
        // all NCH x 4 taps first (one s_load_dwordx16 per channel, all in flight together),
        // then the fmas tap-major so consecutive instructions hit different accumulators
        float4 t[NCH][4];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
          const float4 *gk = gp[j] + (k + r);
#pragma unroll
          for (int q = 0; q < 4; ++q) t[j][q] = gk[q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int j = 0; j < NCH; ++j) acc[j] = __builtin_elementwise_fma(v2f{t[j][q].x, t[j][q].y}, v[q], acc[j]);
#pragma unroll
          for (int j = 0; j < NCH; ++j) acc[j] = __builtin_elementwise_fma(v2f{t[j][q].z, t[j][q].w}, v[q].yx, acc[j]);
        }
      }
      for (; r < run; ++r) {
        const float2 w = wp[-r * ldw];
        const v2f v = {w.x, w.y};
#pragma unroll
        for (int j = 0; j < NCH; ++j) acc[j] = tap_mac(acc[j], gp[j][k + r], v);
      }
      k += run;
      row = D - 1;
      colofs -= 1;
    }
    // ---- de-rotate to baseband and store (coalesced over m) ----
    const long long m_rel = tile_m0 + ml;
    if (ml < ge.MT && m_rel < n_out) {
      const uint64_t n = (m_first + (uint64_t)m_rel) * (uint64_t)D;
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int c = c0 + j;
        if (c < ge.nchan) {
          float cs, sn;
          sd::phasor_u32(phase0[c] + (uint32_t)(n * (uint64_t)dphase[c]), cs, sn);
          const c32 r = sd::cmul_cs(c32{acc[j].x, acc[j].y}, cs, sn);
          y[(long long)c * yv.cs + m_rel * yv.ms] = float2{r.re, r.im};
        }
      }
    }
  }
}

// new history = last (ntaps-1) samples of [old hist ; x]
__global__ void update_hist_kernel(float2 *hist, const float2 *__restrict__ x, long long len, int hl)
{
  // single workgroup; two phases so that in-place shifting is safe
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2 *tmp = reinterpret_cast<float2 *>(smem);
  for (int i = threadIdx.x; i < hl; i += blockDim.x) {
    const long long src = (long long)i + len;          // index into [hist ; x]
    tmp[i] = src < hl ? hist[src] : x[src - hl];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < hl; i += blockDim.x) hist[i] = tmp[i];
}

inline unsigned grid_for(long long n, int block) {
  long long g = (n + block - 1) / block;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (unsigned)g;
}

template <int NCH>
hipError_t launch_fir(const sdk::ChanFeedArgs &a, const FirGeom &ge, size_t lds, unsigned ntiles, hipStream_t st)
{
  auto kern = chan_fir_kernel<NCH>;
  static size_t attr_lds = 0;
  if (lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_lds = lds;
  }
  hipLaunchKernelGGL(kern, dim3(ntiles), dim3(FIR_THREADS), lds, st,
                     reinterpret_cast<const float2 *>(a.x), reinterpret_cast<const float2 *>(a.hist), a.len, a.n0,
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

hipError_t chan_modulate_taps(const float *h, int ntaps, const uint32_t *dphase, int nchan, void *g, hipStream_t st)
{
  const int total = ntaps * nchan;
  hipLaunchKernelGGL(modulate_taps_kernel, dim3((total + 255) / 256), dim3(256), 0, st, h, ntaps, dphase, nchan,
                     reinterpret_cast<float4 *>(g));
  return hipGetLastError();
}

hipError_t chan_feed(const ChanFeedArgs &a, hipStream_t st)
{
  if (a.n_out <= 0) return hipSuccess;
  FirGeom ge;
  ge.D = (int)a.D; ge.ntaps = a.ntaps; ge.nchan = a.nchan;
  ge.KD = ((a.ntaps - 1 + ge.D - 1) / ge.D) * ge.D;
  // tile: as many 64-output sub-tiles as fit a ~48 KiB window (keeps >=3 workgroups per CU)
  const int budget = 6144;                                         // samples
  int mt = ((budget - ge.KD) / ge.D) / 64 * 64;
  if (mt < 64) mt = 64;
  if (mt > 1024) mt = 1024;
  // don't over-tile tiny problems
  while (mt > 64 && (long long)(mt - 64) >= a.n_out) mt -= 64;
  ge.MT = mt;
  ge.COLS = ge.MT + ge.KD / ge.D;
  ge.LDW = ge.COLS | 1;                                            // odd pitch: conflict-free transposed writes
  const size_t lds = (size_t)ge.D * ge.LDW * sizeof(float2);
  if (lds > 160 * 1024) return hipErrorInvalidValue;               // decimation too large for one tile
  const unsigned ntiles = (unsigned)((a.n_out + ge.MT - 1) / ge.MT);
  if (a.nchan >= 4) return launch_fir<4>(a, ge, lds, ntiles, st);
  if (a.nchan >= 2) return launch_fir<2>(a, ge, lds, ntiles, st);
  return launch_fir<1>(a, ge, lds, ntiles, st);
}

hipError_t chan_update_hist(void *hist, const void *x, long long len, int ntaps, hipStream_t st)
{
  const int hl = ntaps - 1;
  if (hl <= 0 || len <= 0) return hipSuccess;
  hipLaunchKernelGGL(update_hist_kernel, dim3(1), dim3(256), (size_t)hl * sizeof(float2), st,
                     reinterpret_cast<float2 *>(hist), reinterpret_cast<const float2 *>(x), len, hl);
  return hipGetLastError();
}

}  // namespace sdk
