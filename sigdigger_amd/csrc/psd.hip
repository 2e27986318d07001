// psd.hip -- main-spectrum PSD for gfx950: windowed N-point FFT, |X|^2, frame averaging,
// optional fused fftshift + dB (PSDMessage ctor) -- rows A2/A3/A4/A9 of SURVEY.md section 8(a).
//
// One workgroup transforms one output frame (navg consecutive input frames, accumulated in
// registers).  The N-point FFT is a Stockham autosort FFT whose passes (radix 8/16) run on
// registers; the frame lives in ONE LDS buffer (in place: all of a pass's operands are
// pulled into VGPRs, barrier, results are written back to the autosorted positions).  The
// first pass reads HBM directly (coalesced float2), the last writes power straight to HBM,
// so the HBM traffic is the compulsory 8 B/sample in and 4 B/bin/output-frame out.
//
// LDS indices are padded by one element every 16 (idx + idx>>4) which keeps the stride-R
// stores of the first passes at <=2-way bank conflicts.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "kernels.hpp"

namespace {

// complex = one aligned VGPR pair; arithmetic written so that it maps 1:1 onto v_pk_add_f32 /
// v_pk_mul_f32 / v_pk_fma_f32 with op_sel / neg modifiers (no register shuffling)
typedef float cf __attribute__((ext_vector_type(2)));

__device__ __forceinline__ cf cadd(cf a, cf b) { return a + b; }
__device__ __forceinline__ cf csub(cf a, cf b) { return a - b; }
// The rotations by -j / +j and the complex product are single VOP3P instructions once the operand
// halves are picked with op_sel / op_sel_hi and negated with neg_lo / neg_hi; the compiler builds
// the swapped / negated pair with v_mov + v_xor instead (a quarter of the loop's VALU work).
__device__ __forceinline__ cf add_mj(cf t, cf d)          // t + (-j) d = (t.x + d.y, t.y - d.x)
{
  cf r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(t), "v"(d));
  return r;
}
__device__ __forceinline__ cf sub_mj(cf t, cf d)          // t - (-j) d = (t.x - d.y, t.y + d.x)
{
  cf r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(t), "v"(d));
  return r;
}
__device__ __forceinline__ cf cmul(cf a, cf b)
{
  // (a.x b.x - a.y b.y, a.x b.y + a.y b.x)
  cf t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
  return r;
}

// forward DFTs on registers, natural order in, natural order out (DIT, even/odd split)
__device__ __forceinline__ void dft2(cf &a, cf &b) { cf t = a; a = cadd(t, b); b = csub(t, b); }

__device__ __forceinline__ void dft4(cf &a0, cf &a1, cf &a2, cf &a3)
{
  cf t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), d = csub(a1, a3);
  a0 = cadd(t0, t2); a1 = add_mj(t1, d); a2 = csub(t0, t2); a3 = sub_mj(t1, d);
}

__device__ __forceinline__ cf mul_w8_1(cf a) { return add_mj(a, a) * 0.70710678118654752440f; }            // * (1 - j)/sqrt2
__device__ __forceinline__ cf mul_w8_3(cf a) { return add_mj(-a, a) * 0.70710678118654752440f; }           // * (-1 - j)/sqrt2

__device__ __forceinline__ void dft8(cf *v)
{
  cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
  cf o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
  dft4(e0, e1, e2, e3);
  dft4(o0, o1, o2, o3);
  o1 = mul_w8_1(o1);
  o3 = mul_w8_3(o3);
  v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
  v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
  v[2] = add_mj(e2, o2); v[6] = sub_mj(e2, o2);          // o2 * (-j) folded into the butterfly
  v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}

__device__ __forceinline__ void dft16(cf *v)
{
  cf e[8], o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
  dft8(e);
  dft8(o);
  const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;
  o[1] = cmul(o[1], cf{ c1, -s1});
  o[2] = mul_w8_1(o[2]);
  o[3] = cmul(o[3], cf{ s1, -c1});
  o[5] = cmul(o[5], cf{-s1, -c1});
  o[6] = mul_w8_3(o[6]);
  o[7] = cmul(o[7], cf{-c1, -s1});
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i == 4) { v[4] = add_mj(e[4], o[4]); v[12] = sub_mj(e[4], o[4]); }
    else { v[i] = cadd(e[i], o[i]); v[i + 8] = csub(e[i], o[i]); }
  }
}

template <int R> __device__ __forceinline__ void dftR(cf *v);
template <> __device__ __forceinline__ void dftR<2>(cf *v)  { dft2(v[0], v[1]); }
template <> __device__ __forceinline__ void dftR<4>(cf *v)  { dft4(v[0], v[1], v[2], v[3]); }
template <> __device__ __forceinline__ void dftR<8>(cf *v)  { dft8(v); }
template <> __device__ __forceinline__ void dftR<16>(cf *v) { dft16(v); }

__device__ __forceinline__ int lpad(int i) { return i + (i >> 4); }

// compile-time pass plan: ceil(bits/4) passes, the first (bits % P) passes one bit wider
template <int LOG2N> struct Plan {
  static constexpr int P     = (LOG2N + 3) / 4;
  static constexpr int BASE  = LOG2N / P;
  static constexpr int EXTRA = LOG2N % P;
  static constexpr int bits(int p) { return BASE + (p < EXTRA ? 1 : 0); }
  static constexpr int ns_log2(int p) { int s = 0; for (int i = 0; i < p; ++i) s += bits(i); return s; }
};

// twiddle multiply for one butterfly: v[q] *= W_N^(q*tw) for q = 1..R-1 (tw already scaled
// to the N-point table); the powers 1,2,4,8 are table look-ups, the rest one product each.
template <int R>
__device__ __forceinline__ void apply_twiddles(cf *v, const cf *__restrict__ tw, int idx, int nmask)
{
  cf w[R];
  w[1] = tw[idx & nmask];
  if (R > 2) w[2] = tw[(2 * idx) & nmask];
  if (R > 4) w[4] = tw[(4 * idx) & nmask];
  if (R > 8) w[8] = tw[(8 * idx) & nmask];
  if (R > 2) w[3] = cmul(w[1], w[2]);
  if (R > 4) { w[5] = cmul(w[1], w[4]); w[6] = cmul(w[2], w[4]); w[7] = cmul(w[3], w[4]); }
  if (R > 8) {
#pragma unroll
    for (int q = 9; q < 16; ++q) w[q] = cmul(w[q - 8], w[8]);
  }
#pragma unroll
  for (int q = 1; q < R; ++q) v[q] = cmul(v[q], w[q]);
}

// Base twiddles of one thread: W^(k), W^(2k), W^(4k), W^(8k) for each of its butterflies in each
// pass.  They depend on the lane id only, so they are loaded ONCE per workgroup (before the frame
// loop) and stay in registers; the other powers are re-derived per frame (one complex product each).
// (W^(4k), W^(8k) are squared from W^(2k) per frame: two instructions each, and 12 VGPRs fewer
// than keeping them -- the kernel sits exactly at the 128-VGPR budget of two workgroups per CU)
constexpr int MAXP = 4, MAXNB = 2;
struct TwBase { cf w[MAXP][MAXNB][2]; };

template <int LOG2N, int THREADS, int PASS>
__device__ __forceinline__ void load_tw_base(TwBase &tb, const cf *__restrict__ tw, int tid)
{
  using PL = Plan<LOG2N>;
  constexpr int N = 1 << LOG2N, E = N / THREADS;
  if constexpr (PASS < PL::P) {
    constexpr int RB = PL::bits(PASS), R = 1 << RB, NB = E / R, NSL = PL::ns_log2(PASS), NS = 1 << NSL;
    static_assert(PL::P <= MAXP && NB <= MAXNB, "TwBase too small");
    if constexpr (PASS > 0) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int j = tid + b * THREADS;
        const int idx = (j & (NS - 1)) << (LOG2N - NSL - RB);
        tb.w[PASS][b][0] = tw[idx & (N - 1)];
        if (R > 2) tb.w[PASS][b][1] = tw[(2 * idx) & (N - 1)];
      }
    }
    load_tw_base<LOG2N, THREADS, PASS + 1>(tb, tw, tid);
  }
}

__device__ __forceinline__ cf opaque(cf a)
{
  // keeps LICM from hoisting the derived twiddle powers of every pass out of the frame loop
  asm volatile("" : "+v"(a));
  return a;
}

template <int R>
__device__ __forceinline__ void apply_twiddles_base(cf *v, const cf *base)
{
  cf w[R];
  w[1] = opaque(base[0]);
  if (R > 2) w[2] = opaque(base[1]);
  if (R > 4) w[4] = cmul(w[2], w[2]);
  if (R > 8) w[8] = cmul(w[4], w[4]);
  if (R > 2) w[3] = cmul(w[1], w[2]);
  if (R > 4) { w[5] = cmul(w[1], w[4]); w[6] = cmul(w[2], w[4]); w[7] = cmul(w[3], w[4]); }
  if (R > 8) {
#pragma unroll
    for (int q = 9; q < 16; ++q) w[q] = cmul(w[q - 8], w[8]);
  }
#pragma unroll
  for (int q = 1; q < R; ++q) v[q] = cmul(v[q], w[q]);
}

template <int LOG2N, int THREADS, int PASS>
__device__ __forceinline__ void fft_pass(cf *v /*[E]*/, cf *lds, const TwBase &tb, int tid,
                                         float *pw /*[E]*/)
{
  using PL = Plan<LOG2N>;
  constexpr int N  = 1 << LOG2N;
  constexpr int E  = N / THREADS;
  constexpr int RB = PL::bits(PASS);
  constexpr int R  = 1 << RB;
  constexpr int NB = E / R;                       // butterflies per thread
  constexpr int NSL = PL::ns_log2(PASS);
  constexpr int NS = 1 << NSL;
  constexpr bool LAST = (PASS == PL::P - 1);
  static_assert(NB >= 1, "radix larger than per-thread element count");

  if (PASS > 0) {
    // gather this pass's operands: element q of butterfly j sits at j + q*N/R
    // (N/R is a multiple of 16, so lpad(j + q*N/R) = lpad(j) + q*lpad(N/R): one base address and
    // immediate offsets)
    static_assert((N / R) % 16 == 0, "gather offsets are compile-time constants only for N/R % 16 == 0");
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const cf *gp = lds + lpad(tid + b * THREADS);
#pragma unroll
      for (int q = 0; q < R; ++q) v[b * R + q] = gp[q * ((N / R) + (N / R) / 16)];
    }
    __syncthreads();                              // everyone has read: LDS may be overwritten
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int j = tid + b * THREADS;
    const int k = j & (NS - 1);
    cf *vb = v + b * R;
    if (PASS > 0) apply_twiddles_base<R>(vb, tb.w[PASS][b]);
    dftR<R>(vb);
    const int j0 = ((j - k) << RB) + k;
    if (!LAST) {
      if constexpr (NS % 16 == 0) {               // lpad(j0 + q*NS) = lpad(j0) + q*lpad(NS)
        cf *sp = lds + lpad(j0);
#pragma unroll
        for (int q = 0; q < R; ++q) sp[q * (NS + NS / 16)] = vb[q];
      } else if constexpr (PASS == 0 && R <= 16) {   // j0 = j*R, q < R <= 16: no carry into the pad term
        cf *sp = lds + lpad(j0);
#pragma unroll
        for (int q = 0; q < R; ++q) sp[q] = vb[q];
      } else {
#pragma unroll
        for (int q = 0; q < R; ++q) lds[lpad(j0 + q * NS)] = vb[q];
      }
    } else {
      // last pass: NS == N/R, j0 == j, output index j + q*N/R; accumulate power
#pragma unroll
      for (int q = 0; q < R; ++q) pw[b * R + q] += vb[q].x * vb[q].x + vb[q].y * vb[q].y;
    }
  }
  if (!LAST) __syncthreads();
}

template <int LOG2N, int THREADS, int PASS>
struct PassRunner {
  static __device__ __forceinline__ void run(cf *v, cf *lds, const TwBase &tb, int tid, float *pw)
  {
    fft_pass<LOG2N, THREADS, PASS>(v, lds, tb, tid, pw);
    if constexpr (PASS + 1 < Plan<LOG2N>::P) PassRunner<LOG2N, THREADS, PASS + 1>::run(v, lds, tb, tid, pw);
  }
};

// grid.x = number of output frames, grid.y = S (split of the navg frames of one output over S
// workgroups; S > 1 writes unscaled partial sums to `partial`, reduced by psd_reduce_kernel in a
// fixed order so the result is deterministic)
template <int LOG2N, int THREADS>
// second launch bound: two workgroups per CU must fit the register file (N = 16384 is LDS-limited to one)
__global__ __launch_bounds__(THREADS, (THREADS >= 1024 ? 4 : (THREADS >= 128 ? THREADS / 128 : 1))) void psd_kernel(const cf *__restrict__ x, long long hop, int navg,
                                                      const float *__restrict__ window,
                                                      const cf *__restrict__ tw, float scale, int mode,
                                                      float *__restrict__ out, float *__restrict__ partial)
{
  __builtin_amdgcn_s_setprio(3);   // ahead of the resident recurrence wavefronts (see chan_fir_kernel)
  using PL = Plan<LOG2N>;
  constexpr int N  = 1 << LOG2N;
  constexpr int E  = N / THREADS;
  constexpr int R0 = 1 << PL::bits(0);
  constexpr int NB0 = E / R0;
  constexpr int RL = 1 << PL::bits(PL::P - 1);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cf *lds = reinterpret_cast<cf *>(smem);
  const int tid0 = threadIdx.x;
  const long long o = blockIdx.x;

  float pw[E];
#pragma unroll
  for (int i = 0; i < E; ++i) pw[i] = 0.0f;
  TwBase tb;
  load_tw_base<LOG2N, THREADS, 0>(tb, tw, tid0);

  const int S = gridDim.y;
  const int fps = (navg + S - 1) / S;
  const int f_begin = blockIdx.y * fps;
  const int f_end = (f_begin + fps < navg) ? f_begin + fps : navg;
  // Software pipeline over the frames of this workgroup: the raw samples of frame f+1 are requested
  // right after pass 0 of frame f has moved its operands to LDS, so the HBM latency hides behind
  // passes 1..P-1 (with two workgroups per CU there is not enough other work to hide it otherwise:
  // one-frame workgroups reached 51 % of the HBM peak where the in-loop version stayed at 38 %).
  // (measured: pays for N = 4096 / 8192; N = 2048 loses a wave of occupancy to the 32 extra VGPRs,
  // N = 16384 spills, smaller frames have enough workgroups per CU anyway)
  constexpr bool PREFETCH = (LOG2N == 12 || LOG2N == 13);
  cf nxt[E];
  auto request = [&](int f) {
    const cf *fr = x + (o * navg + f) * hop;
#pragma unroll
    for (int b = 0; b < NB0; ++b) {
      const int j = tid0 + b * THREADS;
#pragma unroll
      for (int q = 0; q < R0; ++q) nxt[b * R0 + q] = fr[j + q * (N / R0)];
    }
  };
  if (PREFETCH && f_begin < f_end) request(f_begin);
  for (int f = f_begin; f < f_end; ++f) {
    if (!PREFETCH) request(f);
    // Make the lane id opaque per frame: otherwise LICM hoists every twiddle (and its derived
    // powers) and every LDS address of all passes out of the frame loop and the kernel needs
    // >256 VGPRs.  Re-deriving them per frame costs a few integer ops and L1-resident loads.
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    cf v[E];
    // pass 0 operands: the prefetched samples, window applied on the fly
#pragma unroll
    for (int b = 0; b < NB0; ++b) {
      const int j = tid + b * THREADS;
#pragma unroll
      for (int q = 0; q < R0; ++q) {
        const int i = j + q * (N / R0);
        v[b * R0 + q] = nxt[b * R0 + q] * window[i];
      }
    }
    // (the barrier after the previous frame's last gather already ordered LDS reuse)
    fft_pass<LOG2N, THREADS, 0>(v, lds, tb, tid, pw);
    if (PREFETCH && f + 1 < f_end) request(f + 1);
    PassRunner<LOG2N, THREADS, 1>::run(v, lds, tb, tid, pw);
  }

  // epilogue: thread holds power of bins j + q*N/RL (last-pass geometry)
  const float sc = (S > 1) ? 1.0f : scale / (float)navg;
  if (S > 1) mode = 0;
  float *dst = (S > 1) ? partial + (o * S + blockIdx.y) * N : out + o * N;
  constexpr int NBL = E / RL;
#pragma unroll
  for (int b = 0; b < NBL; ++b) {
    const int j = tid0 + b * THREADS;
#pragma unroll
    for (int q = 0; q < RL; ++q) {
      const int i = j + q * (N / RL);
      float p = pw[b * RL + q] * sc;
      if (mode == 0) {
        dst[i] = p;
      } else {
        // Suscan/Messages/PSDMessage.cpp:29-38: out[(i + N/2) mod N] = 10 log10(p + 1e-8)
        dst[(i + N / 2) & (N - 1)] = 10.0f * log10f(p + 1e-8f);
      }
    }
  }
}

// out[o][.] = (scale/navg) * sum_s partial[o][s][.], optional shift + dB.  A block handles 64 bins;
// its 4 waves each sum a quarter of the S partials (ascending s), the quarters are combined as
// (q0 + q1) + (q2 + q3): a fixed order, so the result is deterministic.
__global__ __launch_bounds__(256) void psd_reduce_kernel(const float *__restrict__ partial, int S, int n, float sc,
                                                         int mode, float *__restrict__ out)
{
  __shared__ float part[4][64];
  const long long o = blockIdx.y;
  const int ix = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + ix;
  const int s0 = (int)((long long)S * g / 4), s1 = (int)((long long)S * (g + 1) / 4);
  float acc = 0.0f;
  if (i < n) {
    const float *p = partial + (o * S) * n + i;
    int s2 = s0;
    for (; s2 + 8 <= s1; s2 += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(long long)(s2 + u) * n];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; s2 < s1; ++s2) acc += p[(long long)s2 * n];
  }
  part[g][ix] = acc;
  __syncthreads();
  if (g == 0 && i < n) {
    const float pw = ((part[0][ix] + part[1][ix]) + (part[2][ix] + part[3][ix])) * sc;
    if (mode == 0) out[o * n + i] = pw;
    else out[o * n + ((i + n / 2) & (n - 1))] = 10.0f * log10f(pw + 1e-8f);
  }
}

template <int LOG2N, int THREADS>
hipError_t launch_psd(const void *x, long long hop, int navg, const float *window, const void *tw,
                      float scale, int mode, float *out, long long nout, float *partial, int S, hipStream_t st)
{
  constexpr int N = 1 << LOG2N;
  const size_t lds = sizeof(cf) * (size_t)(N + (N >> 4) + 1);
  auto kern = psd_kernel<LOG2N, THREADS>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nout, (unsigned)S), dim3(THREADS), lds, st,
                     reinterpret_cast<const cf *>(x), hop, navg, window, reinterpret_cast<const cf *>(tw),
                     scale, mode, out, partial);
  if (S > 1) {
    hipLaunchKernelGGL(psd_reduce_kernel, dim3((N + 63) / 64, (unsigned)nout), dim3(256), 0, st,
                       partial, S, N, scale / (float)navg, mode, out);
  }
  return hipGetLastError();
}

// ---- element-wise post-processing -------------------------------------------------------

// PSDMessage ctor loop, in place on nframes frames of n floats
__global__ void psd_shift_db_kernel(float *psd, long long n, long long total_half)
{
  const long long half = n >> 1;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total_half;
       t += (long long)gridDim.x * blockDim.x) {
    const long long fr = t / half, i = t - fr * half;
    float *p = psd + fr * n;
    const float lo = p[i], hi = p[i + half];
    p[i + half] = 10.0f * log10f(lo + 1e-8f);
    p[i]        = 10.0f * log10f(hi + 1e-8f);
  }
}

// Averager::feed
__global__ void averager_kernel(float *last, const float *x, long long n, float alpha, int blend)
{
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float v = x[i];
    if (blend) {
      const float l = last[i];
      last[i] = l + alpha * (v - l);
    } else {
      last[i] = v;
    }
  }
}

// GenericInspector SPECTRUM case: dB (raw, +1e-20) of every element, then the swap loop
// data[i] <-> data[len/2 + i], i < len/2 (its wrapping p never wraps: p <= len-1).  For odd
// len the last element is converted to dB and stays in place.
__global__ void insp_spectrum_kernel(float *data, long long len, long long nspec)
{
  const long long half = len >> 1;
  const long long total = nspec * half;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
    const long long fr = t / half, i = t - fr * half;
    float *p = data + fr * len;
    const float lo = p[i], hi = p[i + half];
    p[i]        = 10.0f * log10f(hi + 1e-20f);
    p[i + half] = 10.0f * log10f(lo + 1e-20f);
    if ((len & 1) && i == 0) p[len - 1] = 10.0f * log10f(p[len - 1] + 1e-20f);
  }
}

inline unsigned grid_for(long long n, int block) {
  long long g = (n + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

namespace sdk {

// how many workgroups share the navg frames of one output: enough to put >= ~1024 workgroups
// on the chip, at least 2 frames each
int psd_split(long long nout, int navg)
{
  // two resident workgroups per CU (512 on the chip), at least `minf` frames per workgroup: the
  // register-resident twiddles are reused and the partial sums stay a fraction of the input (PMC
  // showed 2.1x the algorithmic HBM traffic with one frame per workgroup).  Only matters for the
  // small analyzer blocks (4 Mi samples: 8192-pt 25.6 us at minf = 2, 30.6 us at 4; 16384-pt 28.8
  // vs 44.8 us); a capture-sized input has nout >= target and never splits.
  static int target = 0, minf = 0;
  if (target == 0) {
    const char *e = getenv("SUAMD_PSD_SPLIT_TARGET");
    target = e ? atoi(e) : 512;
    e = getenv("SUAMD_PSD_MIN_FRAMES");
    minf = e ? atoi(e) : 2;
    if (target < 1) target = 1;
    if (minf < 1) minf = 1;
  }
  if (nout <= 0 || navg < 2 * minf || nout >= target) return 1;
  long long s = (target + nout - 1) / nout;
  if (s > navg / minf) s = navg / minf;
  if (s < 1) s = 1;
  return (int)s;
}

hipError_t psd_frames(int log2n, const void *x, long long hop, int navg, const float *window,
                      const void *tw, float scale, int mode, float *out, long long nout, float *partial,
                      hipStream_t st)
{
  if (nout <= 0) return hipSuccess;
  const int S = partial ? psd_split(nout, navg) : 1;
  switch (log2n) {
    // 16 points per thread (8 for N = 512): one radix-16 or two radix-8 butterflies per pass
    case 9:  return launch_psd<9, 64>(x, hop, navg, window, tw, scale, mode, out, nout, partial, S, st);
    case 10: return launch_psd<10, 64>(x, hop, navg, window, tw, scale, mode, out, nout, partial, S, st);
    case 11: return launch_psd<11, 128>(x, hop, navg, window, tw, scale, mode, out, nout, partial, S, st);
    case 12: return launch_psd<12, 256>(x, hop, navg, window, tw, scale, mode, out, nout, partial, S, st);
    case 13: return launch_psd<13, 512>(x, hop, navg, window, tw, scale, mode, out, nout, partial, S, st);
    case 14: return launch_psd<14, 1024>(x, hop, navg, window, tw, scale, mode, out, nout, partial, S, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t psd_shift_db(float *psd, long long n, long long nframes, hipStream_t st)
{
  const long long total = nframes * (n >> 1);
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(psd_shift_db_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, psd, n, total);
  return hipGetLastError();
}

hipError_t averager_feed(float *last, const float *x, long long n, float alpha, int blend, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(averager_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, last, x, n, alpha, blend);
  return hipGetLastError();
}

hipError_t insp_spectrum_db_shift(float *data, long long len, long long nspec, hipStream_t st)
{
  if (len <= 0 || nspec <= 0) return hipSuccess;
  hipLaunchKernelGGL(insp_spectrum_kernel, dim3(grid_for(nspec * (len >> 1), 256)), dim3(256), 0, st,
                     data, len, nspec);
  return hipGetLastError();
}

}  // namespace sdk
