// specttuner_wave.hip -- FFT channeliser for narrow channels (inverse transform of 8..64 points), one WAVEFRONT per
// window (SPEC.md section C2; rows T2 / N2; same arithmetic contract as specttuner.hip, which keeps the wider channels).
//
// W = 4096 = 64 x 64: a wavefront's 64 lanes hold 64 samples each, so the forward transform is two 64-point DFTs on
// REGISTERS (8 x 8, twiddles are compile-time constants) around one transposition through LDS, and -- there being no
// second wavefront in the workgroup -- no barrier anywhere: LDS operations of one wavefront execute in order.
//   lane t, register r:  x[t + 64 r]                      (loads: 512 contiguous bytes per register)
//   DFT64 over r         A_t[k2]
//   twiddle              B_t[k2] = A_t[k2] W_4096^(t k2)  (six per-lane base powers, the rest one product each)
//   LDS transposition    lane l gets B_t[l], t = 0..63    (pitch 65: both sides conflict-free)
//   DFT64 over t         X[l + 64 k1] in register k1
//   spectrum -> LDS      natural order, first 32 bins repeated after the end (a channel half never wraps)
// Channel stage: lane = channel (64 / size groups of 64 channels per wavefront).  A lane reads its `size` bins from the
// LDS spectrum (two contiguous runs), multiplies by its response (table transposed on the host: lanes read
// consecutive addresses), runs the inverse transform on its own registers (a forward DFT read backwards), cross-fades
// with the previous window's second half (registers) and stores: time-major output is 512 contiguous bytes per
// instant.  Nothing is exchanged between lanes after the spectrum.
//
// A workgroup (= one wavefront) owns a run of consecutive windows; the cross-fade partner at the seam between two runs is
// handed over through global memory (see handoff_store); 4 workgroups fit a CU (33 KB of LDS each), one per SIMD, each with the full 512-register file:
// the instruction-level parallelism of 64 independent points per lane is what keeps the SIMD busy.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include "kernels.hpp"
#include "fft_core.hpp"
#include "sd_math.hpp"

namespace {
using namespace fftcore;

// W_64^m = exp(-2 pi i m / 64), binary32-rounded
__device__ constexpr float kC64[64] = {
    1.000000000e+00f, 9.951847196e-01f, 9.807852507e-01f, 9.569403529e-01f, 9.238795042e-01f, 8.819212914e-01f, 8.314695954e-01f, 7.730104327e-01f,
    7.071067691e-01f, 6.343932748e-01f, 5.555702448e-01f, 4.713967443e-01f, 3.826834261e-01f, 2.902846634e-01f, 1.950903237e-01f, 9.801714122e-02f,
    0.0f, -9.801714122e-02f, -1.950903237e-01f, -2.902846634e-01f, -3.826834261e-01f, -4.713967443e-01f, -5.555702448e-01f, -6.343932748e-01f,
    -7.071067691e-01f, -7.730104327e-01f, -8.314695954e-01f, -8.819212914e-01f, -9.238795042e-01f, -9.569403529e-01f, -9.807852507e-01f, -9.951847196e-01f,
    -1.000000000e+00f, -9.951847196e-01f, -9.807852507e-01f, -9.569403529e-01f, -9.238795042e-01f, -8.819212914e-01f, -8.314695954e-01f, -7.730104327e-01f,
    -7.071067691e-01f, -6.343932748e-01f, -5.555702448e-01f, -4.713967443e-01f, -3.826834261e-01f, -2.902846634e-01f, -1.950903237e-01f, -9.801714122e-02f,
    0.0f, 9.801714122e-02f, 1.950903237e-01f, 2.902846634e-01f, 3.826834261e-01f, 4.713967443e-01f, 5.555702448e-01f, 6.343932748e-01f,
    7.071067691e-01f, 7.730104327e-01f, 8.314695954e-01f, 8.819212914e-01f, 9.238795042e-01f, 9.569403529e-01f, 9.807852507e-01f, 9.951847196e-01f};
__device__ constexpr float kS64[64] = {
    0.0f, -9.801714122e-02f, -1.950903237e-01f, -2.902846634e-01f, -3.826834261e-01f, -4.713967443e-01f, -5.555702448e-01f, -6.343932748e-01f,
    -7.071067691e-01f, -7.730104327e-01f, -8.314695954e-01f, -8.819212914e-01f, -9.238795042e-01f, -9.569403529e-01f, -9.807852507e-01f, -9.951847196e-01f,
    -1.000000000e+00f, -9.951847196e-01f, -9.807852507e-01f, -9.569403529e-01f, -9.238795042e-01f, -8.819212914e-01f, -8.314695954e-01f, -7.730104327e-01f,
    -7.071067691e-01f, -6.343932748e-01f, -5.555702448e-01f, -4.713967443e-01f, -3.826834261e-01f, -2.902846634e-01f, -1.950903237e-01f, -9.801714122e-02f,
    0.0f, 9.801714122e-02f, 1.950903237e-01f, 2.902846634e-01f, 3.826834261e-01f, 4.713967443e-01f, 5.555702448e-01f, 6.343932748e-01f,
    7.071067691e-01f, 7.730104327e-01f, 8.314695954e-01f, 8.819212914e-01f, 9.238795042e-01f, 9.569403529e-01f, 9.807852507e-01f, 9.951847196e-01f,
    1.000000000e+00f, 9.951847196e-01f, 9.807852507e-01f, 9.569403529e-01f, 9.238795042e-01f, 8.819212914e-01f, 8.314695954e-01f, 7.730104327e-01f,
    7.071067691e-01f, 6.343932748e-01f, 5.555702448e-01f, 4.713967443e-01f, 3.826834261e-01f, 2.902846634e-01f, 1.950903237e-01f, 9.801714122e-02f};

// sin^2(pi i / 64): the cross-fade window of a 64-point block (a size-S block uses every (64/S)-th value)
__device__ constexpr float kWin64[64] = {
    0.000000000e+00f, 2.407636726e-03f, 9.607359767e-03f, 2.152983285e-02f, 3.806023300e-02f, 5.903936923e-02f, 8.426519483e-02f, 1.134947762e-01f, 1.464466155e-01f, 1.828033626e-01f, 2.222148776e-01f, 2.643016279e-01f, 3.086582720e-01f, 3.548576534e-01f, 4.024548531e-01f, 4.509914219e-01f, 5.000000000e-01f, 5.490085483e-01f, 5.975451469e-01f, 6.451423168e-01f, 6.913416982e-01f, 7.356983423e-01f, 7.777851224e-01f, 8.171966672e-01f, 8.535534143e-01f, 8.865052462e-01f, 9.157348275e-01f, 9.409606457e-01f, 9.619397521e-01f, 9.784701467e-01f, 9.903926253e-01f, 9.975923896e-01f, 1.000000000e+00f, 9.975923896e-01f, 9.903926253e-01f, 9.784701467e-01f, 9.619397521e-01f, 9.409606457e-01f, 9.157348275e-01f, 8.865052462e-01f, 8.535534143e-01f, 8.171966672e-01f, 7.777851224e-01f, 7.356983423e-01f, 6.913416982e-01f, 6.451423168e-01f, 5.975451469e-01f, 5.490085483e-01f, 5.000000000e-01f, 4.509914219e-01f, 4.024548531e-01f, 3.548576534e-01f, 3.086582720e-01f, 2.643016279e-01f, 2.222148776e-01f, 1.828033626e-01f, 1.464466155e-01f, 1.134947762e-01f, 8.426519483e-02f, 5.903936923e-02f, 3.806023300e-02f, 2.152983285e-02f, 9.607359767e-03f, 2.407636726e-03f};

// a * b with b wave-uniform (a compile-time constant): the constant travels in an SGPR pair, not in VGPRs
__device__ __forceinline__ cf cmul_u(cf a, cf b)
{
  cf t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "s"(b));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "s"(b), "v"(t));
  return r;
}

// al cur + be prv with ONE rounding pattern wherever it is written (the seam block must equal the in-run block bit for bit)
__device__ __forceinline__ cf xfade(float al, cf cur, float be, cf prv)
{
  const cf t = cur * al;
  return __builtin_elementwise_fma(cf{be, be}, prv, t);
}

// a * W_64^m, m a constant after unrolling
__device__ __forceinline__ cf mul_w64(cf a, int m)
{
  m &= 63;
  if (m == 0) return a;
  if (m == 16) return cf{a.y, -a.x};
  if (m == 32) return -a;
  if (m == 48) return cf{-a.y, a.x};
  if (m == 8) return mul_w8_1(a);
  if (m == 24) return mul_w8_3(a);
  if (m == 40) return -mul_w8_1(a);
  if (m == 56) return -mul_w8_3(a);
  return cmul_u(a, cf{kC64[m], kS64[m]});
}

// N = R1 * R2 points on registers, natural order in and out: n = n1 + R1 n2, k = k2 + R2 k1
template <int R1, int R2>
__device__ __forceinline__ void dft_2f(const cf *in, cf *out)
{
  constexpr int N = R1 * R2;
  cf mid[N];
#pragma unroll
  for (int n1 = 0; n1 < R1; ++n1) {
    cf a[R2];
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) a[n2] = in[n1 + R1 * n2];
    dftR<R2>(a);
#pragma unroll
    for (int k2 = 0; k2 < R2; ++k2) mid[n1 + R1 * k2] = mul_w64(a[k2], n1 * k2 * (64 / N));
  }
#pragma unroll
  for (int k2 = 0; k2 < R2; ++k2) {
    cf b[R1];
#pragma unroll
    for (int n1 = 0; n1 < R1; ++n1) b[n1] = mid[n1 + R1 * k2];
    dftR<R1>(b);
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) out[k2 + R2 * k1] = b[k1];
  }
}

template <int LOG2N> __device__ __forceinline__ void dft_reg(const cf *in, cf *out)
{
  constexpr int N = 1 << LOG2N;
  if constexpr (LOG2N == 6) dft_2f<8, 8>(in, out);
  else if constexpr (LOG2N == 5) dft_2f<4, 8>(in, out);
  else {
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = in[i];
    dftR<N>(out);
  }
}

__device__ __forceinline__ long long opaque_ll(long long v) { asm volatile("" : "+v"(v)); return v; }
typedef __attribute__((address_space(1))) cf gcf;
typedef unsigned v2u __attribute__((ext_vector_type(2)));         // global (not flat) stores: they must not count against lgkmcnt
constexpr int WAVE = 64;
constexpr int WV_W = 4096, WV_H = 2048;
constexpr int WV_PITCH = 65;                                   // transposition pitch (elements)
constexpr int WV_REP = 32;                                     // spectrum bins repeated after the end (>= size/2)
constexpr int WV_HK = WV_W + WV_REP > WAVE * WV_PITCH ? WV_W + WV_REP : WAVE * WV_PITCH;   // uniform response: 64 elements after the spectrum
constexpr int WV_LDS = (WV_HK + 64) * 8;

// Hand-off between consecutive runs (no window is transformed twice).  Output block w needs the first half of window w
// and the second half of window w - 1; the block at the seam between run b and run b + 1 is emitted by run b: run b + 1
// publishes the (unweighted) first half of its first window as soon as it has it -- its very first step -- and run b
// picks it up after its own last window, long after.  Payload and flag are sc1 (agent-scope) accesses on both sides
// (they bypass the non-coherent L1 / per-XCD L2), the flag goes out after the producer's s_waitcnt vmcnt(0)
// (MI355X_MICROARCH.md, inter-workgroup visibility: "sc1 payload -> asm vmcnt(0) -> sc1 flag").  The consumer clears
// the flag: every launch leaves the flag array zeroed.  A run only ever waits for the NEXT workgroup in dispatch order,
// and only for that workgroup's first step: when workgroups queue for a slot the wait ends as soon as any resident
// one retires.
constexpr int AUX_SC1 = 16;                                    // cache-policy bit of the raw buffer builtins: sc1 (agent scope)

template <int LOG2S, bool UNIFORM>
__global__ __launch_bounds__(WAVE, 1) void stw_kernel(sdk::StArgs a)
{
  __builtin_amdgcn_s_setprio(3);   // ahead of the resident recurrence wavefronts
  constexpr int W = WV_W, H = WV_H, S = 1 << LOG2S, HS = S / 2, NG = WAVE / S;
  static_assert(HS <= WV_REP && S >= 4, "size out of range for the wavefront kernel");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cf *buf = reinterpret_cast<cf *>(smem);
  const int t = threadIdx.x;

  // W_4096^(t 2^j): exact table values; every other power is at most five products away
  cf wb[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) wb[j] = reinterpret_cast<const cf *>(a.tw_w)[t << j];

  const long long w_begin = (long long)blockIdx.x * a.run, w_end = (w_begin + a.run < a.nwin) ? w_begin + a.run : a.nwin;
  const cf *x = reinterpret_cast<const cf *>(a.x), *hist = reinterpret_cast<const cf *>(a.hist);
  const long long off = a.have_hist ? H : 0;                   // virtual stream = hist (H samples) ++ x
  const int kbase = blockIdx.y * (NG * WAVE) + t;              // this lane's channel in group g: kbase + 64 g
  const long long slot = (long long)blockIdx.x * gridDim.y + blockIdx.y;          // this run's hand-off slot
  cf *const ho = reinterpret_cast<cf *>(a.handoff);
  constexpr long long HO = (long long)NG * HS * WAVE;          // elements per slot

  cf prev[NG][HS];                                             // y_{w-1}[i + S/2] of this lane's channels
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int k = kbase + g * WAVE;
#pragma unroll
    for (int i = 0; i < HS; ++i)
      prev[g][i] = (w_begin == 0 && k < a.nchan) ? reinterpret_cast<const cf *>(a.prev_in)[(long long)k * HS + i] : cf{0.f, 0.f};
  }

  // per lane and group, fixed for the launch: centre bin, output base, residual-NCO flag (loads inside the window loop
  // would have to be waited for with vmcnt(0), i.e. together with the prefetch)
  int center[NG];
  cf *ybase[NG];
  bool precise[NG];
  uint32_t dphase[NG], phase0[NG];                             // residual NCO: step and (n0 - n_open) mod 2^32
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int k = kbase + g * WAVE;
    const sdk::StChan cd = a.chans[k < a.nchan ? k : 0];
    center[g] = cd.center;
    precise[g] = cd.precise != 0;
    dphase[g] = cd.dphase;
    phase0[g] = (uint32_t)(a.n0 - cd.n_open);
    ybase[g] = a.rows ? static_cast<cf *>(const_cast<void *>(a.rows[cd.row])) : reinterpret_cast<cf *>(a.y) + (long long)cd.row * a.yv.cs;
  }
  const long long yms = a.rows ? 1 : a.yv.ms;

  // output block `wo` of group g: residual NCO ("precise"), store through the view
  auto store_block = [&](int g, long long wo, cf *o) {
    const int k = kbase + g * WAVE;
    if (k >= a.nchan) return;
    if (precise[g]) {
#pragma unroll
      for (int i = 0; i < HS; ++i) {
        const uint32_t m = (uint32_t)((unsigned long long)wo * HS + i);
        float c, s;
        sd::phasor_u32((phase0[g] + m) * dphase[g], c, s);
        o[i] = cf{__builtin_fmaf(o[i].x, c, -(o[i].y * s)), __builtin_fmaf(o[i].x, s, o[i].y * c)};   // one rounding pattern at every call site
      }
    }
    gcf *yp = (gcf *)(ybase[g] + (long long)((unsigned long long)wo * HS) * yms);
    // a running pointer: 32 separately hoisted 64-bit addresses would not fit the register budget
    const long long ystep = opaque_ll(yms);
#pragma unroll
    for (int i = 0; i < HS; ++i) { *yp = o[i]; yp += ystep; }
  };

  // The window's samples are requested one window ahead straight into the registers the first DFT reads, by 64 buffer
  // loads (wave-uniform descriptor + 32-bit lane offset + immediate: no per-load address arithmetic).  The same 64
  // instructions serve three purposes, chosen by the descriptors: next window / the seam payload of the next run (first 32,
  // sc1) / nothing (zero-length descriptor: returns 0 without touching memory) -- so the number of loads in flight is
  // the same on every path and the waits are exact.
  cf nxt[WAVE];
  auto issue = [&](const cf *pa, unsigned na, const cf *pb, unsigned nb) {
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf *>(pa), 0, na, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf *>(pb), 0, nb, 0x00020000);
#pragma unroll
    for (int r = 0; r < WAVE / 2; ++r) nxt[r] = __builtin_bit_cast(cf, __builtin_amdgcn_raw_buffer_load_b64(ra, t * 8, r * WAVE * 8, AUX_SC1));
#pragma unroll
    for (int r = 0; r < WAVE / 2; ++r) nxt[r + WAVE / 2] = __builtin_bit_cast(cf, __builtin_amdgcn_raw_buffer_load_b64(rb, t * 8, r * WAVE * 8, 0));
  };
  auto request = [&](long long w) {
    issue((w == 0 && a.have_hist) ? hist : x + (w * H - off), H * 8, x + (w * H + H - off), H * 8);
  };
  const bool final_run = w_end == a.nwin;
  const bool single = w_end - w_begin == 1;                    // degenerate run: publish and pick-up both after the loop
  const long long nslot = slot + gridDim.y;
  // what follows window w in the load registers
  auto issue_next = [&](long long w) {
    if (w + 1 < w_end) request(w + 1);
    else if (!final_run && !single) {
      // the seam block's first half, published by the next run at its first step (its second window's top): long ago
      while (__hip_atomic_load(a.flags + nslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(8);
      asm volatile("" ::: "memory");
      issue(ho + nslot * HO, (unsigned)HO * 8, x, 0);
    } else issue(x, 0, x, 0);
  };

  if constexpr (UNIFORM) {
    // one response for every channel of the launch: 512 bytes of LDS, read as a broadcast
    if (t < S) buf[WV_HK + t] = reinterpret_cast<const cf *>(a.hk)[(long long)a.chans[0].hsel * S + t];
  }
  request(w_begin);
  bool publish = false;
  for (long long w = w_begin; w < w_end; ++w) {
    cf v[WAVE], A[WAVE];
    // ---- forward transform ----
    dft_reg<6>(nxt, A);
    if (publish) {
      // the wait for this window's samples (needed here anyway) also drains the seam stores issued before them
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (t == 0) __hip_atomic_store(a.flags + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      publish = false;
    }
    if constexpr (UNIFORM) issue_next(w);                      // nxt is dead after the first DFT: no copies, no second set
    {
      cf lo[8], hi[8];
      lo[1] = opaque(wb[0]); lo[2] = opaque(wb[1]); lo[4] = opaque(wb[2]);
      lo[3] = cmul(lo[1], lo[2]); lo[5] = cmul(lo[1], lo[4]); lo[6] = cmul(lo[2], lo[4]); lo[7] = cmul(lo[3], lo[4]);
      hi[1] = opaque(wb[3]); hi[2] = opaque(wb[4]); hi[4] = opaque(wb[5]);
      hi[3] = cmul(hi[1], hi[2]); hi[5] = cmul(hi[1], hi[4]); hi[6] = cmul(hi[2], hi[4]); hi[7] = cmul(hi[3], hi[4]);
#pragma unroll
      for (int k2 = 1; k2 < WAVE; ++k2) {
        const int l = k2 & 7, h = k2 >> 3;
        const cf wk = h == 0 ? lo[l] : (l == 0 ? hi[h] : cmul(hi[h], lo[l]));
        A[k2] = cmul(A[k2], wk);
      }
    }
    {
      cf *wr = buf + t * WV_PITCH;
#pragma unroll
      for (int k2 = 0; k2 < WAVE; ++k2) wr[k2] = A[k2];
      __builtin_amdgcn_wave_barrier();
      const cf *rd = buf + t;
#pragma unroll
      for (int tt = 0; tt < WAVE; ++tt) v[tt] = rd[tt * WV_PITCH];
    }
    dft_reg<6>(v, A);                                          // A[k1] = X[t + 64 k1]
    cf hkr[UNIFORM ? 1 : NG][UNIFORM ? 1 : S];
    if constexpr (!UNIFORM) {
      // per-channel responses: one batch of loads (L2 hits) issued behind the spectrum's way to LDS; VMEM returns in
      // order, so they go BEFORE the prefetch and can be waited for with its 64 loads still in flight
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<cf *>(reinterpret_cast<const cf *>(a.hkt)) + (long long)(blockIdx.y * NG + g) * (S * WAVE), 0, S * WAVE * 8, 0x00020000);
#pragma unroll
        for (int i = 0; i < S; ++i) hkr[g][i] = __builtin_bit_cast(cf, __builtin_amdgcn_raw_buffer_load_b64(rh, t * 8, i * WAVE * 8, 0));
      }
      issue_next(w);
    }
    __builtin_amdgcn_wave_barrier();
    {
      cf *sp = buf + t;
#pragma unroll
      for (int k1 = 0; k1 < WAVE; ++k1) sp[k1 * WAVE] = A[k1];
      if (t < WV_REP) buf[W + t] = A[0];
    }
    __builtin_amdgcn_wave_barrier();
    // ---- channel stage ----
    const bool seam = w == w_begin && w_begin > 0;             // this block belongs to the previous run: publish, do not emit
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int c0 = center[g], c1 = (center[g] - HS) & (W - 1);
      cf u[S], F[S];
#pragma unroll
      for (int i = 0; i < S; i += 2) {
        // two bins per read: the centre bin is even, so is i
        const float4 X2 = *reinterpret_cast<const float4 *>(buf + (i < HS ? c0 + i : c1 + (i - HS)));
        if constexpr (UNIFORM) {
          const float4 H2 = *reinterpret_cast<const float4 *>(buf + WV_HK + i);
          u[i] = cmul(cf{X2.x, X2.y}, cf{H2.x, H2.y});
          u[i + 1] = cmul(cf{X2.z, X2.w}, cf{H2.z, H2.w});
        } else {
          u[i] = cmul(cf{X2.x, X2.y}, hkr[g][i]);
          u[i + 1] = cmul(cf{X2.z, X2.w}, hkr[g][i + 1]);
        }
      }
      dft_reg<LOG2S>(u, F);                                    // y[n] = F[(S - n) mod S]: the inverse transform read backwards
      if (seam) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ho + slot * HO, 0, (int)HO * 8, 0x00020000);
#pragma unroll
        for (int i = 0; i < HS; ++i) {
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, F[(S - i) & (S - 1)]), rs, t * 8, (g * HS + i) * WAVE * 8, AUX_SC1);
          prev[g][i] = F[HS - i];
        }
      } else {
        cf o[HS];
#pragma unroll
        for (int i = 0; i < HS; ++i) {
          constexpr int ws = 64 / S;
          const float al = kWin64[i * ws], be = kWin64[(i + HS) * ws];   // compile-time constants
          o[i] = xfade(al, F[(S - i) & (S - 1)], be, prev[g][i]);
          prev[g][i] = F[HS - i];
        }
        store_block(g, w, o);
      }
    }
    if (seam) publish = true;                                  // the flag follows once the stores have drained (next window's top)
    __builtin_amdgcn_wave_barrier();                           // the next window's transposition overwrites the spectrum
  }
  if (publish) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (t == 0) __hip_atomic_store(a.flags + slot, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (final_run) {
    // carry the last window's second half to the next feed
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int k = kbase + g * WAVE;
      if (k < a.nchan) {
#pragma unroll
        for (int i = 0; i < HS; ++i) reinterpret_cast<cf *>(a.prev_out)[(long long)k * HS + i] = prev[g][i];
      }
    }
  } else {
    if (single) {
      while (__hip_atomic_load(a.flags + nslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(8);
      asm volatile("" ::: "memory");
      issue(ho + nslot * HO, (unsigned)HO * 8, x, 0);
    }
    // the seam block: nxt[g HS + i] = first half of the next run's first window
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      cf o[HS];
#pragma unroll
      for (int i = 0; i < HS; ++i) {
        constexpr int ws = 64 / S;
        const float al = kWin64[i * ws], be = kWin64[(i + HS) * ws];
        o[i] = xfade(al, nxt[g * HS + i], be, prev[g][i]);
      }
      store_block(g, w_end, o);
    }
    if (t == 0) __hip_atomic_store(a.flags + nslot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int LOG2S, bool UNIFORM>
hipError_t launch_stw_u(const sdk::StArgs &a, hipStream_t st)
{
  constexpr int NG = WAVE >> LOG2S;
  auto kern = stw_kernel<LOG2S, UNIFORM>;
  const unsigned nruns = (unsigned)((a.nwin + a.run - 1) / a.run);
  const unsigned ny = (unsigned)((a.nchan + NG * WAVE - 1) / (NG * WAVE));
  hipLaunchKernelGGL(kern, dim3(nruns, ny), dim3(WAVE), WV_LDS, st, a);
  return hipGetLastError();
}

template <int LOG2S>
hipError_t launch_stw(const sdk::StArgs &a, hipStream_t st)
{
  return a.hk_uniform ? launch_stw_u<LOG2S, true>(a, st) : launch_stw_u<LOG2S, false>(a, st);
}

}  // namespace

namespace sdk {

int stw_channels_per_wave(int log2s) { return (WAVE >> log2s) * WAVE; }

hipError_t specttuner_feed_wave(int log2s, const StArgs &a, hipStream_t st)
{
  if (a.nwin <= 0 || a.nchan <= 0) return hipSuccess;
  if (!a.hkt) return hipErrorInvalidValue;
  switch (log2s) {
    case 3: return launch_stw<3>(a, st);
    case 4: return launch_stw<4>(a, st);
    case 5: return launch_stw<5>(a, st);
    case 6: return launch_stw<6>(a, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace sdk
