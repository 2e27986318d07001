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
// phase clocks (STW_TSTAMP) and the wrong-result timing experiments (STP_UNSAFE_*) exist in the instrumented build only
#ifndef SUAMD_INSTRUMENT
#undef STW_TSTAMP
#undef STP_UNSAFE_NO_EXCHANGE
#undef STP_UNSAFE_NO_ALIAS_BARRIERS
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "kernels.hpp"
#include "fft_core.hpp"
#include "fft_reg.hpp"
#include "sd_math.hpp"

namespace {
using namespace fftcore;

// sin^2(pi i / 64): the cross-fade window of a 64-point block (a size-S block uses every (64/S)-th value)
__device__ constexpr float kWin64[64] = {
    0.000000000e+00f, 2.407636726e-03f, 9.607359767e-03f, 2.152983285e-02f, 3.806023300e-02f, 5.903936923e-02f, 8.426519483e-02f, 1.134947762e-01f, 1.464466155e-01f, 1.828033626e-01f, 2.222148776e-01f, 2.643016279e-01f, 3.086582720e-01f, 3.548576534e-01f, 4.024548531e-01f, 4.509914219e-01f, 5.000000000e-01f, 5.490085483e-01f, 5.975451469e-01f, 6.451423168e-01f, 6.913416982e-01f, 7.356983423e-01f, 7.777851224e-01f, 8.171966672e-01f, 8.535534143e-01f, 8.865052462e-01f, 9.157348275e-01f, 9.409606457e-01f, 9.619397521e-01f, 9.784701467e-01f, 9.903926253e-01f, 9.975923896e-01f, 1.000000000e+00f, 9.975923896e-01f, 9.903926253e-01f, 9.784701467e-01f, 9.619397521e-01f, 9.409606457e-01f, 9.157348275e-01f, 8.865052462e-01f, 8.535534143e-01f, 8.171966672e-01f, 7.777851224e-01f, 7.356983423e-01f, 6.913416982e-01f, 6.451423168e-01f, 5.975451469e-01f, 5.490085483e-01f, 5.000000000e-01f, 4.509914219e-01f, 4.024548531e-01f, 3.548576534e-01f, 3.086582720e-01f, 2.643016279e-01f, 2.222148776e-01f, 1.828033626e-01f, 1.464466155e-01f, 1.134947762e-01f, 8.426519483e-02f, 5.903936923e-02f, 3.806023300e-02f, 2.152983285e-02f, 9.607359767e-03f, 2.407636726e-03f};

// al cur + be prv with ONE rounding pattern wherever it is written (the seam block must equal the in-run block bit for bit)
__device__ __forceinline__ cf xfade(float al, cf cur, float be, cf prv)
{
  const cf t = cur * al;
  return __builtin_elementwise_fma(cf{be, be}, prv, t);
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
// (MI355X_MICROARCH.md, inter-workgroup visibility: "sc1 payload -> asm vmcnt(0) -> sc1 flag").  The flag's value
// is the launch's epoch (a host counter), so nothing is ever cleared.  A run only waits for the NEXT workgroup in
// dispatch order, only for that workgroup's first step, and only for a bounded time: if the payload has not appeared
// after ~60 us (that workgroup has not started -- every other slot of the device is held by something else), the run
// transforms the seam window itself (one more iteration of its loop) and ignores the payload when it comes.  No launch
// can hang on residency.
#ifdef STW_TSTAMP
#define TS(n) do { __builtin_amdgcn_sched_barrier(0); ts[n] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TS(n) do { } while (0)
#endif
#ifndef STW_WPB
#define STW_WPB 1
#endif
#ifndef STW_EARLY_LDS
#define STW_EARLY_LDS 1
#endif
constexpr bool EARLY_LDS = STW_EARLY_LDS != 0;                 // transposition writes / reads interleaved with the arithmetic around them
constexpr int AUX_NT = 2;                                      // nt: the outputs are written once and not read back here (64 ch x 4 Mi: 36.1 -> 34.4 us)
constexpr int AUX_SC1 = 16;                                    // cache-policy bit of the raw buffer builtins: sc1 (agent scope)

// ROTCAP: which channels of the launch are precise -- 0 none, 2 all, 1 some (specttuner_pair.hip: the forms that carry ONE form of
// the channel stage; here the whole stage -- bins, response, inverse transform, outputs -- is in each form, and the kernel with all
// three is 76 - 86 KB of code for a 64 KB instruction cache)
template <int LOG2S, bool UNIFORM, bool Y32, int ROTCAP>
__global__ __launch_bounds__(WAVE * STW_WPB, 1) void stw_kernel(sdk::StArgs a)
{
#ifdef STW_TSTAMP
  const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
  __builtin_amdgcn_s_setprio(3);   // ahead of the resident recurrence wavefronts
  constexpr int W = WV_W, H = WV_H, S = 1 << LOG2S, HS = S / 2, NG = WAVE / S;
  static_assert(HS <= WV_REP && S >= 4, "size out of range for the wavefront kernel");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // STW_WPB independent wavefronts per workgroup (nothing is shared, no barrier): fewer, fatter workgroups to dispatch
  const int wv = STW_WPB > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : 0;
  const long long bx = (long long)blockIdx.x * STW_WPB + wv;
  if (STW_WPB > 1 && bx * a.run >= a.nwin) return;
  cf *buf = reinterpret_cast<cf *>(smem + wv * WV_LDS);
  const int t = threadIdx.x & (WAVE - 1);

  const long long w_begin = bx * a.run, w_end = (w_begin + a.run < a.nwin) ? w_begin + a.run : a.nwin;
  const cf *x = reinterpret_cast<const cf *>(a.x), *hist = reinterpret_cast<const cf *>(a.hist);
  const long long off = a.have_hist ? H : 0;                   // virtual stream = hist (H samples) ++ x
  const int kbase = blockIdx.y * (NG * WAVE) + t;              // this lane's channel in group g: kbase + 64 g
  const long long slot = bx * gridDim.y + blockIdx.y;          // this run's hand-off slot
  cf *const ho = reinterpret_cast<cf *>(a.handoff);
  constexpr long long HO = (long long)NG * HS * WAVE;          // elements per slot

  // The window's samples are requested one window ahead straight into the registers the first DFT reads, by 64 buffer
  // loads (wave-uniform descriptor + 32-bit lane offset + immediate: no per-load address arithmetic).  The same 64
  // instructions serve three purposes, chosen by the descriptors: next window / the seam payload of the next run (first 32,
  // sc1) / nothing (zero-length descriptor: returns 0 without touching memory) -- so the number of loads in flight is
  // the same on every path and the waits are exact.
  cf nxt[WAVE];
  __amdgpu_buffer_rsrc_t ra, rb;                               // descriptors of the request in progress
  auto aim = [&](const cf *pa, unsigned na, const cf *pb, unsigned nb) {
    ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf *>(pa), 0, na, 0x00020000);
    rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf *>(pb), 0, nb, 0x00020000);
  };
  auto load_one = [&](int r) {
    if (r < WAVE / 2) nxt[r] = __builtin_bit_cast(cf, __builtin_amdgcn_raw_buffer_load_b64(ra, t * 8, r * WAVE * 8, AUX_SC1));
    else nxt[r] = __builtin_bit_cast(cf, __builtin_amdgcn_raw_buffer_load_b64(rb, t * 8, (r - WAVE / 2) * WAVE * 8, 0));
  };
  // (in the order the first transform consumes them -- its first-stage sub-transform n1 reads registers n1 + 8 n2 -- so
  // that a wavefront whose requests trickle in, as at the start of a launch when every wavefront asks at once, can begin
  // after the first eight)
  auto issue_all = [&]() {
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1)
#pragma unroll
      for (int n2 = 0; n2 < 8; ++n2) load_one(n1 + 8 * n2);
  };
  auto aim_window = [&](long long w) { aim((w == 0 && a.have_hist) ? hist : x + (w * H - off), H * 8, x + (w * H + H - off), H * 8); };
  const bool final_run = w_end == a.nwin;
  const bool single = w_end - w_begin == 1;                    // degenerate run: publish and pick-up both at the end of its only window
  const long long nslot = slot + gridDim.y;
  const unsigned epoch = a.epoch;                              // flag value of THIS launch
  long long w_stop = w_end;                                    // w_end + 1 when this run ends up transforming the seam window itself
  bool self_seam = false;
  // Has the next run published its seam payload?  Bounded wait: if that workgroup has not even started (every other
  // slot of the chip held by something else), this run transforms the seam window itself instead of waiting -- no
  // launch can hang on residency, whatever else occupies the device.  a.seam_polls (256) polls x ~0.25 us.
  auto seam_ready = [&]() -> bool {
    for (int n = 0; n < a.seam_polls; ++n) {
      const unsigned f = __builtin_amdgcn_readfirstlane(__hip_atomic_load(a.flags + nslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if (f == epoch) return true;
      __builtin_amdgcn_s_sleep(8);
    }
    return false;
  };
  // what follows window w in the load registers (branch-free: the loads themselves are spread over the window's arithmetic)
  auto aim_next = [&](long long w) {
    const bool more = w + 1 < w_stop, pay = !more && !final_run && !single && !self_seam;
    const long long wn = more ? w + 1 : w;
    const cf *pa = pay ? ho + nslot * HO : ((wn == 0 && a.have_hist) ? hist : x + (wn * H - off));
    aim(pa, more ? H * 8 : (pay ? (unsigned)HO * 8 : 0u), x + (wn * H + H - off), more ? H * 8 : 0u);
  };

  // the first window's samples before anything else: every other load of the prologue then travels in their shadow
  aim_window(w_begin);
  issue_all();
  __builtin_amdgcn_sched_barrier(0);


  // W_4096^(t 2^j): exact table values; every other power is at most five products away
  cf wb[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) wb[j] = reinterpret_cast<const cf *>(a.tw_w)[t << j];

  cf prev[NG][HS];                                             // y_{w-1}[i + S/2] of this lane's channels
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int k = kbase + g * WAVE;
#pragma unroll
    for (int i = 0; i < HS; ++i)
      prev[g][i] = (w_begin == 0 && k < a.nchan) ? reinterpret_cast<const cf *>(a.prev_in)[(long long)k * HS + i] : cf{0.f, 0.f};
  }

  // per lane and group, fixed for the launch: centre bin, output base, residual-NCO parameters (loads inside the window
  // loop would have to be waited for with vmcnt(0), i.e. together with the prefetch)
  int center[NG];
  cf *ybase[NG];                                               // 64-bit addressing (row pointers, or views beyond 2 GiB)
  unsigned yvoff[NG];                                          // 32-bit addressing: the lane's byte offset, out of range for lanes without a channel
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
    // (Y32 with a row table: the caller has promised every row within 2 GiB of a.y -- offsets from there)
    yvoff[g] = k >= a.nchan ? 0x80000000u : a.rows ? (unsigned)(reinterpret_cast<const char *>(a.rows[cd.row]) - reinterpret_cast<const char *>(a.y))
                                                   : (unsigned)((long long)cd.row * a.yv.cs * 8);
  }
  const long long yms = a.rows ? 1 : a.yv.ms;
  const unsigned yms8 = (unsigned)(yms * 8);

  // sample i of output block `wo` of group g: residual NCO ("precise"), store through the view.  With 32-bit offsets the
  // store is a buffer store: wave-uniform descriptor (the block's first instant) + the lane's offset + a scalar i * stride
  // Y32: host-checked, every byte offset of the view fits 31 bits.  `rot`: some lane of this wavefront has a residual NCO
  // (then every lane rotates, the others by exactly 1 + 0j: no divergent branch per sample)
  bool any_precise = ROTCAP == 2;
  if constexpr (ROTCAP == 1) {
#pragma unroll
    for (int g = 0; g < NG; ++g) any_precise |= __builtin_amdgcn_ballot_w64(precise[g]) != 0;
  }
  auto emit_one = [&](auto rot, int g, long long wo, int i, cf o) {
    if constexpr (decltype(rot)::value) {
      const uint32_t m = (uint32_t)((unsigned long long)wo * HS + i);
      sd::v2f_ cs = sd::phasor_pk((phase0[g] + m) * dphase[g]);          // (packed: specttuner_pair.hip's emit_one)
      if constexpr (ROTCAP != 2) { cs.x = precise[g] ? cs.x : 1.0f; cs.y = precise[g] ? cs.y : 0.0f; }
      o = sd::mix_rot(o, cs);                                            // one rounding pattern at every call site
    }
    if constexpr (Y32) {
      const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<cf *>(a.y) + (long long)((unsigned long long)wo * HS) * yms, 0, 0x7fffffff, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, o), ry, yvoff[g], (unsigned)i * yms8, AUX_NT);
    } else if (kbase + g * WAVE < a.nchan) {
      *(gcf *)(ybase[g] + ((long long)((unsigned long long)wo * HS) + i) * yms) = o;
    }
  };

  if constexpr (UNIFORM) {
    // one response for every channel of the launch: 512 bytes of LDS, read as a broadcast
    if (t < S) buf[WV_HK + t] = reinterpret_cast<const cf *>(a.hk)[t];    // a uniform launch has one response: selector 0
  }
  bool publish = false;
  for (long long w = w_begin; w < w_stop; ++w) {
    cf v[WAVE], A[WAVE];
#ifdef STW_TSTAMP
    unsigned long long ts[16] = {0};
#endif
    TS(0);
    if (publish) {
      // the wait for this window's samples (needed here anyway) also drains the seam stores issued before them
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (t == 0) __hip_atomic_store(a.flags + slot, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      publish = false;
    }
    if (final_run && w + 1 == w_end && a.hist_out != nullptr && blockIdx.y == 0) {
      // the stream's last half window is the next feed's history: this wavefront holds it (no separate copy launch)
      const __amdgpu_buffer_rsrc_t rhs = __builtin_amdgcn_make_buffer_rsrc(a.hist_out, 0, H * 8, 0x00020000);
#pragma unroll
      for (int r = 0; r < WAVE / 2; ++r)
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, nxt[WAVE / 2 + r]), rhs, t * 8, r * WAVE * 8, 0);
    }
    // ---- forward transform ----
    dft_reg<6>(nxt, A);
    TS(1);
    if (w + 1 == w_end && !final_run && !single) {
      // the seam block's first half, published by the next run at its first step (its second window's top);
      // this run's own flag went out above, BEFORE this wait: no chain of runs waiting for each other
      if (!seam_ready()) { self_seam = true; w_stop = w_end + 1; }
      asm volatile("" ::: "memory");
    }
    // nxt is dead after the first DFT: the next request goes into the same registers (no copies, no second set).  Its
    // 64 loads are SPREAD over the arithmetic that follows (4 per 8 twiddles, 2 per sub-transform of the second DFT):
    // a burst of 64 x 512 B fills the CU's memory pipeline and the lone wavefront then sits in the issue stage
    if constexpr (UNIFORM) aim_next(w);
    TS(2);
    {
      cf lo[8], hi[8];
      lo[1] = opaque(wb[0]); lo[2] = opaque(wb[1]); lo[4] = opaque(wb[2]);
      lo[3] = cmul(lo[1], lo[2]); lo[5] = cmul(lo[1], lo[4]); lo[6] = cmul(lo[2], lo[4]); lo[7] = cmul(lo[3], lo[4]);
      hi[1] = opaque(wb[3]); hi[2] = opaque(wb[4]); hi[4] = opaque(wb[5]);
      hi[3] = cmul(hi[1], hi[2]); hi[5] = cmul(hi[1], hi[4]); hi[6] = cmul(hi[2], hi[4]); hi[7] = cmul(hi[3], hi[4]);
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        // W^(t (8h + l)) = hi[h] lo[l]; two elements per asm statement
        if (h == 0) {
          A[1] = cmul1(A[1], lo[1]);
#pragma unroll
          for (int l = 2; l < 8; l += 2) cmul1x2(A[l], lo[l], A[l + 1], lo[l + 1], A[l], A[l + 1]);
        } else {
          A[8 * h] = cmul1(A[8 * h], hi[h]);
          A[8 * h + 1] = cmul3(A[8 * h + 1], hi[h], lo[1]);
#pragma unroll
          for (int l = 2; l < 8; l += 2) cmul3x2(A[8 * h + l], hi[h], lo[l], A[8 * h + l + 1], hi[h], lo[l + 1], A[8 * h + l], A[8 * h + l + 1]);
        }
        // this group's eight products go to the transposition buffer right away: the LDS writes travel under the next
        // group's arithmetic instead of in one burst behind the last
        if constexpr (EARLY_LDS) {
          cf *wr = buf + t * WV_PITCH;
#pragma unroll
          for (int l = 0; l < 8; ++l) wr[8 * h + l] = A[8 * h + l];
        }
        if constexpr (UNIFORM) {
#pragma unroll
          for (int r = 0; r < 4; ++r) load_one(4 * h + r);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    TS(3);
    {
      if constexpr (!EARLY_LDS) {
        cf *wr = buf + t * WV_PITCH;
#pragma unroll
        for (int k2 = 0; k2 < WAVE; ++k2) wr[k2] = A[k2];
      }
      __builtin_amdgcn_wave_barrier();
      const cf *rd = buf + t;
      if constexpr (EARLY_LDS) {
        // in the order the second transform consumes them (its first-stage sub-transform n1 reads v[n1 + 8 n2]): the first
        // sub-transform can start after 8 reads instead of 57
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1)
#pragma unroll
          for (int n2 = 0; n2 < 8; ++n2) v[n1 + 8 * n2] = rd[(n1 + 8 * n2) * WV_PITCH];
      } else {
#pragma unroll
        for (int tt = 0; tt < WAVE; ++tt) v[tt] = rd[tt * WV_PITCH];
      }
    }
    TS(4);
    // (EARLY_LDS: the rows of the spectrum a second-stage sub-transform completes -- X[t + 64 (k2 + 8 k1)], k1 = 0..7 --
    // go to LDS behind it; every lane's reads of the transposition buffer are done by then: the second stage starts after
    // the first has consumed all 64 values)
    auto spectrum_rows = [&](int step) {
      if constexpr (EARLY_LDS) {
        if (step >= 8) {
          if (step == 8) __builtin_amdgcn_wave_barrier();       // all lanes are through with the transposition buffer
          const int k2 = step - 8;
          cf *sp = buf + t;
#pragma unroll
          for (int k1 = 0; k1 < 8; ++k1) sp[(k2 + 8 * k1) * WAVE] = A[k2 + 8 * k1];
          if (k2 == 0 && t < WV_REP) buf[W + t] = A[0];
        }
      }
    };
    if constexpr (UNIFORM) {
      dft_reg<6>(v, A, [&](int step) {                         // A[k1] = X[t + 64 k1]
        if (step < 8) {
#pragma unroll
          for (int r = 0; r < 4; ++r) load_one(WAVE / 2 + 4 * step + r);
          __builtin_amdgcn_sched_barrier(0);
        }
        spectrum_rows(step);
      });
    } else dft_reg<6>(v, A, spectrum_rows);
    TS(5);
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
      aim_next(w);
      issue_all();
    }
    if constexpr (!EARLY_LDS) {
      __builtin_amdgcn_wave_barrier();
      cf *sp = buf + t;
#pragma unroll
      for (int k1 = 0; k1 < WAVE; ++k1) sp[k1 * WAVE] = A[k1];
      if (t < WV_REP) buf[W + t] = A[0];
    }
    __builtin_amdgcn_wave_barrier();
    TS(6);
    // ---- channel stage ----
    const bool seam = w == w_begin && w_begin > 0;             // this block belongs to the previous run: publish, do not emit
    auto chan = [&](auto seam_tag, auto rot) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int c0 = center[g], c1 = (center[g] - HS) & (W - 1);
      cf u[S], F[S];
      if constexpr (UNIFORM) {
        // all the bin reads first (two bins per ds_read_b128: the centre bin is even), then the response in chunks of 16
        // bins, each requested one chunk ahead of its products: left to itself the compiler alternates read / wait / use
        // and the lone wavefront pays the LDS latency 32 times
        float4 X2[S / 2];
#pragma unroll
        for (int i = 0; i < S; i += 2) X2[i / 2] = *reinterpret_cast<const float4 *>(buf + (i < HS ? c0 + i : c1 + (i - HS)));
        __builtin_amdgcn_sched_barrier(0);
        constexpr int CH = S / 2 < 8 ? S / 2 : 8, NCH = (S / 2) / CH;
        float4 Hq[2][CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) Hq[0][j] = *reinterpret_cast<const float4 *>(buf + WV_HK + 2 * j);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          if (c + 1 < NCH) {
#pragma unroll
            for (int j = 0; j < CH; ++j) Hq[(c + 1) & 1][j] = *reinterpret_cast<const float4 *>(buf + WV_HK + 2 * ((c + 1) * CH + j));
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            const float4 X = X2[c * CH + j], Hh = Hq[c & 1][j];
            cmul1x2(cf{X.x, X.y}, cf{Hh.x, Hh.y}, cf{X.z, X.w}, cf{Hh.z, Hh.w}, u[2 * (c * CH + j)], u[2 * (c * CH + j) + 1]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < S; i += 2) {
          const float4 X2 = *reinterpret_cast<const float4 *>(buf + (i < HS ? c0 + i : c1 + (i - HS)));
          u[i] = cmul(cf{X2.x, X2.y}, hkr[g][i]);
          u[i + 1] = cmul(cf{X2.z, X2.w}, hkr[g][i + 1]);
        }
      }
      if (g == 0) TS(7);
      // y[n] = F[(S - n) mod S] (the inverse transform is a forward DFT read backwards).  Output i needs F[(S - i) mod S]
      // (this block's first half) and F[S/2 - i] (its second half, the next block's partner): both come out of the same
      // second-stage sub-transform, so each sub-transform is followed by ITS cross-fades and stores -- the stores are
      // spread over the arithmetic like the loads
      auto finish = [&](int i) {
        const cf cur = F[(S - i) & (S - 1)], nx = F[HS - i];
        if constexpr (decltype(seam_tag)::value) {
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ho + slot * HO, 0, (int)HO * 8, 0x00020000);
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, cur), rs, t * 8, (g * HS + i) * WAVE * 8, AUX_SC1);
        } else {
          constexpr int ws = 64 / S;
          emit_one(rot, g, w, i, xfade(kWin64[i * ws], cur, kWin64[(i + HS) * ws], prev[g][i]));
        }
        prev[g][i] = nx;
      };
      if constexpr (LOG2S >= 5) {
        dft_reg<LOG2S>(u, F, [&](int step) {
          constexpr int R1 = S / 8;
          if (step >= R1) {
            const int k2 = step - R1;
#pragma unroll
            for (int m = 0; m < HS / 8; ++m) finish(((8 - k2) & 7) + 8 * m);
            __builtin_amdgcn_sched_barrier(0);
          }
        });
      } else {
        dft_reg<LOG2S>(u, F);
#pragma unroll
        for (int i = 0; i < HS; ++i) finish(i);
      }
      if (g == 0) TS(8);
    }
    };
    if (seam) chan(std::true_type{}, std::false_type{});
    else if constexpr (ROTCAP == 2) chan(std::false_type{}, std::true_type{});
    else if constexpr (ROTCAP == 0) chan(std::false_type{}, std::false_type{});
    else if (any_precise) chan(std::false_type{}, std::bool_constant<ROTCAP != 0>{});
    else chan(std::false_type{}, std::false_type{});
    if (seam) publish = true;                                  // the flag follows once the stores have drained (next window's top)
    if (single && !final_run && w + 1 == w_end) {
      // a run of one window: its own flag first (nobody may wait for a run that is itself waiting), then the pick-up
      if (publish) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (t == 0) __hip_atomic_store(a.flags + slot, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        publish = false;
      }
      if (seam_ready()) aim(ho + nslot * HO, (unsigned)HO * 8, x, 0);
      else { self_seam = true; w_stop = w_end + 1; aim_window(w_end); }
      asm volatile("" ::: "memory");
      issue_all();
    }
    TS(9);
#ifdef STW_TSTAMP
    if (a.tstamp && t == 0) {
      unsigned long long *tp = a.tstamp + ((long long)slot * a.run + (w - w_begin)) * 16;
      for (int n = 0; n < 10; ++n) tp[n] = ts[n];
    }
#endif
    __builtin_amdgcn_wave_barrier();                           // the next window's transposition overwrites the spectrum
  }
  if (publish) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (t == 0) __hip_atomic_store(a.flags + slot, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
  } else if (!self_seam) {
    // the seam block: nxt[g HS + i] = first half of the next run's first window
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
      for (int i = 0; i < HS; ++i) {
        constexpr int ws = 64 / S;
        const cf o = xfade(kWin64[i * ws], nxt[g * HS + i], kWin64[(i + HS) * ws], prev[g][i]);
        if constexpr (ROTCAP == 2) emit_one(std::true_type{}, g, w_end, i, o);
        else if constexpr (ROTCAP == 0) emit_one(std::false_type{}, g, w_end, i, o);
        else if (any_precise) emit_one(std::bool_constant<ROTCAP != 0>{}, g, w_end, i, o);
        else emit_one(std::false_type{}, g, w_end, i, o);
      }
    }
  }
#ifdef STW_TSTAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (a.tstamp && t == 0) { unsigned long long *tp = a.tstamp + ((long long)slot * a.run) * 16; tp[10] = t_entry; tp[11] = __builtin_amdgcn_s_memtime(); }
#endif
}

template <int LOG2S, bool UNIFORM, bool Y32, int ROTCAP>
hipError_t launch_stw_u(const sdk::StArgs &a, hipStream_t st)
{
  constexpr int NG = WAVE >> LOG2S;
  auto kern = stw_kernel<LOG2S, UNIFORM, Y32, ROTCAP>;
  const unsigned nruns = (unsigned)((a.nwin + a.run - 1) / a.run);
  const unsigned ny = (unsigned)((a.nchan + NG * WAVE - 1) / (NG * WAVE));
  if constexpr (WV_LDS * STW_WPB > 65536) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, WV_LDS * STW_WPB);
    if (attr != hipSuccess) return attr;
  }
  sdk::launch_timed("stw_kernel", kern, dim3((nruns + STW_WPB - 1) / STW_WPB, ny), dim3(WAVE * STW_WPB), WV_LDS * STW_WPB, st, a);
  return hipGetLastError();
}

template <int LOG2S>
hipError_t launch_stw(const sdk::StArgs &a, hipStream_t st)
{
  // (32-bit offsets come in the three forms of ROTCAP; the 64-bit fallback in the general one)
  if (a.y32 && a.any_precise == 0) return a.hk_uniform ? launch_stw_u<LOG2S, true, true, 0>(a, st) : launch_stw_u<LOG2S, false, true, 0>(a, st);
  if (a.y32 && a.any_precise == 2) return a.hk_uniform ? launch_stw_u<LOG2S, true, true, 2>(a, st) : launch_stw_u<LOG2S, false, true, 2>(a, st);
  if (a.y32) return a.hk_uniform ? launch_stw_u<LOG2S, true, true, 1>(a, st) : launch_stw_u<LOG2S, false, true, 1>(a, st);
  return a.hk_uniform ? launch_stw_u<LOG2S, true, false, 1>(a, st) : launch_stw_u<LOG2S, false, false, 1>(a, st);
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
