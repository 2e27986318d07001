// specttuner_pair.hip -- the FFT channeliser for banks of narrow channels (8 .. 64 bins) with one response, TWO wavefronts per window
// (SPEC.md section C2; rows T2 / N2).  Same arithmetic, operation for operation, as specttuner_wave.hip (the oracle's
// binary32 statement: both kernels equal it bit for bit); what changes is who holds what.
//
// specttuner_wave.hip keeps a whole 4096-point window in ONE wavefront's registers (64 points per lane, 512 registers): one
// wavefront per SIMD, and a lone wavefront issues an instruction every ~6 cycles at best (no second wavefront to fill
// the slots its scalar / LDS / memory instructions and their waits leave).  Here a window belongs to a workgroup of two
// wavefronts with 32 points per lane each (<= 256 registers: two wavefronts per SIMD, eight per CU -- the LDS still
// holds four windows per CU).  Every 64-point DFT of the window (8 x 8 on registers) is split down the middle:
//   wavefront p runs the first-stage sub-transforms n1 = 4p .. 4p+3 (inputs n1 + 8 n2),
//   the halves swap 16 values per lane through LDS,
//   wavefront p runs the second-stage sub-transforms k2 = 4p .. 4p+3 (outputs k2 + 8 k1).
// Each wavefront runs its own specialisation of the code (p is a template parameter), so the W_64 twiddles stay the
// compile-time constants they are in the one-wavefront kernel.  Loads, transposition, spectrum, bin gather and stores
// can hand any lane any element, so only the three mid-DFT swaps are new traffic.
// phase clocks (STW_TSTAMP) and the wrong-result timing experiments (STP_UNSAFE_*) exist in the instrumented build only
#ifndef SUAMD_INSTRUMENT
#undef STW_TSTAMP
#undef STP_UNSAFE_NO_EXCHANGE
#undef STP_UNSAFE_NO_ALIAS_BARRIERS
#undef STP_UNSAFE_NO_STORES
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "kernels.hpp"
#include "tuning.hpp"
#include "fft_core.hpp"
#include "fft_reg.hpp"
#include "sd_math.hpp"

namespace {
using namespace fftcore;

// sin^2(pi i / 64): the cross-fade window of a 64-point block
__device__ constexpr float kWinP[64] = {
    0.000000000e+00f, 2.407636726e-03f, 9.607359767e-03f, 2.152983285e-02f, 3.806023300e-02f, 5.903936923e-02f, 8.426519483e-02f, 1.134947762e-01f, 1.464466155e-01f, 1.828033626e-01f, 2.222148776e-01f, 2.643016279e-01f, 3.086582720e-01f, 3.548576534e-01f, 4.024548531e-01f, 4.509914219e-01f, 5.000000000e-01f, 5.490085483e-01f, 5.975451469e-01f, 6.451423168e-01f, 6.913416982e-01f, 7.356983423e-01f, 7.777851224e-01f, 8.171966672e-01f, 8.535534143e-01f, 8.865052462e-01f, 9.157348275e-01f, 9.409606457e-01f, 9.619397521e-01f, 9.784701467e-01f, 9.903926253e-01f, 9.975923896e-01f, 1.000000000e+00f, 9.975923896e-01f, 9.903926253e-01f, 9.784701467e-01f, 9.619397521e-01f, 9.409606457e-01f, 9.157348275e-01f, 8.865052462e-01f, 8.535534143e-01f, 8.171966672e-01f, 7.777851224e-01f, 7.356983423e-01f, 6.913416982e-01f, 6.451423168e-01f, 5.975451469e-01f, 5.490085483e-01f, 5.000000000e-01f, 4.509914219e-01f, 4.024548531e-01f, 3.548576534e-01f, 3.086582720e-01f, 2.643016279e-01f, 2.222148776e-01f, 1.828033626e-01f, 1.464466155e-01f, 1.134947762e-01f, 8.426519483e-02f, 5.903936923e-02f, 3.806023300e-02f, 2.152983285e-02f, 9.607359767e-03f, 2.407636726e-03f};

// al cur + be prv, the rounding pattern of specttuner_wave.hip's xfade
__device__ __forceinline__ cf xfade_p(float al, cf cur, float be, cf prv)
{
  const cf t = cur * al;
  return __builtin_elementwise_fma(cf{be, be}, prv, t);
}

typedef __attribute__((address_space(1))) cf gcf;
// the channel table as the compiler may treat it: written by the host before the launch, constant while it runs (loads from
// the constant address space are not clobbered by the kernel's stores)
typedef const __attribute__((address_space(4))) sdk::StChan *cchan;
typedef unsigned v2u __attribute__((ext_vector_type(2)));
constexpr int WAVE = 64;
constexpr int PW_W = 4096, PW_H = 2048;
constexpr int PW_PITCH = 65;                                   // transposition pitch (elements)
constexpr int PW_REP = 32;                                     // spectrum bins repeated after the end
constexpr int PW_FLAG = WAVE * PW_PITCH;                       // one word the wavefronts of a pair share (seam decision), behind the transposition buffer
constexpr int PW_TAB = PW_FLAG + 2;                            // the launch's response tables: nsel x (S + 2) elements (the pad
                                                               // shifts each table by four banks: lanes with different responses
                                                               // reading the same bin do not meet in a bank)
constexpr int PW_MAXSEL = 12;                                  // 64-bin tables that still leave four workgroups per CU (39 KB: the LDS is handed out in blocks)
constexpr int PW_EX = 1024;                                    // elements of one direction of a mid-DFT swap (16 x 64)
constexpr int AUX_NT = 2;
constexpr int AUX_SC1 = 16;
// The first half of a window is the half the run's previous window requested as its second: its last use.  Requested
// non-temporal (-DSTP_FIRST_HALF_AUX=2: a hit does not renew the line, the L2 -- 4 MiB per XCD, about what an XCD's 128
// workgroups stream per window -- keeps the second halves, which ARE used again, one window longer) the kernel moves the bytes
// its run structure gives by construction: FETCH_SIZE 80.7 MiB per 16 Mi block against 100.1 with the default policy
// (traffic 1.19 x against 1.34 x of the algorithmic bytes) -- and takes LONGER: 80.4 against 78.6 us (rocprofv3, same box,
// alternating, twice; sc0 | nt: 80.4, sc0: 78.6).  HBM traffic is not what bounds this kernel, the non-temporal requests
// cost more than the re-reads they save: the default policy is the default again (it was non-temporal for most of round 6).
#ifndef STP_FIRST_HALF_AUX
#define STP_FIRST_HALF_AUX 0
#endif

#ifdef STW_TSTAMP
#define TS(n) do { __builtin_amdgcn_sched_barrier(0); ts[n] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define TS(n) do { } while (0)
#endif

// LDS operations of the two wavefronts are only ordered by a barrier; the prefetch (vmcnt) stays in flight across it
// (barriers that only keep a swap area from being overwritten early: STP_UNSAFE_NO_ALIAS_BARRIERS drops them -- a timing
// experiment, results are wrong)
__device__ __forceinline__ void alias_barrier();
__device__ __forceinline__ void pair_barrier()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void alias_barrier()
{
#ifndef STP_UNSAFE_NO_ALIAS_BARRIERS
  pair_barrier();
#endif
}

// One 64-point DFT, split between the wavefronts of the pair.  in[a * 8 + n2] = x[(4P + a) + 8 n2]; on return
// out[b * 8 + k1] = X[(4P + b) + 8 k1].  `ex`: 2 x PW_EX elements of LDS nobody else touches between the barrier
// inside and the caller's next barrier.  hook(step) runs after each of the eight sub-transforms.
template <int P, class Hook>
__device__ __forceinline__ void dft64_pair(const cf *in, cf *out, cf *ex, int t, Hook hook)
{
  constexpr int Q = 1 - P;
  cf keep[16];
#ifdef STP_UNSAFE_NO_EXCHANGE
  cf unsafe[16];
#endif
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    cf a8[8];
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) a8[n2] = in[a * 8 + n2];
    dftR<8>(a8);
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) {
      const cf m = mul_w64(a8[k2], (4 * P + a) * k2);
      if (k2 / 4 == P) keep[a * 4 + (k2 & 3)] = m;
#ifdef STP_UNSAFE_NO_EXCHANGE
      else unsafe[a * 4 + (k2 & 3)] = m;                       // timing experiment: the partner's half never travels (results wrong)
#else
      else ex[(P * 16 + a * 4 + (k2 & 3)) * WAVE + t] = m;
#endif
    }
    hook(a);
  }
#ifndef STP_UNSAFE_NO_EXCHANGE
  pair_barrier();
#endif
  // the partner's sixteen values, requested in one go (left alone the compiler alternates read / wait / use and a lone
  // wavefront pays the LDS latency every time)
  cf recv[16];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int aq = 0; aq < 4; ++aq)
#ifdef STP_UNSAFE_NO_EXCHANGE
      recv[b * 4 + aq] = unsafe[aq * 4 + b];
#else
      recv[b * 4 + aq] = ex[(Q * 16 + aq * 4 + b) * WAVE + t];
#endif
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    cf b8[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) b8[n1] = (n1 / 4 == P) ? keep[(n1 & 3) * 4 + b] : recv[b * 4 + (n1 & 3)];
    dftR<8>(b8);
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) out[b * 8 + k1] = b8[k1];
    hook(4 + b);
  }
}

// SEP: the forward swaps have 16 KiB of their own behind the buffer (a launch of at most three workgroups per CU can
// afford 50 KB each): the three barriers that only keep them off the transposition data go away
// Channels of S = 2^LOG2S bins.  S = 64: a lane's channel is split between the wavefronts like the forward DFTs (third
// swap).  S < 64: a lane serves 64 / S channels, wavefront p takes half of them whole -- their inverse transforms are
// lane-local, no third swap.  Either way a wavefront owns 16 outputs per lane and block (`slot` o = 0..15 below).
// ROWT (64-bin channels, 32-bit row offsets): the outputs go to per-channel ROWS -- the live analyzer's layout, one row per
// inspector -- through a transposition in LDS.  Lane = channel means a store instruction touches 64 rows with 8 bytes each:
// 118 us per 2 Mi-sample block in the 64-inspector analyzer (rocprofv3, round 6) against 17 for time-major output.  Staged,
// a block's 64 x 32 outputs leave as 16-byte pieces, sixteen lanes covering 256 contiguous bytes of a row.
// ROTCAP: which channels of the launch are precise (residual NCO on the outputs) -- 0: none, 2: all, 1: some (each wavefront
// looks at its own lanes).  0 and 2 leave the other form of the channel stage out of the kernel altogether: merely present --
// never executed -- the residual-NCO code cost a bank without precise channels 1.8 us per 16 Mi block (78.6 -> 76.8 us,
// same-box A / B, three times in turn, round 6: code size and register allocation of the common path).
template <int P, int LOG2S, bool Y32, bool SEP, bool ROWT, int ROTCAP>
__device__ __forceinline__ void stp_body(const sdk::StArgs &a, cf *buf, const int t)
{
  static_assert(!ROWT || (LOG2S == 6 && Y32), "row-transposed stores: 64-bin channels, 32-bit offsets");
  constexpr int TP = (1 << LOG2S) + 2;                         // pitch of a response table
  cf *const tab = buf + PW_TAB;
  cf *const exf = SEP ? tab + a.nsel * TP : buf;               // swap area of the two forward DFTs
  unsigned *const rowoff = reinterpret_cast<unsigned *>(tab + a.nsel * TP + (SEP ? 2 * PW_EX : 0));   // ROWT: byte offset of every lane's row
  constexpr int W = PW_W, H = PW_H;
  constexpr int S = 1 << LOG2S, HS = S / 2, NG = WAVE / S, NGW = LOG2S == 6 ? 1 : NG / 2, WS = 64 / S;
  static_assert(LOG2S >= 3 && LOG2S <= 6, "channel size out of range for the two-wavefront kernel");
#ifdef STW_TSTAMP
  const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
  const long long w_begin = (long long)blockIdx.x * a.run, w_end = (w_begin + a.run < a.nwin) ? w_begin + a.run : a.nwin;
  const cf *x = reinterpret_cast<const cf *>(a.x), *hist = reinterpret_cast<const cf *>(a.hist);
  const long long off = a.have_hist ? H : 0;                   // virtual stream = hist (H samples) ++ x
  const int kbase = blockIdx.y * (NG * WAVE) + t;              // this lane's channel in group g: kbase + 64 g
  const long long slot = (long long)blockIdx.x * gridDim.y + blockIdx.y;
  cf *const ho = reinterpret_cast<cf *>(a.handoff);
  constexpr long long HO = 32ll * WAVE;                        // elements per hand-off slot: 16 rows per wavefront
  unsigned *const shared_flag = reinterpret_cast<unsigned *>(buf + PW_FLAG);

  // Register (a, n2) of the request holds sample t + 64 r, r = 4P + a + 8 n2: rows r < 32 come through descriptor `ra`
  // (first half window, or the seam payload, or nothing), the others through `rb`
  cf nxt[32];
  __amdgpu_buffer_rsrc_t ra, rb;
  auto aim = [&](const cf *pa, unsigned na, const cf *pb, unsigned nb) {
    ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf *>(pa), 0, na, 0x00020000);
    rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<cf *>(pb), 0, nb, 0x00020000);
  };
  // (`pay`: the first-half request fetches a seam payload another workgroup -- another XCD -- has just written: agent scope.
  // Samples are requested with the default policy: the half window this run's previous window already fetched is then
  // found in the L2.  One policy for both would have to be sc1 -- and was, in the one-wavefront kernel: every first half
  // came past the L2.)
  auto load_one = [&](auto pay, int q) {                       // q = a * 8 + n2
    const int r = 4 * P + (q >> 3) + 8 * (q & 7);
    if (r < WAVE / 2) nxt[q] = __builtin_bit_cast(cf, __builtin_amdgcn_raw_buffer_load_b64(ra, t * 8, r * WAVE * 8, decltype(pay)::value ? AUX_SC1 : STP_FIRST_HALF_AUX));
    else nxt[q] = __builtin_bit_cast(cf, __builtin_amdgcn_raw_buffer_load_b64(rb, t * 8, (r - WAVE / 2) * WAVE * 8, 0));
  };
  bool pay_now = false;                                        // what the request in progress fetches through `ra`
  auto aim_window = [&](long long w) { aim((w == 0 && a.have_hist) ? hist : x + (w * H - off), H * 8, x + (w * H + H - off), H * 8); };
  const bool final_run = w_end == a.nwin;
  const long long nslot = slot + gridDim.y;
  const unsigned epoch = a.epoch;
  long long w_stop = w_end;
  bool self_seam = false;
  auto seam_ready = [&]() -> bool {
    for (int n = 0; n < a.seam_polls; ++n) {
      const unsigned f = __builtin_amdgcn_readfirstlane(__hip_atomic_load(a.flags + nslot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if (f == epoch) return true;
      __builtin_amdgcn_s_sleep(8);
    }
    return false;
  };
  auto aim_next = [&](long long w) {
    const bool more = w + 1 < w_stop, pay = !more && !final_run && !self_seam;
    pay_now = pay;
    const long long wn = more ? w + 1 : w;
    const cf *pa = pay ? ho + nslot * HO : ((wn == 0 && a.have_hist) ? hist : x + (wn * H - off));
    aim(pa, more ? H * 8 : (pay ? (unsigned)HO * 8 : 0u), x + (wn * H + H - off), more ? H * 8 : 0u);
  };

  aim_window(w_begin);
#pragma unroll
  for (int q = 0; q < 32; ++q) load_one(std::false_type{}, q);
  __builtin_amdgcn_sched_barrier(0);

  // W_4096^(t 2^j)
  cf wb[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) wb[j] = reinterpret_cast<const cf *>(a.tw_w)[t << j];

  // The 16 outputs per lane and block this wavefront owns, slot o = 0..15:
  //   S = 64: output i needs F[(64 - i) & 63] and F[32 - i], both of second-stage sub-transform k2 = (-i) mod 8, so the
  //           wavefront owns i(b, m) = ((8 - (4P + b)) & 7) + 8 m, o = 4 b + m;
  //   S < 64: outputs 0 .. S/2 - 1 of its NGW channels, o = gl S/2 + i.
  // Slot o's seam payload travels in row 4P + (o >> 2) + 8 (o & 3) of the hand-off slot, i.e. in request register
  // (a, n2) = (o >> 2, o & 3) of the consumer.
  auto out_index = [](int b, int m) { return ((8 - (4 * P + b)) & 7) + 8 * m; };
  auto slot_group = [](int o) { return LOG2S == 6 ? 0 : o / HS; };                  // local group gl
  auto slot_out = [&](int o) { return LOG2S == 6 ? out_index(o >> 2, o & 3) : o % HS; };
  auto slot_q = [](int o) { return (o >> 2) * 8 + (o & 3); };
  auto slot_row = [](int o) { return 4 * P + (o >> 2) + 8 * (o & 3); };
  auto group_of = [](int gl) { return LOG2S == 6 ? 0 : P * NGW + gl; };
  cf prev[16];
#pragma unroll
  for (int o = 0; o < 16; ++o) {
    const int kk = kbase + group_of(slot_group(o)) * WAVE;
    prev[o] = (w_begin == 0 && kk < a.nchan) ? reinterpret_cast<const cf *>(a.prev_in)[(long long)kk * HS + slot_out(o)] : cf{0.f, 0.f};
  }

  // (the channel record is read where it is used: the residual-NCO fields only on the path that rotates -- held across the
  // window loop they would be spilled)
  const sdk::StChan *cdp[NGW];
  int center[NGW];
  cf *ybase[NGW];
  unsigned yvoff[NGW];
  bool any_precise = ROTCAP == 2;
#pragma unroll
  for (int gl = 0; gl < NGW; ++gl) {
    const int kk = kbase + group_of(gl) * WAVE;
    cdp[gl] = a.chans + (kk < a.nchan ? kk : 0);
    center[gl] = cdp[gl]->center;
    const int row = cdp[gl]->row;
    ybase[gl] = a.rows ? static_cast<cf *>(const_cast<void *>(a.rows[row])) : reinterpret_cast<cf *>(a.y) + (long long)row * a.yv.cs;
    // (Y32 with a row table: the caller has promised every row within 2 GiB of a.y -- offsets from there)
    yvoff[gl] = kk >= a.nchan ? 0x80000000u : a.rows ? (unsigned)(reinterpret_cast<const char *>(a.rows[row]) - reinterpret_cast<const char *>(a.y))
                                                      : (unsigned)((long long)row * a.yv.cs * 8);
    if constexpr (ROTCAP == 1) any_precise |= __builtin_amdgcn_ballot_w64(cdp[gl]->precise != 0) != 0;
  }
  const long long yms = a.rows ? 1 : a.yv.ms;
  const unsigned yms8 = (unsigned)(yms * 8);
  if constexpr (ROWT) { if (P == 0) rowoff[t] = yvoff[0]; }      // (read behind the first window's barriers)
  // (S < 64: the wavefronts serve different channels and may differ in `any_precise` -- their channel stages have no barrier)

  // ROWT staging: element (channel t, sample i) of the block at stg[t * 32 + (i ^ ((t & 15) << 1))] -- an even swizzle keeps
  // the pair (2 m, 2 m + 1) adjacent and in order for the 16-byte reads, and spreads a store's 64 lanes over the banks
  cf *const stg = buf;
  auto stage_flush = [&](long long wo) {
    pair_barrier();                                              // both wavefronts' halves of the block are in
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<cf *>(a.y) + (long long)((unsigned long long)wo * HS), 0, 0x7fffffff, 0x00020000);
    const int m2 = (t & 15) * 2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = 32 * P + 4 * j + (t >> 4);
      const float4 v = *reinterpret_cast<const float4 *>(stg + r * 32 + (m2 ^ ((r & 15) << 1)));
      typedef unsigned v4u __attribute__((ext_vector_type(4)));
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), ry, rowoff[r], m2 * 8, AUX_NT);
    }
    if constexpr (!SEP) alias_barrier();                         // the next window's first swap lands where the staging was read
  };
  // The residual NCO of a channel opened `precise` (the live analyzer's inspectors are): step and start phase, read ONCE per
  // window (rot_load, at the head of the channel stage).  Round 6: they used to be read inside emit_one, from plain global
  // memory -- the compiler, which must assume that an output store may have changed the channel table, re-read n_open before
  // EVERY store and waited for it with s_waitcnt vmcnt(0): sixteen times per window a wavefront waited for its previous
  // store to be acknowledged (and for the next window's prefetch).  Held across the whole window loop they would be spilled
  // (the forward transform leaves no register), hence per window, behind a pointer the optimiser cannot see through.
  struct RotZ { uint32_t dphase, phase0; bool precise; };
  RotZ rz[NGW];
  auto rot_load = [&]() {
#pragma unroll
    for (int gl = 0; gl < NGW; ++gl) {
      const sdk::StChan *q = cdp[gl];
      asm volatile("" : "+v"(q));
      const cchan cc = (cchan)q;
      rz[gl] = RotZ{cc->dphase, (uint32_t)(a.n0 - cc->n_open), cc->precise != 0};
    }
  };
  auto emit_one = [&](auto rot, int gl, long long wo, int i, cf o) {
    if constexpr (decltype(rot)::value) {
      const uint32_t m = (uint32_t)((unsigned long long)wo * HS + i);
      [[maybe_unused]] const bool precise = rz[gl].precise;
      const uint32_t dphase = rz[gl].dphase, phase0 = rz[gl].phase0;
      // (the packed forms of sd_math.hpp: the same binary32 operations as phasor_u32 and the two fmas, sixteen instructions
      // instead of twenty-nine per output -- and sixteen outputs per wavefront and window)
      sd::v2f_ cs = sd::phasor_pk((phase0 + m) * dphase);
      if constexpr (ROTCAP != 2) { cs.x = precise ? cs.x : 1.0f; cs.y = precise ? cs.y : 0.0f; }
      o = sd::mix_rot(o, cs);
    }
    if constexpr (ROWT) {
      stg[t * 32 + (i ^ ((t & 15) << 1))] = o;
    } else if constexpr (Y32) {
      const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<cf *>(a.y) + (long long)((unsigned long long)wo * HS) * yms, 0, 0x7fffffff, 0x00020000);
#ifdef STP_UNSAFE_NO_STORES
      if (a.epoch == 0xffffffffu)                              // (never true: the stores stay in the code and move no data -- a timing experiment)
#endif
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, o), ry, yvoff[gl], (unsigned)i * yms8, AUX_NT);
    } else if (kbase + group_of(gl) * WAVE < a.nchan) {
      *(gcf *)(ybase[gl] + ((long long)((unsigned long long)wo * HS) + i) * yms) = o;
    }
  };

  // the launch's responses (one per distinct pass-band width: a.nsel <= PW_MAXSEL), each lane's table by its channel's selector
  for (int e = P * WAVE + t; e < a.nsel * S; e += 2 * WAVE) tab[(e / S) * TP + (e % S)] = reinterpret_cast<const cf *>(a.hk)[e];
  const cf *htab[NGW];
#pragma unroll
  for (int gl = 0; gl < NGW; ++gl) htab[gl] = tab + cdp[gl]->hsel * TP;
  bool publish = false;
  for (long long w = w_begin; w < w_stop; ++w) {
    cf v[32], A[32];
#ifdef STW_TSTAMP
    unsigned long long ts[16] = {0};
#endif
    TS(0);
    if (publish) {
      // the wait for this window's samples (needed here anyway) also drains the seam stores issued before them; the flag
      // goes out once BOTH wavefronts are there -- and before this run asks for its successor's (no chain of runs waiting
      // for each other)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      pair_barrier();
      if (P == 0 && t == 0) __hip_atomic_store(a.flags + slot, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      publish = false;
    }
    if (final_run && w + 1 == w_end && a.hist_out != nullptr && blockIdx.y == 0) {
      const __amdgpu_buffer_rsrc_t rhs = __builtin_amdgcn_make_buffer_rsrc(a.hist_out, 0, H * 8, 0x00020000);
#pragma unroll
      for (int q = 0; q < 32; ++q) {
        const int r = 4 * P + (q >> 3) + 8 * (q & 7);
        if (r >= WAVE / 2) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, nxt[q]), rhs, t * 8, (r - WAVE / 2) * WAVE * 8, 0);
      }
    }
    const bool last_of_run = w + 1 == w_end && !final_run;
    if (P == 0 && last_of_run) {
      // has the next run published the seam block's first half?  (bounded wait; otherwise this run transforms the seam
      // window itself.)  One wavefront asks, both act on the answer.
      const bool ok = seam_ready();
      if (t == 0) *shared_flag = ok ? 1u : 0u;
    }
    // ---- forward transform, columns ----
    // (the swap area is the first 16 KiB of the buffer: the previous window's spectrum, which both wavefronts have left)
    dft64_pair<P>(nxt, A, exf, t, [](int) {});
    TS(1);
    if (last_of_run) {
      if (*shared_flag == 0u) { self_seam = true; w_stop = w_end + 1; }
      asm volatile("" ::: "memory");
    }
    aim_next(w);
    {
      // W_4096^(t kk), kk = 8 h + l: hi[h] lo[l]; this wavefront's l = 4P + b
      cf lo[8], hi[8];
      lo[1] = opaque(wb[0]); lo[2] = opaque(wb[1]); lo[4] = opaque(wb[2]);
      lo[3] = cmul(lo[1], lo[2]); lo[5] = cmul(lo[1], lo[4]); lo[6] = cmul(lo[2], lo[4]); lo[7] = cmul(lo[3], lo[4]);
      hi[1] = opaque(wb[3]); hi[2] = opaque(wb[4]); hi[4] = opaque(wb[5]);
      hi[3] = cmul(hi[1], hi[2]); hi[5] = cmul(hi[1], hi[4]); hi[6] = cmul(hi[2], hi[4]); hi[7] = cmul(hi[3], hi[4]);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int l = 4 * P + b;
#pragma unroll
        for (int h = 0; h < 8; ++h) {
          cf &e = A[b * 8 + h];
          if (h == 0 && l == 0) continue;
          if (h == 0) e = cmul1(e, lo[l]);
          else if (l == 0) e = cmul1(e, hi[h]);
          else e = cmul3(e, hi[h], lo[l]);
        }
      }
    }
    TS(2);
    if constexpr (!SEP) alias_barrier();                       // the partner has taken its half of the swap
    {
      cf *wr = buf + t * PW_PITCH;
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int h = 0; h < 8; ++h) wr[4 * P + b + 8 * h] = A[b * 8 + h];
    }
    if (pay_now) {
#pragma unroll
      for (int q = 0; q < 16; ++q) load_one(std::true_type{}, q);
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) load_one(std::false_type{}, q);
    }
    pair_barrier();
    {
      const cf *rd = buf + t;
#pragma unroll
      for (int aa = 0; aa < 4; ++aa)
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) v[aa * 8 + n2] = rd[(4 * P + aa + 8 * n2) * PW_PITCH];
      __builtin_amdgcn_sched_barrier(0);
    }
    TS(3);
    if constexpr (!SEP) alias_barrier();                       // both have read the transposition buffer: the swap may overwrite it
    TS(4);
    // ---- forward transform, rows: A[b * 8 + k1] = X[t + 64 (4P + b + 8 k1)] ----
    dft64_pair<P>(v, A, exf, t, [&](int step) {
      if (step < 4) {
        if (pay_now) {
#pragma unroll
          for (int r = 0; r < 4; ++r) load_one(std::true_type{}, 16 + 4 * step + r);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) load_one(std::false_type{}, 16 + 4 * step + r);
        }
      }
    });
    TS(5);
    if constexpr (!SEP) alias_barrier();                       // swap consumed
    {
      cf *sp = buf + t;
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) sp[(4 * P + b + 8 * k1) * WAVE] = A[b * 8 + k1];
      if (P == 0 && t < PW_REP) buf[W + t] = A[0];
    }
    pair_barrier();
    TS(6);
    // ---- channel stage: lane = channel ----
    const bool seam = w == w_begin && w_begin > 0;
    if constexpr (ROTCAP != 0) { if (any_precise && !seam) rot_load(); }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ho + slot * HO, 0, (int)HO * 8, 0x00020000);
    // slot o: this block's sample `cur` and the next block's partner `nx`
    auto slot_done = [&](auto seam_tag, auto rot, int o, cf cur, cf nx) {
      const int i = slot_out(o);
      if constexpr (decltype(seam_tag)::value) {
        // the payload sits where the consumer's request registers expect it
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, cur), rs, t * 8, slot_row(o) * WAVE * 8, AUX_SC1);
      } else {
        emit_one(rot, slot_group(o), w, i, xfade_p(kWinP[i * WS], cur, kWinP[(i + HS) * WS], prev[o]));
      }
      prev[o] = nx;
    };
    if constexpr (LOG2S == 6) {
      // this wavefront's bins i = 4P + a + 8 n2 of the lane's one channel
      const int c0 = center[0], c1 = (center[0] - HS) & (W - 1);
      cf u[32];
      // all sixteen bin pairs first, then the response in chunks of four pairs, each requested one chunk ahead
      float4 X2[16], Hq[2][4];
      auto pair_index = [](int j) { return 4 * P + 2 * (j >> 3) + 8 * (j & 7); };   // j = (aa / 2) * 8 + n2 -> bin i (even)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int i = pair_index(j);
        X2[j] = *reinterpret_cast<const float4 *>(buf + (i < HS ? c0 + i : c1 + (i - HS)));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) Hq[0][j] = *reinterpret_cast<const float4 *>(htab[0] + pair_index(j));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (c + 1 < 4) {
#pragma unroll
          for (int j = 0; j < 4; ++j) Hq[(c + 1) & 1][j] = *reinterpret_cast<const float4 *>(htab[0] + pair_index(4 * (c + 1) + j));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int jj = 4 * c + j, aa = 2 * (jj >> 3), n2 = jj & 7;
          const float4 X = X2[jj], Hh = Hq[c & 1][j];
          cmul1x2(cf{X.x, X.y}, cf{Hh.x, Hh.y}, cf{X.z, X.w}, cf{Hh.z, Hh.w}, u[aa * 8 + n2], u[(aa + 1) * 8 + n2]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      TS(7);
      alias_barrier();                                         // both have their bins: the swap may overwrite the spectrum
      cf F[32];
      // (the swap of the inverse transform lives in the SECOND 16 KiB: the next window's first swap must not run into it)
      dft64_pair<P>(u, F, buf + 2 * PW_EX, t, [](int) {});
      TS(8);
      // F[b * 8 + k1] = F_nat[(4P + b) + 8 k1];  output i: cur = F_nat[(64 - i) & 63], next block's partner = F_nat[32 - i]
      auto chan_out = [&](auto seam_tag, auto rot) {
#pragma unroll
        for (int o = 0; o < 16; ++o) {
          const int b = o >> 2, i = slot_out(o);
          const int fc = (64 - i) & 63, fn = 32 - i;
          slot_done(seam_tag, rot, o, F[b * 8 + (fc - (4 * P + b)) / 8], F[b * 8 + (fn - (4 * P + b)) / 8]);
        }
      };
      if (seam) chan_out(std::true_type{}, std::false_type{});
      else if constexpr (ROTCAP == 2) chan_out(std::false_type{}, std::true_type{});
      else if constexpr (ROTCAP == 0) chan_out(std::false_type{}, std::false_type{});
      else if (any_precise) chan_out(std::false_type{}, std::bool_constant<ROTCAP != 0>{});
      else chan_out(std::false_type{}, std::false_type{});
      if constexpr (ROWT) { if (!seam) stage_flush(w); }
    } else {
      // this wavefront's NGW channels per lane, whole: bins, response, inverse transform on the lane's own registers
      auto chan_out = [&](auto seam_tag, auto rot) {
#pragma unroll
        for (int gl = 0; gl < NGW; ++gl) {
          const int c0 = center[gl], c1 = (center[gl] - HS) & (W - 1);
          cf u[S], F[S];
          // all the bin pairs first, then the response in chunks of four pairs, each requested one chunk ahead of its products
          constexpr int CH = LOG2S == 4 ? 2 : (S / 2 < 4 ? S / 2 : 4), NCH = (S / 2) / CH;   // (16 bins: two pairs at a time -- that instantiation has no register to spare)
          float4 X2[S / 2], Hq[2][CH];
#pragma unroll
          for (int i = 0; i < S; i += 2) X2[i / 2] = *reinterpret_cast<const float4 *>(buf + (i < HS ? c0 + i : c1 + (i - HS)));
#pragma unroll
          for (int j = 0; j < CH; ++j) Hq[0][j] = *reinterpret_cast<const float4 *>(htab[gl] + 2 * j);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            if (c + 1 < NCH) {
#pragma unroll
              for (int j = 0; j < CH; ++j) Hq[(c + 1) & 1][j] = *reinterpret_cast<const float4 *>(htab[gl] + 2 * ((c + 1) * CH + j));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < CH; ++j) {
              const float4 X = X2[c * CH + j], Hh = Hq[c & 1][j];
              cmul1x2(cf{X.x, X.y}, cf{Hh.x, Hh.y}, cf{X.z, X.w}, cf{Hh.z, Hh.w}, u[2 * (c * CH + j)], u[2 * (c * CH + j) + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          dft_reg<LOG2S>(u, F);                                // y[n] = F[(S - n) mod S]
#pragma unroll
          for (int i = 0; i < HS; ++i) slot_done(seam_tag, rot, gl * HS + i, F[(S - i) & (S - 1)], F[HS - i]);
          __builtin_amdgcn_sched_barrier(0);                   // one channel after the other: their operands would not fit side by side
        }
      };
      if (seam) chan_out(std::true_type{}, std::false_type{});
      else if constexpr (ROTCAP == 2) chan_out(std::false_type{}, std::true_type{});
      else if constexpr (ROTCAP == 0) chan_out(std::false_type{}, std::false_type{});
      else if (any_precise) chan_out(std::false_type{}, std::bool_constant<ROTCAP != 0>{});
      else chan_out(std::false_type{}, std::false_type{});
      // both wavefronts are through with the spectrum before the next window's first swap lands in it
      if constexpr (!SEP) alias_barrier();
      TS(8);
    }
    if (seam) publish = true;
    TS(9);
#ifdef STW_TSTAMP
    if (a.tstamp && t == 0 && P == 0) {
      unsigned long long *tp = a.tstamp + ((long long)slot * a.run + (w - w_begin)) * 16;
      for (int n = 0; n < 10; ++n) tp[n] = ts[n];
    }
#endif
    // (no barrier here: the next window's first swap writes the first 16 KiB, this window's last swap was read from the second)
  }
  if (publish) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (P == 0 && t == 0) __hip_atomic_store(a.flags + slot, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (final_run) {
#pragma unroll
    for (int o = 0; o < 16; ++o) {
      const int kk = kbase + group_of(slot_group(o)) * WAVE;
      if (kk < a.nchan) reinterpret_cast<cf *>(a.prev_out)[(long long)kk * HS + slot_out(o)] = prev[o];
    }
  } else if (!self_seam) {
    // the seam block: request register slot_q(o) holds the next run's unweighted sample of slot o
    if constexpr (ROWT) pair_barrier();                          // (the partner may still be reading the last block's staging)
    if constexpr (ROTCAP != 0) { if (any_precise) rot_load(); }
#pragma unroll
    for (int o = 0; o < 16; ++o) {
      const int i = slot_out(o);
      const cf v = xfade_p(kWinP[i * WS], nxt[slot_q(o)], kWinP[(i + HS) * WS], prev[o]);
      if constexpr (ROTCAP == 2) emit_one(std::true_type{}, slot_group(o), w_end, i, v);
      else if constexpr (ROTCAP == 0) emit_one(std::false_type{}, slot_group(o), w_end, i, v);
      else if (any_precise) emit_one(std::bool_constant<ROTCAP != 0>{}, slot_group(o), w_end, i, v);
      else emit_one(std::false_type{}, slot_group(o), w_end, i, v);
    }
    if constexpr (ROWT) stage_flush(w_end);
  }
#ifdef STW_TSTAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (a.tstamp && t == 0 && P == 0) { unsigned long long *tp = a.tstamp + ((long long)slot * a.run) * 16; tp[10] = t_entry; tp[11] = __builtin_amdgcn_s_memtime(); }
#endif
}

template <int LOG2S, bool Y32, bool SEP, bool ROWT = false, int ROTCAP = 1>
__global__ __launch_bounds__(2 * WAVE, 2) void stp_kernel(sdk::StArgs a)
{
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cf *buf = reinterpret_cast<cf *>(smem);
  const int t = threadIdx.x & (WAVE - 1);
  if (__builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) == 0) stp_body<0, LOG2S, Y32, SEP, ROWT, ROTCAP>(a, buf, t);
  else stp_body<1, LOG2S, Y32, SEP, ROWT, ROTCAP>(a, buf, t);
}

template <int LOG2S>
hipError_t launch_stp(const sdk::StArgs &a, hipStream_t st)
{
  constexpr int NG = WAVE >> LOG2S;
  const unsigned nruns = (unsigned)((a.nwin + a.run - 1) / a.run);
  const unsigned ny = (unsigned)((a.nchan + NG * WAVE - 1) / (NG * WAVE));
  // LDS: transposition / spectrum buffer, flag word, response tables -- and, when the launch has at most three
  // workgroups per CU anyway (768 of the 1024 window slots) and a third of the LDS holds it, the forward swaps' own 16 KiB
  const bool no_sep = sdk::tuning().st_pair_sep == 0;
  const int base = PW_TAB + a.nsel * ((1 << LOG2S) + 2);
  // (52 KB, not 160 / 3 = 53.3: the LDS is handed out in blocks, and a workgroup of 53.2 KB measured TWO per CU)
  const bool sep = (unsigned long long)nruns * ny <= 768 && !no_sep && (base + 2 * PW_EX) * 8 <= 52 * 1024;
  // per-channel rows with 32-bit offsets (the live analyzer): 64-bin channels leave through the LDS transposition (ROWT);
  // sdk::tuning().st_row_stage = 0 keeps the lane-per-row stores (A / B)
  bool rowt = false;
  if constexpr (LOG2S == 6) rowt = a.rows != nullptr && a.y32 && sdk::tuning().st_row_stage != 0;
  const int lds = (base + (sep ? 2 * PW_EX : 0)) * 8 + (rowt ? 256 : 0);
  auto go = [&](auto kern) {
    sdk::launch_timed("stp_kernel", kern, dim3(nruns, ny), dim3(2 * WAVE), (size_t)lds, st, a);
  };
  if constexpr (LOG2S == 6) {
    if (rowt) { if (sep) go(stp_kernel<LOG2S, true, true, true>); else go(stp_kernel<LOG2S, true, false, true>); return hipGetLastError(); }
  }
  // (32-bit offsets -- every bank of the harness and of the analyzer's slabs -- come in the three forms of ROTCAP; the 64-bit
  // form, a fallback for outputs beyond 2 GiB, and the per-channel rows only in the general one)
  if (a.y32 && a.any_precise == 0) { if (sep) go(stp_kernel<LOG2S, true, true, false, 0>); else go(stp_kernel<LOG2S, true, false, false, 0>); }
  else if (a.y32 && a.any_precise == 2) { if (sep) go(stp_kernel<LOG2S, true, true, false, 2>); else go(stp_kernel<LOG2S, true, false, false, 2>); }
  else if (sep) { if (a.y32) go(stp_kernel<LOG2S, true, true>); else go(stp_kernel<LOG2S, false, true>); }
  else { if (a.y32) go(stp_kernel<LOG2S, true, false>); else go(stp_kernel<LOG2S, false, false>); }
  return hipGetLastError();
}

}  // namespace

namespace sdk {

int stp_max_responses() { return PW_MAXSEL; }

// channels of 8 .. 64 bins, at most stp_max_responses() distinct responses in the launch, runs of at least two windows
hipError_t specttuner_feed_pair(int log2s, const StArgs &a, hipStream_t st)
{
  if (a.nwin <= 0 || a.nchan <= 0) return hipSuccess;
  if (a.nsel < 1 || a.nsel > PW_MAXSEL || a.run < 2) return hipErrorInvalidValue;
  switch (log2s) {
    case 3: return launch_stp<3>(a, st);
    case 4: return launch_stp<4>(a, st);
    case 5: return launch_stp<5>(a, st);
    case 6: return launch_stp<6>(a, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace sdk
