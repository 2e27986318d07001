// specttuner.hip -- FFT channeliser for gfx950 (SPEC.md section C2; rows T2 / N2): what su_specttuner does for
// Tasks/LPFTask.cpp:52-69,83-87 and for every inspector channel of the analyzer, as ONE launch per channel size.
//
// Overlap-save filter bank: the wideband stream is cut into windows of W samples that advance by H = W/2; a window is
// transformed ONCE (W-point forward FFT) and every channel takes what it needs from that spectrum: `size` bins around
// its (even) centre bin, times its frequency response, a `size`-point inverse FFT back to the time domain at the
// decimated rate W/size, and a sin^2 / cos^2 cross-fade of the first half of this window's block with the second half of
// the previous window's.  Compulsory HBM traffic 8 B per input sample + 8 B per output sample; per channel the work no
// longer scales with the tap count but with size log size per W/2 input samples.
//
// One workgroup (256 threads) owns a run of R consecutive windows (after a warm-up window that only provides the
// "previous half" of the run's first block; the first run takes it from the carried state instead).  Per window:
//   1. forward FFT: three radix-16 passes on registers (fft_core.hpp, as psd.hip), spectrum left in LDS in natural order;
//      the samples of the NEXT window are requested from HBM right after pass 0 (software pipeline);
//   2. channel stage: the workgroup's threads split into groups of TPI = size/16 threads, one channel per group,
//      256/TPI channels side by side (grid.y covers more): gather the channel's bins from the LDS spectrum, multiply by
//      k h[i], inverse transform (conjugate, forward passes on the group's own LDS scratch, conjugate), cross-fade with
//      the previous window's half (kept in registers: in the last pass's geometry a thread holds y[i] and y[i + size/2]
//      for the same i), optional residual NCO ("precise"), store through the view.
//   Banks with more channels than fit side by side (NGL > 1, round 4): the workgroup keeps the window's spectrum in the
//   registers that hold the last forward pass's output and runs step 2 for up to NGL channel groups in turn, rewriting
//   the LDS spectrum from those registers before each -- ONE forward transform per window for NGL x 256 / TPI channels
//   instead of one per group (NGL = 2: 64 channels of 256 bins 164 -> 142 us per 4 Mi samples).  Every channel's
//   operation sequence is unchanged, so are the bits.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "kernels.hpp"
#include "tuning.hpp"
#include "fft_core.hpp"
#include "sd_math.hpp"

namespace sdk { bool st_two_turns(int log2s, int nchan); }

namespace {
using namespace fftcore;

constexpr int ST_THREADS = 256;

// compile-time loop over the turns (the per-turn arrays must be indexed by constants to stay in registers)
template <int N, int I = 0, typename F> __device__ __forceinline__ void static_for_g(F &&f)
{
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for_g<N, I + 1>(f); }
}

template <int LOG2S> struct StGeomT {
  static constexpr int S    = 1 << LOG2S;
  static constexpr int TPI  = (LOG2S == 5) ? 4 : (LOG2S < 4 ? 1 : S / 16);   // threads per inverse transform (16 points each; 8 for S = 32; S for S < 16)
  static constexpr int E    = S / TPI;
  static constexpr int CPP  = ST_THREADS / TPI;                // channels side by side in one workgroup
  static constexpr int PADS = S + S / 16 + 1;                  // a group's LDS scratch (elements)
};

// One output sample: sin^2 / cos^2 cross-fade of this window's first half with the previous window's second half, then
// the channel's residual NCO ("precise").  One function for st_kernel and st_seam_kernel: the same expression, the same
// instruction selection, the same bits.
__device__ __forceinline__ cf st_emit(float al, float be, cf cur, cf prev, const sdk::StChan &cd, unsigned long long n0, unsigned long long m)
{
  float orr = al * cur.x + be * prev.x, oi = al * cur.y + be * prev.y;
  if (cd.precise) {
    float c, s;
    sd::phasor_u32((uint32_t)(n0 - cd.n_open + m) * cd.dphase, c, s);
    const float tr = orr * c - oi * s, ti = orr * s + oi * c;
    orr = tr; oi = ti;
  }
  return cf{orr, oi};
}

// OCC: workgroups per CU the register budget is cut for (the LDS of one workgroup is ~35 KB: up to 4 fit);
// PREFETCH: request the next window's samples right after pass 0 of this one (32 VGPRs)
template <int LOG2W, int LOG2S, int OCC, bool PREFETCH, int NGL>
__global__ __launch_bounds__(ST_THREADS, OCC) void st_kernel(sdk::StArgs a)
{
  __builtin_amdgcn_s_setprio(3);   // ahead of the resident recurrence wavefronts (see chan_fir_kernel)
  using G = StGeomT<LOG2S>;
  using PW = Plan<LOG2W>;
  using PS = Plan<LOG2S>;
  constexpr int W = 1 << LOG2W, H = W / 2, S = G::S, HS = S / 2;
  constexpr int EW = W / ST_THREADS;                           // forward: points per thread
  constexpr int R0W = 1 << PW::bits(0), NB0W = EW / R0W;
  constexpr int RLW = 1 << PW::bits(PW::P - 1), NBLW = EW / RLW;
  constexpr int R0S = 1 << PS::bits(0), NB0S = G::E / R0S;
  constexpr int RLS = 1 << PS::bits(PS::P - 1), NBLS = G::E / RLS;
  static_assert(RLS >= 2, "the cross-fade pairs outputs q and q + RL/2 of one butterfly");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cf *spec = reinterpret_cast<cf *>(smem);                     // W + W/16 + 1: forward passes in place, then the spectrum,
  cf *scratch = spec;                                          // then (once every group holds its bins) the groups' scratch
  const int tid0 = threadIdx.x;
  const int grp = tid0 / G::TPI, tl0 = tid0 % G::TPI;
  // this workgroup's channel groups: (blockIdx.y * NGL + g) * CPP + grp is the channel of this thread's group in turn g
  const int gbase = blockIdx.y * NGL;
  const int ngl = min(NGL, (a.nchan - gbase * G::CPP + G::CPP - 1) / G::CPP);     // turns that have a channel at all (uniform)
  int chv[NGL];
  bool livev[NGL];
  sdk::StChan cdv[NGL];
#pragma unroll
  for (int g = 0; g < NGL; ++g) {
    chv[g] = (gbase + g) * G::CPP + grp;
    livev[g] = chv[g] < a.nchan;
    cdv[g] = a.chans[livev[g] ? chv[g] : 0];
  }
  cf *gscr = scratch + grp * G::PADS;

  TwBase tbw, tbs;
  load_tw_base<LOG2W, ST_THREADS, 0>(tbw, reinterpret_cast<const cf *>(a.tw_w), tid0);
  load_tw_base<LOG2S, G::TPI, 0>(tbs, reinterpret_cast<const cf *>(a.tw_s), tl0);

  // run of windows [w_begin, w_end); w_begin - 1 is the warm-up window (none for the first run: carried state)
  const long long w_begin = (long long)blockIdx.x * a.run, w_end = (w_begin + a.run < a.nwin) ? w_begin + a.run : a.nwin;
  // a.handoff set (round 4): no warm-up window.  A run writes the raw first half of its first block and leaves its last
  // window's second half in a.handoff[run][channel][size/2]; st_seam_kernel completes the first block of every later run
  // from its predecessor's entry.  A third of the transforms of a 3-window run were warm-up.
  const bool seam = a.handoff != nullptr;
  const long long w_first = (seam || w_begin == 0) ? w_begin : w_begin - 1;
  const cf *x = reinterpret_cast<const cf *>(a.x), *hist = reinterpret_cast<const cf *>(a.hist);
  const long long off = a.have_hist ? H : 0;                   // virtual stream = hist (H samples) ++ x

  cf prev[NGL][NBLS][RLS / 2];                                 // y_{w-1}[i + S/2] for this thread's i, per turn
#pragma unroll
  for (int g = 0; g < NGL; ++g)
#pragma unroll
    for (int b = 0; b < NBLS; ++b)
#pragma unroll
      for (int q = 0; q < RLS / 2; ++q) {
        const int i = tl0 + b * G::TPI + q * (S / RLS);
        prev[g][b][q] = (w_begin == 0 && livev[g]) ? reinterpret_cast<const cf *>(a.prev_in)[(long long)chv[g] * HS + i] : cf{0.f, 0.f};
      }

  cf nxt[EW];
  auto request = [&](long long w) {
    const cf *pa = (w == 0 && a.have_hist) ? hist : x + (w * H - off);
    const cf *pb = x + (w * H + H - off);
#pragma unroll
    for (int b = 0; b < NB0W; ++b) {
      const int j = tid0 + b * ST_THREADS;
#pragma unroll
      for (int q = 0; q < R0W; ++q) {
        const int i = j + q * (W / R0W);                         // j < W / R0W: the first half of the window is q < R0W / 2
        nxt[b * R0W + q] = q < R0W / 2 ? pa[i] : pb[i - H];
      }
    }
  };
  if (PREFETCH) request(w_first);
  for (long long w = w_first; w < w_end; ++w) {
    if (!PREFETCH) request(w);
    int tid = tid0, tl = tl0;
    asm volatile("" : "+v"(tid), "+v"(tl));                    // see psd_kernel: keeps LICM from hoisting every derived twiddle
    cf v[EW];
#pragma unroll
    for (int i = 0; i < EW; ++i) v[i] = nxt[i];
    // ---- 1. forward transform (no window function: su_specttuner's forward FFT is rectangular) ----
    fft_pass<LOG2W, ST_THREADS, 0, 1>(v, spec, tbw, tid, nullptr);
    if (PREFETCH && w + 1 < w_end) request(w + 1);
    PassRunner<LOG2W, ST_THREADS, 1, 1>::run(v, spec, tbw, tid, nullptr);
    // v[b*RL + q] = X[j + q*W/RL]; the last pass's gather was followed by a barrier: spec may take the spectrum
    static_for_g<NGL>([&](auto gc) {
    constexpr int g = decltype(gc)::value;
    if (g >= ngl) return;                                      // (uniform: every thread of the workgroup skips the turn)
    const sdk::StChan &cd = cdv[g];
    const bool live = livev[g];
    // (turn g > 0: the previous turn ended with a barrier behind its last LDS reads; the spectrum comes back from v)
#pragma unroll
    for (int b = 0; b < NBLW; ++b) {
      cf *sp = spec + lpad(tid + b * ST_THREADS);
#pragma unroll
      for (int q = 0; q < RLW; ++q) sp[q * ((W / RLW) + (W / RLW) / 16)] = v[b * RLW + q];
    }
    __syncthreads();
    // ---- 2. channel stage ----
    cf u[G::E];
    {
      const cf *hk = reinterpret_cast<const cf *>(a.hk) + (long long)cd.hsel * S;
#pragma unroll
      for (int b = 0; b < NB0S; ++b) {
        const int j = tl + b * G::TPI;
#pragma unroll
        for (int q = 0; q < R0S; ++q) {
          const int i = j + q * (S / R0S);
          const int idx = (cd.center + i + (i < HS ? 0 : W - S)) & (W - 1);
          const cf X = spec[lpad(idx)], hh = hk[i];
          // binary32 products as SPEC.md C2 states them (unfused), conjugated for the inverse transform
          const float yr = X.x * hh.x - X.y * hh.y, yi = X.x * hh.y + X.y * hh.x;
          u[b * R0S + q] = cf{yr, -yi};
        }
      }
    }
    if constexpr (PS::P > 1) __syncthreads();                // every group holds its bins: the spectrum's LDS becomes scratch
    PassRunner<LOG2S, G::TPI, 0, 1>::run(u, gscr, tbs, tl, nullptr);
    // u[b*RL + q] = conj(y[j + q*S/RL]), j = tl + b*TPI: q < RL/2 is the first half of the block, q + RL/2 its partner
    const bool emit = w >= w_begin;
    const bool raw = seam && w == w_begin && w_begin > 0;
    const float *win = a.win;
    // time-major output ([m][channel] in memory, what the one-lane-per-channel loops stream): the block goes through an
    // LDS tile [i][channel] so that a wavefront stores 64 channels of one instant = 512 contiguous bytes; otherwise
    // (channel-major rows) a thread stores its own samples (32-byte runs per 4 lanes)
    constexpr bool CAN_TILE = G::CPP >= 16;                      // tile = S/2 rows x (CPP + 4): fits the LDS region
    constexpr int TP = G::CPP + 4;                               // row pitch = 4 mod 32 elements: 16-lane store groups hit 16 bank pairs
    const bool tile = CAN_TILE && a.yv.cs == 1 && !a.rows;
    cf *ybase = a.rows ? static_cast<cf *>(const_cast<void *>(a.rows[cd.row])) : reinterpret_cast<cf *>(a.y) + (long long)cd.row * a.yv.cs;
    const long long yms = a.rows ? 1 : a.yv.ms;
    if (tile && PS::P == 1) __syncthreads();                     // (P > 1: the last pass's gather ended with a barrier) the LDS becomes the tile
#pragma unroll
    for (int b = 0; b < NBLS; ++b) {
#pragma unroll
      for (int q = 0; q < RLS / 2; ++q) {
        const int i = tl + b * G::TPI + q * (S / RLS);
        const cf cur = cf{u[b * RLS + q].x, -u[b * RLS + q].y};
        const cf nx = cf{u[b * RLS + q + RLS / 2].x, -u[b * RLS + q + RLS / 2].y};
        if (emit && live) {
          const unsigned long long m = (unsigned long long)w * HS + i;      // output index within this feed
          // (raw: the first block of a later run -- its previous half is the run before's: st_seam_kernel)
          const cf o = raw ? cur : st_emit(win[i], win[i + HS], cur, prev[g][b][q], cd, a.n0, m);
          if (tile) spec[i * TP + grp] = o;
          else ybase[(long long)m * yms] = o;
        }
        prev[g][b][q] = nx;
      }
    }
    if (tile) {
      __syncthreads();
      if (emit) {
        const int col = tid % G::CPP, r0 = tid / G::CPP;
        const int cch = (gbase + g) * G::CPP + col;
        if (cch < a.nchan) {
          const long long orow = a.chans[cch].row;
          cf *yb = reinterpret_cast<cf *>(a.y) + orow * a.yv.cs + (long long)((unsigned long long)w * HS) * a.yv.ms;
#pragma unroll
          for (int k = 0; k < (HS * G::CPP) / ST_THREADS; ++k) {
            const int r = r0 + k * (ST_THREADS / G::CPP);
            yb[(long long)r * a.yv.ms] = spec[r * TP + col];
          }
        }
      }
    }
    // the group scratch and the spectrum are rewritten by the next turn / the next window's passes (its pass 0 ends with
    // a barrier only after writing spec): order this turn's LDS reads before that
    __syncthreads();
    });
  }
  // the last window's second half: to the next run (seam) ...
  if (seam && w_end != a.nwin) {
#pragma unroll
    for (int g = 0; g < NGL; ++g)
      if (livev[g]) {
        cf *sp = reinterpret_cast<cf *>(a.handoff) + ((long long)blockIdx.x * a.nchan + chv[g]) * HS;
#pragma unroll
        for (int b = 0; b < NBLS; ++b)
#pragma unroll
          for (int q = 0; q < RLS / 2; ++q) sp[tl0 + b * G::TPI + q * (S / RLS)] = prev[g][b][q];
      }
  }
  // ... or to the next feed
  if (w_end == a.nwin) {
#pragma unroll
    for (int g = 0; g < NGL; ++g)
      if (livev[g]) {
#pragma unroll
        for (int b = 0; b < NBLS; ++b)
#pragma unroll
          for (int q = 0; q < RLS / 2; ++q) {
            const int i = tl0 + b * G::TPI + q * (S / RLS);
            reinterpret_cast<cf *>(a.prev_out)[(long long)chv[g] * HS + i] = prev[g][b][q];
          }
      }
  }
}

// completes the first block of the runs 1 .. nruns-1 of an st_kernel launch with a.handoff: raw first half (where the run left
// it) x sin^2 + the previous run's last second half x cos^2, residual NCO -- st_emit, as inside the run.  grid.y = run - 1.
__global__ __launch_bounds__(256) void st_seam_kernel(sdk::StArgs a, int log2s)
{
  const int HS = (1 << log2s) / 2;
  const long long r = (long long)blockIdx.y + 1, w = r * a.run;
  const cf *seam = reinterpret_cast<const cf *>(a.handoff) + (r - 1) * (long long)a.nchan * HS;
  const long long total = (long long)a.nchan * HS;
  // neighbouring threads along the output's unit stride: channels for time-major views, time for rows / channel-major ones
  const bool tm = !a.rows && a.yv.cs == 1 && a.yv.ms != 1;
  for (long long t = blockIdx.x * 256ll + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int c = tm ? (int)(t % a.nchan) : (int)(t / HS), i = tm ? (int)(t / a.nchan) : (int)(t - (long long)c * HS);
    const sdk::StChan cd = a.chans[c];
    const unsigned long long m = (unsigned long long)w * HS + i;
    cf *yp = a.rows ? static_cast<cf *>(const_cast<void *>(a.rows[cd.row])) + m
                    : reinterpret_cast<cf *>(a.y) + (long long)cd.row * a.yv.cs + (long long)m * a.yv.ms;
    *yp = st_emit(a.win[i], a.win[i + HS], *yp, seam[(long long)c * HS + i], cd, a.n0, m);
  }
}

template <int LOG2W, int LOG2S, int OCC, bool PREFETCH, int NGL>
hipError_t launch_st_v(const sdk::StArgs &a, hipStream_t st)
{
  using G = StGeomT<LOG2S>;
  constexpr int W = 1 << LOG2W;
  const size_t lds = sizeof(cf) * std::max((size_t)(W + W / 16 + 1), (size_t)G::CPP * G::PADS);
  auto kern = st_kernel<LOG2W, LOG2S, OCC, PREFETCH, NGL>;
  static bool attr_done_dev[64] = {};                        // a function attribute belongs to a device
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  bool &attr_done = attr_done_dev[dev_ & 63];
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  const unsigned nruns = (unsigned)((a.nwin + a.run - 1) / a.run);
  const unsigned ngroups = (unsigned)((a.nchan + G::CPP - 1) / G::CPP);
  sdk::launch_timed("st_kernel", kern, dim3(nruns, (ngroups + NGL - 1) / NGL), dim3(ST_THREADS), lds, st, a);
  if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  if (a.handoff && nruns > 1) {
    const long long total = (long long)a.nchan * (1 << (LOG2S - 1));
    sdk::launch_timed("st_seam_kernel", st_seam_kernel, dim3((unsigned)std::min<long long>(64, (total + 255) / 256), nruns - 1), dim3(256), 0, st, a, LOG2S);
  }
  return hipGetLastError();
}

template <int LOG2W, int LOG2S>
hipError_t launch_st(const sdk::StArgs &a, hipStream_t st)
{
  const int variant = (int)sdk::tuning().st_variant;
  switch (variant) {
    case 1:  return launch_st_v<LOG2W, LOG2S, 2, true, 1>(a, st);
    case 2:  return launch_st_v<LOG2W, LOG2S, 4, false, 1>(a, st);
    case 3:  return launch_st_v<LOG2W, LOG2S, 3, false, 1>(a, st);
    default: break;
  }
  // two channel groups per workgroup, in turn, on ONE forward transform (SUAMD_ST_NGL=1: a workgroup per group, rounds 2-3).
  // Measured per 4 Mi-sample block (tools/st_wide.py, same box, back to back): 64 x 256 bins 163.5 -> 142.4 us, 128 x 256
  // 333 -> 275, 64 x 128 84.1 -> 77.7, 128 x 128 148.8 -> 140.9; compiled for two workgroups per CU -- with three the
  // second turn's cross-fade state spills to scratch memory and the gain turns into a loss (246 us), and four turns per
  // workgroup lose to two even without spills (190 us): the channel stage, not the forward transform, is most of a window.
  if constexpr (LOG2S >= 7 && LOG2S <= 11) {
    if (sdk::st_two_turns(LOG2S, a.nchan)) return launch_st_v<LOG2W, LOG2S, 2, true, 2>(a, st);
  }
  // Banks of ONE channel group of 256- or 2048-bin channels run without the next window's prefetch (same arithmetic, same
  // bits): measured per size, same box, alternating (round 6; 4 Mi / 16 Mi samples): 1 x 256 bins 43.8 -> 40.4 / 131 -> 125 us
  // (C2's FFT variant), 16 x 256 51.9 -> 47.8 / 143.6 -> 139.7, 1 x 2048 50.0 -> 45.8 / 145 -> 136; 128-, 512-, 1024- and
  // 4096-bin channels keep it (1 x 128: 122.9 with, 127.1 without per 16 Mi).  Four workgroups per CU WITH the prefetch spills (59 us).
  if constexpr (LOG2S == 8 || LOG2S == 11) {
    if (a.nchan <= sdk::st_channels_per_group(LOG2S)) return launch_st_v<LOG2W, LOG2S, 3, false, 1>(a, st);
  }
  return launch_st_v<LOG2W, LOG2S, 3, true, 1>(a, st);
}

}  // namespace

namespace sdk {

int st_channels_per_group(int log2s) { return log2s == 5 ? 64 : (log2s < 4 ? 256 : (256 * 16) >> log2s); }

bool st_two_turns(int log2s, int nchan)
{
  const int ngl_env = (int)sdk::tuning().st_ngl;               // (read on every launch: A / B tests switch it in-process)
  const int cpp = st_channels_per_group(log2s);
  const int ngroups = (nchan + cpp - 1) / cpp;
  return log2s >= 7 && log2s <= 11 && (ngl_env ? ngl_env : (ngroups >= 2 ? 2 : 1)) >= 2;
}

// windows per workgroup of the workgroup kernel: the launch fills ONE round of the chip's resident workgroups (three per CU,
// two for the two-turn instantiation), a tenth of them left to whatever else is running -- a launch of 2.7 rounds runs as
// long as one of 3, and a workgroup that finds no slot waits for a whole run of another.  At least 3 (a run re-transforms
// the window before its first).  Measured per 4 Mi-sample block (tools/st_wide.py): 64 x 256 bins 144 us at 3 windows per
// workgroup (1366 workgroups on 512 slots), 129 at 4, 121 at 8 (512 workgroups); one channel: 46 at 3, 45 at 4, 56 at 6.
int st_plan_run(int log2s, int nchan, long long nwin)
{
  const int cpp = st_channels_per_group(log2s);
  const int ngroups = (nchan + cpp - 1) / cpp;
  const bool two = st_two_turns(log2s, nchan);
  const long long ny = two ? (ngroups + 1) / 2 : ngroups;
  const long long slots = (two ? 2 : 3) * 256 * 9 / 10;
  const long long run = (nwin * ny + slots - 1) / slots;
  return (int)std::min<long long>(64, std::max<long long>(3, run));
}

hipError_t specttuner_feed(int log2w, int log2s, const StArgs &a, hipStream_t st)
{
  if (log2w != 12) return hipErrorInvalidValue;
  if (a.nwin <= 0 || a.nchan <= 0) return hipSuccess;
  switch (log2s) {
    case 1:  return launch_st<12, 1>(a, st);
    case 2:  return launch_st<12, 2>(a, st);
    case 3:  return launch_st<12, 3>(a, st);
    case 4:  return launch_st<12, 4>(a, st);
    case 5:  return launch_st<12, 5>(a, st);
    case 6:  return launch_st<12, 6>(a, st);
    case 7:  return launch_st<12, 7>(a, st);
    case 8:  return launch_st<12, 8>(a, st);
    case 9:  return launch_st<12, 9>(a, st);
    case 10: return launch_st<12, 10>(a, st);
    case 11: return launch_st<12, 11>(a, st);
    case 12: return launch_st<12, 12>(a, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace sdk
