// chan_stream.hip -- the translate + FIR + decimate bank (SPEC.md section C; rows T1 / T2, BASELINE configs[1]: "1 PSK
// inspector, 255-tap LPF") for ONE or TWO channels at decimation 8 or 16, as a stream through the chip.  Same arithmetic
// contract as chan.hip -- every output is the SPEC's chain of binary32 fmas, taps ascending, then the de-rotation -- so the
// results are bit-identical to chan_fir_kernel and to the oracle (tests/test_gpu_parity.py::test_chanbank_stream_kernel_...);
// what differs is how the samples get to the lanes.
//
// With one channel the stage is HBM-bound (8 B in + 8 / D B out per input sample against 8 T / D flop) and
// chan_fir_kernel, built for many channels per staged window, leaves most of the chip idle: a workgroup stages, waits,
// computes, and every tap is re-read from LDS as a broadcast.  Here
//   * a persistent workgroup (one per CU, 8 wavefronts) walks a contiguous range of TILES of 1024 outputs;
//   * a lane computes TWO adjacent outputs m and m + 1: they share all but D of their T + D samples, so a run of eight
//     samples (four 16-byte LDS reads) feeds two independent fma chains -- tap k of output m + 1 and tap k - D of output m --
//     half the LDS reads and tap loads per output, and the chains fill each other's dependent-issue slots;
//   * LDS layout: PAIR-BLOCKS of 2 D samples, one per lane -- samples (m + 1) D - (2 D - 1) .. (m + 1) D -- at a pitch of
//     2 D + PAD samples with (2 D + PAD) / 2 odd: a wavefront's 16-byte reads sit 4 x odd dwords apart, no two lanes of a
//     ds_read_b128 group share a bank;
//   * the tile takes the whole LDS once; the NEXT tile's samples wait in registers -- D plain 16-byte loads per thread,
//     issued a few at a time in front of the first groups of runs (in one burst they hold every wavefront in the issue stage
//     for 3000 ticks at the same moment), written to the LDS between two barriers when the arithmetic ends; the history
//     pair-blocks in front of a tile are the tail of the previous one, an LDS to LDS copy;
//   * taps are wave-uniform: compact (re, im) pairs fetched by scalar loads one run ahead and used straight from SGPR pairs
//     as the packed operand of v_pk_fma_f32 (32 of them per run in ONE asm statement); outputs are stored one tile late so
//     that nobody waits for a store.
// What was built on the way and measured slower -- an LDS-DMA ring (global_load_lds_dwordx4, one output per lane, two tiles
// in the LDS), a loader wavefront, requests spread through the arithmetic -- is in profiles/r04_fir_stream_phases.txt with
// its phase clocks; tools/ubench/{dma_rate,issue,lds_write,glds_probe}.hip are the probes behind the numbers quoted here.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <vector>
#include <stdio.h>
#include "kernels.hpp"
#include "tuning.hpp"
#include "sd_math.hpp"

namespace {

using sd::c32;
typedef float v2f __attribute__((ext_vector_type(2)));
// taps are read through the constant address space: uniform loads from it are scalar loads whatever else the kernel
// stores or clobbers (a plain global pointer loses its s_load as soon as an asm statement with a memory clobber is around)
typedef const __attribute__((address_space(4))) float *ctaps;     // (re, im) interleaved

constexpr int pad_for(int D) { int p = 0; while (((D + p) & 3) != 2) ++p; return p; }     // (D + PAD) / 2 odd

struct StreamArgs {
  sdk::ChanFeedArgs a;
  const float2 *g2;      // [nchan][ntaps] (re, im), 64 spare entries on either side
  int ntiles, tiles_per_wg;
#ifdef SUAMD_INSTRUMENT
  unsigned long long *ts; // phase clocks (instrumented build + SUAMD_FIR_STREAM_TS=1)
#endif
};

// The tile loop's two structural branches ("prefetch the next tile in the run loop", "run the runs at all") must stay
// branches: once the optimiser knows both are always taken it folds the prefetch branches into the tile loop's own and the
// kernel takes 33 instead of 20 us per 4 Mi samples (round 4, measured three ways).  Rounds 4-5 kept a dead kernel
// argument for that; this is the same thing said to the compiler directly: a zero it cannot see through.
__device__ __forceinline__ int opaque_zero()
{
  int z;
  asm volatile("s_mov_b32 %0, 0" : "=s"(z));
  return z;
}

// One tap against one sample as two v_pk_fma_f32, the tap straight from its SGPR pair: op_sel picks re for both halves,
// then im against the swapped sample with the real part's product negated -- the SPEC's operation pair
//   acc = fma((re, re), (x.re, x.im), acc); acc = fma((-im, im), (x.im, x.re), acc)
// bit for bit (tools/ubench/glds_probe.hip).  Left to the compiler every tap costs four issue slots instead of two: an
// s_xor to build (-im, im) in the SGPR pair and an s_nop behind each dependent packed operation.
#define SD_TAP(T, X) "v_pk_fma_f32 %0, %" #T ", %" #X ", %0 op_sel_hi:[0,1,1]\n\t" \
                     "v_pk_fma_f32 %0, %" #T ", %" #X ", %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
__device__ __forceinline__ v2f mac1(v2f acc, float2 t, v2f x)
{
  const v2f tv = {t.x, t.y};
  asm(SD_TAP(1, 2) : "+v"(acc) : "s"(tv), "v"(x));
  return acc;
}
#undef SD_TAP


// =====================================================================================================================
// Second shape (D <= 16): lane = TWO adjacent outputs, the next tile staged through registers.
//
// What the first shape measured (profiles/r04_fir_stream_phases.txt): with one output per lane and two wavefronts per SIMD
// -- all a two-tile LDS ring leaves room for -- the run loop waits for its own LDS reads and tap loads 40 % of the time, and
// every 1 KiB LDS-DMA request holds its wavefront ~370 ticks wherever it is issued.  Here
//   * a lane computes outputs m and m + 1: they share all but D of their T + D samples, so a run of eight samples feeds
//     TWO independent fma chains (tap k of output m + 1, tap k - D of output m) -- half the LDS reads and tap loads per
//     output, twice the arithmetic between two waits, and the two chains fill each other's dependent-issue slots;
//   * 8 wavefronts x 64 lanes x 2 outputs = 1024 outputs per tile take the whole LDS ONCE (pair-blocks of 2 D samples at a
//     pitch of 2 D + PAD, (2 D + PAD) / 2 odd: conflict-free 16-byte reads as before); the NEXT tile's samples wait in
//     registers -- D plain 16-byte loads per thread, issued when the arithmetic starts, written to the LDS between two
//     barriers when it ends.  The history pair-blocks in front of a tile are the tail of the previous one: an LDS to LDS copy.
// Same operation chains per output as everywhere else: same bits.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

constexpr int hbp_for(int D, int T) { return T > D ? (T - D + 2 * D - 1) / (2 * D) : 0; }

__device__ __forceinline__ float4 load16_a8(const float2 *p)
{
  typedef float v4u __attribute__((ext_vector_type(4), aligned(8)));
  const v4u t = *reinterpret_cast<const v4u *>(p);
  return float4{t.x, t.y, t.z, t.w};
}

// eight samples against taps t1 (chain a1) and t0 (chain a0), interleaved: 32 v_pk_fma_f32, two independent chains
#define SD_TAP2(A, T, X) "v_pk_fma_f32 %" #A ", %" #T ", %" #X ", %" #A " op_sel_hi:[0,1,1]\n\t"
#define SD_TAP2B(A, T, X) "v_pk_fma_f32 %" #A ", %" #T ", %" #X ", %" #A " op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
#define SD_STEP(T1, T0, X) SD_TAP2(0, T1, X) SD_TAP2(1, T0, X) SD_TAP2B(0, T1, X) SD_TAP2B(1, T0, X)
__device__ __forceinline__ void mac8x2(v2f &a1, v2f &a0, const float2 *t1, const float2 *t0, const float4 *s)
{
  const v2f p0 = {t1[0].x, t1[0].y}, p1 = {t1[1].x, t1[1].y}, p2 = {t1[2].x, t1[2].y}, p3 = {t1[3].x, t1[3].y},
            p4 = {t1[4].x, t1[4].y}, p5 = {t1[5].x, t1[5].y}, p6 = {t1[6].x, t1[6].y}, p7 = {t1[7].x, t1[7].y};
  const v2f q0 = {t0[0].x, t0[0].y}, q1 = {t0[1].x, t0[1].y}, q2 = {t0[2].x, t0[2].y}, q3 = {t0[3].x, t0[3].y},
            q4 = {t0[4].x, t0[4].y}, q5 = {t0[5].x, t0[5].y}, q6 = {t0[6].x, t0[6].y}, q7 = {t0[7].x, t0[7].y};
  const v2f x0 = {s[0].z, s[0].w}, x1 = {s[0].x, s[0].y}, x2 = {s[1].z, s[1].w}, x3 = {s[1].x, s[1].y},
            x4 = {s[2].z, s[2].w}, x5 = {s[2].x, s[2].y}, x6 = {s[3].z, s[3].w}, x7 = {s[3].x, s[3].y};
  asm(SD_STEP(2, 10, 18) SD_STEP(3, 11, 19) SD_STEP(4, 12, 20) SD_STEP(5, 13, 21)
      SD_STEP(6, 14, 22) SD_STEP(7, 15, 23) SD_STEP(8, 16, 24) SD_STEP(9, 17, 25)
      : "+v"(a1), "+v"(a0)
      : "s"(p0), "s"(p1), "s"(p2), "s"(p3), "s"(p4), "s"(p5), "s"(p6), "s"(p7),
        "s"(q0), "s"(q1), "s"(q2), "s"(q3), "s"(q4), "s"(q5), "s"(q6), "s"(q7),
        "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));
}
#undef SD_STEP
#undef SD_TAP2
#undef SD_TAP2B

// a run of the pair kernel: eight samples, the taps of both chains
template <int NPB> struct Run2T {
  float4 s[NPB][4];      // the lane's NPB pair-blocks (NPB x 64 NW pair-blocks apart): the same taps serve all of them
  float2 t1[8], t0[8];
};
template <int D, int NPB>
__device__ __forceinline__ void load_run2(Run2T<NPB> &R, const float4 *__restrict__ top, int pb_stride16, ctaps tp, int k0)
{
#pragma unroll
  for (int b = 0; b < NPB; ++b)
#pragma unroll
    for (int j = 0; j < 4; ++j) R.s[b][j] = top[b * pb_stride16 - j];
#pragma unroll
  for (int q = 0; q < 8; ++q) R.t1[q] = float2{tp[2 * (k0 + q)], tp[2 * (k0 + q) + 1]};
#pragma unroll
  for (int q = 0; q < 8; ++q) R.t0[q] = float2{tp[2 * (k0 - D + q)], tp[2 * (k0 - D + q) + 1]};   // (front padding of the table: k0 < D)
}
template <int NPB>
__device__ __forceinline__ void touch_run2(const Run2T<NPB> &R)
{
#pragma unroll
  for (int b = 0; b < NPB; ++b) asm volatile("" :: "v"(R.s[b][0].x), "v"(R.s[b][3].w));
  asm volatile("" :: "s"(R.t1[0].x), "s"(R.t1[7].y), "s"(R.t0[0].x), "s"(R.t0[7].y));
}
// taps KK .. cnt-1 of one chain, one at a time
template <int KK>
__device__ __forceinline__ void mac_tail2(v2f &acc, const float2 (&t)[8], const float4 *s, int cnt)
{
  if constexpr (KK < 8) {
    if (KK < cnt) {
      const float4 sp = s[KK >> 1];
      acc = mac1(acc, t[KK], (KK & 1) ? v2f{sp.x, sp.y} : v2f{sp.z, sp.w});
      mac_tail2<KK + 1>(acc, t, s, cnt);
    }
  }
}

template <int D, int NW, int NCH, int NPB>
__global__ __launch_bounds__(64 * NW, NW >= 4 ? NW / 4 : 1) void chan_pair_kernel(StreamArgs sa)
{
  static_assert(D % 8 == 0 && D <= 16, "pair kernel: D in {8, 16}");
  constexpr int PB = 2 * D, PPAD = pad_for(PB), PP = PB + PPAD, PPB = PP * 8;   // pair-block: samples, pad, pitch, pitch in bytes
  constexpr int CPB = PP / 2;                                                   // 16-byte chunks per pair-block (D of them data)
  constexpr int NP = 64 * NW * NPB, TO = 2 * NP, NT = 64 * NW, GUARD = 2 * PPB;
  constexpr int NPF = NPB * D;                                  // staged 16-byte chunks per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __builtin_amdgcn_s_setprio(3);
  const sdk::ChanFeedArgs &a = sa.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.ntaps, HBP = hbp_for(D, T);
  const float2 *__restrict__ x = reinterpret_cast<const float2 *>(a.x);
  const float2 *__restrict__ hist = reinterpret_cast<const float2 *>(a.hist);
  const long long n0 = (long long)a.n0, len = a.len, hist0 = n0 - (T - 1);

  if (blockIdx.x == gridDim.x - 1) {                            // history for the next block (ping-pong buffer)
    float2 *hist_next = reinterpret_cast<float2 *>(a.hist_next);
    const int hl = T - 1;
    for (int i = tid; i < hl; i += NT) {
      const long long src = (long long)i + len;
      hist_next[i] = src < hl ? hist[src] : x[src - hl];
    }
  }
  const int t_begin = blockIdx.x * sa.tiles_per_wg;
  const int t_end = t_begin + sa.tiles_per_wg < sa.ntiles ? t_begin + sa.tiles_per_wg : sa.ntiles;
  if (t_begin >= t_end) return;
#ifdef SUAMD_INSTRUMENT
  unsigned long long tsv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SD_TS(i) do { if (sa.ts) { __builtin_amdgcn_sched_barrier(0); tsv[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define SD_TS(i) do { } while (0)
#endif
  const int shape = opaque_zero();

  // pair-block p of tile t ends at sample (m_first + t TO + 2 p + 1) D; its first new sample:
  auto tile_n1 = [&](int t) { return ((long long)a.m_first + (long long)t * TO - 1) * D + 1; };
  auto fetch = [&](long long n) -> float2 {
    if (n >= n0) return n < n0 + len ? x[n - n0] : float2{0.f, 0.f};
    return n >= hist0 ? hist[n - hist0] : float2{0.f, 0.f};
  };
  unsigned char *const data0 = smem + GUARD + (size_t)HBP * PPB;   // pair-block 0 of the tile
  // chunk c of a tile's new samples (D chunks per pair-block): thread tid stages chunks tid, tid + NT, ...: D of them
  float4 pf[NPF];
  auto prefetch = [&](int t) {
    const long long N1 = tile_n1(t);
    if (N1 >= n0 && N1 + (long long)NP * PB <= n0 + len) {        // wave-uniform: the whole tile inside x
      const float2 *src = x + (N1 - n0) + 2 * tid;
#pragma unroll
      for (int i = 0; i < NPF; ++i) pf[i] = load16_a8(src + (size_t)i * 2 * NT);
    } else {
#pragma unroll
      for (int i = 0; i < NPF; ++i) {
        const long long n = N1 + 2 * ((long long)i * NT + tid);
        const float2 s0 = fetch(n), s1 = fetch(n + 1);
        pf[i] = float4{s0.x, s0.y, s1.x, s1.y};
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
      const int c = i * NT + tid, p = c / D, pos = c - p * D;
      *reinterpret_cast<float4 *>(data0 + (size_t)p * PPB + pos * 16) = pf[i];
    }
  };

  ctaps tp[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) tp[c] = (ctaps)reinterpret_cast<const float *>(sa.g2 + (long long)(c < a.nchan ? c : a.nchan - 1) * T);
  uint32_t ph0[NCH], dph[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) { const int cc = c < a.nchan ? c : a.nchan - 1; ph0[c] = a.phase0[cc]; dph[c] = a.dphase[cc]; }
  float2 *const y = reinterpret_cast<float2 *>(a.y);
  const long long ycs = a.yv.cs, yms = a.yv.ms;
  const int nch = a.nchan;
  c32 pend[NCH][NPB][2];                                         // stored one tile late (see the first shape)
  long long pend_m = -1;
  auto flush = [&]() {
    if (pend_m >= 0) {
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int b = 0; b < NPB; ++b) {
          const long long m = pend_m + (long long)b * 2 * 64 * NW;
          if (c < nch && m < a.n_out) {
            y[(long long)c * ycs + m * yms] = float2{pend[c][b][0].re, pend[c][b][0].im};
            if (m + 1 < a.n_out) y[(long long)c * ycs + (m + 1) * yms] = float2{pend[c][b][1].re, pend[c][b][1].im};
          }
        }
    }
  };

  // ---- the first tile: history from wherever it lies (hist / x straddle: sample by sample), new samples through the
  // same registers as every other tile ----
  {
    const long long N1 = tile_n1(t_begin);
    for (int q = tid; q < HBP * D; q += NT) {
      const int p = q / D, pos = q - p * D;
      const long long n = N1 - (long long)HBP * PB + (long long)p * PB + 2 * pos;
      const float2 s0 = fetch(n), s1 = fetch(n + 1);
      *reinterpret_cast<float4 *>(smem + GUARD + (size_t)p * PPB + pos * 16) = float4{s0.x, s0.y, s1.x, s1.y};
    }
    prefetch(t_begin);
    stage();
    __syncthreads();
  }
  // Runs of eight steps; G of them read one pair-block.  Chain 1 (output m + 1) runs tap 8 r + q at step q of run r, chain 0
  // (output m) tap 8 r - D + q.  A GROUP of G runs in which both chains run whole (r >= D / 8, 8 r + 8 <= T) is straight-line
  // code: wait, four LDS reads and two tap loads for the next run, 32 fmas -- the LDS address moves by one add per group,
  // the tap pointer by one scalar add; the groups at either end of the tap range test every run.  The next tile's D
  // loads per thread go out one per run in the first D runs (statically unrolled groups: pf[] keeps constant indices):
  // issued in one burst they hold every wavefront in the issue stage for 3000 ticks at the same time.
  constexpr int G = PB / 8;
  const int RR = (shape & 2) ? 0 : (T + D + 7) / 8;                // runs
  const int NG = (RR + G - 1) / G;                               // groups
  const int g_lo = (D / 8 + G - 1) / G, g_hi = (T / 8) / G;      // groups g_lo .. g_hi - 1 are whole
  for (int t = t_begin; t < t_end; ++t) {
#ifdef SUAMD_INSTRUMENT
    tsv[3] = tsv[5] = tsv[6] = 0;
#endif
    SD_TS(0);
    flush();                                                     // the previous tile's outputs
    const bool more = t + 1 < t_end;
    const bool pre = more && !(shape & 1);
    const long long N1n = tile_n1(t + 1);
    const bool pre_whole = N1n >= n0 && N1n + (long long)NP * PB <= n0 + len;   // the next tile lies inside x
    const float2 *const psrc = x + (N1n - n0) + 2 * tid;
    // (a tile at the block's end, not entirely inside x, is fetched sample by sample, here and now; the others in the loop)
    if (pre && !pre_whole) prefetch(t + 1);
    const bool pre_loop = pre && pre_whole;
    auto prefetch_one = [&](auto idx) __attribute__((always_inline)) {                          // chunk idx NT + tid of the next tile
      constexpr int i = decltype(idx)::value;
      pf[i] = load16_a8(psrc + (size_t)i * 2 * NT);
    };
    SD_TS(1);
    const int lp = wave * 64 + lane;
    const long long m_rel = (long long)t * TO + 2 * lp;
    const uint64_t nabs = (a.m_first + (uint64_t)m_rel) * (uint64_t)D;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      v2f a1[NPB], a0[NPB];                                      // outputs m + 1 and m of each of the lane's pair-blocks
#pragma unroll
      for (int b = 0; b < NPB; ++b) { a1[b] = v2f{0.f, 0.f}; a0[b] = v2f{0.f, 0.f}; }
      Run2T<NPB> A, B;
      constexpr int PBS16 = 64 * NW * PPB / 16;                  // the lane's next pair-block, in 16-byte units
      const unsigned char *gb = data0 + (size_t)lp * PPB;        // the pair-block the current group reads
      ctaps tq = tp[c];                                          // taps of the current group's first run (chain 1)
      // run i of the group at gb: its first 16-byte pair sits at gb + (PB - 2 - 8 i) * 8
      load_run2<D, NPB>(A, reinterpret_cast<const float4 *>(gb + (PB - 2) * 8), PBS16, tq, 0);
      // one run: CUR is consumed, NXT loaded for the run behind it (I: the run's index in its group, a literal)
#define SD_RUN(CUR, NXT, I, WHOLE)                                                                                          \
      do {                                                                                                                  \
        touch_run2<NPB>(CUR);                                                                                               \
        if constexpr ((I) + 1 < G) load_run2<D, NPB>(NXT, reinterpret_cast<const float4 *>(gb + (PB - 2 - 8 * ((I) + 1)) * 8), PBS16, tq, 8 * ((I) + 1)); \
        else load_run2<D, NPB>(NXT, reinterpret_cast<const float4 *>(gb - PPB + (PB - 2) * 8), PBS16, tq, 8 * G);           \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
        if (WHOLE) { _Pragma("unroll") for (int b_ = 0; b_ < NPB; ++b_) mac8x2(a1[b_], a0[b_], CUR.t1, CUR.t0, CUR.s[b_]); } \
        else {                                                                                                              \
          const int r_ = g * G + (I), c1_ = T - 8 * r_, s0_ = 8 * r_ - D, c0_ = T - s0_;                                    \
          _Pragma("unroll") for (int b_ = 0; b_ < NPB; ++b_) {                                                              \
            if (c1_ >= 8 && s0_ >= 0 && c0_ >= 8) mac8x2(a1[b_], a0[b_], CUR.t1, CUR.t0, CUR.s[b_]);                        \
            else { mac_tail2<0>(a1[b_], CUR.t1, CUR.s[b_], c1_); if (s0_ >= 0) mac_tail2<0>(a0[b_], CUR.t0, CUR.s[b_], c0_); } \
          }                                                                                                                 \
        }                                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
      } while (0)
#define SD_GROUP(WHOLE)                                                                                                     \
      do {                                                                                                                  \
        SD_RUN(A, B, 0, WHOLE);                                                                                             \
        SD_RUN(B, A, 1, WHOLE);                                                                                             \
        if constexpr (G > 2) { SD_RUN(A, B, 2, WHOLE); SD_RUN(B, A, 3, WHOLE); }                                            \
        gb -= PPB;                                                                                                          \
        tq += 2 * 8 * G;                                                                                                    \
      } while (0)
      // the next tile's loads go out G at a time in front of the first D / G groups (constant indices into pf[])
      constexpr int NGP = (NPF + G - 1) / G;
      for (int g = 0; g < NG || (c == 0 && pre_loop && g < NGP); ++g) {
        if (c == 0 && pre_loop && g < NGP) {
          static_for<0, NGP>([&](auto gg_tag) __attribute__((always_inline)) {
            constexpr int gg = decltype(gg_tag)::value;
            if (g == gg) {
              static_for<0, G>([&](auto i_tag) __attribute__((always_inline)) {
                constexpr int i = decltype(i_tag)::value;
                if constexpr (gg * G + i < NPF) prefetch_one(std::integral_constant<int, gg * G + i>{});
              });
            }
          });
        }
        if (g < NG) {
          if (g >= g_lo && g < g_hi) SD_GROUP(true);
          else SD_GROUP(false);
        }
      }
#undef SD_GROUP
#undef SD_RUN
      touch_run2<NPB>(A);                                        // nothing in flight into registers past this point
      touch_run2<NPB>(B);
#pragma unroll
      for (int b = 0; b < NPB; ++b) {
        const uint64_t nb = nabs + (uint64_t)b * 2 * 64 * NW * D;
        float cs, sn;
        sd::phasor_u32(ph0[c] + (uint32_t)(nb * (uint64_t)dph[c]), cs, sn);
        pend[c][b][0] = sd::cmul_cs(c32{a0[b].x, a0[b].y}, cs, sn);
        sd::phasor_u32(ph0[c] + (uint32_t)((nb + (uint64_t)D) * (uint64_t)dph[c]), cs, sn);
        pend[c][b][1] = sd::cmul_cs(c32{a1[b].x, a1[b].y}, cs, sn);
      }
    }
    pend_m = m_rel;                                              // (flush tests every output against n_out)
    SD_TS(2);
    if (more) {
      // next tile: its history = this tile's last HBP pair-blocks (read them before anybody overwrites them), then the
      // staged samples
      float4 hc[(64 * CPB + NT - 1) / NT > 0 ? (64 * CPB + NT - 1) / NT : 1];   // up to 64 history pair-blocks
      const int nh = HBP * CPB;
      __syncthreads();                                           // everybody has finished reading this tile
      SD_TS(3);
#pragma unroll
      for (int i = 0; i < (int)(sizeof(hc) / sizeof(hc[0])); ++i) {
        const int q = i * NT + tid;
        hc[i] = float4{0.f, 0.f, 0.f, 0.f};
        if (q < nh) hc[i] = *reinterpret_cast<const float4 *>(data0 + (size_t)(NP - HBP) * PPB + (size_t)q * 16);
      }
      __syncthreads();
      SD_TS(5);
#pragma unroll
      for (int i = 0; i < (int)(sizeof(hc) / sizeof(hc[0])); ++i) {
        const int q = i * NT + tid;
        if (q < nh) *reinterpret_cast<float4 *>(smem + GUARD + (size_t)q * 16) = hc[i];
      }
      stage();
      SD_TS(6);
      __syncthreads();
    }
    SD_TS(4);
#ifdef SUAMD_INSTRUMENT
    if (sa.ts && tid == 0) {
      unsigned long long *tp2 = sa.ts + ((size_t)blockIdx.x * sa.tiles_per_wg + (t - t_begin)) * 8;
#pragma unroll
      for (int i = 0; i < 7; ++i) tp2[i] = tsv[i];
    }
    if (sa.ts && lane == 0 && blockIdx.x == 3 && t == t_begin + 1) {     // one tile of one workgroup: every wavefront's clocks
      unsigned long long *tp3 = sa.ts + (size_t)4096 * 64 * 8 - 64 * 8 + wave * 8;
#pragma unroll
      for (int i = 0; i < 7; ++i) tp3[i] = tsv[i];
    }
#endif
  }
#undef SD_TS
  flush();
}

template <int D, int NW, int NCH, int NPB>
hipError_t launch_pair(const StreamArgs &sa, size_t lds, unsigned grid, hipStream_t st)
{
  auto kern = chan_pair_kernel<D, NW, NCH, NPB>;
  static size_t attr_lds_dev[64] = {};
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  size_t &attr_lds = attr_lds_dev[dev_ & 63];
  if (lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_lds = lds;
  }
  sdk::launch_timed("chan_pair_kernel", kern, dim3(grid), dim3(64 * NW), lds, st, sa);
  return hipGetLastError();
}
template <int D, int NCH>
hipError_t launch_pair_nw(int nw, const StreamArgs &sa, size_t lds, unsigned grid, hipStream_t st)
{
  switch (nw) {
    case 8: return launch_pair<D, 8, NCH, 1>(sa, lds, grid, st);
    case 4: return launch_pair<D, 4, NCH, 1>(sa, lds, grid, st);
    case 2: return launch_pair<D, 2, NCH, 1>(sa, lds, grid, st);
    default: return launch_pair<D, 1, NCH, 1>(sa, lds, grid, st);
  }
}

}  // namespace

namespace sdk {

#ifdef SUAMD_INSTRUMENT
static unsigned long long *ts_buffer()
{
  static unsigned long long *d_ts = nullptr;
  if (!getenv("SUAMD_FIR_STREAM_TS")) return nullptr;
  if (!d_ts) (void)hipMalloc((void **)&d_ts, sizeof(unsigned long long) * 8 * 4096 * 64);
  return d_ts;
}
// measurement aid (SUAMD_FIR_STREAM_TS=1): phase clocks (s_memtime ticks) of wavefront 0 of every workgroup, averaged per
// phase, and every wavefront's clocks of one tile
static void ts_report(const StreamArgs &sa, unsigned grid, int nw, hipStream_t st)
{
  (void)hipStreamSynchronize(st);
  const size_t n = (size_t)grid * sa.tiles_per_wg;
  std::vector<unsigned long long> h(n * 8);
  (void)hipMemcpy(h.data(), sa.ts, h.size() * 8, hipMemcpyDeviceToHost);
  // stamps: 0 tile start, 1 loads set up, 2 loops done, 3 behind barrier 1, 5 behind barrier 2, 6 LDS writes issued, 4 behind barrier 3
  double ph[6] = {0, 0, 0, 0, 0, 0}, tile = 0; size_t cnt = 0, ct = 0;
  for (size_t i = 0; i < n; ++i) {
    const unsigned long long *p = &h[i * 8];
    if ((i % sa.tiles_per_wg) + 1 < (size_t)sa.tiles_per_wg && h[(i + 1) * 8] > p[0] && p[0]) { tile += (double)(h[(i + 1) * 8] - p[0]); ++ct; }
    if (!p[0] || p[3] < p[2] || p[5] < p[3] || p[6] < p[5] || p[4] < p[6]) continue;     // (a workgroup's last tile has no boundary)
    ph[0] += (double)(p[1] - p[0]); ph[1] += (double)(p[2] - p[1]); ph[2] += (double)(p[3] - p[2]);
    ph[3] += (double)(p[5] - p[3]); ph[4] += (double)(p[6] - p[5]); ph[5] += (double)(p[4] - p[6]); ++cnt;
  }
  const double c = cnt ? (double)cnt : 1.0;
  fprintf(stderr, "chan_pair_kernel phases (ticks, wavefront 0, mean over %zu tiles): flush + setup %.0f; run loops (+ the next tile's loads) "
                  "+ de-rotate %.0f; barrier 1 %.0f; history reads + barrier 2 %.0f; history + staging writes issued %.0f; barrier 3 %.0f; "
                  "tile to tile %.0f\n", cnt, ph[0] / c, ph[1] / c, ph[2] / c, ph[3] / c, ph[4] / c, ph[5] / c, tile / (ct ? ct : 1));
  std::vector<unsigned long long> w(64 * 8);
  (void)hipMemcpy(w.data(), sa.ts + (size_t)4096 * 64 * 8 - 64 * 8, w.size() * 8, hipMemcpyDeviceToHost);
  fprintf(stderr, "  one tile, per wavefront (ticks from wavefront 0's start): loops start / loops end / behind barrier 1 / behind barrier 2 / writes issued / behind barrier 3\n");
  for (int wv = 0; wv < nw; ++wv) {
    const unsigned long long *q = &w[wv * 8], z = w[0];
    fprintf(stderr, "    wave %d: %6lld %6lld %6lld %6lld %6lld %6lld\n", wv, (long long)(q[1] - z), (long long)(q[2] - z), (long long)(q[3] - z), (long long)(q[5] - z), (long long)(q[6] - z), (long long)(q[4] - z));
  }
}

#endif  // SUAMD_INSTRUMENT

// true if this feed is taken (launched); false if the shape is not this kernel's (the caller goes on to chan_fir_kernel):
// one or two channels, D = 8 or 16, at least 128 outputs, a tile (with its history pair-blocks) within the LDS.
bool chan_stream_feed(const ChanFeedArgs &a, const void *g2, hipStream_t st, hipError_t *err)
{
  *err = hipSuccess;
  const sdk::Tuning &tn = sdk::tuning();                      // (fir_stream = 0: off -- read per call: the tests compare the two kernels)
  if (tn.fir_stream == 0 || !g2 || a.nchan < 1 || a.nchan > 2 || a.n_out <= 0 || a.ntaps < 1) return false;
  const int D = (int)a.D;
  if (D != 8 && D != 16) return false;
  static const int ncu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
  const int PP = 2 * D + pad_for(2 * D), HBP = hbp_for(D, a.ntaps);
  auto pbytes = [&](int w) { return (size_t)2 * PP * 8 + (size_t)(HBP + 64 * w) * PP * 8; };   // w wavefronts
  // Default shape: independent tiles of 256 outputs (2 wavefronts per workgroup, four workgroups per CU, one tile each).
  // The dispatcher overlaps one workgroup's loads with another's arithmetic, and a CU that is partly taken by other
  // kernels costs a quarter tile, not a run of whole ones.  The persistent shape (SUAMD_FIR_PAIR_NW=8: one workgroup of 8
  // wavefronts per CU streaming its run of 1024-output tiles, the next tile's loads under the current tile's arithmetic)
  // is 3-12 % faster on an idle chip at 8-16 Mi samples and up to 1.6 x slower inside the analyzer pipeline, where the
  // recurrence wavefronts of earlier blocks hold three CUs for the whole step: it needs a CU's whole LDS and register
  // file, so its last three workgroups start when the first ones finish.  C = 1, D = 16, T = 255, alone / in the
  // pipeline: 4 Mi 19.9 / 31.0 us against 20.9 / 38.3; 8 Mi alone 46 against 41; 16 Mi 72 / 75 against 70 / 122
  // (profiles/r04_fir_tile_shapes.txt).
  // Measured and NOT taken: one wavefront per SIMD with two pair-blocks per lane (four fma chains on the same taps) --
  // 35 k against 20 k ticks per 1024 outputs, the compiler's SGPR spill code in its loop.
  // Round 5: a caller that HAS the chip to itself while the feed runs (suamd_chanbank_set_exclusive: the stream pipeline's
  // transform window, an offline pass) gets the persistent shape for long feeds -- 16 Mi samples inside the window: 71.1
  // against 75.0 us, 4 Mi: 25.6 against 22.1 (the independent tiles keep the short feeds).
  int nw = 2, tpw = 1;
  if (a.exclusive && a.n_out >= (1ll << 19)) { nw = 8; tpw = 0; }
  if (const int v = (int)tn.fir_pair_nw; v == 1 || v == 2 || v == 4 || v == 8) { nw = v; tpw = 0; }
  while (nw > 1 && a.n_out < 128ll * nw) nw >>= 1;
  while (nw > 1 && pbytes(nw) > 160 * 1024) nw >>= 1;
  if (pbytes(nw) > 160 * 1024 || HBP > 64 || HBP > 64 * nw || a.n_out < 128 * nw) return false;
  StreamArgs sa;
  sa.a = a; sa.g2 = reinterpret_cast<const float2 *>(g2);
  const int TO = 128 * nw;
  sa.ntiles = (int)((a.n_out + TO - 1) / TO);
  const int slots = ncu * (8 / nw);                              // resident workgroups: 8 wavefronts and the LDS of one CU
  sa.tiles_per_wg = tpw ? tpw : (sa.ntiles + slots - 1) / slots;
  if (tn.fir_pair_tpw >= 1) sa.tiles_per_wg = (int)tn.fir_pair_tpw;
  const unsigned grid = (unsigned)((sa.ntiles + sa.tiles_per_wg - 1) / sa.tiles_per_wg);
#ifdef SUAMD_INSTRUMENT
  sa.ts = (size_t)grid * sa.tiles_per_wg <= 4096 * 64 ? ts_buffer() : nullptr;
#endif
  const size_t lds = pbytes(nw);
  if (D == 8) *err = a.nchan == 1 ? launch_pair_nw<8, 1>(nw, sa, lds, grid, st) : launch_pair_nw<8, 2>(nw, sa, lds, grid, st);
  else *err = a.nchan == 1 ? launch_pair_nw<16, 1>(nw, sa, lds, grid, st) : launch_pair_nw<16, 2>(nw, sa, lds, grid, st);
#ifdef SUAMD_INSTRUMENT
  if (sa.ts && *err == hipSuccess) ts_report(sa, grid, nw, st);
#endif
  return true;
}

}  // namespace sdk
