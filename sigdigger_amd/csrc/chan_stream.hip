// chan_stream.hip -- the translate + FIR + decimate bank (SPEC.md section C; rows T1 / T2, BASELINE configs[1]: "1 PSK
// inspector, 255-tap LPF") for FEW channels, as a stream through the chip.  Same arithmetic contract as chan.hip -- every
// output is the SPEC's chain of binary32 fmas, taps ascending, then the de-rotation -- so the results are bit-identical
// to chan_fir_kernel and to the oracle; what differs is how the samples get to the lanes.
//
// With one channel the stage is HBM-bound (8 B in + 8/D B out per input sample against 8 T / D flop), so the
// kernel is built around the read of the wideband block:
//   * a persistent workgroup (one per CU, NW wavefronts) walks a contiguous range of TILES of TO = 64 NW outputs;
//   * a tile's samples go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write pass, nothing for the
//     wavefront to wait for), one tile AHEAD of the arithmetic, into the other half of a two-tile ring: the read of tile
//     j + 1 is in flight for the whole of tile j's arithmetic, one barrier per tile;
//   * LDS layout: one BLOCK of D samples per output m -- samples m D - (D - 1) .. m D, the last one being the output's
//     tap 0 -- at a pitch of D + PAD samples with (D + PAD) / 2 odd: lane = output, so a wavefront's 16-byte reads (two
//     samples) sit 4 x odd dwords apart and no two lanes of a ds_read_b128 group share a bank; the DMA writes LDS
//     linearly (lane x 16 B), so the padding is made on the SOURCE side -- chunk q of the ring is block q / CB, pair q % CB,
//     and the pad chunk of every block is simply not requested;
//   * a tile carries its own history blocks in front (HB = ceil(T / D) - 1 blocks, re-read from the last-level cache:
//     3 % more requests), so every tile is self-contained and the two halves of the ring never read each other;
//   * taps are wave-uniform: compact (re, im) pairs fetched by scalar loads one run of D taps ahead and used straight
//     from SGPR pairs as the packed operand (v_pk_fma_f32 op_sel picks re for both halves, then im / -im against the
//     swapped sample); a lane reads D / 2 x 16 bytes per D taps.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include <stdio.h>
#include "kernels.hpp"
#include "sd_math.hpp"

namespace {

using sd::c32;
typedef float v2f __attribute__((ext_vector_type(2)));
// taps are read through the constant address space: uniform loads from it are scalar loads whatever else the kernel
// stores or clobbers (a plain global pointer loses its s_load as soon as an asm statement with a memory clobber is around)
typedef const __attribute__((address_space(4))) float *ctaps;     // (re, im) interleaved

constexpr int pad_for(int D) { int p = 0; while (((D + p) & 3) != 2) ++p; return p; }     // (D + PAD) / 2 odd
constexpr int guard_for(int D) { return (2 + 32 / D) * (D + pad_for(D)) * 8; }                        // bytes in front of the ring (see the run loop)

struct StreamArgs {
  sdk::ChanFeedArgs a;
  const float2 *g2;      // [nchan][ntaps] (re, im)
  int HB;                // history blocks in front of a tile
  int ntiles, tiles_per_wg;
  unsigned long long *ts; // phase clocks of wavefront 0 of every workgroup (SUAMD_FIR_STREAM_TS=1; measurement aid)
  int dbg;               // timing experiments (SUAMD_FIR_STREAM_DBG; wrong results): 1 = no requests, 2 = no arithmetic
};

// one run of RT consecutive taps of NCH channels: the samples (RT / 2 pairs, descending addresses) and the taps
template <int RT, int NCH> struct Run {
  float4 s[RT / 2];
  float2 t[NCH][RT];
};

// top: the 16-byte pair that holds the run's first tap (its later sample)
template <int RT, int NCH>
__device__ __forceinline__ void load_run(Run<RT, NCH> &R, const float4 *__restrict__ top, const ctaps (&tp)[NCH], int k0, int dbg = 0)
{
  if (!(dbg & 8)) {
#pragma unroll
  for (int j = 0; j < RT / 2; ++j) R.s[j] = top[-j];                 // descending positions = ascending taps
  }
  if (!(dbg & 4)) {
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int q = 0; q < RT; ++q) R.t[c][q] = float2{tp[c][2 * (k0 + q)], tp[c][2 * (k0 + q) + 1]};
  }
}

// see chan.hip fir_touch: an empty asm reading the run's registers makes the compiler wait for it here, i.e. BEFORE
// the next run's loads are issued (scalar loads return out of order and share lgkmcnt with LDS: every wait is 0)
template <int RT, int NCH>
__device__ __forceinline__ void touch_run(const Run<RT, NCH> &R)
{
  asm volatile("" :: "v"(R.s[0].x), "v"(R.s[RT / 2 - 1].w));
#pragma unroll
  for (int c = 0; c < NCH; ++c) asm volatile("" :: "s"(R.t[c][0].x), "s"(R.t[c][RT - 1].y));
}

// acc = fma((re, re), (x.re, x.im), acc); acc = fma((-im, im), (x.im, x.re), acc)   (SPEC.md section C, exact fmas)
__device__ __forceinline__ v2f tap_mac2(v2f acc, float2 t, v2f x)
{
  acc = __builtin_elementwise_fma(v2f{t.x, t.x}, x, acc);
  acc = __builtin_elementwise_fma(v2f{-t.y, t.y}, x.yx, acc);
  return acc;
}

// Eight taps of one channel as ONE asm statement: 16 dependent v_pk_fma_f32, the tap straight from its SGPR pair
// (op_sel picks re for both halves, then im against the swapped sample with the real part's product negated -- the
// operation pair of tap_mac2, bit for bit: tools/ubench/glds_probe.hip).  Left to the compiler every tap costs four issue
// slots instead of two: an s_xor to build (-im, im) in the SGPR pair and an s_nop behind each dependent packed operation
// (SQ counters: as many scalar as vector instructions in the loop, profiles/r04_fir_stream_sq.txt).
#define SD_TAP(T, X) "v_pk_fma_f32 %0, %" #T ", %" #X ", %0 op_sel_hi:[0,1,1]\n\t" \
                     "v_pk_fma_f32 %0, %" #T ", %" #X ", %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
__device__ __forceinline__ v2f mac8(v2f acc, const float2 *t, const float4 *s)
{
  const v2f t0 = {t[0].x, t[0].y}, t1 = {t[1].x, t[1].y}, t2 = {t[2].x, t[2].y}, t3 = {t[3].x, t[3].y},
            t4 = {t[4].x, t[4].y}, t5 = {t[5].x, t[5].y}, t6 = {t[6].x, t[6].y}, t7 = {t[7].x, t[7].y};
  // tap 2j multiplies the LATER sample of pair j (.zw), tap 2j + 1 the earlier one (.xy)
  const v2f x0 = {s[0].z, s[0].w}, x1 = {s[0].x, s[0].y}, x2 = {s[1].z, s[1].w}, x3 = {s[1].x, s[1].y},
            x4 = {s[2].z, s[2].w}, x5 = {s[2].x, s[2].y}, x6 = {s[3].z, s[3].w}, x7 = {s[3].x, s[3].y};
  asm(SD_TAP(1, 9) SD_TAP(2, 10) SD_TAP(3, 11) SD_TAP(4, 12) SD_TAP(5, 13) SD_TAP(6, 14) SD_TAP(7, 15) SD_TAP(8, 16)
      : "+v"(acc)
      : "s"(t0), "s"(t1), "s"(t2), "s"(t3), "s"(t4), "s"(t5), "s"(t6), "s"(t7),
        "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));
  return acc;
}
__device__ __forceinline__ v2f mac1(v2f acc, float2 t, v2f x)
{
  const v2f tv = {t.x, t.y};
  asm(SD_TAP(1, 2) : "+v"(acc) : "s"(tv), "v"(x));
  return acc;
}
#undef SD_TAP

template <int RT, int NCH>
__device__ __forceinline__ void mac_run(v2f (&acc)[NCH], const Run<RT, NCH> &R)
{
  if constexpr (RT % 8 == 0) {
#pragma unroll
    for (int g = 0; g < RT / 8; ++g)
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc[c] = mac8(acc[c], &R.t[c][8 * g], &R.s[4 * g]);
  } else {
#pragma unroll
    for (int j = 0; j < RT / 2; ++j) {
      const v2f x0 = {R.s[j].z, R.s[j].w}, x1 = {R.s[j].x, R.s[j].y};   // tap 2j: the later sample of the pair
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc[c] = tap_mac2(acc[c], R.t[c][2 * j], x0);
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc[c] = tap_mac2(acc[c], R.t[c][2 * j + 1], x1);
    }
  }
}

// taps KK .. rem-1 of a run, one at a time behind wave-uniform tests (register indices are compile-time constants)
template <int KK, int RT, int NCH>
__device__ __forceinline__ void mac_tail(v2f (&acc)[NCH], const Run<RT, NCH> &R, int rem)
{
  if constexpr (KK < RT - 1) {
    if (KK < rem) {
      const float4 sp = R.s[KK >> 1];
      const v2f xv = (KK & 1) ? v2f{sp.x, sp.y} : v2f{sp.z, sp.w};
#pragma unroll
      for (int c = 0; c < NCH; ++c) acc[c] = mac1(acc[c], R.t[c][KK], xv);
      mac_tail<KK + 1, RT, NCH>(acc, R, rem);
    }
  }
}

// a whole run or its first cnt taps
template <int RT, int NCH>
__device__ __forceinline__ void mac_n(v2f (&acc)[NCH], const Run<RT, NCH> &R, int cnt)
{
  if (cnt >= RT) mac_run<RT, NCH>(acc, R);
  else mac_tail<0, RT, NCH>(acc, R, cnt);
}

// Requests of one tile, one at a time: wavefront w takes the 64-chunk groups w, w + NW, ... of the (linear) LDS image.
// Chunk q is pair q % CB of block q / CB, the pad chunks (q % CB >= D / 2) are not requested; a lane keeps its chunk's
// pair index and 32-bit source offset incrementally (a step of 64 NW chunks = QB blocks and QR pairs): a dozen
// instructions per request, none of them a vector load.
template <int D, int NW> struct Requests {
  static constexpr int PAD = pad_for(D), CB = (D + PAD) / 2;
  static constexpr int QB = (64 * NW) / CB, QR = 64 * NW - QB * CB, STEP = QB * D * 8 + QR * 16;
  const char *src;       // the tile's first sample
  unsigned lds;          // LDS byte address of this wavefront's next group
  int lo, hi;            // source offsets of the chunks entirely inside x
  int q0, nchunk;        // next group's first chunk
  int r, goff;           // per lane

  __device__ __forceinline__ void begin(const char *src_, int lo_, int hi_, unsigned lds_half, int nchunk_, int wave, int lane)
  {
    src = src_; lo = lo_; hi = hi_; nchunk = nchunk_;
    q0 = wave * 64;
    lds = lds_half + (unsigned)q0 * 16u;
    const int q = q0 + lane, b = q / CB;
    r = q - b * CB;
    goff = b * D * 8 + r * 16;
  }
  __device__ __forceinline__ bool pending() const { return q0 < nchunk; }
  __device__ __forceinline__ void step(int lane)
  {
    const bool ins = r < D / 2 && q0 + lane < nchunk && goff >= lo && goff <= hi;
    if (ins)
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(goff), "s"(src), "s"(lds) : "memory", "m0");
    q0 += 64 * NW;
    lds += 1024u * NW;
    r += QR;
    goff += STEP;
    if (r >= CB) { r -= CB; goff -= PAD * 8; }
  }
};

// NW wavefronts, lane = output.  Every wavefront requests its share of the NEXT tile while it works on this one, one
// request per run of taps: issued in one burst the 76 requests of a tile keep their wavefronts in the issue stage for
// 3600 of the tile's 15000 ticks (the CU's address pipeline takes a 1 KiB request every ~45 ticks) and the barrier behind
// them adds the skew; a single loader wavefront cannot issue them fast enough at all (130-340 ticks per request from one
// wavefront: tools/ubench/dma_rate.hip; profiles/r04_fir_stream_phases.txt has the phase clocks of the three shapes).
template <int D, int NW, int NCH>
__global__ __launch_bounds__(64 * NW, NW >= 4 ? NW / 4 : 1) void chan_stream_kernel(StreamArgs sa)
{
  constexpr int PAD = pad_for(D), BP = D + PAD;                 // block pitch (samples)
  constexpr int CB = BP / 2;                                    // 16-byte chunks per block, D / 2 of them data
  constexpr int TO = 64 * NW, NT = 64 * NW, GUARD = guard_for(D);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __builtin_amdgcn_s_setprio(3);                                // ahead of resident recurrence wavefronts (see chan.hip)
  const sdk::ChanFeedArgs &a = sa.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int T = a.ntaps, HB = sa.HB;
  const int nblk = HB + TO;                                     // blocks per ring half
  const int nchunk = nblk * CB;
  const unsigned half_bytes = (unsigned)nblk * BP * 8u;
  const float2 *__restrict__ x = reinterpret_cast<const float2 *>(a.x);
  const float2 *__restrict__ hist = reinterpret_cast<const float2 *>(a.hist);
  const long long n0 = (long long)a.n0, len = a.len, hist0 = n0 - (T - 1);

  // ---- carry: history for the next block = last T-1 samples of [hist ; x] (ping-pong buffer) ----
  if (blockIdx.x == gridDim.x - 1) {
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
  unsigned long long tsv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define SD_TS(i) do { if (sa.ts) { __builtin_amdgcn_sched_barrier(0); tsv[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)

  // absolute index of the first sample of tile t's ring half: block HB + o ends at sample (m_first + t TO + o) D
  auto tile_n0 = [&](int t) { return ((long long)a.m_first + (long long)t * TO - HB) * D - (D - 1); };
  auto fetch = [&](long long n) -> float2 {
    if (n >= n0) return n < n0 + len ? x[n - n0] : float2{0.f, 0.f};
    return n >= hist0 ? hist[n - hist0] : float2{0.f, 0.f};
  };
  Requests<D, NW> rq;
  // start tile t's requests into ring half h.  Chunks that are not entirely inside x (the feed's first history blocks
  // straddle hist / x; the last tile runs past the block's end) go through registers, here and now: a vector load inside
  // a loop that also requests makes the compiler wait vmcnt(0) on its back edge.
  auto request_begin = [&](int t, int h) {
    const long long N0 = tile_n0(t);
    unsigned char *dst = smem + GUARD + h * half_bytes;
    const long long lo64 = (n0 - N0) * 8, hi64 = (n0 + len - 2 - N0) * 8;
    const int lo = lo64 < 0 ? 0 : (lo64 > 0x7fffffff ? 0x7fffffff : (int)lo64);
    const int hi = hi64 < 0 ? -1 : (hi64 > 0x7fffffff ? 0x7fffffff : (int)hi64);
    rq.begin(reinterpret_cast<const char *>(x) + (N0 - n0) * 8, lo, hi, (unsigned)(uintptr_t)dst, nchunk, wave, lane);
    if (!(lo == 0 && hi >= (nblk * D - 2) * 8)) {
      for (int q = tid; q < nchunk; q += NT) {
        const int bb = q / CB, rr = q - bb * CB;
        const long long n = N0 + (long long)bb * D + 2 * rr;
        if (rr < D / 2 && !(n >= n0 && n + 1 < n0 + len)) {
          const float2 s0 = fetch(n), s1 = fetch(n + 1);
          *reinterpret_cast<float4 *>(dst + (size_t)q * 16) = float4{s0.x, s0.y, s1.x, s1.y};
        }
      }
    }
  };

  ctaps tp[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) tp[c] = (ctaps)reinterpret_cast<const float *>(sa.g2 + (long long)(c < a.nchan ? c : a.nchan - 1) * T);
  // de-rotation parameters once (inside the tile loop every load sits behind the asm statements' memory clobbers)
  uint32_t ph0[NCH], dph[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) { const int cc = c < a.nchan ? c : a.nchan - 1; ph0[c] = a.phase0[cc]; dph[c] = a.dphase[cc]; }
  float2 *const y = reinterpret_cast<float2 *>(a.y);
  const long long ycs = a.yv.cs, yms = a.yv.ms;
  const int nch = a.nchan;
  // a tile's outputs are stored one tile LATE, behind the next tile's barrier: the store's latency is nobody's wait
  c32 pend[NCH];
  long long pend_m = -1;
  auto flush = [&]() {
    if (pend_m >= 0) {
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        if (c < nch) y[(long long)c * ycs + pend_m * yms] = float2{pend[c].re, pend[c].im};
    }
  };
  // runs of RT taps, one ahead of the arithmetic (ping-pong, no register copies); a block of D samples is D / RT runs.
  // The loop is branch-free as far as loads go: every step loads the NEXT run whether or not it exists -- up to 2 RT taps
  // past the table's end (the rows of g2 are padded) and a block or two below the tile's first history block (the ring
  // sits GUARD bytes into the LDS) -- because a conditional load makes the compiler merge the two arithmetic phases
  // behind both waits (seen in the ISA: no overlap left).  The last run may be partial: its taps run one at a time.
  constexpr int RT = D >= 16 && NCH == 1 ? 16 : (D < 8 ? D : 8);
  const int R = (sa.dbg & 2) ? 0 : (T + RT - 1) / RT;

  request_begin(t_begin, 0);
  while (rq.pending()) rq.step(lane);                            // the first tile: nothing to hide it behind
  for (int t = t_begin; t < t_end; ++t) {
    const int h = (t - t_begin) & 1;
    SD_TS(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wavefront's requests of tile t have landed ...
    __syncthreads();                                             // ... and everybody's; the other half is free
    SD_TS(1);
    flush();                                                     // the previous tile's outputs
    const bool more = t + 1 < t_end && !(sa.dbg & 1);
    if (more) request_begin(t + 1, h ^ 1);
    SD_TS(2);
    const int o = wave * 64 + lane;
    const unsigned char *lb = smem + GUARD + h * half_bytes + (size_t)(HB + o) * BP * 8;       // this output's block
    v2f acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc[c] = v2f{0.f, 0.f};
    Run<RT, NCH> A, B;
    auto top_of = [&](int k0) {                                  // the pair holding tap k0: block k0 / D back, position D-1 - k0 % D
      const int b = k0 / D, kk = k0 - b * D;
      return reinterpret_cast<const float4 *>(lb - (size_t)b * BP * 8 + (size_t)(D - 2 - kk) * 8);
    };
    load_run<RT, NCH>(A, top_of(0), tp, 0);
    for (int r = 0; r < R; r += 2) {
      touch_run<RT, NCH>(A);
      // (the request goes out while nothing of this wavefront is in flight to the LDS: behind the run's reads it waited
      // for them -- 380 ticks per request; here it is a dozen instructions)
      if (more && rq.pending()) {
        if (sa.ts) {
          __builtin_amdgcn_sched_barrier(0);
          const unsigned long long q0 = __builtin_amdgcn_s_memtime();
          __builtin_amdgcn_sched_barrier(0);
          rq.step(lane);
          __builtin_amdgcn_sched_barrier(0);
          tsv[5] += __builtin_amdgcn_s_memtime() - q0;
          __builtin_amdgcn_sched_barrier(0);
        } else rq.step(lane);
      }
      __builtin_amdgcn_sched_barrier(0);
      load_run<RT, NCH>(B, top_of((r + 1) * RT), tp, (r + 1) * RT, sa.dbg);
      __builtin_amdgcn_sched_barrier(0);
      mac_n<RT, NCH>(acc, A, T - r * RT);
      __builtin_amdgcn_sched_barrier(0);
      touch_run<RT, NCH>(B);
      if (more && rq.pending()) rq.step(lane);
      __builtin_amdgcn_sched_barrier(0);
      load_run<RT, NCH>(A, top_of((r + 2) * RT), tp, (r + 2) * RT, sa.dbg);
      __builtin_amdgcn_sched_barrier(0);
      mac_n<RT, NCH>(acc, B, T - (r + 1) * RT);                  // <= 0 behind the last run: nothing
      __builtin_amdgcn_sched_barrier(0);
    }
    touch_run<RT, NCH>(A);                                       // nothing may be in flight into registers past this point
    if (more) while (rq.pending()) rq.step(lane);                // (few runs, many requests: the rest)
    SD_TS(3);
    // ---- de-rotate to baseband; the store follows behind the next barrier ----
    const long long m_rel = (long long)t * TO + o;
    pend_m = m_rel < a.n_out ? m_rel : -1;
    const uint64_t n = (a.m_first + (uint64_t)m_rel) * (uint64_t)D;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float cs, sn;
      sd::phasor_u32(ph0[c] + (uint32_t)(n * (uint64_t)dph[c]), cs, sn);
      pend[c] = sd::cmul_cs(c32{acc[c].x, acc[c].y}, cs, sn);
    }
    SD_TS(4);
    if (sa.ts && tid == 0) {
      unsigned long long *tp2 = sa.ts + ((size_t)blockIdx.x * sa.tiles_per_wg + (t - t_begin)) * 8;
      for (int i = 0; i < 6; ++i) tp2[i] = tsv[i];
      tsv[5] = 0;
    }
  }
#undef SD_TS
  flush();
}

template <int D, int NW, int NCH>
hipError_t launch_stream(const StreamArgs &sa, size_t lds, unsigned grid, hipStream_t st)
{
  auto kern = chan_stream_kernel<D, NW, NCH>;
  static size_t attr_lds_dev[64] = {};                       // a function attribute belongs to a device
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  size_t &attr_lds = attr_lds_dev[dev_ & 63];
  if (lds > attr_lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_lds = lds;
  }
  sdk::launch_timed("chan_stream_kernel", kern, dim3(grid), dim3(64 * NW), lds, st, sa);
  return hipGetLastError();
}

template <int D, int NCH>
hipError_t launch_stream_nw(int nw, const StreamArgs &sa, size_t lds, unsigned grid, hipStream_t st)
{
  switch (nw) {
    case 8: return launch_stream<D, 8, NCH>(sa, lds, grid, st);
    case 4: return launch_stream<D, 4, NCH>(sa, lds, grid, st);
    case 2: return launch_stream<D, 2, NCH>(sa, lds, grid, st);
    default: return launch_stream<D, 1, NCH>(sa, lds, grid, st);
  }
}

template <int D>
hipError_t launch_stream_d(int nw, int nch, const StreamArgs &sa, size_t lds, unsigned grid, hipStream_t st)
{
  if (nch == 1) return launch_stream_nw<D, 1>(nw, sa, lds, grid, st);
  return launch_stream_nw<D, 2>(nw, sa, lds, grid, st);
}

}  // namespace

namespace sdk {

// true if this feed is taken (launched); false if the shape is not the stream kernel's (the caller falls back to
// chan_fir_kernel).  Shapes: 1 or 2 channels, D in {8, 16, 32, 64}, a ring of two tiles within the LDS.
bool chan_stream_feed(const ChanFeedArgs &a, const void *g2, hipStream_t st, hipError_t *err)
{
  *err = hipSuccess;
  const char *mode_env = getenv("SUAMD_FIR_STREAM");          // 0: off (read per call: the tests compare the two kernels)
  const int mode = mode_env ? atoi(mode_env) : 1;
  if (!mode || !g2 || a.nchan < 1 || a.nchan > 2 || a.n_out <= 0 || a.ntaps < 1) return false;
  const int D = (int)a.D;
  if (D != 8 && D != 16 && D != 32 && D != 64) return false;
  const int PAD = pad_for(D), BP = D + PAD;
  const int HB = (a.ntaps + D - 1) / D - 1;
  // as many wavefronts (64 outputs each) as a ring of two tiles allows
  int nw = 8;
  auto bytes = [&](int w) { return (size_t)guard_for(D) + 2 * (size_t)(HB + 64 * w) * BP * 8; };
  while (nw > 1 && bytes(nw) > 160 * 1024) nw >>= 1;
  if (bytes(nw) > 160 * 1024) return false;
  if (a.n_out < 64 * nw) return false;                        // tiny feeds: the tiled kernel has less to set up
  StreamArgs sa;
  sa.a = a; sa.g2 = reinterpret_cast<const float2 *>(g2); sa.HB = HB;
  { const char *e = getenv("SUAMD_FIR_STREAM_DBG"); sa.dbg = e ? atoi(e) : 0; }
  const int TO = 64 * nw;
  sa.ntiles = (int)((a.n_out + TO - 1) / TO);
  static const int ncu = [] { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
  sa.tiles_per_wg = (sa.ntiles + ncu - 1) / ncu;
  const unsigned grid = (unsigned)((sa.ntiles + sa.tiles_per_wg - 1) / sa.tiles_per_wg);
  const size_t lds = bytes(nw);
  static unsigned long long *d_ts = nullptr;
  const bool want_ts = getenv("SUAMD_FIR_STREAM_TS") != nullptr;
  if (want_ts && !d_ts) (void)hipMalloc((void **)&d_ts, sizeof(unsigned long long) * 8 * 4096 * 64);
  sa.ts = want_ts && (size_t)grid * sa.tiles_per_wg <= 4096 * 64 ? d_ts : nullptr;
  switch (D) {
    case 8:  *err = launch_stream_d<8>(nw, a.nchan, sa, lds, grid, st); break;
    case 16: *err = launch_stream_d<16>(nw, a.nchan, sa, lds, grid, st); break;
    case 32: *err = launch_stream_d<32>(nw, a.nchan, sa, lds, grid, st); break;
    default: *err = launch_stream_d<64>(nw, a.nchan, sa, lds, grid, st); break;
  }
  if (sa.ts && *err == hipSuccess) {
    // measurement aid: phase clocks (s_memtime ticks) of wavefront 0 of every workgroup, averaged per phase
    (void)hipStreamSynchronize(st);
    const size_t n = (size_t)grid * sa.tiles_per_wg;
    std::vector<unsigned long long> h(n * 8);
    (void)hipMemcpy(h.data(), d_ts, h.size() * 8, hipMemcpyDeviceToHost);
    static const char *names[6] = {"wait + barrier", "flush + request setup", "run loop (+ requests)", "de-rotate", "", ""};
    double acc[6] = {0, 0, 0, 0, 0, 0}, tile = 0; size_t cnt = 0, ct = 0;
    for (size_t i = 0; i < n; ++i) {
      const unsigned long long *p = &h[i * 8];
      if (!p[0] || p[4] < p[0]) continue;
      for (int k = 0; k < 4; ++k) acc[k] += (double)(p[k + 1] - p[k]);
      ++cnt;
      if ((i % sa.tiles_per_wg) + 1 < (size_t)sa.tiles_per_wg && h[(i + 1) * 8] > p[0]) { tile += (double)(h[(i + 1) * 8] - p[0]); ++ct; }
    }
    fprintf(stderr, "chan_stream phases (ticks, mean over %zu tiles):", cnt);
    for (int k = 0; k < 4; ++k) fprintf(stderr, " %s %.0f;", names[k], acc[k] / (cnt ? cnt : 1));
    { double rqt = 0; for (size_t i = 0; i < n; ++i) rqt += (double)h[i * 8 + 5]; fprintf(stderr, " [half of the requests, timed: %.0f per tile]", rqt / (cnt ? cnt : 1)); }
    fprintf(stderr, " tile to tile %.0f\n", tile / (ct ? ct : 1));
  }
  return true;
}

}  // namespace sdk
