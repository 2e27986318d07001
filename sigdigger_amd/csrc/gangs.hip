// gangs.hip -- the recurrences of MANY 1-channel banks side by side, one lane each (the live analyzer's inspectors: their
// own parameters, state, rows and lengths): gather / scatter between rows and packed time-major slabs, the gang kernels on
// those slabs, and the gangs on rows that ARE columns of a slab already (kernels.hpp GangSlab), with the AGC's feed-forward
// steps in tiles of 64 columns.  Device helpers shared with the banks: loops_dev.hpp.  SPEC.md sections E-H.
// Compiled with -ffp-contract=off: the arithmetic is the SPEC's fixed binary32 sequence and matches the CPU oracle bit for bit.
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.hpp"
#include "tuning.hpp"
#include "sd_math.hpp"
#include "loops_dev.hpp"

namespace {

// ---------------------------------------------------------------------------------------
// Gangs: lane j runs item j -- a 1-channel bank with its own parameters, state, rows and length.
// Rows are streamed 16 steps ahead per lane (each lane its own pointer; a chunk is 128 contiguous
// bytes per lane); steps beyond a lane's length are skipped by predication.
// Every lane has its own row somewhere in memory.  Reading it lane by lane would touch 64 different
// cache lines per load instruction (measured: the gang ran 4.6x slower per sample than a bank).  Instead the
// wave moves data in tiles of GT samples x 64 items through LDS: row k's tile is loaded by all 64 lanes
// together (one coalesced GT*sizeof(T)-byte access), transposed through LDS (pitch 65: conflict-free both
// ways), and lane k then reads its own samples from LDS; outputs go back the same way.  The loads of tile
// t+1 are issued before the steps of tile t run (they wait in registers), so their latency is hidden.
constexpr int GT = 32;          // samples per tile and row
constexpr int GP = 65;          // LDS pitch (items per sample row + 1)

template <typename T> struct GangLds { T in[GT * GP]; T out[GT * GP]; const T *xp[64]; T *yp[64]; long long ln[64]; };

// HAS_OUT: step() returns a T that is written to row y; otherwise step() returns nothing
template <bool HAS_OUT, typename T, typename F>
__device__ __forceinline__ void gang_stream(GangLds<T> &lds, const T *__restrict__ x, T *__restrict__ y, long long len,
                                            F step)
{
  const int lane = threadIdx.x;
  const long long maxlen = wave_max(len);
  if (maxlen <= 0) return;
  const long long minlen = -wave_max(len > 0 ? -len : -(1ll << 62));      // shortest non-empty row
  // a tile is GT = 32 samples: the 64 lanes cover two rows per access (lane >> 5 picks the row of the pair).
  // The rows a lane touches (2k + half, k < 32) never change: their pointers and lengths are fetched once,
  // through LDS, into registers -- a single wavefront owns the SIMD's whole register file.
  const int half = lane >> 5, sl = lane & 31;
  lds.xp[lane] = x; lds.yp[lane] = y; lds.ln[lane] = len;     // (single wave: no barrier needed)
  const T *rx[32];
  T *ry[32];
  int rl[32];                                                // rows are shorter than 2^31 samples
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    rx[k] = lds.xp[2 * k + half];
    if constexpr (HAS_OUT) ry[k] = lds.yp[2 * k + half];
    rl[k] = (int)lds.ln[2 * k + half];
  }
  T pre[32];
  auto request = [&](long long s0) {                        // rows 2k + half, samples s0 + sl
#pragma unroll
    for (int k = 0; k < 32; ++k) pre[k] = (s0 + sl < rl[k]) ? rx[k][s0 + sl] : T{};
  };
  request(0);
  for (long long s0 = 0; s0 < maxlen; s0 += GT) {
#pragma unroll
    for (int k = 0; k < 32; ++k) lds.in[sl * GP + 2 * k + half] = pre[k];   // sample (s0 + sl) of row 2k + half
    if (s0 + GT < maxlen) request(s0 + GT);
    // steps in groups of CHUNK: the group's inputs are pulled out of LDS first (independent reads), so the
    // recurrence itself never waits on LDS; groups that lie inside every row's length (the common case)
    // run without per-step predication
    for (int g = 0; g < GT; g += CHUNK) {
      T vin[CHUNK], vout[CHUNK];
#pragma unroll
      for (int j = 0; j < CHUNK; ++j) vin[j] = lds.in[(g + j) * GP + lane];
      if (s0 + g + CHUNK <= minlen) {
        if (len > 0) {                                        // empty rows (and lanes without an item) sit the group out
#pragma unroll
          for (int j = 0; j < CHUNK; ++j) {
            if constexpr (HAS_OUT) vout[j] = step(s0 + g + j, vin[j]);
            else step(s0 + g + j, vin[j]);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < CHUNK; ++j) {
          if (s0 + g + j < len) {
            if constexpr (HAS_OUT) vout[j] = step(s0 + g + j, vin[j]);
            else step(s0 + g + j, vin[j]);
          }
        }
      }
      if constexpr (HAS_OUT) {
#pragma unroll
        for (int j = 0; j < CHUNK; ++j) lds.out[(g + j) * GP + lane] = vout[j];
      }
    }
    if constexpr (HAS_OUT) {
#pragma unroll
      for (int k = 0; k < 32; ++k)
        if (s0 + sl < rl[k]) ry[k][s0 + sl] = lds.out[sl * GP + 2 * k + half];
    }
  }
}

// The transposition above costs the lone wavefront ~90 ns per sample (address arithmetic, predicates, LDS round trips:
// a gang ran 2x slower per sample than a bank, tools/gang_bench.py).  It is throughput work, so it moves out of the
// recurrence: a parallel kernel gathers the 64 rows of a group into a time-major slab tm[m][lane] (LDS-transposed
// tiles, coalesced both ways), the recurrence streams the slab exactly like a 64-channel bank (scalar base +
// immediate offsets, one chunk prefetched ahead) -- in place -- and a second parallel kernel scatters the results to
// the rows.  The slab has whole 64-sample tiles plus one tile of slack for the prefetch.
// (tm / tmo: where the lanes read and write -- the same packed slab for the gather / scatter form; lo / loo: the lane's byte
// offset there.  The slab form of a 64-column pitch runs this loop too, on the producer's slab and with the items' own columns.)
// EVERY lane has work: a lane without an item, or with an empty row, is given a copy of another lane's item by its caller
// (gang_lane) -- it computes and stores exactly what that lane does, and its state is not written back.  That keeps the
// steady-state loop ONE straight path.  With the `if (len > 0)` it used to have around the steps, two paths with different
// numbers of stores in flight met at the back edge, and the compiler's wait there was s_waitcnt vmcnt(0): every chunk waited
// for its own last store to complete (the bank kernels, one path, get counted waits) -- 86 against 77 ns per Costas sample.
// PEEL: the first chunk runs in front of the loop.  The loop's waits for its prefetched chunk sit at its top and have to hold
// for both ways in: from the back edge the chunk's loads are followed by the previous chunk's stores (a counted wait lets
// those stay in flight), from a preheader that has only issued loads nothing follows them, and the merged requirement is
// s_waitcnt vmcnt(0).  With a peeled copy in front both ways in look alike.  Measured per 64 x 65536 samples, both ways, for
// every user (what the scheduler makes of the longer code differs): Costas gangs 5.43 -> 5.18 ms (slab 5.36 -> 5.20) -- taken;
// AGC level gangs 3.46 -> 3.58, the banks' stream_row 5.04 -> 5.19 (Costas) / 3.35 -> 3.63 (AGC), the run-time-pitch loop
// 5.38 -> 5.76 -- not taken.
template <bool HAS_OUT, bool PEEL, typename T, typename F>
__device__ __forceinline__ void gang_stream_tm(const T *tm, T *tmo, const uint32_t lo, const uint32_t loo, long long len, F step)
{
  const long long maxlen = uniform64(wave_max(len));
  if (maxlen <= 0) return;
  const long long minlen = uniform64(-wave_max(-len));
  T cur[CHUNK], nxt[CHUNK];
#pragma unroll
  for (int j = 0; j < CHUNK; ++j) cur[j] = ld_elem(tm, (long long)j * 64, lo);
  long long i = 0;
  if (PEEL && 2 * CHUNK <= minlen) {
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) nxt[j] = ld_elem(tm, (long long)(CHUNK + j) * 64, lo);
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) {
      if constexpr (HAS_OUT) st_elem(tmo, (long long)j * 64, loo, step((long long)j, cur[j]));
      else step((long long)j, cur[j]);
    }
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) cur[j] = nxt[j];
    i = CHUNK;
  }
  for (; i + CHUNK <= minlen; i += CHUNK) {                  // inside every row
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) nxt[j] = ld_elem(tm, (i + CHUNK + j) * 64, lo);
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) {
      if constexpr (HAS_OUT) st_elem(tmo, (i + j) * 64, loo, step(i + j, cur[j]));
      else step(i + j, cur[j]);
    }
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) cur[j] = nxt[j];
  }
  for (; i < maxlen; i += CHUNK) {                           // the rows' ends: step by step
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) nxt[j] = ld_elem(tm, (i + CHUNK + j) * 64, lo);
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) {
      if (i + j < len) {
        if constexpr (HAS_OUT) st_elem(tmo, (i + j) * 64, loo, step(i + j, cur[j]));
        else step(i + j, cur[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) cur[j] = nxt[j];
  }
}

// Which item a lane of a gang works on: its own (lane < count, a row that is not empty), else a copy of the first lane that
// has one.  false: nobody has work.  own: the lane's state goes back to its item.
template <typename Item>
__device__ __forceinline__ bool gang_lane(const Item *__restrict__ items, int count, int j, int *src, bool *own)
{
  const long long mylen = j < count ? items[j].len : 0;
  const unsigned long long has = __ballot(mylen > 0);
  if (has == 0) return false;
  *own = mylen > 0;
  *src = *own ? j : (int)__ffsll((long long)has) - 1;
  return true;
}

// rows of a gang's items <-> slabs: item k of the table (item_bytes apart; its row pointer and length sit at off_ptr /
// off_len) is lane k % 64 of group k / 64.  Tile = 64 samples x 64 rows through LDS (pitch 65).
template <typename T>
__global__ __launch_bounds__(256) void rows_tm_gather_kernel(const char *__restrict__ items, int item_bytes, int off_ptr, int off_len,
                                                             int n, const sdk::GangGroup *__restrict__ groups, T *__restrict__ tm, long long slab)
{
  __builtin_amdgcn_s_setprio(3);
  __shared__ T tile[64][65];
  const int g = blockIdx.y, s = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int first = groups ? groups[g].first : g * 64;
  const int last = groups ? first + groups[g].count : n;
  const long long m0 = (long long)blockIdx.x * 64;
#pragma unroll 4
  for (int p = 0; p < 16; ++p) {
    const int row = p * 4 + q, item = first + row;
    T v = T{};
    if (item < last) {
      const char *it = items + (size_t)item * item_bytes;
      const T *src = *reinterpret_cast<const T *const *>(it + off_ptr);
      const long long len = *reinterpret_cast<const long long *>(it + off_len);
      if (m0 + s < len) v = src[m0 + s];
    }
    tile[row][s] = v;
  }
  __syncthreads();
  T *dst = tm + (size_t)g * slab + m0 * 64;
#pragma unroll 4
  for (int p = 0; p < 16; ++p) { const int ss = p * 4 + q; dst[ss * 64 + s] = tile[s][ss]; }
}

template <typename T>
__global__ __launch_bounds__(256) void rows_tm_scatter_kernel(const char *__restrict__ items, int item_bytes, int off_ptr, int off_len,
                                                              int n, const sdk::GangGroup *__restrict__ groups, const T *__restrict__ tm, long long slab)
{
  __builtin_amdgcn_s_setprio(3);
  __shared__ T tile[64][65];
  const int g = blockIdx.y, s = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int first = groups ? groups[g].first : g * 64;
  const int last = groups ? first + groups[g].count : n;
  const long long m0 = (long long)blockIdx.x * 64;
  const T *src = tm + (size_t)g * slab + m0 * 64;
#pragma unroll 4
  for (int p = 0; p < 16; ++p) { const int ss = p * 4 + q; tile[s][ss] = src[ss * 64 + s]; }
  __syncthreads();
#pragma unroll 4
  for (int p = 0; p < 16; ++p) {
    const int row = p * 4 + q, item = first + row;
    if (item < last) {
      const char *it = items + (size_t)item * item_bytes;
      T *dst = *reinterpret_cast<T *const *>(it + off_ptr);
      const long long len = *reinterpret_cast<const long long *>(it + off_len);
      if (m0 + s < len) dst[m0 + s] = tile[row][s];
    }
  }
}

// (Round 6 tried a UNIFORM form for groups whose items have identical parameters -- 64 inspectors opened alike --, the parameters
// read once and kept in SGPRs like a bank's: SLOWER, 828 against 747 us per 64 x 8192 samples.  A VOP3P instruction reads one
// SGPR operand; thirteen scalar parameters cost more v_mov than the per-lane VGPRs cost anything.  Taken back.)
template <int KIND, int ORDER, bool GAIN1>
__device__ __forceinline__ void costas_gang_body(const sdk::CostasGangItem *__restrict__ items, int count, float2 *tm)
{
  int j; bool live;
  if (!gang_lane(items, count, (int)threadIdx.x, &j, &live)) return;
  const sdk::CostasGangItem it = items[j];
  const sdk::CostasParams p = it.p;                           // per lane: every item its own loop
  const sdk::CostasState s = it.s;
  CostasRegs<ORDER> r;
  r.phase = s.phase[0];
  r.omega = s.omega[0];
#pragma unroll
  for (int i = 1; i <= ORDER; ++i) {
    r.xh[i] = c32{s.xh[(i - 1) * 2 + 0], s.xh[(i - 1) * 2 + 1]};
    r.yh[i] = c32{s.yh[(i - 1) * 2 + 0], s.yh[(i - 1) * 2 + 1]};
  }
  gang_stream_tm<true, true>(tm, tm, j * 8u, j * 8u, it.len, [&](long long, float2 v) { return costas_step<KIND, ORDER, GAIN1>(p, r, v); });
  if (!live) return;
  s.phase[0] = r.phase;
  s.omega[0] = r.omega;
#pragma unroll
  for (int i2 = 1; i2 <= ORDER; ++i2) {
    s.xh[(i2 - 1) * 2 + 0] = r.xh[i2].re; s.xh[(i2 - 1) * 2 + 1] = r.xh[i2].im;
    s.yh[(i2 - 1) * 2 + 0] = r.yh[i2].re; s.yh[(i2 - 1) * 2 + 1] = r.yh[i2].im;
  }
}

// every loop type of a gang in ONE launch: workgroup g runs the items of groups[g] (all of one kind and arm-filter
// order, which selects the compiled-in loop).  The types used to fork onto side streams and join back: four event
// hops and three sets of gather / scatter launches per call.
__global__ __launch_bounds__(64) void costas_gang_kernel(const sdk::CostasGangItem *__restrict__ items,
                                                         const sdk::GangGroup *__restrict__ groups, float2 *tm, long long slab)
{
  const sdk::GangGroup gd = groups[blockIdx.x];
  const sdk::CostasGangItem *mine = items + gd.first;
  float2 *my = tm + (size_t)blockIdx.x * slab;
  switch (gd.kind * 8 + gd.order) {
#define SD_GANG_CASE(K, O) case (K) * 8 + (O): if (gd.gain1) costas_gang_body<K, O, true>(mine, gd.count, my); \
                                               else costas_gang_body<K, O, false>(mine, gd.count, my); break;
    SD_GANG_CASE(1, 0) SD_GANG_CASE(1, 1) SD_GANG_CASE(1, 2) SD_GANG_CASE(1, 3) SD_GANG_CASE(1, 4)
    SD_GANG_CASE(2, 0) SD_GANG_CASE(2, 1) SD_GANG_CASE(2, 2) SD_GANG_CASE(2, 3) SD_GANG_CASE(2, 4)
    SD_GANG_CASE(3, 0) SD_GANG_CASE(3, 1) SD_GANG_CASE(3, 2) SD_GANG_CASE(3, 3) SD_GANG_CASE(3, 4)
#undef SD_GANG_CASE
    default: break;
  }
}

__global__ __launch_bounds__(64) void pll_gang_kernel(const sdk::PllGangItem *__restrict__ items, int n, float2 *tm, long long slab)
{
  const sdk::PllGangItem *mine = items + (size_t)blockIdx.x * 64;
  int j; bool live;
  if (!gang_lane(mine, n - (int)blockIdx.x * 64, (int)threadIdx.x, &j, &live)) return;
  const sdk::PllGangItem it = mine[j];
  const float alpha = it.alpha, beta = it.beta;
  uint32_t phase = it.s.phase[0];
  float omega = it.s.omega[0];
  float2 *my = tm + (size_t)blockIdx.x * slab;
  gang_stream_tm<true, false>(my, my, j * 8u, j * 8u, it.len, [&](long long, float2 v) { return pll_step(alpha, beta, phase, omega, v); });
  if (!live) return;
  it.s.phase[0] = phase;
  it.s.omega[0] = omega;
}

// CMA equalizers (SPEC.md section I) of many inspectors: N weights and the delay line per lane
template <int N>
__global__ __launch_bounds__(64) void cma_gang_kernel(const sdk::CmaGangItem *__restrict__ items, int n)
{
  const int j = blockIdx.x * 64 + threadIdx.x;
  const bool live = j < n;
  const sdk::CmaGangItem it = items[live ? j : 0];
  const float mu = it.mu;
  const bool locked = it.locked != 0;
  float2 *w = reinterpret_cast<float2 *>(it.w), *dl = reinterpret_cast<float2 *>(it.dl);
  c32 wr[N], d[N];
#pragma unroll
  for (int i = 0; i < N; ++i) { const float2 a = w[i], b = dl[i]; wr[i] = c32{a.x, a.y}; d[i] = c32{b.x, b.y}; }
  const long long len = live ? (it.count ? (long long)it.count[0] : it.fixed_len) : 0;
  __shared__ GangLds<float2> lds;
  gang_stream<true>(lds, reinterpret_cast<const float2 *>(it.x), reinterpret_cast<float2 *>(it.y), len,
                    [&](long long, float2 v) {
#pragma unroll
    for (int i = N - 1; i > 0; --i) d[i] = d[i - 1];
    d[0] = c32{v.x, v.y};
    float yr = 0.0f, yi = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      yr = sd::fma_(wr[i].re, d[i].re, yr); yr = sd::fma_(-wr[i].im, d[i].im, yr);
      yi = sd::fma_(wr[i].re, d[i].im, yi); yi = sd::fma_(wr[i].im, d[i].re, yi);
    }
    if (!locked) {
      const float g = sd::fma_(yi, yi, yr * yr) - 1.0f;
      const c32 e = {yr * g, yi * g};
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const c32 t = sd::cmul_conj(e, d[i]);
        wr[i].re = sd::fma_(-mu, t.re, wr[i].re);
        wr[i].im = sd::fma_(-mu, t.im, wr[i].im);
      }
    }
    return float2{yr, yi};
  });
  if (!live) return;
#pragma unroll
  for (int i = 0; i < N; ++i) { w[i] = float2{wr[i].re, wr[i].im}; dl[i] = float2{d[i].re, d[i].im}; }
}
__global__ __launch_bounds__(64) void clock_gang_kernel(const sdk::ClockGangItem *__restrict__ items, int n, float2 *tm, long long slab)
{
  __shared__ float2 lds[RING * 64];                             // clock_ring's two tiles; clock_stream_tm uses the first
  const int j = blockIdx.x * 64 + threadIdx.x;
  const bool live = j < n;
  const sdk::ClockGangItem it = items[live ? j : 0];
  sdk::ClockParams p = it.p;
  // (the host writes one `steps` / `uniform` per group of 64 items; lane 0 of a workgroup is always live)
  const int steps = __builtin_amdgcn_readfirstlane(it.steps);
  if (__builtin_amdgcn_readfirstlane(it.uniform)) {
    p.alpha = uniform_f(p.alpha); p.beta = uniform_f(p.beta); p.gain = uniform_f(p.gain); p.bmin = uniform_f(p.bmin); p.bmax = uniform_f(p.bmax);
  }
  const sdk::ClockState s = it.s;
  ClockRegs r;
  r.phi = s.phi[0]; r.bnor = s.bnor[0];
  r.halfcycle = s.halfcycle[0];
  r.prev = float2{s.prev[0], s.prev[1]};
  r.x0 = float2{s.x0[0], s.x0[1]};
  r.x1 = float2{s.x1[0], s.x1[1]};
  r.x2 = float2{s.x2[0], s.x2[1]};
  r.n = it.count[0];
  const long long len = live ? it.len : 0;
  float2 *out = reinterpret_cast<float2 *>(it.sym);
  // round by round when the group's symbol rates allow it (the banks' schedule: staggered symbol clocks cost nothing), else
  // crossing by crossing
  if (steps > 0) clock_ring(tm + (size_t)blockIdx.x * slab, 64, threadIdx.x * 8u, len, live, steps, p, r, out, lds);
  else clock_stream_tm<true>(tm + (size_t)blockIdx.x * slab, 64, threadIdx.x * 8u, len, p, r, out, lds);
  if (!live) return;
  s.phi[0] = r.phi; s.bnor[0] = r.bnor; s.halfcycle[0] = r.halfcycle;
  s.prev[0] = r.prev.x; s.prev[1] = r.prev.y;
  s.x0[0] = r.x0.x; s.x0[1] = r.x0.y;
  s.x1[0] = r.x1.x; s.x1[1] = r.x1.y;
  s.x2[0] = r.x2.x; s.x2[1] = r.x2.y;
  it.count[0] = r.n;
}

__global__ __launch_bounds__(64) void agc_level_gang_kernel(const sdk::AgcGangItem *__restrict__ items, int n, float *tm, long long slab)
{
  const sdk::AgcGangItem *mine = items + (size_t)blockIdx.x * 64;
  int j; bool live;
  if (!gang_lane(mine, n - (int)blockIdx.x * 64, (int)threadIdx.x, &j, &live)) return;
  const sdk::AgcGangItem it = mine[j];
  const sdk::AgcState s = it.s;
  unsigned hang_n = s.hang_n[0];
  float fast = s.fast_level[0], slow = s.slow_level[0];
  const float far = it.p.fast_alpha_rise, faf = it.p.fast_alpha_fall, sar = it.p.slow_alpha_rise, saf = it.p.slow_alpha_fall;
  const float knee = it.p.knee;
  const unsigned hang_max = it.p.hang_max;
  const long long len = it.len;
  float *my = tm + (size_t)blockIdx.x * slab;
  gang_stream_tm<true, false>(my, my, j * 4u, j * 4u, len, [&](long long, float pk) {
    float d = pk - fast;
    const float fa = d > 0.0f ? far : faf;
    fast = sd::fma_(fa, d, fast);
    d = pk - slow;
    const bool rise = d > 0.0f;
    const bool fall = !rise && hang_n >= hang_max;
    const float sa = rise ? sar : saf;
    const float upd = sd::fma_(sa, d, slow);
    slow = (rise || fall) ? upd : slow;
    hang_n = rise ? 0u : (fall ? hang_n : hang_n + 1u);
    float lvl = fast > slow ? fast : slow;
    if (lvl < knee) lvl = knee;
    return lvl;
  });
  if (!live) return;
  s.hang_n[0] = hang_n; s.fast_level[0] = fast; s.slow_level[0] = slow;
}

// ---------------------------------------------------------------------------------------
// Gangs whose rows are columns of a time-major slab already (kernels.hpp GangSlab): what rows_tm_gather would build is
// what the producer -- the FFT filter bank writing channel c of every time step side by side -- left in memory.  The
// recurrence streams it where it lies: a wave-uniform base, the lane's byte offset from it, a run-time pitch.  The byte
// offsets of a chunk's CHUNK steps are loop invariants and stay in registers (a lone wavefront owns the register file), so
// a load is still scalar base + vector offset and nothing is added per sample.  Per lane the steps are the packed form's.
template <bool HAS_OUT, typename T, typename F>
__device__ __forceinline__ void gang_stream_slab(const T *tin, T *tout, long long pin, long long pout, uint32_t lo_in, uint32_t lo_out,
                                                 long long len, F step)
{
  const long long maxlen = uniform64(wave_max(len));
  if (maxlen <= 0) return;
  const long long minlen = uniform64(-wave_max(-len));       // (every lane has work: gang_lane)
  uint32_t oin[CHUNK], oout[CHUNK];
#pragma unroll
  for (int j = 0; j < CHUNK; ++j) {
    oin[j] = lo_in + (uint32_t)j * (uint32_t)pin * (uint32_t)sizeof(T);
    oout[j] = lo_out + (uint32_t)j * (uint32_t)pout * (uint32_t)sizeof(T);
  }
  T cur[CHUNK], nxt[CHUNK];
#pragma unroll
  for (int j = 0; j < CHUNK; ++j) cur[j] = ld_elem(tin, 0, oin[j]);
  long long i = 0;
  for (; i + CHUNK <= minlen; i += CHUNK) {                  // inside every row: one straight path
    const T *bn = tin + (i + CHUNK) * pin;
    T *bo = tout + i * pout;
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) nxt[j] = ld_elem(bn, 0, oin[j]);
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) {
      if constexpr (HAS_OUT) st_elem(bo, 0, oout[j], step(i + j, cur[j]));
      else step(i + j, cur[j]);
    }
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) cur[j] = nxt[j];
  }
  for (; i < maxlen; i += CHUNK) {
    const T *bn = tin + (i + CHUNK) * pin;
    T *bo = tout + i * pout;
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) nxt[j] = ld_elem(bn, 0, oin[j]);
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) {
      if (i + j < len) {
        if constexpr (HAS_OUT) st_elem(bo, 0, oout[j], step(i + j, cur[j]));
        else step(i + j, cur[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) cur[j] = nxt[j];
  }
}

__device__ __forceinline__ uint32_t slab_offset(const void *p, const void *base)
{
  return (uint32_t)(static_cast<const char *>(p) - static_cast<const char *>(base));
}

template <int KIND, int ORDER, bool GAIN1>
__device__ __forceinline__ void costas_gang_slab_body(const sdk::CostasGangItem *__restrict__ items, int count, const sdk::GangSlab &io)
{
  int j; bool live;
  if (!gang_lane(items, count, (int)threadIdx.x, &j, &live)) return;
  const sdk::CostasGangItem it = items[j];
  const sdk::CostasParams p = it.p;
  const sdk::CostasState s = it.s;
  CostasRegs<ORDER> r;
  r.phase = s.phase[0];
  r.omega = s.omega[0];
#pragma unroll
  for (int i = 1; i <= ORDER; ++i) {
    r.xh[i] = c32{s.xh[(i - 1) * 2 + 0], s.xh[(i - 1) * 2 + 1]};
    r.yh[i] = c32{s.yh[(i - 1) * 2 + 0], s.yh[(i - 1) * 2 + 1]};
  }
  const long long len = it.len;
  auto step = [&](long long, float2 v) { return costas_step<KIND, ORDER, GAIN1>(p, r, v); };
  // a pitch of 64 columns (at most 64 narrow inspectors on the shard -- BASELINE configs[3]'s slice) is the packed slabs' own:
  // the loop with immediate offsets
  if (io.pitch_in == 64 && io.pitch_out == 64)
    gang_stream_tm<true, true>(static_cast<const float2 *>(io.in), static_cast<float2 *>(io.out), slab_offset(it.x, io.in), slab_offset(it.y, io.out), len, step);
  else
    gang_stream_slab<true>(static_cast<const float2 *>(io.in), static_cast<float2 *>(io.out), io.pitch_in, io.pitch_out,
                           slab_offset(it.x, io.in), slab_offset(it.y, io.out), len, step);
  if (!live) return;
  s.phase[0] = r.phase;
  s.omega[0] = r.omega;
#pragma unroll
  for (int i2 = 1; i2 <= ORDER; ++i2) {
    s.xh[(i2 - 1) * 2 + 0] = r.xh[i2].re; s.xh[(i2 - 1) * 2 + 1] = r.xh[i2].im;
    s.yh[(i2 - 1) * 2 + 0] = r.yh[i2].re; s.yh[(i2 - 1) * 2 + 1] = r.yh[i2].im;
  }
}

__global__ __launch_bounds__(64) void costas_gang_slab_kernel(const sdk::CostasGangItem *__restrict__ items,
                                                              const sdk::GangGroup *__restrict__ groups, sdk::GangSlab io)
{
  const sdk::GangGroup gd = groups[blockIdx.x];
  const sdk::CostasGangItem *mine = items + gd.first;
  switch (gd.kind * 8 + gd.order) {
#define SD_GANG_CASE(K, O) case (K) * 8 + (O): if (gd.gain1) costas_gang_slab_body<K, O, true>(mine, gd.count, io); \
                                               else costas_gang_slab_body<K, O, false>(mine, gd.count, io); break;
    SD_GANG_CASE(1, 0) SD_GANG_CASE(1, 1) SD_GANG_CASE(1, 2) SD_GANG_CASE(1, 3) SD_GANG_CASE(1, 4)
    SD_GANG_CASE(2, 0) SD_GANG_CASE(2, 1) SD_GANG_CASE(2, 2) SD_GANG_CASE(2, 3) SD_GANG_CASE(2, 4)
    SD_GANG_CASE(3, 0) SD_GANG_CASE(3, 1) SD_GANG_CASE(3, 2) SD_GANG_CASE(3, 3) SD_GANG_CASE(3, 4)
#undef SD_GANG_CASE
    default: break;
  }
}

__global__ __launch_bounds__(64) void pll_gang_slab_kernel(const sdk::PllGangItem *__restrict__ items, int n, sdk::GangSlab io)
{
  const sdk::PllGangItem *mine = items + (size_t)blockIdx.x * 64;
  int j; bool live;
  if (!gang_lane(mine, n - (int)blockIdx.x * 64, (int)threadIdx.x, &j, &live)) return;
  const sdk::PllGangItem it = mine[j];
  const float alpha = it.alpha, beta = it.beta;
  uint32_t phase = it.s.phase[0];
  float omega = it.s.omega[0];
  const long long len = it.len;
  auto step = [&](long long, float2 v) { return pll_step(alpha, beta, phase, omega, v); };
  if (io.pitch_in == 64 && io.pitch_out == 64)
    gang_stream_tm<true, false>(static_cast<const float2 *>(io.in), static_cast<float2 *>(io.out), slab_offset(it.x, io.in), slab_offset(it.y, io.out), len, step);
  else
    gang_stream_slab<true>(static_cast<const float2 *>(io.in), static_cast<float2 *>(io.out), io.pitch_in, io.pitch_out,
                           slab_offset(it.x, io.in), slab_offset(it.y, io.out), len, step);
  if (!live) return;
  it.s.phase[0] = phase;
  it.s.omega[0] = omega;
}

__global__ __launch_bounds__(64) void clock_gang_slab_kernel(const sdk::ClockGangItem *__restrict__ items, int n, sdk::GangSlab io)
{
  __shared__ float2 lds[RING * 64];
  const int j = blockIdx.x * 64 + threadIdx.x;
  const bool live = j < n;
  const sdk::ClockGangItem it = items[live ? j : 0];
  sdk::ClockParams p = it.p;
  const int steps = __builtin_amdgcn_readfirstlane(it.steps);
  if (__builtin_amdgcn_readfirstlane(it.uniform)) {
    p.alpha = uniform_f(p.alpha); p.beta = uniform_f(p.beta); p.gain = uniform_f(p.gain); p.bmin = uniform_f(p.bmin); p.bmax = uniform_f(p.bmax);
  }
  const sdk::ClockState s = it.s;
  ClockRegs r;
  r.phi = s.phi[0]; r.bnor = s.bnor[0];
  r.halfcycle = s.halfcycle[0];
  r.prev = float2{s.prev[0], s.prev[1]};
  r.x0 = float2{s.x0[0], s.x0[1]};
  r.x1 = float2{s.x1[0], s.x1[1]};
  r.x2 = float2{s.x2[0], s.x2[1]};
  r.n = it.count[0];
  const long long len = live ? it.len : 0;
  float2 *out = reinterpret_cast<float2 *>(it.sym);
  const float2 *base = static_cast<const float2 *>(io.in);
  const uint32_t lo = slab_offset(it.x, io.in);
  if (steps > 0) clock_ring(base, io.pitch_in, lo, len, live, steps, p, r, out, lds);
  else clock_stream_tm<true>(base, io.pitch_in, lo, len, p, r, out, lds);
  if (!live) return;
  s.phi[0] = r.phi; s.bnor[0] = r.bnor; s.halfcycle[0] = r.halfcycle;
  s.prev[0] = r.prev.x; s.prev[1] = r.prev.y;
  s.x0[0] = r.x0.x; s.x0[1] = r.x0.y;
  s.x1[0] = r.x1.x; s.x1[1] = r.x1.y;
  s.x2[0] = r.x2.x; s.x2[1] = r.x2.y;
  it.count[0] = r.n;
}

__global__ __launch_bounds__(64) void agc_level_gang_slab_kernel(const sdk::AgcGangItem *__restrict__ items, int n, sdk::GangSlab io)
{
  const sdk::AgcGangItem *mine = items + (size_t)blockIdx.x * 64;
  int j; bool live;
  if (!gang_lane(mine, n - (int)blockIdx.x * 64, (int)threadIdx.x, &j, &live)) return;
  const sdk::AgcGangItem it = mine[j];
  const sdk::AgcState s = it.s;
  unsigned hang_n = s.hang_n[0];
  float fast = s.fast_level[0], slow = s.slow_level[0];
  const float far = it.p.fast_alpha_rise, faf = it.p.fast_alpha_fall, sar = it.p.slow_alpha_rise, saf = it.p.slow_alpha_fall;
  const float knee = it.p.knee;
  const unsigned hang_max = it.p.hang_max;
  const long long len = it.len;
  const uint32_t lo = slab_offset(it.peak, io.in);
  auto step = [&](long long, float pk) {
    float d = pk - fast;
    const float fa = d > 0.0f ? far : faf;
    fast = sd::fma_(fa, d, fast);
    d = pk - slow;
    const bool rise = d > 0.0f;
    const bool fall = !rise && hang_n >= hang_max;
    const float sa = rise ? sar : saf;
    const float upd = sd::fma_(sa, d, slow);
    slow = (rise || fall) ? upd : slow;
    hang_n = rise ? 0u : (fall ? hang_n : hang_n + 1u);
    float lvl = fast > slow ? fast : slow;
    if (lvl < knee) lvl = knee;
    return lvl;
  };
  if (io.pitch_in == 64 && io.pitch_out == 64) gang_stream_tm<true, false>(static_cast<const float *>(io.in), static_cast<float *>(io.out), lo, lo, len, step);
  else gang_stream_slab<true>(static_cast<const float *>(io.in), static_cast<float *>(io.out), io.pitch_in, io.pitch_out, lo, lo, len, step);
  if (!live) return;
  s.hang_n[0] = hang_n; s.fast_level[0] = fast; s.slow_level[0] = slow;
}

// The AGC's feed-forward steps on a slab: item k of the table is lane k % 64 of workgroup row k / 64, its samples sit in
// column it.lane of x, its magnitudes / peaks / levels in the same column of the work slabs (pitch `pw`).  A tile is 64
// items x SLAB_TM time steps, every access a row of adjacent columns.  Per value the operations (and, for the sliding
// maximum, the order of the comparisons: newest first) are agc_pre_items_kernel's / agc_apply_items_kernel's.
// (SLAB_TM = 32 time steps and a halo of the gang's longest history, not of the 63 steps the longest possible one takes: with
// 128 + 63 rows per tile a 2 Mi-sample block was 256 workgroups of 48 KB of LDS and took 52 us -- three times the row form,
// and the next block's channeliser, which wants the same LDS, ran beside it)
constexpr int SLAB_TM = 32;
__global__ __launch_bounds__(256) void agc_pre_slab_kernel(const sdk::AgcSlabItem *__restrict__ items, int n, const float2 *__restrict__ x,
                                                           long long px, float *__restrict__ db, float *__restrict__ peak, long long pw, int halo)
{
  __builtin_amdgcn_s_setprio(3);
  __shared__ float tile[SLAB_TM + 63][64];
  const int lane = threadIdx.x & 63, rowt = threadIdx.x >> 6;
  const int k = blockIdx.y * 64 + lane;
  const bool live = k < n;
  const sdk::AgcSlabItem it = items[live ? k : 0];
  const long long len = live ? it.len : 0;
  const int hl = (int)it.p.mag_history_size - 1;              // <= halo
  const long long col = it.lane;
  const long long m0 = (long long)blockIdx.x * SLAB_TM;
  for (int r = rowt; r < SLAB_TM + halo; r += 4) {            // tile row r <-> time m0 - halo + r
    const long long m = m0 - halo + r;
    float v = 0.f;
    if (m < len) {
      if (m >= 0) {
        const float2 s = x[m * px + col];
        v = 3.01029995663981195f * sd::log2_(sd::fma_(s.x, s.x, s.y * s.y) + 1e-8f);
        if (m >= m0) db[m * pw + col] = v;
      } else if (m + hl >= 0) v = it.s.mag_history[m + hl];
    }
    tile[r][lane] = v;
  }
  __syncthreads();
  for (int t = rowt; t < SLAB_TM; t += 4) {
    const long long m = m0 + t;
    if (m >= len) break;
    float pk = tile[t + halo][lane];
    for (int i = 1; i <= hl; ++i) {
      const float v = tile[t + halo - i][lane];
      pk = pk > v ? pk : v;
    }
    peak[m * pw + col] = pk;
  }
}

constexpr int APPLY_TM = 16;     // time steps per workgroup (64: an 8192-step sub-range was 128 workgroups and took 11 us; the row form 5)
__global__ __launch_bounds__(256) void agc_apply_slab_kernel(const sdk::AgcSlabItem *__restrict__ items, int n, const float2 *__restrict__ x,
                                                             long long px, float2 *__restrict__ y, long long py,
                                                             const float *__restrict__ lvl, long long pw, long long mlo)
{
  __builtin_amdgcn_s_setprio(3);
  const int lane = threadIdx.x & 63, rowt = threadIdx.x >> 6;
  const int k = blockIdx.y * 64 + lane;
  if (k >= n) return;
  const sdk::AgcSlabItem it = items[k];
  const long long delay = it.p.delay_line_size;
  const float slope = it.p.gain_slope - 1.0f;
  const long long mb = mlo + (long long)blockIdx.x * APPLY_TM;
  for (int r = rowt; r < APPLY_TM; r += 4) {
    const long long m = mb + r;
    if (m < it.m0 || m >= it.m1) continue;
    const float2 xd = m >= delay ? x[(m - delay) * px + it.lane] : float2{it.s.delay_line[m * 2 + 0], it.s.delay_line[m * 2 + 1]};
    const float g_db = lvl[m * pw + it.lane] * slope;
    const float g = sd::exp2_(g_db * 0.166096404744368117f) * 0.7f;
    y[m * py + it.lane_y] = float2{xd.x * g, xd.y * g};
  }
}

}  // namespace

namespace sdk {

hipError_t rows_tm_gather(const void *d_items, int item_bytes, int off_ptr, int off_len, int n, const GangGroup *d_groups, int ngroups,
                          int elem_bytes, void *tm, long long slab, long long maxlen, hipStream_t st)
{
  if (n <= 0 || maxlen <= 0) return hipSuccess;
  const dim3 grid((unsigned)((maxlen + 63) / 64), (unsigned)(d_groups ? ngroups : (n + 63) / 64)), block(256);
  if (elem_bytes == 8)
    hipLaunchKernelGGL(rows_tm_gather_kernel<float2>, grid, block, 0, st, static_cast<const char *>(d_items), item_bytes, off_ptr, off_len, n,
                       d_groups, static_cast<float2 *>(tm), slab);
  else if (elem_bytes == 4)
    hipLaunchKernelGGL(rows_tm_gather_kernel<float>, grid, block, 0, st, static_cast<const char *>(d_items), item_bytes, off_ptr, off_len, n,
                       d_groups, static_cast<float *>(tm), slab);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t rows_tm_scatter(const void *d_items, int item_bytes, int off_ptr, int off_len, int n, const GangGroup *d_groups, int ngroups,
                           int elem_bytes, const void *tm, long long slab, long long maxlen, hipStream_t st)
{
  if (n <= 0 || maxlen <= 0) return hipSuccess;
  const dim3 grid((unsigned)((maxlen + 63) / 64), (unsigned)(d_groups ? ngroups : (n + 63) / 64)), block(256);
  if (elem_bytes == 8)
    hipLaunchKernelGGL(rows_tm_scatter_kernel<float2>, grid, block, 0, st, static_cast<const char *>(d_items), item_bytes, off_ptr, off_len, n,
                       d_groups, static_cast<const float2 *>(tm), slab);
  else if (elem_bytes == 4)
    hipLaunchKernelGGL(rows_tm_scatter_kernel<float>, grid, block, 0, st, static_cast<const char *>(d_items), item_bytes, off_ptr, off_len, n,
                       d_groups, static_cast<const float *>(tm), slab);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t agc_level_gang(const AgcGangItem *d_items, int n, void *tm, long long slab, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(agc_level_gang_kernel, dim3((n + 63) / 64), dim3(64), 0, st, d_items, n, static_cast<float *>(tm), slab);
  return hipGetLastError();
}

hipError_t pll_gang(const PllGangItem *d_items, int n, void *tm, long long slab, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(pll_gang_kernel, dim3((n + 63) / 64), dim3(64), 0, st, d_items, n, static_cast<float2 *>(tm), slab);
  return hipGetLastError();
}

hipError_t cma_gang(const CmaGangItem *d_items, int n, int ntaps, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  const dim3 grid((n + 63) / 64), block(64);
  switch (ntaps) {
#define SD_CMA_GANG(N) case N: hipLaunchKernelGGL(cma_gang_kernel<N>, grid, block, 0, st, d_items, n); break;
    SD_CMA_GANG(1) SD_CMA_GANG(2) SD_CMA_GANG(3) SD_CMA_GANG(4) SD_CMA_GANG(5) SD_CMA_GANG(6) SD_CMA_GANG(7) SD_CMA_GANG(8)
    SD_CMA_GANG(9) SD_CMA_GANG(10) SD_CMA_GANG(11) SD_CMA_GANG(12) SD_CMA_GANG(13) SD_CMA_GANG(14) SD_CMA_GANG(15) SD_CMA_GANG(16)
#undef SD_CMA_GANG
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t clock_gang(const ClockGangItem *d_items, int n, void *tm, long long slab, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(clock_gang_kernel, dim3((n + 63) / 64), dim3(64), 0, st, d_items, n, static_cast<float2 *>(tm), slab);
  return hipGetLastError();
}

hipError_t costas_gang(const CostasGangItem *d_items, const GangGroup *d_groups, int ngroups, void *tm, long long slab, hipStream_t st)
{
  if (ngroups <= 0) return hipSuccess;
  hipLaunchKernelGGL(costas_gang_kernel, dim3((unsigned)ngroups), dim3(64), 0, st, d_items, d_groups, static_cast<float2 *>(tm), slab);
  return hipGetLastError();
}

hipError_t costas_gang_slab(const CostasGangItem *d_items, const GangGroup *d_groups, int ngroups, GangSlab io, hipStream_t st)
{
  if (ngroups <= 0) return hipSuccess;
  hipLaunchKernelGGL(costas_gang_slab_kernel, dim3((unsigned)ngroups), dim3(64), 0, st, d_items, d_groups, io);
  return hipGetLastError();
}

hipError_t pll_gang_slab(const PllGangItem *d_items, int n, GangSlab io, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(pll_gang_slab_kernel, dim3((n + 63) / 64), dim3(64), 0, st, d_items, n, io);
  return hipGetLastError();
}

hipError_t clock_gang_slab(const ClockGangItem *d_items, int n, GangSlab io, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(clock_gang_slab_kernel, dim3((n + 63) / 64), dim3(64), 0, st, d_items, n, io);
  return hipGetLastError();
}

hipError_t agc_level_gang_slab(const AgcGangItem *d_items, int n, GangSlab io, hipStream_t st)
{
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(agc_level_gang_slab_kernel, dim3((n + 63) / 64), dim3(64), 0, st, d_items, n, io);
  return hipGetLastError();
}

hipError_t agc_pre_slab(const AgcSlabItem *d_items, int n, const void *x, long long px, float *db, float *peak, long long pw,
                        long long max_len, int halo, hipStream_t st)
{
  if (n <= 0 || max_len <= 0) return hipSuccess;
  if (halo < 0 || halo > 63) return hipErrorInvalidValue;
  hipLaunchKernelGGL(agc_pre_slab_kernel, dim3((unsigned)((max_len + SLAB_TM - 1) / SLAB_TM), (unsigned)((n + 63) / 64)), dim3(256), 0, st,
                     d_items, n, static_cast<const float2 *>(x), px, db, peak, pw, halo);
  return hipGetLastError();
}

hipError_t agc_apply_slab(const AgcSlabItem *d_items, int n, const void *x, long long px, void *y, long long py, const float *lvl,
                          long long pw, long long mlo, long long mhi, hipStream_t st)
{
  if (n <= 0 || mhi <= mlo) return hipSuccess;
  hipLaunchKernelGGL(agc_apply_slab_kernel, dim3((unsigned)((mhi - mlo + APPLY_TM - 1) / APPLY_TM), (unsigned)((n + 63) / 64)), dim3(256), 0, st,
                     d_items, n, static_cast<const float2 *>(x), px, static_cast<float2 *>(y), py, lvl, pw, mlo);
  return hipGetLastError();
}

}  // namespace sdk
