// psd.hip -- main-spectrum PSD for gfx950: windowed N-point FFT, |X|^2, frame averaging,
// optional fused fftshift + dB (PSDMessage ctor) -- rows A2/A3/A4/A9 of SURVEY.md section 8(a).
//
// One workgroup transforms one output frame (navg consecutive input frames, accumulated in
// registers).  The N-point FFT is a Stockham autosort FFT whose passes (radix 8/16, 16/32 for the
// 8192- and 16384-point frames: three passes, 32 points per thread) run on registers; the frame lives in ONE LDS buffer (in place: all of a pass's operands are
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
#include "tuning.hpp"

#include "fft_core.hpp"

#ifdef PSD_TSTAMP
// debug build only: per-phase clock stamps of one wavefront (tools/psd_tstamp.py reads them)
__device__ unsigned long long g_psd_ts[64 * 8];
#define PTS(n) do { __builtin_amdgcn_sched_barrier(0); ts[n] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PTS(n) do { } while (0)
#endif

// cache policy of the sample requests of a capture-sized input: sc0 | nt -- read once, do not displace the window and
// the twiddles from L1 / L2 (256 Mi samples, 8192-pt: 516-521 us against 558-580 us with the default policy; 4096-pt
// 457-463 against 472-475).  An analyzer block (32 MiB, just written by the ingest kernel and read by the channeliser
// too) still sits in the last-level cache, and there the default policy is the faster one (4 Mi samples: 17.1 against
// 18.9 us at 4096-pt), so the host picks by input size.
constexpr int AUX_DEFAULT = 0, AUX_STREAM = 3;
namespace {
using namespace fftcore;
typedef float __attribute__((ext_vector_type(4))) f4;
typedef float __attribute__((ext_vector_type(2))) f2;

// grid.x = number of output frames, grid.y = S (split of the navg frames of one output over S
// workgroups; S > 1 writes unscaled partial sums to `partial`, reduced by psd_reduce_kernel in a
// fixed order so the result is deterministic)
// HALVES (round 3): a frame of 2 N points on this N-point kernel, in ONE trip through HBM.  Decimation in frequency by two:
//   X[2k]     = DFT_N( w[n] x[n] + w[n+N] x[n+N] )[k]
//   X[2k + 1] = DFT_N( (w[n] x[n] - w[n+N] x[n+N]) W_2N^n )[k]
// Two workgroups per output (blockIdx.x = 2 o + r) each read the WHOLE frame -- the second read of a frame comes out of
// the last-level cache, not HBM -- and form their half's operands on the way in; everything behind pass 0's operands is
// the N-point kernel unchanged, bin k of half r lands at 2 k + r.  `tw2`: W_2N table (2 N entries).
template <int LOG2N, int THREADS, bool STREAM, bool HALVES = false>
// second launch bound: two workgroups per CU must fit the register file (N = 16384 is LDS-limited to one)
__global__ __launch_bounds__(THREADS, ((1 << LOG2N) / THREADS >= 32 ? 2 : (THREADS >= 1024 ? 4 : (THREADS >= 128 ? THREADS / 128 : 1)))) void psd_kernel(const cf *__restrict__ x, long long hop, int navg,
                                                      const float *__restrict__ window,
                                                      const cf *__restrict__ tw, float scale, int mode,
                                                      float *__restrict__ out, float *__restrict__ partial,
                                                      const cf *__restrict__ tw2 = nullptr)
{
  __builtin_amdgcn_s_setprio(3);   // ahead of the resident recurrence wavefronts (see chan_fir_kernel)
  using PL = PlanFor<LOG2N, THREADS>;
  constexpr int N  = 1 << LOG2N;
  constexpr int E  = N / THREADS;
  constexpr int R0 = 1 << PL::bits(0);
  constexpr int NB0 = E / R0;
  constexpr int RL = 1 << PL::bits(PL::P - 1);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cf *lds = reinterpret_cast<cf *>(smem);
  const int tid0 = threadIdx.x;
  // HALVES: the two workgroups of an output sit 8 apart in dispatch order -- workgroups go round the 8 XCDs, so the pair
  // shares an L2 and the frame's second reader finds it there (when the grid's width is a multiple of 16; any other
  // width pairs neighbours: correct, just no shared L2)
  const bool paired = HALVES && (gridDim.x & 15) == 0;
  const long long o = !HALVES ? (long long)blockIdx.x : paired ? 8ll * (blockIdx.x >> 4) + (blockIdx.x & 7) : (long long)(blockIdx.x >> 1);
  const int half = !HALVES ? 0 : paired ? (int)((blockIdx.x >> 3) & 1) : (int)(blockIdx.x & 1);   // which of the two interleaved half spectra
  constexpr int NT = HALVES ? 2 * N : N;                       // points of a frame / bins of an output

  float pw[E];
#pragma unroll
  for (int i = 0; i < E; ++i) pw[i] = 0.0f;
  TwBase tb;
  load_tw_base<LOG2N, THREADS, 0>(tb, tw, tid0);

  const int S = gridDim.y;
  const int fps = (navg + S - 1) / S;
  const int f_begin = blockIdx.y * fps;
  const int f_end = (f_begin + fps < navg) ? f_begin + fps : navg;
  // Software pipeline over the frames of this workgroup (see the frame loop): the raw samples of frame f+1 are
  // requested while frame f is in passes 1 and 2.  With two workgroups per CU there is not enough other work to hide
  // the HBM latency otherwise: one-frame workgroups reached 51 % of the HBM peak where the in-loop version stayed at
  // 38 %.  (measured: pays for N = 4096 / 8192; N = 2048 loses a wave of occupancy to the 32 extra VGPRs, N = 16384
  // spills, smaller frames have enough workgroups per CU anyway)
  constexpr bool PREFETCH = (LOG2N == 12 || LOG2N == 13);
  cf nxt[E];
  // one request of frame `fr`: operand i = b * R0 + q of pass 0 (a buffer load: the lane offset is the only VGPR, the
  // rest of the address is scalar; a zero-length descriptor makes the request of the frame after the last a no-op)
  auto descr = [&](int f) {
    const cf *fr = x + (o * navg + f) * hop;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<cf *>(fr), 0, f < f_end ? NT * 8 : 0, 0x00020000);
  };
  // (PAIR0 plans: the thread's NB0 = 2 butterflies are neighbours, so request q fetches operand q of both: 16 bytes)
  constexpr int NREQ = PL::PAIR0 ? R0 : E;
  static_assert(!PL::PAIR0 || NB0 == 2, "pair requests assume two butterflies per thread in pass 0");
  auto request_one = [&](__amdgpu_buffer_rsrc_t r, int i) {
    if constexpr (PL::PAIR0) {
      const f4 s2 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, tid0 * 16, i * (N / R0) * 8, STREAM ? AUX_STREAM : AUX_DEFAULT));
      nxt[i] = s2.xy; nxt[R0 + i] = s2.zw;
    } else {
      const int b = i / R0, q = i % R0;
      nxt[i] = __builtin_bit_cast(cf, __builtin_amdgcn_raw_buffer_load_b64(r, tid0 * 8, (b * THREADS + q * (N / R0)) * 8, STREAM ? AUX_STREAM : AUX_DEFAULT));
    }
  };
  auto request = [&](int f) {
    const __amdgpu_buffer_rsrc_t r = descr(f);
#pragma unroll
    for (int i = 0; i < NREQ; ++i) request_one(r, i);
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
#ifdef PSD_TSTAMP
    unsigned long long ts[8] = {0};
#endif
    PTS(0);
#ifdef PSD_TSTAMP
    if (PREFETCH) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
    PTS(1);
    // pass 0 operands: the prefetched samples, window applied on the fly
    if constexpr (HALVES) {
      static_assert(!HALVES || PL::PAIR0, "the two-halves front is written for the 32-point threads");
      // the frame's second half: requested now, combined with the first as it arrives
      const __amdgpu_buffer_rsrc_t r = descr(f);
#pragma unroll
      for (int q = 0; q < R0; ++q) {
        const f4 s2 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, tid0 * 16 + N * 8, q * (N / R0) * 8, STREAM ? AUX_STREAM : AUX_DEFAULT));
        v[q] = s2.xy; v[R0 + q] = s2.zw;
      }
      const f2 *w2 = reinterpret_cast<const f2 *>(window) + tid;
      const f4 *t2 = reinterpret_cast<const f4 *>(tw2) + tid;            // W_2N^(2 tid + q N/R0), W_2N^(2 tid + 1 + q N/R0)
#pragma unroll
      for (int q = 0; q < R0; ++q) {
        const f2 wa = w2[q * (N / R0 / 2)], wb = w2[(N / 2) + q * (N / R0 / 2)];
        const cf a0 = nxt[q] * wa.x, a1 = nxt[R0 + q] * wa.y, b0 = v[q] * wb.x, b1 = v[R0 + q] * wb.y;
        if (half == 0) { v[q] = a0 + b0; v[R0 + q] = a1 + b1; }
        else {
          const f4 tq = t2[q * (N / R0 / 2)];
          v[q] = cmul(a0 - b0, tq.xy); v[R0 + q] = cmul(a1 - b1, tq.zw);
        }
      }
    } else if constexpr (PL::PAIR0) {
      const f2 *w2 = reinterpret_cast<const f2 *>(window) + tid;        // window[2 tid + q N/R0], window[2 tid + 1 + ...]
#pragma unroll
      for (int q = 0; q < R0; ++q) {
        const f2 wq = w2[q * (N / R0 / 2)];
        v[q] = nxt[q] * wq.x; v[R0 + q] = nxt[R0 + q] * wq.y;
      }
    } else {
#pragma unroll
      for (int b = 0; b < NB0; ++b) {
        const int j = tid + b * THREADS;
#pragma unroll
        for (int q = 0; q < R0; ++q) {
          const int i = j + q * (N / R0);
          v[b * R0 + q] = nxt[b * R0 + q] * window[i];
        }
      }
    }
    PTS(2);
    // (the barrier after the previous frame's last gather already ordered LDS reuse)
    fft_pass<LOG2N, THREADS, 0>(v, lds, tb, tid, pw);
    PTS(3);
    if constexpr (PREFETCH) {
      // Software pipeline over the frames of this workgroup: the raw samples of frame f+1 are requested while passes 1
      // and 2 of frame f run, ONE request between two twiddle products, half of them in each pass.  Issued back to
      // back, the requests of a wavefront fill the CU's miss queue and the wavefront sits in the issue stage for as
      // long as a whole pass takes (measured with 32 x 512 B: 2900 of the frame's 10900 clocks); spread out, the
      // queue has drained by the time the next request comes.  (1 Gi samples, 8192-pt: 596 us back to back after pass
      // 0, 554 us all in pass 1, 551 us split; 4096-pt: 526 / 516 / 472 us)
      static_assert(PL::P == 3, "the prefetching sizes have three passes");
      const __amdgpu_buffer_rsrc_t r = descr(f + 1);
      auto hooked = [&](int i) {
        __builtin_amdgcn_sched_barrier(0);
        request_one(r, i);
        __builtin_amdgcn_sched_barrier(0);
      };
      fft_pass<LOG2N, THREADS, 1>(v, lds, tb, tid, pw, [&](int i) {
        if (i % 2 == 0 && i / 2 < NREQ / 2) hooked(i / 2);
      });
      PTS(5);
      fft_pass<LOG2N, THREADS, 2>(v, lds, tb, tid, pw, [&](int i) {
        if (i % 2 == 0 && i / 2 < NREQ - NREQ / 2) hooked(NREQ / 2 + i / 2);
      });
      PTS(6);
#ifdef PSD_TSTAMP
      if (blockIdx.x == 3 && blockIdx.y == 0 && tid0 == 0 && f - f_begin < 64) {
        for (int n = 0; n < 8; ++n) g_psd_ts[(f - f_begin) * 8 + n] = ts[n];
      }
#endif
    } else {
      PassRunner<LOG2N, THREADS, 1>::run(v, lds, tb, tid, pw);
    }
  }

  // epilogue: thread holds power of bins j + q*N/RL (last-pass geometry)
  const float sc = (S > 1) ? 1.0f : scale / (float)navg;
  if (S > 1) mode = 0;
  float *dst = (S > 1) ? partial + (o * S + blockIdx.y) * NT : out + o * NT;
  constexpr int NBL = E / RL;
#pragma unroll
  for (int b = 0; b < NBL; ++b) {
    const int j = tid0 + b * THREADS;
#pragma unroll
    for (int q = 0; q < RL; ++q) {
      const int i = HALVES ? 2 * (j + q * (N / RL)) + half : j + q * (N / RL);     // bin of the output frame
      float p = pw[b * RL + q] * sc;
      if (mode == 0) {
        dst[i] = p;
      } else {
        // Suscan/Messages/PSDMessage.cpp:29-38: out[(i + N/2) mod N] = 10 log10(p + 1e-8)
        dst[(i + NT / 2) & (NT - 1)] = 10.0f * log10f(p + 1e-8f);
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

template <int LOG2N, int THREADS, bool STREAM, bool HALVES = false>
hipError_t launch_psd_p(const void *x, long long hop, int navg, const float *window, const void *tw,
                      float scale, int mode, float *out, long long nout, float *partial, int S, hipStream_t st,
                      const void *tw2 = nullptr)
{
  constexpr int N = 1 << LOG2N, NT = HALVES ? 2 * N : N;
  const size_t lds = sizeof(cf) * (size_t)(N + (N >> 4) + 1);
  auto kern = psd_kernel<LOG2N, THREADS, STREAM, HALVES>;
  static bool attr_done_dev[64] = {};                        // a function attribute belongs to a device
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  bool &attr_done = attr_done_dev[dev_ & 63];
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  sdk::launch_timed("psd_kernel", kern, dim3((unsigned)(HALVES ? 2 * nout : nout), (unsigned)S), dim3(THREADS), lds, st,
                     reinterpret_cast<const cf *>(x), hop, navg, window, reinterpret_cast<const cf *>(tw),
                     scale, mode, out, partial, reinterpret_cast<const cf *>(tw2));
  if (S > 1) {
    sdk::launch_timed("psd_reduce_kernel", psd_reduce_kernel, dim3((NT + 63) / 64, (unsigned)nout), dim3(256), 0, st,
                       partial, S, NT, scale / (float)navg, mode, out);
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

#ifdef PSD_TSTAMP
extern "C" __attribute__((visibility("default"))) int suamd_debug_psd_ts(unsigned long long *out)
{
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_psd_ts), sizeof(unsigned long long) * 64 * 8);
}
#endif

// inputs larger than this are read with the streaming policy (the last-level cache holds 256 MiB)
constexpr long long PSD_STREAM_BYTES = 128ll << 20;

template <int LOG2N, int THREADS>
hipError_t launch_psd(const void *x, long long hop, int navg, const float *window, const void *tw,
                      float scale, int mode, float *out, long long nout, float *partial, int S, hipStream_t st)
{
  const int force = (int)sdk::tuning().psd_stream;           // 0 / 1 pins the policy (measurements)
  const bool stream = force >= 0 ? force != 0 : nout * navg * hop * 8 > PSD_STREAM_BYTES;
  return stream ? launch_psd_p<LOG2N, THREADS, true>(x, hop, navg, window, tw, scale, mode, out, nout, partial, S, st)
                : launch_psd_p<LOG2N, THREADS, false>(x, hop, navg, window, tw, scale, mode, out, nout, partial, S, st);
}

namespace sdk {

// how many workgroups share the navg frames of one output: enough to fill the chip's resident-workgroup slots for
// that frame size, at least `minf` frames each
// a caller's plan for the launches of this thread (suamd_psd_set_split_target): 0 = the defaults below
static thread_local int t_split_target = 0;
void psd_split_target(int target) { t_split_target = target > 0 ? target : 0; }

int psd_split(long long nout, int navg, int log2n)
{
  // Resident workgroups on the chip (256 CUs): 4 per CU for N <= 4096, 2 for 8192, 1 for 16384 (LDS).  At least `minf`
  // frames per workgroup: the register-resident twiddles are reused and the partial sums stay a fraction of the input
  // (PMC showed 2.1x the algorithmic HBM traffic with one frame per workgroup).  Measured on 256 Mi samples, 128
  // outputs: 4096-pt 463 us split over 512 workgroups, 407 us over 1024; 8192-pt 540-585 / 565 / 543 us over 512 /
  // 1024 / 2048; 16384-pt 570 / 586 / 605 us.  On a 4 Mi analyzer block minf decides (8192-pt 25.6 us at minf = 2,
  // 30.6 us at 4; 16384-pt 28.8 vs 44.8 us).  An input with nout >= target never splits.
  const int target_env = sdk::tuning().psd_split_target > 0 ? (int)sdk::tuning().psd_split_target : -1;
  const int minf = sdk::tuning().psd_min_frames > 0 ? (int)sdk::tuning().psd_min_frames : 2;
  // (round 4: 8192 points 512 -> 256 and 16384 points 512 -> 256 workgroups.  16384 points fit ONE workgroup per CU, so 512
  // was two rounds: 48.5 -> 42.1 us per 16 Mi samples.  8192 points fit two per CU, and 512 workgroups of 4 frames for
  // exactly 512 slots is what the analyzer pipeline cannot give: its three recurrence wavefronts take a slot each, the
  // displaced workgroups start when the first ones end -- 55.9 + 5.1 us inside the pipeline against 34.2 + 6.5 alone; 256
  // workgroups of 8 frames: 49.6 + 4.8 inside, 38.2 + 5.3 alone, and half the partial sums (1.125 x instead of 1.25 x the
  // algorithmic bytes).  Blocks of 4 Mi samples and captures are not affected: the first is bounded by `minf`, the second
  // never splits.  tools/psd_block.py, tools/inpipe_sweep.sh, profiles/r04_inpipe_penalty.txt)
  const int target = t_split_target > 0 ? t_split_target : target_env > 0 ? target_env : (log2n <= 12 ? 1024 : 256);
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
  const int S = partial ? psd_split(nout, navg, log2n) : 1;
  switch (log2n) {
    // 16 points per thread (8 for N = 512): one radix-16 or two radix-8 butterflies per pass; 32 points per thread for
    // N = 8192 / 16384: three passes of radix 16 / 32, pass-0 operands requested as 16-byte pairs
    case 9:  return launch_psd<9, 64>(x, hop, navg, window, tw, scale, mode, out, nout, partial, S, st);
    case 10: return launch_psd<10, 64>(x, hop, navg, window, tw, scale, mode, out, nout, partial, S, st);
    case 11: return launch_psd<11, 128>(x, hop, navg, window, tw, scale, mode, out, nout, partial, S, st);
    case 12: return launch_psd<12, 256>(x, hop, navg, window, tw, scale, mode, out, nout, partial, S, st);
    case 13: return launch_psd<13, 256>(x, hop, navg, window, tw, scale, mode, out, nout, partial, S, st);
    case 14: return launch_psd<14, 512>(x, hop, navg, window, tw, scale, mode, out, nout, partial, S, st);
    default: return hipErrorInvalidValue;
  }
}

// 32768-point frames in one trip: two workgroups per output on the 16384-point kernel (see psd_kernel, HALVES).
// tw: W_16384 table, tw2: W_32768 table; partial: nout * psd_split(2 nout, navg, 14) * 32768 floats (or nullptr)
hipError_t psd_frames_32k(const void *x, long long hop, int navg, const float *window, const void *tw, const void *tw2,
                          float scale, int mode, float *out, long long nout, float *partial, hipStream_t st)
{
  if (nout <= 0) return hipSuccess;
  const int S = partial ? psd_split(2 * nout, navg, 14) : 1;
  const int force = (int)sdk::tuning().psd_stream;
  const bool stream = force >= 0 ? force != 0 : nout * navg * hop * 8 > PSD_STREAM_BYTES;
  return stream ? launch_psd_p<14, 512, true, true>(x, hop, navg, window, tw, scale, mode, out, nout, partial, S, st, tw2)
                : launch_psd_p<14, 512, false, true>(x, hop, navg, window, tw, scale, mode, out, nout, partial, S, st, tw2);
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
