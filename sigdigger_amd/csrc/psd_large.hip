// psd_large.hip -- main-spectrum PSD of frames beyond the LDS (N = 2^15 .. 2^20: the scanner's nextPow2(fs / 1 kHz),
// Panoramic/Scanner.cpp:323; FFTWidget's sizes up to 2^20, Default/FFT/FFTWidget.cpp:350-351) -- rows A2 / P1.
//
// Four-step transform in TWO trips through HBM instead of the four or five radix-16 passes of fft.hip (64 B of traffic per
// sample there against 12 B algorithmic):  N = N1 N2,  n = n1 + N1 n2,  k = k2 + N2 k1,
//
//   X[k2 + N2 k1] = sum_n1 W_N1^(n1 k1) [ W_N^(n1 k2) sum_n2 w[n] x[n1 + N1 n2] W_N2^(n2 k2) ]
//
//   pass A  (psdl_cols_kernel):  one THREAD per column n1: its N2 = 32 / 64 samples (stride N1: a wavefront reads 512
//           contiguous bytes per n2) times the window, DFT_N2 on registers, times W_N^(n1 k2), stored k2-major:
//           A[f][k2][n1] -- again 512 contiguous bytes per wavefront and k2.  No LDS, no barrier.   8 B in, 8 B out.
//   pass B  (psdl_rows_kernel):  one WORKGROUP per (output, chunk, k2): the row A[f][k2][.] is a contiguous N1-point frame
//           (1024 .. 16384 points) transformed in LDS by fft_core.hpp's register passes, exactly as psd.hip transforms a
//           frame; |X|^2 accumulates in registers over the chunk's frames (frame order) and goes out once per chunk as
//           P[o][c][k2][k1].                                                                      8 B in, 4 B / chunk out.
//   finish  (psdl_finish_kernel): out[o][k2 + N2 k1] = scale / navg * sum_c P[o][c][k2][k1] (chunk order) through an
//           LDS tile (both sides in whole rows), optional fftshift + dB (PSDMessage.cpp:26-39).
//
// The frames go through in batches small enough for the A buffer to be read back out of the last-level cache (256 MiB)
// instead of HBM.  An output's navg frames are cut into chunks of `ch` consecutive frames -- a function of navg alone --
// and a chunk that straddles a batch boundary keeps its partial sums in P and continues: the per-bin sum is
// sum over chunks (in order) of the sum over the chunk's frames (in order) whatever the batch size, so the result does
// not depend on it (tests/test_gpu_parity.py::test_psd_large_frames_batches_do_not_change_the_bits).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "kernels.hpp"
#include "tuning.hpp"
#include "fft_core.hpp"
#include "fft_reg.hpp"

namespace {
using namespace fftcore;

// v[q] *= W_N^(q idx), q = 1 .. R - 1: the powers of two are table values, every other power one product of two lower ones
template <int R>
__device__ __forceinline__ void col_twiddles(cf *v, const cf *__restrict__ tw, unsigned idx, unsigned nmask)
{
  cf w[R];
#pragma unroll
  for (int q = 1; q < R; q <<= 1) w[q] = tw[(idx * (unsigned)q) & nmask];
#pragma unroll
  for (int q = 3; q < R; ++q) {
    if (q & (q - 1)) {
      const int hi = 1 << (31 - __builtin_clz(q));
      w[q] = cmul(w[q - hi], w[hi]);
    }
  }
#pragma unroll
  for (int q = 1; q < R; ++q) v[q] = cmul(v[q], w[q]);
}

// pass A.  grid = (N1 / 256, frames of the batch); x: first frame of the batch (frames `hop` samples apart)
template <int LOG2N2>
__global__ __launch_bounds__(256) void psdl_cols_kernel(const cf *__restrict__ x, long long hop, int log2n1,
                                                        const float *__restrict__ window, const cf *__restrict__ tw,
                                                        cf *__restrict__ A)
{
  __builtin_amdgcn_s_setprio(3);
  constexpr int N2 = 1 << LOG2N2;
  const long long N1 = 1ll << log2n1;
  const unsigned n1 = blockIdx.x * 256u + threadIdx.x;
  const cf *fr = x + (long long)blockIdx.y * hop + n1;
  const float *wp = window + n1;
  cf v[N2];
  float wv[N2];
#pragma unroll
  for (int n2 = 0; n2 < N2; ++n2) v[n2] = __builtin_nontemporal_load(fr + (long long)n2 * N1);    // read once
#pragma unroll
  for (int n2 = 0; n2 < N2; ++n2) wv[n2] = wp[(long long)n2 * N1];
#pragma unroll
  for (int n2 = 0; n2 < N2; ++n2) v[n2] = v[n2] * wv[n2];
  cf *dst = A + ((long long)blockIdx.y << (log2n1 + LOG2N2)) + n1;
  const unsigned nmask = (1u << (log2n1 + LOG2N2)) - 1u;
  if constexpr (LOG2N2 == 6) {
    cf o[N2];
    dft_reg<6>(v, o);
    col_twiddles<N2>(o, tw, n1, nmask);
#pragma unroll
    for (int k2 = 0; k2 < N2; ++k2) dst[(long long)k2 * N1] = o[k2];
  } else {
    dftR<N2>(v);
    col_twiddles<N2>(v, tw, n1, nmask);
#pragma unroll
    for (int k2 = 0; k2 < N2; ++k2) dst[(long long)k2 * N1] = v[k2];
  }
}

// pass B.  grid.x = (chunks touched by the batch) * N2.  Output o (absolute index) owns the frames [o navg, (o + 1) navg),
// its chunk c the frames [o navg + c ch, o navg + min((c + 1) ch, navg)); global chunk index g = o cpo + c (cpo chunks
// per output); the batch holds the frames [F0, F0 + nb).  P: ring of `pring` chunk slots of N floats.
template <int LOG2N1, int THREADS>
__global__ __launch_bounds__(THREADS, ((1 << LOG2N1) / THREADS >= 32 ? 2 : (THREADS >= 1024 ? 4 : (THREADS >= 128 ? THREADS / 128 : 1))))
void psdl_rows_kernel(const cf *__restrict__ A, int log2n2, long long F0, int nb, int navg, int ch, int cpo, long long g_first,
                      const cf *__restrict__ tw, float *__restrict__ P, int pring)
{
  __builtin_amdgcn_s_setprio(3);
  using PL = PlanFor<LOG2N1, THREADS>;
  constexpr int N1 = 1 << LOG2N1;
  constexpr int E = N1 / THREADS;
  constexpr int R0 = 1 << PL::bits(0);
  constexpr int NB0 = E / R0;
  constexpr int RL = 1 << PL::bits(PL::P - 1);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cf *lds = reinterpret_cast<cf *>(smem);
  const int tid0 = threadIdx.x;
  const int k2 = blockIdx.x & ((1 << log2n2) - 1);
  const long long g = g_first + (blockIdx.x >> log2n2), o = g / cpo;
  const int c = (int)(g - o * cpo);
  // this chunk's frames, and those of them inside the batch
  const long long c0 = o * navg + (long long)c * ch, c1 = (c + 1) * ch < navg ? c0 + ch : (o + 1) * navg;
  const long long fa = c0 > F0 ? c0 : F0, fe = c1 < F0 + nb ? c1 : F0 + nb;
  const bool cont = fa > c0;                                    // the chunk began in an earlier batch: continue its sums
  float *prow = P + (((long long)(g % pring) << log2n2) + k2) * N1;

  TwBase tb;
  load_tw_base<LOG2N1, THREADS, 0>(tb, tw, tid0);
  constexpr int NBL = E / RL;
  float pw[E];
#pragma unroll
  for (int b = 0; b < NBL; ++b) {
#pragma unroll
    for (int q = 0; q < RL; ++q) pw[b * RL + q] = cont ? prow[tid0 + b * THREADS + q * (N1 / RL)] : 0.0f;
  }
  // the next frame's row is requested before this one is transformed (16 points per thread: 32 VGPRs; the 32-point
  // threads have no registers to spare and two workgroups per CU to cover for each other)
  constexpr bool PF = (E == 16);
  cf nxt[E];
  auto request = [&](long long f, int tid) {
    const cf *row = A + ((((f - F0) << log2n2) + k2) << LOG2N1);
    if constexpr (PL::PAIR0) {
      typedef float __attribute__((ext_vector_type(4))) f4;
      static_assert(NB0 == 2, "pair requests assume two butterflies per thread in pass 0");
#pragma unroll
      for (int q = 0; q < R0; ++q) {
        const f4 s2 = *reinterpret_cast<const f4 *>(row + 2 * tid + q * (N1 / R0));
        nxt[q] = s2.xy; nxt[R0 + q] = s2.zw;
      }
    } else {
#pragma unroll
      for (int b = 0; b < NB0; ++b) {
#pragma unroll
        for (int q = 0; q < R0; ++q) nxt[b * R0 + q] = row[tid + b * THREADS + q * (N1 / R0)];
      }
    }
  };
  if (PF && fa < fe) request(fa, tid0);
  for (long long f = fa; f < fe; ++f) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));                               // see psd_kernel: keeps LICM from hoisting every pass's twiddles
    if (!PF) request(f, tid);
    cf v[E];
#pragma unroll
    for (int i = 0; i < E; ++i) v[i] = nxt[i];
    if (PF && f + 1 < fe) request(f + 1, tid);
    fft_pass<LOG2N1, THREADS, 0>(v, lds, tb, tid, pw);
    PassRunner<LOG2N1, THREADS, 1>::run(v, lds, tb, tid, pw);
  }
#pragma unroll
  for (int b = 0; b < NBL; ++b) {
#pragma unroll
    for (int q = 0; q < RL; ++q) prow[tid0 + b * THREADS + q * (N1 / RL)] = pw[b * RL + q];
  }
}

// finish.  grid = (N1 / 64, outputs to finish); block 256.  A block owns 64 consecutive k1 of every k2: the run
// out[N2 k1_0 .. N2 (k1_0 + 64)) is contiguous.  The output's chunks are summed in order; a thread reads 4 consecutive
// k1 of one k2 (16 bytes) from every chunk, all chunks' requests in flight together.
template <int LOG2N2>
__global__ __launch_bounds__(256) void psdl_finish_kernel(const float *__restrict__ P, int log2n1, long long o0, int cpo,
                                                          int pring, float sc, int mode, float *__restrict__ out)
{
  typedef float __attribute__((ext_vector_type(4))) f4;
  constexpr int N2 = 1 << LOG2N2, PITCH = 65, PER = N2 * 16 / 256;   // float4 loads per thread and chunk
  __shared__ float tile[N2 * PITCH];
  const long long N1 = 1ll << log2n1, N = N1 << LOG2N2;
  const long long o = o0 + blockIdx.y;
  const int k10 = blockIdx.x * 64;
  f4 acc[PER];
  for (int c0 = 0; c0 < cpo; c0 += 4) {
    f4 v[4][PER];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (c0 + u < cpo) {
        const float *p = P + ((long long)((o * cpo + c0 + u) % pring) << (log2n1 + LOG2N2)) + k10;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
          const int i = threadIdx.x + 256 * j, k2 = i >> 4, q = i & 15;
          v[u][j] = *reinterpret_cast<const f4 *>(p + (long long)k2 * N1 + 4 * q);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (c0 + u < cpo) {
#pragma unroll
        for (int j = 0; j < PER; ++j) acc[j] = (c0 + u == 0) ? v[u][j] : acc[j] + v[u][j];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = threadIdx.x + 256 * j, k2 = i >> 4, q = i & 15;
    float *t = tile + k2 * PITCH + 4 * q;
    t[0] = acc[j].x; t[1] = acc[j].y; t[2] = acc[j].z; t[3] = acc[j].w;
  }
  __syncthreads();
  float *dst = out + o * N;
  for (int i = threadIdx.x; i < N2 * 64; i += 256) {
    const int k2 = i & (N2 - 1), k1 = i >> LOG2N2;
    const float ps = tile[k2 * PITCH + k1] * sc;
    const long long k = (long long)k10 * N2 + i;                // = k2 + N2 (k10 + k1)
    if (mode == 0) dst[k] = ps;
    else dst[(k + N / 2) & (N - 1)] = 10.0f * log10f(ps + 1e-8f);
  }
}

template <int LOG2N1, int THREADS>
hipError_t launch_rows(const void *A, int log2n2, long long F0, int nb, int navg, int ch, int cpo, long long g_first, long long ng,
                       const void *tw, float *P, int pring, hipStream_t st)
{
  constexpr int N1 = 1 << LOG2N1;
  const size_t lds = sizeof(cf) * (size_t)(N1 + (N1 >> 4) + 1);
  auto kern = psdl_rows_kernel<LOG2N1, THREADS>;
  static bool attr_done_dev[64] = {};
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  bool &attr_done = attr_done_dev[dev_ & 63];
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(ng << log2n2)), dim3(THREADS), lds, st, reinterpret_cast<const cf *>(A), log2n2, F0, nb,
                     navg, ch, cpo, g_first, reinterpret_cast<const cf *>(tw), P, pring);
  return hipGetLastError();
}

}  // namespace

namespace sdk {

// how N = 2^log2n splits: pass A transforms 2^log2n2 points per thread on registers, pass B rows of 2^(log2n - log2n2)
int psd_large_log2n2(int log2n)
{
  const int force = (int)sdk::tuning().psd_large_n2;           // measurements
  if ((force == 5 && log2n <= 19) || force == 6) return force;
  return log2n <= 17 ? 5 : 6;
}

// frames per chunk of an output's sum: a function of navg alone (the result must not depend on the batch size).  Short
// chunks are what fills the chip when a batch holds few outputs; 1/ch of the input's size goes through P.
int psd_large_chunk(int navg) { return navg >= 128 ? 16 : (navg >= 32 ? 8 : (navg >= 16 ? 4 : (navg >= 4 ? 2 : 1))); }

// a: the A buffer, `batch` frames of N complex; P: pring chunk slots of N floats (pring >= chunks a batch can touch
// plus those of an output still open: batch / ch + 2 cpo + 2); tw_n: W_N table (N entries), tw_row: W_N1 table (N1 entries)
hipError_t psd_frames_large2(int log2n, const void *x, long long hop, int navg, const float *window, const void *tw_n,
                             const void *tw_row, float scale, int mode, float *out, long long nout, void *a, float *P, int pring,
                             int batch, hipStream_t st)
{
  const int l2 = psd_large_log2n2(log2n), l1 = log2n - l2;
  const long long n1 = 1ll << l1, total = nout * navg;
  const int ch = psd_large_chunk(navg), cpo = (navg + ch - 1) / ch;
  auto chunk_of = [&](long long f) { const long long o = f / navg; return o * cpo + (f - o * navg) / ch; };
  const cf *xx = reinterpret_cast<const cf *>(x);
  if (batch < 1) batch = 1;
  for (long long F0 = 0; F0 < total; F0 += batch) {
    const int nb = total - F0 < batch ? (int)(total - F0) : batch;
    const dim3 ga((unsigned)(n1 / 256), (unsigned)nb);
    if (l2 == 5) hipLaunchKernelGGL(psdl_cols_kernel<5>, ga, dim3(256), 0, st, xx + F0 * hop, hop, l1, window, reinterpret_cast<const cf *>(tw_n), reinterpret_cast<cf *>(a));
    else hipLaunchKernelGGL(psdl_cols_kernel<6>, ga, dim3(256), 0, st, xx + F0 * hop, hop, l1, window, reinterpret_cast<const cf *>(tw_n), reinterpret_cast<cf *>(a));
    const long long o_first = F0 / navg;
    const long long g_first = chunk_of(F0), ng = chunk_of(F0 + nb - 1) - g_first + 1;
    // slots are addressed g % pring: the chunks from the first one of the batch's first output (read again by the finish
    // kernel when that output completes) to the batch's last must not share a slot
    if (chunk_of(F0 + nb - 1) - o_first * cpo + 1 > pring) return hipErrorInvalidValue;
    hipError_t e;
    switch (l1) {
      case 9:  e = launch_rows<9, 64>(a, l2, F0, nb, navg, ch, cpo, g_first, ng, tw_row, P, pring, st); break;
      case 10: e = launch_rows<10, 64>(a, l2, F0, nb, navg, ch, cpo, g_first, ng, tw_row, P, pring, st); break;
      case 11: e = launch_rows<11, 128>(a, l2, F0, nb, navg, ch, cpo, g_first, ng, tw_row, P, pring, st); break;
      case 12: e = launch_rows<12, 256>(a, l2, F0, nb, navg, ch, cpo, g_first, ng, tw_row, P, pring, st); break;
      case 13: e = launch_rows<13, 256>(a, l2, F0, nb, navg, ch, cpo, g_first, ng, tw_row, P, pring, st); break;
      case 14: e = launch_rows<14, 512>(a, l2, F0, nb, navg, ch, cpo, g_first, ng, tw_row, P, pring, st); break;
      default: return hipErrorInvalidValue;
    }
    if (e != hipSuccess) return e;
    // the outputs whose last frame is in this batch
    const long long done_first = o_first, done_last = (F0 + nb) / navg - 1;      // last COMPLETE output so far
    if (done_last >= done_first) {
      const dim3 gf((unsigned)(n1 / 64), (unsigned)(done_last - done_first + 1));
      if (l2 == 5) hipLaunchKernelGGL(psdl_finish_kernel<5>, gf, dim3(256), 0, st, P, l1, done_first, cpo, pring, scale / (float)navg, mode, out);
      else hipLaunchKernelGGL(psdl_finish_kernel<6>, gf, dim3(256), 0, st, P, l1, done_first, cpo, pring, scale / (float)navg, mode, out);
    }
  }
  return hipGetLastError();
}

}  // namespace sdk
