// fft_core.hpp -- the register-resident Stockham FFT passes shared by psd.hip (main spectrum) and specttuner.hip (FFT
// channeliser): VOP3P complex arithmetic, radix 2/4/8/16 DFTs on registers, the compile-time pass plan, per-thread base
// twiddles and one pass (gather from LDS -> twiddles -> DFT -> autosort scatter to LDS).
//
// A transform of N = 2^LOG2N points is run by THREADS cooperating threads (N / THREADS points each); `tid` is the
// thread's index inside that group and `lds` the group's buffer of N + N/16 (+1) complex elements (index padded by
// lpad()), so several groups of one workgroup can run independent transforms side by side -- the barriers are the
// workgroup's (every group executes the same pass at the same time).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fftcore {


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
// (Between two consecutive inline-asm statements the compiler inserts an s_nop -- it cannot see inside them -- and for a
// lone wavefront an s_nop is a full issue slot: instructions that belong together are ONE statement.)
__device__ __forceinline__ cf cmul(cf a, cf b)
{
  // (a.x b.x - a.y b.y, a.x b.y + a.y b.x)
  cf r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=&v"(r) : "v"(a), "v"(b));
  return r;
}
// p = t + (-j) d, m = t - (-j) d
__device__ __forceinline__ void addsub_mj(cf t, cf d, cf &p, cf &m)
{
  asm("v_pk_add_f32 %0, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n\t"
      "v_pk_add_f32 %1, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=&v"(p), "=&v"(m) : "v"(t), "v"(d));
}

// forward DFTs on registers, natural order in, natural order out (DIT, even/odd split)
__device__ __forceinline__ void dft2(cf &a, cf &b) { cf t = a; a = cadd(t, b); b = csub(t, b); }

__device__ __forceinline__ void dft4(cf &a0, cf &a1, cf &a2, cf &a3)
{
  cf t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), d = csub(a1, a3);
  a0 = cadd(t0, t2); a2 = csub(t0, t2); addsub_mj(t1, d, a1, a3);
}

__device__ __forceinline__ cf mul_w8_1(cf a) { return add_mj(a, a) * 0.70710678118654752440f; }            // * (1 - j)/sqrt2
__device__ __forceinline__ cf mul_w8_3(cf a) { return add_mj(-a, a) * 0.70710678118654752440f; }           // * (-1 - j)/sqrt2
// e +- o W8^1 and e +- o W8^3 as explicit fused multiply-adds: the same roundings in every instantiation and under
// every -ffp-contract setting (different instantiations of one transform must agree bit for bit)
__device__ __forceinline__ cf fmac(cf a, float c, cf b) { return __builtin_elementwise_fma(a, cf{c, c}, b); }
__device__ __forceinline__ void bfly_w8_1(cf e, cf o, cf &p, cf &m)
{
  const cf r = add_mj(o, o);
  p = fmac(r, 0.70710678118654752440f, e); m = fmac(r, -0.70710678118654752440f, e);
}
__device__ __forceinline__ void bfly_w8_3(cf e, cf o, cf &p, cf &m)
{
  const cf r = add_mj(-o, o);
  p = fmac(r, 0.70710678118654752440f, e); m = fmac(r, -0.70710678118654752440f, e);
}

__device__ __forceinline__ void dft8(cf *v)
{
  cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
  cf o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
  dft4(e0, e1, e2, e3);
  dft4(o0, o1, o2, o3);
  v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
  bfly_w8_1(e1, o1, v[1], v[5]);
  addsub_mj(e2, o2, v[2], v[6]);                         // o2 * (-j) folded into the butterfly
  bfly_w8_3(e3, o3, v[3], v[7]);
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
  o[3] = cmul(o[3], cf{ s1, -c1});
  o[5] = cmul(o[5], cf{-s1, -c1});
  o[7] = cmul(o[7], cf{-c1, -s1});
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i == 4) addsub_mj(e[4], o[4], v[4], v[12]);
    else if (i == 2) bfly_w8_1(e[2], o[2], v[2], v[10]);
    else if (i == 6) bfly_w8_3(e[6], o[6], v[6], v[14]);
    else { v[i] = cadd(e[i], o[i]); v[i + 8] = csub(e[i], o[i]); }
  }
}

// 32 points: two 16-point transforms of the even/odd inputs, odd half times W32^k, one butterfly
__device__ __forceinline__ void dft32(cf *v)
{
  cf e[16], o[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
  dft16(e);
  dft16(o);
  // W32^k = (cos, -sin)(2 pi k / 32), k = 1..15 (k = 4, 8, 12 fold into the butterfly)
  constexpr float C[16] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                           0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f,
                           0.19509032201612826785f, 0.0f, -0.19509032201612826785f, -0.38268343236508977173f,
                           -0.55557023301960222474f, -0.70710678118654752440f, -0.83146961230254523708f,
                           -0.92387953251128675613f, -0.98078528040323044913f};
  constexpr float S[16] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f,
                           0.70710678118654752440f, 0.83146961230254523708f, 0.92387953251128675613f,
                           0.98078528040323044913f, 1.0f, 0.98078528040323044913f, 0.92387953251128675613f,
                           0.83146961230254523708f, 0.70710678118654752440f, 0.55557023301960222474f,
                           0.38268343236508977173f, 0.19509032201612826785f};
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (i == 0) { v[0] = cadd(e[0], o[0]); v[16] = csub(e[0], o[0]); }
    else if (i == 8) addsub_mj(e[8], o[8], v[8], v[24]);
    else if (i == 4) bfly_w8_1(e[4], o[4], v[4], v[20]);
    else if (i == 12) bfly_w8_3(e[12], o[12], v[12], v[28]);
    else { const cf t = cmul(o[i], cf{C[i], -S[i]}); v[i] = cadd(e[i], t); v[i + 16] = csub(e[i], t); }
  }
}

template <int R> __device__ __forceinline__ void dftR(cf *v);
template <> __device__ __forceinline__ void dftR<2>(cf *v)  { dft2(v[0], v[1]); }
template <> __device__ __forceinline__ void dftR<4>(cf *v)  { dft4(v[0], v[1], v[2], v[3]); }
template <> __device__ __forceinline__ void dftR<8>(cf *v)  { dft8(v); }
template <> __device__ __forceinline__ void dftR<16>(cf *v) { dft16(v); }
template <> __device__ __forceinline__ void dftR<32>(cf *v) { dft32(v); }

__device__ __forceinline__ int lpad(int i) { return i + (i >> 4); }

// compile-time pass plan: ceil(bits/MB) passes, the first (bits % P) passes one bit wider
// (MB = 4: radix <= 16, sixteen points per thread; MB = 5: radix <= 32, thirty-two points per thread)
// MB = 5 puts the wider passes LAST and gives a thread ADJACENT butterflies in pass 0 (j = NB tid + b instead of
// tid + b THREADS): its pass-0 operands are then pairs of neighbouring samples, one 16-byte request each (see psd.hip)
template <int LOG2N, int MB = 4> struct Plan {
  static constexpr int P     = (LOG2N + MB - 1) / MB;
  static constexpr int BASE  = LOG2N / P;
  static constexpr int EXTRA = LOG2N % P;
  static constexpr bool PAIR0 = (MB == 5);
  static constexpr int bits(int p) { return BASE + ((PAIR0 ? p >= P - EXTRA : p < EXTRA) ? 1 : 0); }
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
template <int LOG2N, int THREADS> using PlanFor = Plan<LOG2N, (((1 << LOG2N) / THREADS >= 32) ? 5 : 4)>;
struct TwBase { cf w[MAXP][MAXNB][2]; };

template <int LOG2N, int THREADS, int PASS>
__device__ __forceinline__ void load_tw_base(TwBase &tb, const cf *__restrict__ tw, int tid)
{
  using PL = PlanFor<LOG2N, THREADS>;
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

// a place to slip other work (memory requests) between the steps of a transform
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };

// hook(q) is called once per operand q = 0..R-1, spread over the twiddle products
template <int R, class Hook = NoHook>
__device__ __forceinline__ void apply_twiddles_base(cf *v, const cf *base, Hook hook = Hook())
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
  if (R > 16) {
    w[16] = cmul(w[8], w[8]);
#pragma unroll
    for (int q = 17; q < 32; ++q) w[q] = cmul(w[q - 16], w[16]);
  }
  hook(0);
#pragma unroll
  for (int q = 1; q < R; ++q) { v[q] = cmul(v[q], w[q]); hook(q); }
}

// LASTMODE 0: the last pass accumulates |X|^2 into pw (PSD); 1: it leaves the spectrum in v -- v[b*R + q] = X[j + q*N/R],
// j = tid + b*THREADS -- for the caller
// hook(i), i = 0..E-1: called once per operand of the pass, between its twiddle products (passes > 0 only)
template <int LOG2N, int THREADS, int PASS, int LASTMODE = 0, class Hook = NoHook>
__device__ __forceinline__ void fft_pass(cf *v /*[E]*/, cf *lds, const TwBase &tb, int tid,
                                         float *pw /*[E]*/, Hook hook = Hook())
{
  using PL = PlanFor<LOG2N, THREADS>;
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
    if constexpr ((N / R) % 16 == 0) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const cf *gp = lds + lpad(tid + b * THREADS);
#pragma unroll
        for (int q = 0; q < R; ++q) v[b * R + q] = gp[q * ((N / R) + (N / R) / 16)];
      }
    } else {                                      // small transforms (N < 16 R): the pad term is not linear in q
#pragma unroll
      for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int q = 0; q < R; ++q) v[b * R + q] = lds[lpad(tid + b * THREADS + q * (N / R))];
      }
    }
    __syncthreads();                              // everyone has read: LDS may be overwritten
  }
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int j = (PASS == 0 && PL::PAIR0) ? tid * NB + b : tid + b * THREADS;
    const int k = j & (NS - 1);
    cf *vb = v + b * R;
    if (PASS > 0) apply_twiddles_base<R>(vb, tb.w[PASS][b], [&](int q) { hook(b * R + q); });
    dftR<R>(vb);
    const int j0 = ((j - k) << RB) + k;
    if (!LAST) {
      if constexpr (NS % 16 == 0) {               // lpad(j0 + q*NS) = lpad(j0) + q*lpad(NS)
        cf *sp = lds + lpad(j0);
#pragma unroll
        for (int q = 0; q < R; ++q) sp[q * (NS + NS / 16)] = vb[q];
      } else if constexpr (PASS == 0 && PL::PAIR0 && R == 16 && NB == 2) {
        // the thread's two butterflies fill 34 consecutive slots (lane stride 68 dwords): as 8-byte stores lanes t and
        // t + 16 meet in a bank (25 M conflict cycles per 256 Mi samples); as 16-byte stores 16 lanes cover the 64 banks
        // exactly.  The second butterfly starts on an odd slot, so its first and last results go out alone.
        typedef float __attribute__((ext_vector_type(4))) f4;
        cf *sp = lds + lpad(j0);
        if (b == 0) {
#pragma unroll
          for (int q = 0; q < 16; q += 2) *reinterpret_cast<f4 *>(sp + q) = f4{vb[q].x, vb[q].y, vb[q + 1].x, vb[q + 1].y};
        } else {
          sp[0] = vb[0];
#pragma unroll
          for (int q = 1; q < 15; q += 2) *reinterpret_cast<f4 *>(sp + q) = f4{vb[q].x, vb[q].y, vb[q + 1].x, vb[q + 1].y};
          sp[15] = vb[15];
        }
      } else if constexpr (PASS == 0 && R <= 32) {   // j0 = j*R, q < R: the pad term of q is a constant
        cf *sp = lds + lpad(j0);
#pragma unroll
        for (int q = 0; q < R; ++q) sp[q + (q >> 4)] = vb[q];
      } else {
#pragma unroll
        for (int q = 0; q < R; ++q) lds[lpad(j0 + q * NS)] = vb[q];
      }
    } else {
      // last pass: NS == N/R, j0 == j, output index j + q*N/R; accumulate power
      if constexpr (LASTMODE == 0) {
#pragma unroll
        for (int q = 0; q < R; ++q) pw[b * R + q] += vb[q].x * vb[q].x + vb[q].y * vb[q].y;
      }
    }
  }
  if (!LAST) __syncthreads();
}

template <int LOG2N, int THREADS, int PASS, int LASTMODE = 0>
struct PassRunner {
  static __device__ __forceinline__ void run(cf *v, cf *lds, const TwBase &tb, int tid, float *pw)
  {
    fft_pass<LOG2N, THREADS, PASS, LASTMODE>(v, lds, tb, tid, pw);
    if constexpr (PASS + 1 < PlanFor<LOG2N, THREADS>::P) PassRunner<LOG2N, THREADS, PASS + 1, LASTMODE>::run(v, lds, tb, tid, pw);
  }
};


}  // namespace fftcore
