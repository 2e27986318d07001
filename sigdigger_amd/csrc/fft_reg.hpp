// fft_reg.hpp -- transforms of up to 64 points held entirely in one lane's registers (8 x 8 with compile-time twiddles in
// SGPR pairs), and the fused complex-product helpers of the one-wavefront-per-transform kernels (specttuner_wave.hip,
// psd_wave.hip).  See fft_core.hpp for the building blocks (dft2/4/8/16, cmul, add_mj).
#pragma once
#include "fft_core.hpp"

namespace fftcore {

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

// a * b with b wave-uniform (a compile-time constant): the constant travels in an SGPR pair, not in VGPRs
__device__ __forceinline__ cf cmul_u(cf a, cf b)
{
  cf r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=&v"(r) : "v"(a), "s"(b));
  return r;
}

// a * b and a * (b * c) as ONE asm statement each: between two dependent asm statements the compiler inserts an s_nop
// (it cannot see inside them), and for a lone wavefront an s_nop costs a full issue slot like any instruction
__device__ __forceinline__ cf cmul1(cf a, cf b)
{
  cf r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=&v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ cf cmul3(cf a, cf b, cf c)
{
  cf w, r;
  asm("v_pk_mul_f32 %1, %3, %4 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_fma_f32 %1, %3, %4, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
      "v_pk_mul_f32 %0, %2, %1 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_fma_f32 %0, %2, %1, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=&v"(r), "=&v"(w) : "v"(a), "v"(b), "v"(c));
  return r;
}

// two independent a * (b * c) in one statement
__device__ __forceinline__ void cmul3x2(cf a0, cf b0, cf c0, cf a1, cf b1, cf c1, cf &r0, cf &r1)
{
  cf w0, w1;
  asm("v_pk_mul_f32 %2, %6, %7 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %3, %8, %9 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_fma_f32 %2, %6, %7, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
      "v_pk_fma_f32 %3, %8, %9, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
      "v_pk_mul_f32 %0, %4, %2 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %1, %5, %3 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_fma_f32 %0, %4, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
      "v_pk_fma_f32 %1, %5, %3, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
      : "=&v"(r0), "=&v"(r1), "=&v"(w0), "=&v"(w1) : "v"(a0), "v"(a1), "v"(b0), "v"(c0), "v"(b1), "v"(c1));
}
// two independent a * b in one statement
__device__ __forceinline__ void cmul1x2(cf a0, cf b0, cf a1, cf b1, cf &r0, cf &r1)
{
  asm("v_pk_mul_f32 %0, %2, %3 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_mul_f32 %1, %4, %5 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
      "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
      "v_pk_fma_f32 %1, %4, %5, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
      : "=&v"(r0), "=&v"(r1) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
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

// `hook(step)` is called after each of the R1 + R2 sub-transforms: a place to slip other work (memory requests) in
template <int R1, int R2, class Hook = NoHook>
__device__ __forceinline__ void dft_2f(const cf *in, cf *out, Hook hook = Hook())
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
    hook(n1);
  }
#pragma unroll
  for (int k2 = 0; k2 < R2; ++k2) {
    cf b[R1];
#pragma unroll
    for (int n1 = 0; n1 < R1; ++n1) b[n1] = mid[n1 + R1 * k2];
    dftR<R1>(b);
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) out[k2 + R2 * k1] = b[k1];
    hook(R1 + k2);
  }
}

template <int LOG2N, class Hook = NoHook> __device__ __forceinline__ void dft_reg(const cf *in, cf *out, Hook hook = Hook())
{
  constexpr int N = 1 << LOG2N;
  if constexpr (LOG2N == 6) dft_2f<8, 8>(in, out, hook);
  else if constexpr (LOG2N == 5) dft_2f<4, 8>(in, out, hook);
  else {
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = in[i];
    dftR<N>(out);
  }
}

}  // namespace fftcore
