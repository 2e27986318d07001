// sd_math.hpp -- deterministic binary32 primitives for gfx950 device code (SPEC.md section D).
//
// Every function is a fixed sequence of IEEE-754 binary32 operations; the translation unit
// MUST be compiled with -ffp-contract=off so that only the __builtin_fmaf calls below fuse.
// hipcc's default float division / sqrt are correctly rounded
// (-fhip-fp32-correctly-rounded-divide-sqrt), which the SPEC relies on.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// SD_HD: the primitives that are plain binary32 C++ (no gfx950 instruction) also compile for the host, where
// csrc/sigutils_host.cpp runs them one sample at a time behind libsigutils' by-value su_* calls -- the SAME sequence of
// operations on both sides (the host pass must be compiled with -ffp-contract=off as well).
#define SD_HD __host__ __device__ __forceinline__

namespace sd {

struct c32 { float re, im; };

SD_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
SD_HD uint32_t f2u(float v) { return __builtin_bit_cast(uint32_t, v); }
SD_HD float u2f(uint32_t v) { return __builtin_bit_cast(float, v); }

// D1: 32-bit phase (2^32 per turn) -> cos/sin. Quadrant reduction + Cephes minimax kernels.
SD_HD void phasor_u32(uint32_t p, float &c, float &s)
{
  // r = p - (q << 30) with q = (p + 0x20000000) >> 30 is the low 30 bits of p, sign-extended:
  // one v_bfe_i32 instead of add / and / sub on the loop-carried phase -> sample path
  const int32_t  r = (int32_t)(p << 2) >> 2;
  const float x  = (float)r * 1.46291807926715968e-9f;
  const float z  = x * x;
  float sp = fma_(z, -1.9515295891e-4f, 8.3321608736e-3f);
  sp = fma_(sp, z, -1.6666654611e-1f);
  const float sn = fma_(sp * z, x, x);
  float cp = fma_(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  cp = fma_(cp, z, 4.166664568298827e-2f);
  const float cs = fma_(cp * z, z, fma_(z, -0.5f, 1.0f));
  // rotate by q quarter turns -- q=0:(cs,sn) 1:(-sn,cs) 2:(-cs,-sn) 3:(sn,-cs) -- with integer
  // bit operations only (a compare + v_cndmask pair costs ~4x a bit-op on gfx950 because of the
  // VCC write->read hazard); the values are exactly the selected / negated polynomials.
  const uint32_t t    = p + 0x20000000u;                       // bits 31:30 = q
  const uint32_t swap = (uint32_t)((int32_t)(t << 1) >> 31);   // all ones when q is odd
  const uint32_t ics = f2u(cs), isn = f2u(sn);
  const uint32_t a = (isn & swap) | (ics & ~swap);             // q odd ? sn : cs
  const uint32_t b = (ics & swap) | (isn & ~swap);             // q odd ? cs : sn
  c = u2f(a ^ ((t ^ (t << 1)) & 0x80000000u));                 // negate for q = 1, 2
  s = u2f(b ^ (t & 0x80000000u));                              // negate for q = 2, 3
}

typedef float v2f_ __attribute__((ext_vector_type(2)));

// phasor_u32 with the two polynomials evaluated as one packed chain, result as the pair (cos, sin):
// the same binary32 operations as phasor_u32 (each half of a v_pk_fma_f32 is an IEEE fma), laid out
// so that no register shuffling is needed between the steps -- for the one-wavefront recurrence
// kernels every instruction is 4 issue cycles.
__device__ __forceinline__ v2f_ phasor_pk(uint32_t p)
{
  const int32_t r = (int32_t)(p << 2) >> 2;
  const float x = (float)r * 1.46291807926715968e-9f;
  const float z = x * x;
  // everything hangs off the one pair P = (x, z): (z, z) is P's high half broadcast (an operand modifier), and
  // (x, fma(z, -0.5, 1)) is fma(P, (1, -0.5), (0, 1)) -- x * 1 + 0 is x exactly (x is never -0: it comes from an
  // integer) -- so no register is copied to build a packed operand
  const v2f_ P = {x, z};
  // the five packed operations as ONE asm statement: the compiler puts an s_nop behind every v_pk whose result the next
  // instruction reads, and a lone wavefront pays a full issue slot for it (the hardware interlocks by itself:
  // tools/ubench/issue.hip measures the same rate for dependent and independent chains)
  v2f_ q, xc, sc;
  asm("v_pk_fma_f32 %0, %3, %4, %5 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"        // q  = (z,z) K1 + K2
      "v_pk_fma_f32 %1, %3, %8, %9\n\t"                                          // xc = P (1,-0.5) + (0,1)
      "v_pk_fma_f32 %0, %0, %3, %6 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"        // q  = q (z,z) + K3
      "v_pk_mul_f32 %0, %0, %3 op_sel:[0,1] op_sel_hi:[1,1]\n\t"                // q  = q (z,z)
      "v_pk_fma_f32 %2, %0, %3, %1"                                               // sc = q P + xc
      : "=&v"(q), "=&v"(xc), "=&v"(sc)
      : "v"(P), "s"(v2f_{-1.9515295891e-4f, 2.443315711809948e-5f}), "v"(v2f_{8.3321608736e-3f, -1.388731625493765e-3f}),
        "s"(v2f_{-1.6666654611e-1f, 4.166664568298827e-2f}), "v"(0), "s"(v2f_{1.0f, -0.5f}), "v"(v2f_{0.0f, 1.0f}));
  const uint32_t t = p + 0x20000000u;                           // bits 31:30 = quadrant
  // swap (sn, cs) for odd quadrants with three-input bit operations (v_bitop3): a compare + two selects costs a lone
  // wavefront a VCC write -> read stall (s_nop 1) on top of its three instructions
  const uint32_t t1 = t << 1;
  const uint32_t m = (uint32_t)((int32_t)t1 >> 31);             // all ones when the quadrant is odd
  const uint32_t isn = __float_as_uint(sc.x), ics = __float_as_uint(sc.y);
  // bitop3 truth tables with a = 0xF0, b = 0xCC, c = 0xAA:  a ^ ((a ^ b) & c) = 0xD8,  a ^ (b & c) = 0x78
  const uint32_t ia = __builtin_amdgcn_bitop3_b32(ics, isn, m, 0xD8);   // q odd ? sn : cs
  const uint32_t ib = __builtin_amdgcn_bitop3_b32(isn, ics, m, 0xD8);   // q odd ? cs : sn
  v2f_ cs;
  cs.x = __uint_as_float(__builtin_amdgcn_bitop3_b32(ia, t ^ t1, 0x80000000u, 0x78));   // negate for q = 1, 2
  cs.y = __uint_as_float(__builtin_amdgcn_bitop3_b32(ib, t, 0x80000000u, 0x78));        // negate for q = 2, 3
  return cs;
}

// x * conj(ref), ref = (cos, sin):  re = fma(x.im, sin, x.re * cos),  im = fma(x.im, cos, -(x.re * sin))
// as two VOP3P instructions (operand halves picked with op_sel, the negation is neg_hi)
__device__ __forceinline__ v2f_ mix_conj(v2f_ x, v2f_ cs)
{
  v2f_ t, m;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(x), "v"(cs));            // (x.re cos, x.re sin)
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,0,1]" : "=v"(m) : "v"(x), "v"(cs), "v"(t));
  return m;
}

// x * ref, ref = (cos, sin):  re = fma(x.re, cos, -(x.im * sin)),  im = fma(x.re, sin, x.im * cos) -- the residual NCO of a
// precise channel (SPEC.md C2) -- as two VOP3P instructions, one statement (no s_nop between them)
__device__ __forceinline__ v2f_ mix_rot(v2f_ x, v2f_ cs)
{
  v2f_ t, m;
  asm("v_pk_mul_f32 %0, %2, %3 op_sel:[1,1] op_sel_hi:[1,0]\n\t"                              // (x.im sin, x.im cos)
      "v_pk_fma_f32 %1, %2, %3, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[0,0,1]"            // (x.re cos - t.lo, x.re sin + t.hi)
      : "=&v"(t), "=&v"(m) : "v"(x), "v"(cs));
  return m;
}

// D2: atan2 (radians), Cephes atanf kernel on min/max.
SD_HD float atan2_(float y, float x)
{
  const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
  const float mx = ax > ay ? ax : ay;
  const float mn = ax > ay ? ay : ax;
  if (mx == 0.0f) return 0.0f;
  float t  = mn / mx;
  float y0 = 0.0f;
  if (t > 0.4142135623730950f) {
    y0 = 0.78539816339744830962f;
    t  = (t - 1.0f) / (t + 1.0f);
  }
  const float z = t * t;
  float p = fma_(8.05374449538e-2f, z, -1.38776856032e-1f);
  p = fma_(p, z, 1.99777106478e-1f);
  p = fma_(p, z, -3.33329491539e-1f);
  float a = fma_(p * z, t, t) + y0;
  if (ay > ax)  a = 1.57079632679489661923f - a;
  if (x < 0.0f) a = 3.14159265358979323846f - a;
  if (y < 0.0f) a = -a;
  return a;
}

// D3: log2 of a normal positive float.
SD_HD float log2_(float x)
{
  const uint32_t bits = f2u(x);
  int32_t e = (int32_t)(bits >> 23) - 127;
  float   m = u2f((bits & 0x007FFFFFu) | 0x3F800000u);
  if (m > 1.41421356237309504880f) { m = m * 0.5f; e = e + 1; }
  const float f = m - 1.0f;
  const float z = f * f;
  float y = 7.0376836292e-2f;
  y = fma_(y, f, -1.1514610310e-1f);
  y = fma_(y, f,  1.1676998740e-1f);
  y = fma_(y, f, -1.2420140846e-1f);
  y = fma_(y, f,  1.4249322787e-1f);
  y = fma_(y, f, -1.6668057665e-1f);
  y = fma_(y, f,  2.0000714765e-1f);
  y = fma_(y, f, -2.4999993993e-1f);
  y = fma_(y, f,  3.3333331174e-1f);
  y = (y * f) * z;
  y = fma_(-0.5f, z, y);
  const float ln = f + y;
  return fma_(ln, 1.44269504088896341f, (float)e);
}

// D4: 2^x, x clamped to [-126, 126].
SD_HD float exp2_(float x)
{
  if (x >  126.0f) x =  126.0f;
  if (x < -126.0f) x = -126.0f;
  const float fl = __builtin_floorf(x + 0.5f);
  const int32_t n = (int32_t)fl;
  const float t = (x - fl) * 0.693147180559945309417f;
  const float z = t * t;
  float y = 1.9875691500e-4f;
  y = fma_(y, t, 1.3981999507e-3f);
  y = fma_(y, t, 8.3334519073e-3f);
  y = fma_(y, t, 4.1665795894e-2f);
  y = fma_(y, t, 1.6666665459e-1f);
  y = fma_(y, t, 5.0000001201e-1f);
  y = fma_(y, z, t) + 1.0f;
  return y * u2f((uint32_t)(n + 127) << 23);
}

SD_HD int32_t rad_to_dphase(float d)
{
  // clamp to (-pi, pi): one v_med3_f32 (identical to the two compares of the SPEC for non-NaN d)
#if defined(__HIP_DEVICE_COMPILE__)
  d = __builtin_amdgcn_fmed3f(d, -3.1415925f, 3.1415925f);
#else
  if (d > 3.1415925f) d = 3.1415925f;
  if (d < -3.1415925f) d = -3.1415925f;
#endif
  return (int32_t)(d * 683565275.57643158978f);
}

SD_HD float phase_to_rad(uint32_t p)
{
  return (float)(int32_t)p * 1.46291807926715968e-9f;
}

// (a.re + j a.im)(c + j s)
SD_HD c32 cmul_cs(c32 a, float c, float s)
{
  c32 r;
  r.re = fma_(-a.im, s, a.re * c);
  r.im = fma_( a.im, c, a.re * s);
  return r;
}

// a * conj(b)
SD_HD c32 cmul_conj(c32 a, c32 b)
{
  c32 r;
  r.re = fma_(a.im, b.im, a.re * b.re);
  r.im = fma_(a.im, b.re, -(a.re * b.im));
  return r;
}

SD_HD float sgn(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

// (sgn(a), sgn(b)) without compares: a v_cmp -> v_cndmask pair costs a lone wavefront about four
// dependent-op slots (VCC hazard) and the four pairs of a QPSK detector serialise on VCC.
// v * 2^126 * 2^126 is >= 1 in magnitude for every non-zero binary32 (denormals included; inf stays
// inf), so clamping to [-1, 1] gives exactly +-1, and 0 stays 0; the "+ 0.0f" of the fma turns a
// -0 into the +0 the compare form returns.  Bit-identical to sgn() for all non-NaN inputs.
__device__ __forceinline__ void sgn2(float a, float b, float &sa, float &sb)
{
  typedef float v2f __attribute__((ext_vector_type(2)));
  const v2f big = {8.5070591730234616e37f, 8.5070591730234616e37f};   // 2^126
  v2f s = v2f{a, b} * big;
  s = __builtin_elementwise_fma(s, big, v2f{0.0f, 0.0f});
  sa = __builtin_amdgcn_fmed3f(s.x, -1.0f, 1.0f);
  sb = __builtin_amdgcn_fmed3f(s.y, -1.0f, 1.0f);
}

// atan2_ without its two data-dependent branches, for the one-wavefront recurrences (a divergent `if` costs such a
// wavefront ~100 cycles of EXEC bookkeeping per sample whether or not a lane takes it): both range reductions are
// computed and selected.  The operations behind the selected values are atan2_'s, so are the bits (the unselected
// side may be inf / NaN: it is never used).
__device__ __forceinline__ float atan2_nb_(float y, float x)
{
  const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
  const float mx = ax > ay ? ax : ay;
  const float mn = ax > ay ? ay : ax;
  const float t0 = mn / mx;
  const bool hi = t0 > 0.4142135623730950f;
  const float t1 = (t0 - 1.0f) / (t0 + 1.0f);
  const float t = hi ? t1 : t0;
  const float y0 = hi ? 0.78539816339744830962f : 0.0f;
  const float z = t * t;
  float p = fma_(8.05374449538e-2f, z, -1.38776856032e-1f);
  p = fma_(p, z, 1.99777106478e-1f);
  p = fma_(p, z, -3.33329491539e-1f);
  float a = fma_(p * z, t, t) + y0;
  if (ay > ax)  a = 1.57079632679489661923f - a;
  if (x < 0.0f) a = 3.14159265358979323846f - a;
  if (y < 0.0f) a = -a;
  return mx == 0.0f ? 0.0f : a;
}

}  // namespace sd
