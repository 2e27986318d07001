/*
 * oracle/sdo.c -- CPU ORACLE (test infrastructure only; see sdo.h header comment).
 * "parity unpinned" vs upstream sigutils/suscan; [REF-PINNED] parts follow /root/reference.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -mfma -fPIC -shared  (see Makefile)
 */
#define _GNU_SOURCE                                      /* sincos(): see st_fft64 */
#include "sdo.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <assert.h>

#define SDO_PI 3.14159265358979323846

/* ===================================================================================== */
/* D. deterministic math primitives (SPEC.md section D)                                  */
/* ===================================================================================== */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float    u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* D1: phase (2^32 per turn) -> (cos, sin).  Quadrant reduction to |x| <= pi/4, then the
 * Cephes sinf/cosf minimax kernels; every operation is listed, all in binary32. */
void sdo_phasor_u32(uint32_t p, float *c, float *s)
{
  uint32_t q = (p + 0x20000000u) >> 30;                 /* nearest quadrant 0..3 */
  int32_t  r = (int32_t)(p - (q << 30));                /* residual in [-2^29, 2^29) */
  float x  = (float)r * 1.46291807926715968e-9f;        /* * 2 pi / 2^32 */
  float z  = x * x;
  float sp = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  sp = fmaf(sp, z, -1.6666654611e-1f);
  float sn = fmaf(sp * z, x, x);
  float cp = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  cp = fmaf(cp, z, 4.166664568298827e-2f);
  float cs = fmaf(cp * z, z, fmaf(z, -0.5f, 1.0f));
  switch (q) {
    case 0:  *c =  cs; *s =  sn; break;
    case 1:  *c = -sn; *s =  cs; break;
    case 2:  *c = -cs; *s = -sn; break;
    default: *c =  sn; *s = -cs; break;
  }
}

/* D2: atan2 in radians, Cephes atanf kernel on t = min/max in [0,1]. */
float sdo_atan2f(float y, float x)
{
  float ax = fabsf(x), ay = fabsf(y);
  float mx = ax > ay ? ax : ay;
  float mn = ax > ay ? ay : ax;
  float a;
  if (mx == 0.0f)
    return 0.0f;
  float t  = mn / mx;
  float y0 = 0.0f;
  if (t > 0.4142135623730950f) {                        /* tan(pi/8) */
    y0 = 0.78539816339744830962f;
    t  = (t - 1.0f) / (t + 1.0f);
  }
  float z = t * t;
  float p = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
  p = fmaf(p, z, 1.99777106478e-1f);
  p = fmaf(p, z, -3.33329491539e-1f);
  a = fmaf(p * z, t, t) + y0;
  if (ay > ax)  a = 1.57079632679489661923f - a;
  if (x < 0.0f) a = 3.14159265358979323846f - a;
  if (y < 0.0f) a = -a;
  return a;
}

/* D3: log2 for normal positive x (Cephes logf kernel). */
float sdo_log2f(float x)
{
  uint32_t bits = f2u(x);
  int32_t  e = (int32_t)(bits >> 23) - 127;
  float    m = u2f((bits & 0x007FFFFFu) | 0x3F800000u);  /* [1,2) */
  if (m > 1.41421356237309504880f) { m = m * 0.5f; e = e + 1; }
  float f = m - 1.0f;
  float z = f * f;
  float y = 7.0376836292e-2f;
  y = fmaf(y, f, -1.1514610310e-1f);
  y = fmaf(y, f,  1.1676998740e-1f);
  y = fmaf(y, f, -1.2420140846e-1f);
  y = fmaf(y, f,  1.4249322787e-1f);
  y = fmaf(y, f, -1.6668057665e-1f);
  y = fmaf(y, f,  2.0000714765e-1f);
  y = fmaf(y, f, -2.4999993993e-1f);
  y = fmaf(y, f,  3.3333331174e-1f);
  y = (y * f) * z;
  y = fmaf(-0.5f, z, y);
  float ln = f + y;
  return fmaf(ln, 1.44269504088896341f, (float)e);
}

/* D4: 2^x, x clamped to [-126, 126] (Cephes expf kernel). */
float sdo_exp2f(float x)
{
  if (x >  126.0f) x =  126.0f;
  if (x < -126.0f) x = -126.0f;
  float fl = floorf(x + 0.5f);
  int32_t n = (int32_t)fl;
  float t = (x - fl) * 0.693147180559945309417f;
  float z = t * t;
  float y = 1.9875691500e-4f;
  y = fmaf(y, t, 1.3981999507e-3f);
  y = fmaf(y, t, 8.3334519073e-3f);
  y = fmaf(y, t, 4.1665795894e-2f);
  y = fmaf(y, t, 1.6666665459e-1f);
  y = fmaf(y, t, 5.0000001201e-1f);
  y = fmaf(y, z, t) + 1.0f;
  return y * u2f((uint32_t)(n + 127) << 23);
}

uint32_t sdo_fnor_to_dphase(double fnor)
{
  /* normalised frequency fnor = 2 f / fs  ->  turns/sample = fnor / 2 */
  long long v = llrint(fnor * 2147483648.0);
  return (uint32_t)(v & 0xFFFFFFFFll);
}

/* radians -> signed phase step, |d| clamped below pi; truncation toward zero */
static inline int32_t rad_to_dphase(float d)
{
  if (d >  3.1415925f) d =  3.1415925f;
  if (d < -3.1415925f) d = -3.1415925f;
  return (int32_t)(d * 683565275.57643158978f);         /* 2^32 / 2 pi */
}

static inline float phase_to_rad(uint32_t p)
{
  return (float)(int32_t)p * 1.46291807926715968e-9f;   /* [-pi, pi) */
}

/* ===================================================================================== */
/* A3/A4/A9: reference-owned PSD post-processing  [REF-PINNED]                            */
/* ===================================================================================== */

/* SU_POWER_DB(p) = 10*log10(p + SUFLOAT_MIN_REF_MAG), SUFLOAT_MIN_REF_MAG = 1e-8
 * (upstream recollection; SU_LOG is log10 per Suscan/Library.cpp:115-119) */
static inline float su_power_db(float p)     { return 10.0f * log10f(p + 1e-8f); }
static inline float su_power_db_raw(float p) { return 10.0f * log10f(p); }

/* Suscan/Messages/PSDMessage.cpp:29-38 */
void sdo_psd_shift_db(float *psd, size_t n)
{
  size_t half = n / 2, i;
  for (i = 0; i < half; ++i) {
    float tmp = psd[i + half];
    psd[i + half] = su_power_db(psd[i]);
    psd[i] = su_power_db(tmp);
  }
}

/* Misc/Averager.cpp:25-50 */
int sdo_averager_feed(float *last, size_t *bufsiz, const float *x, size_t n, float alpha)
{
  int blend = alpha < 1.f;
  size_t i;
  if (*bufsiz != n) { *bufsiz = n; blend = 0; }
  if (blend) {
    for (i = 0; i < n; ++i)
      last[i] += alpha * (x[i] - last[i]);
    return 0;
  }
  memcpy(last, x, n * sizeof(float));
  return 1;
}

/* Default/GenericInspector/GenericInspector.cpp:231-247 */
void sdo_inspector_spectrum_db_shift(float *data, size_t len)
{
  size_t p = len / 2, i;
  for (i = 0; i < len; ++i)
    data[i] = su_power_db_raw(data[i] + 1e-20f);
  for (i = 0; i < len / 2; ++i) {
    float x = data[i];
    data[i] = data[p];
    data[p] = x;
    if (++p == len)
      p = 0;
  }
}

/* ===================================================================================== */
/* A2: windowed FFT power spectrum  [SPEC]                                                */
/* ===================================================================================== */

void sdo_window(int type, float *w, size_t n)
{
  size_t i;
  double d = (double)(n - 1);
  for (i = 0; i < n; ++i) {
    double t = 2.0 * SDO_PI * (double)i / d, v = 1.0;
    switch (type) {
      case SDO_WIN_HAMMING: v = 0.54 - 0.46 * cos(t); break;
      case SDO_WIN_HANN:    v = 0.5 - 0.5 * cos(t); break;
      case SDO_WIN_FLAT_TOP:
        v = 0.21557895 - 0.41663158 * cos(t) + 0.277263158 * cos(2 * t)
          - 0.083578947 * cos(3 * t) + 0.006947368 * cos(4 * t);
        break;
      case SDO_WIN_BLACKMANN_HARRIS:
        v = 0.35875 - 0.48829 * cos(t) + 0.14128 * cos(2 * t) - 0.01168 * cos(3 * t);
        break;
      default: v = 1.0;
    }
    w[i] = (float)v;
  }
}

void sdo_fft_f64(double *re, double *im, size_t n)
{
  /* iterative radix-2 DIT; twiddles for the last size used are cached per thread */
  static _Thread_local size_t tw_n = 0;                 /* per thread: the bench's CPU baseline and the full-size tests run ranges side by side */
  static _Thread_local double *twr = NULL, *twi = NULL;
  size_t i, j, len;
  if (tw_n != n) {
    free(twr); free(twi);
    twr = malloc(sizeof(double) * (n / 2 + 1));
    twi = malloc(sizeof(double) * (n / 2 + 1));
    for (i = 0; i < n / 2; ++i) {
      double ang = -2.0 * SDO_PI * (double)i / (double)n;
      twr[i] = cos(ang); twi[i] = sin(ang);
    }
    tw_n = n;
  }
  for (i = 1, j = 0; i < n; ++i) {                      /* bit reversal */
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) {
      double t = re[i]; re[i] = re[j]; re[j] = t;
      t = im[i]; im[i] = im[j]; im[j] = t;
    }
  }
  for (len = 2; len <= n; len <<= 1) {
    size_t half = len >> 1, k, step = n / len;
    for (i = 0; i < n; i += len) {
      for (k = 0; k < half; ++k) {
        double wr = twr[k * step], wi = twi[k * step];
        double ur = re[i + k], ui = im[i + k];
        double vr = re[i + k + half] * wr - im[i + k + half] * wi;
        double vi = re[i + k + half] * wi + im[i + k + half] * wr;
        re[i + k] = ur + vr; im[i + k] = ui + vi;
        re[i + k + half] = ur - vr; im[i + k + half] = ui - vi;
      }
    }
  }
}

void sdo_psd_frames(const sdo_c32 *x, size_t nframes, size_t n, size_t hop,
                    const float *window, size_t navg, float scale, float *out)
{
  double *re = malloc(sizeof(double) * n), *im = malloc(sizeof(double) * n);
  double *acc = malloc(sizeof(double) * n);
  size_t nout = nframes / navg, o, f, i;
  for (o = 0; o < nout; ++o) {
    for (i = 0; i < n; ++i) acc[i] = 0;
    for (f = 0; f < navg; ++f) {
      const sdo_c32 *fr = x + (o * navg + f) * hop;
      for (i = 0; i < n; ++i) {
        /* the window multiply is a binary32 product, as on the device */
        re[i] = (double)(fr[i].re * window[i]);
        im[i] = (double)(fr[i].im * window[i]);
      }
      sdo_fft_f64(re, im, n);
      for (i = 0; i < n; ++i) acc[i] += re[i] * re[i] + im[i] * im[i];
    }
    for (i = 0; i < n; ++i)
      out[o * n + i] = (float)(acc[i] * (double)scale / (double)navg);
  }
  free(re); free(im); free(acc);
}

/* ===================================================================================== */
/* T1/K4: NCO translate  [SPEC]  (sign conventions: Tasks/CarrierXlator.cpp:36-37,57-60)  */
/* ===================================================================================== */

static inline sdo_c32 cmul_spec(sdo_c32 a, float c, float s)
{
  /* (a.re + j a.im)(c + j s): re = a.re*c - a.im*s ; im = a.im*c + a.re*s */
  sdo_c32 r;
  r.re = fmaf(-a.im, s, a.re * c);
  r.im = fmaf( a.im, c, a.re * s);
  return r;
}

void sdo_xlate_bulk(const sdo_c32 *x, sdo_c32 *y, size_t len, uint32_t p0, uint32_t dp, uint64_t n0)
{
  size_t i;
  for (i = 0; i < len; ++i) {
    uint32_t p = p0 + (uint32_t)((n0 + i) * (uint64_t)dp);
    float c, s;
    sdo_phasor_u32(p, &c, &s);
    y[i] = cmul_spec(x[i], c, s);
  }
}

/* ===================================================================================== */
/* K4+K5: translate + decimating low-pass as a complex band-pass polyphase FIR  [SPEC]    */
/* ===================================================================================== */

void sdo_lpf_design(float *h, size_t ntaps, double fc)
{
  /* brick-wall low-pass (cut-off fc, 1 = Nyquist), Hamming window, unit DC gain */
  double *d = malloc(sizeof(double) * ntaps), sum = 0;
  size_t i;
  for (i = 0; i < ntaps; ++i) {
    double t = (double)i - 0.5 * (double)(ntaps - 1);
    double a = SDO_PI * fc * t;
    double sinc = fabs(a) < 1e-12 ? 1.0 : sin(a) / a;
    double w = ntaps > 1 ? 0.54 - 0.46 * cos(2.0 * SDO_PI * (double)i / (double)(ntaps - 1)) : 1.0;
    d[i] = fc * sinc * w;
    sum += d[i];
  }
  for (i = 0; i < ntaps; ++i) h[i] = (float)(d[i] / sum);
  free(d);
}

void sdo_chan_modulate_taps(const float *h, size_t ntaps, uint32_t dp, sdo_c32 *g)
{
  size_t k;
  for (k = 0; k < ntaps; ++k) {
    float c, s;
    sdo_phasor_u32((uint32_t)0 - (uint32_t)k * dp, &c, &s);
    g[k].re = h[k] * c;
    g[k].im = h[k] * s;
  }
}

size_t sdo_chan_feed(const sdo_c32 *hist, const sdo_c32 *x, size_t len, uint64_t n0,
                     const sdo_c32 *g, size_t ntaps, uint32_t D, uint32_t p0, uint32_t dp,
                     sdo_c32 *y)
{
  size_t i, k, nout = 0;
  for (i = 0; i < len; ++i) {
    uint64_t n = n0 + i;
    if (n % D) continue;
    float ar = 0.0f, ai = 0.0f;
    for (k = 0; k < ntaps; ++k) {
      /* sample n-k: index i-k into x, or into hist (hist[ntaps-2] is sample n0-1) */
      sdo_c32 v = (k <= i) ? x[i - k] : hist[(ntaps - 1) - (k - i)];
      ar = fmaf( g[k].re, v.re, ar);
      ar = fmaf(-g[k].im, v.im, ar);
      ai = fmaf( g[k].re, v.im, ai);
      ai = fmaf( g[k].im, v.re, ai);
    }
    float c, s;
    sdo_phasor_u32(p0 + (uint32_t)(n * (uint64_t)dp), &c, &s);
    sdo_c32 a = { ar, ai };
    y[nout++] = cmul_spec(a, c, s);
  }
  return nout;
}

/* ===================================================================================== */
/* T5/T7/T11 element-wise demodulators                                                    */
/* ===================================================================================== */

static inline sdo_c32 cmul_conj(sdo_c32 a, sdo_c32 b)    /* a * conj(b) */
{
  sdo_c32 r;
  r.re = fmaf(a.im, b.im, a.re * b.re);
  r.im = fmaf(a.im, b.re, -(a.re * b.im));
  return r;
}

/* Tasks/QuadDemodTask.cpp:44-60: dest[p] = SU_I * (1/pi) * arg(x * conj(prev)) */
void sdo_quad_demod(const sdo_c32 *x, sdo_c32 *y, size_t len, sdo_c32 prev, int first)
{
  const float k = (float)(1. / SDO_PI);
  size_t p;
  for (p = 0; p < len; ++p) {
    if (p < 1 && first) {
      y[p].re = 0; y[p].im = 0;
    } else {
      sdo_c32 d = cmul_conj(x[p], prev);
      y[p].re = 0.0f;
      y[p].im = k * sdo_atan2f(d.im, d.re);
    }
    prev = x[p];
  }
}

/* Tasks/DelayedConjTask.cpp:70-84: the circular delay line holds x[p-delay] */
void sdo_delayed_conj(const sdo_c32 *x, sdo_c32 *y, size_t len, size_t delay)
{
  size_t p;
  for (p = 0; p < len; ++p) {
    if (p < delay) {
      y[p].re = 0; y[p].im = 0;
    } else {
      sdo_c32 prev = x[p - delay];
      float mag  = sqrtf(fmaf(prev.re, prev.re, prev.im * prev.im));
      float kinv = 1.0f / (mag + 1e-3f);
      sdo_c32 d = cmul_conj(x[p], prev);
      y[p].re = kinv * d.re;
      y[p].im = kinv * d.im;
    }
  }
}

/* Tasks/HistogramFeeder.cpp:45-66 */
size_t sdo_histogram_feed(const sdo_c32 *x, size_t len, int space, float *out)
{
  size_t p, q = 0;
  switch (space) {
    case 0:
      for (p = 0; p < len; ++p)
        out[q++] = sqrtf(fmaf(x[p].re, x[p].re, x[p].im * x[p].im));
      break;
    case 1:
      for (p = 0; p < len; ++p)
        out[q++] = sdo_atan2f(x[p].im, x[p].re);
      break;
    default:
      for (p = 0; p < len; ++p)
        if (p > 0) {
          sdo_c32 d = cmul_conj(x[p], x[p - 1]);
          out[q++] = sdo_atan2f(d.im, d.re);
        }
  }
  return q;
}

/* ===================================================================================== */
/* IIR helper + Butterworth design                                                        */
/* ===================================================================================== */

void sdo_butter_lp(int order, double fc, float *b, float *a)
{
  /* analog Butterworth prototype, pre-warped, bilinear transform (T = 2).
   * poles p_i = wc * exp(j pi (2i + n + 1) / (2n)); digital pole z_i = (1 + p_i)/(1 - p_i) */
  double wc = tan(0.5 * SDO_PI * fc);
  double ar[SDO_IIR_MAX_ORDER + 1] = { 1 }, ai[SDO_IIR_MAX_ORDER + 1] = { 0 };
  int i, k, n = order;
  assert(order >= 0 && order <= SDO_IIR_MAX_ORDER);
  for (i = 0; i < n; ++i) {
    double th = SDO_PI * (2.0 * i + n + 1.0) / (2.0 * n), cth, sth;
    sincos(th, &sth, &cth);                               /* explicitly (see st_fft64): the coefficients are bit-pinned */
    double pr = wc * cth, pi = wc * sth;
    /* z = (1 + p) / (1 - p) */
    double dr = 1.0 - pr, di = -pi, nr = 1.0 + pr, ni = pi;
    double den = dr * dr + di * di;
    double zr = (nr * dr + ni * di) / den, zi = (ni * dr - nr * di) / den;
    /* poly *= (1 - z q) with q = z^-1 */
    for (k = i + 1; k >= 1; --k) {
      double tr = ar[k] - (zr * ar[k - 1] - zi * ai[k - 1]);
      double ti = ai[k] - (zr * ai[k - 1] + zi * ar[k - 1]);
      ar[k] = tr; ai[k] = ti;
    }
  }
  /* numerator (1 + q)^n scaled to unit DC gain */
  double bn[SDO_IIR_MAX_ORDER + 1] = { 1 }, sa = 0, sb = 0;
  for (i = 0; i < n; ++i)
    for (k = i + 1; k >= 1; --k)
      bn[k] += bn[k - 1];
  for (k = 0; k <= n; ++k) { sa += ar[k]; sb += bn[k]; }
  for (k = 0; k <= SDO_IIR_MAX_ORDER; ++k) { b[k] = 0; a[k] = 0; }
  for (k = 0; k <= n; ++k) {
    b[k] = (float)(bn[k] * sa / sb);
    a[k] = (float)ar[k];
  }
}

static inline sdo_c32 iir_feed(sdo_iir *f, sdo_c32 x)
{
  /* direct form I; the history part is summed first so that the new sample enters last */
  float tr = 0.0f, ti = 0.0f;
  int i;
  sdo_c32 y;
  for (i = f->order; i >= 1; --i) {
    tr = fmaf(f->b[i], f->xh[i].re, tr);
    ti = fmaf(f->b[i], f->xh[i].im, ti);
  }
  for (i = f->order; i >= 1; --i) {
    tr = fmaf(-f->a[i], f->yh[i].re, tr);
    ti = fmaf(-f->a[i], f->yh[i].im, ti);
  }
  y.re = fmaf(f->b[0], x.re, tr);
  y.im = fmaf(f->b[0], x.im, ti);
  for (i = f->order; i >= 2; --i) { f->xh[i] = f->xh[i - 1]; f->yh[i] = f->yh[i - 1]; }
  if (f->order >= 1) { f->xh[1] = x; f->yh[1] = y; }
  return y;
}

/* ===================================================================================== */
/* K6: Costas loop [SPEC] (init: Tasks/CostasRecoveryTask.cpp:36-41; feed: :58-61)         */
/* ===================================================================================== */

int sdo_costas_init(sdo_costas *c, int kind, float fhint, float arm_bw, unsigned arm_order, float loop_bw)
{
  memset(c, 0, sizeof *c);
  if (kind < SDO_COSTAS_BPSK || kind > SDO_COSTAS_8PSK) return 0;
  if (arm_order == 0) arm_order = 1;
  if (arm_order - 1 > SDO_IIR_MAX_ORDER) return 0;
  c->kind  = kind;
  c->a     = (float)(SDO_PI * (double)loop_bw);          /* SU_NORM2ANG_FREQ(loop_bw) */
  c->b     = 0.5f * c->a * c->a;
  c->gain  = 1.0f;
  c->omega = (float)(SDO_PI * (double)fhint);
  c->phase = 0;
  c->af.order = (int)arm_order - 1;
  sdo_butter_lp(c->af.order, (double)arm_bw, c->af.b, c->af.a);
  return 1;
}

static inline float sgnf(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

sdo_c32 sdo_costas_feed(sdo_costas *c, sdo_c32 x)
{
  float cs, sn, e;
  sdo_c32 m, z;
  sdo_phasor_u32(c->phase, &cs, &sn);
  /* mix: x * conj(ref) */
  m.re = fmaf(x.im, sn, x.re * cs);
  m.im = fmaf(x.im, cs, -(x.re * sn));
  z = iir_feed(&c->af, m);
  z.re = c->gain * z.re;
  z.im = c->gain * z.im;
  switch (c->kind) {
    case SDO_COSTAS_BPSK:
      e = z.re * z.im;
      break;
    case SDO_COSTAS_QPSK:
      e = sgnf(z.re) * z.im - sgnf(z.im) * z.re;
      break;
    default: /* 8PSK */
      if (fabsf(z.re) >= fabsf(z.im))
        e = sgnf(z.re) * z.im - (sgnf(z.im) * z.re) * 0.41421356237309504880f;
      else
        e = (sgnf(z.re) * z.im) * 0.41421356237309504880f - sgnf(z.im) * z.re;
  }
  /* phi[n+1] = phi[n] + omega[n] + a e ;  omega[n+1] = omega[n] + b e */
  float dphi = fmaf(c->a, e, c->omega);
  c->omega   = fmaf(c->b, e, c->omega);
  c->phase  += (uint32_t)rad_to_dphase(dphi);
  return z;
}

void sdo_costas_feed_bulk(sdo_costas *c, const sdo_c32 *x, sdo_c32 *y, size_t len)
{
  size_t i;
  for (i = 0; i < len; ++i) y[i] = sdo_costas_feed(c, x[i]);
}

/* ===================================================================================== */
/* K7: PLL [SPEC] (init: Tasks/PLLSyncTask.cpp:36; track: :53-56)                          */
/* ===================================================================================== */

int sdo_pll_init(sdo_pll *p, float fhint, float fc)
{
  double w = SDO_PI * (double)fc;
  double dinv = 1.0 / (1.0 + 2.0 * 0.707 * w + w * w);
  memset(p, 0, sizeof *p);
  p->alpha = (float)(4.0 * w * w * dinv);
  p->beta  = (float)(4.0 * 0.707 * w * dinv);
  p->omega = (float)(SDO_PI * (double)fhint);
  return 1;
}

sdo_c32 sdo_pll_track(sdo_pll *p, sdo_c32 x)
{
  float cs, sn;
  sdo_c32 m;
  sdo_phasor_u32(p->phase, &cs, &sn);
  m.re = fmaf(x.im, sn, x.re * cs);
  m.im = fmaf(x.im, cs, -(x.re * sn));
  float err = sdo_atan2f(x.im, x.re) - phase_to_rad(p->phase);
  if (err >  3.14159265358979323846f) err -= 6.28318530717958647692f;
  if (err < -3.14159265358979323846f) err += 6.28318530717958647692f;
  float dphi = fmaf(p->beta, err, p->omega);
  p->omega   = fmaf(p->alpha, err, p->omega);
  p->phase  += (uint32_t)rad_to_dphase(dphi);
  return m;
}

void sdo_pll_track_bulk(sdo_pll *p, const sdo_c32 *x, sdo_c32 *y, size_t len)
{
  size_t i;
  for (i = 0; i < len; ++i) y[i] = sdo_pll_track(p, x[i]);
}

/* ===================================================================================== */
/* K8: Gardner clock recovery [SPEC] (Tasks/WaveSampler.cpp:60-65, :177-213)               */
/* ===================================================================================== */

int sdo_clock_init(sdo_clock *cd, float loop_gain, float bhint)
{
  memset(cd, 0, sizeof *cd);
  if (!(bhint > 0.0f)) return -1;
  cd->alpha = 2e-1f;                                     /* SU_PREFERED_CLOCK_ALPHA */
  cd->beta  = 1.2e-4f;                                   /* SU_PREFERED_CLOCK_BETA  */
  cd->gain  = loop_gain;
  cd->phi   = 0.25f;
  cd->bnor  = bhint;
  cd->bmin  = 0.5f * bhint;
  cd->bmax  = bhint > 0.5f ? 1.0f : 2.0f * bhint;
  return 0;
}

size_t sdo_clock_feed_bulk(sdo_clock *cd, const sdo_c32 *x, size_t len, sdo_c32 *out)
{
  size_t i, n = 0;
  for (i = 0; i < len; ++i) {
    sdo_c32 v = x[i];
    cd->phi = cd->phi + cd->bnor;
    if (cd->phi >= 0.5f) {
      /* the half-symbol instant fell mu samples before v, 0 <= mu < 1 */
      float mu = (cd->phi - 0.5f) / cd->bnor;
      sdo_c32 p;
      p.re = fmaf(mu, cd->prev.re - v.re, v.re);
      p.im = fmaf(mu, cd->prev.im - v.im, v.im);
      cd->phi = cd->phi - 0.5f;
      cd->halfcycle = !cd->halfcycle;
      if (!cd->halfcycle) {
        cd->x2 = cd->x0;
        cd->x0 = p;
        float dr = cd->x0.re - cd->x2.re, di = cd->x0.im - cd->x2.im;
        float e  = cd->gain * fmaf(cd->x1.im, di, cd->x1.re * dr);
        cd->phi  = fmaf(cd->alpha, e, cd->phi);
        float b  = fmaf(cd->beta, e, cd->bnor);
        if (b < cd->bmin) b = cd->bmin;
        if (b > cd->bmax) b = cd->bmax;
        cd->bnor = b;
        out[n++] = p;
      } else {
        cd->x1 = p;
      }
    }
    cd->prev = v;
  }
  return n;
}

/* ===================================================================================== */
/* K9: AGC [SPEC] (Tasks/AGCTask.cpp:22-28,41-53,70-73)                                    */
/* ===================================================================================== */

const sdo_agc_params sdo_agc_params_default = { -100.f, 6.f, 100, 20, 20, 2.f, 4.f, 20.f, 40.f };

void sdo_agc_params_from_tau(sdo_agc_params *p, float tau)
{
  /* Tasks/AGCTask.cpp:22-28,43-47: DELAY_LINE_FRAC / MAG_HISTORY_FRAC are defined but
   * never applied, so the INITIALIZER's delay-line and history sizes stay */
  const double rise = 2 * 3.9062e-1;
  *p = sdo_agc_params_default;
  p->fast_rise_t = (float)(tau * rise);
  p->fast_fall_t = (float)(tau * 2 * rise);
  p->slow_rise_t = (float)(tau * 10 * rise);
  p->slow_fall_t = (float)(tau * 10 * 2 * rise);
  p->hang_max    = (unsigned)(tau * rise * 5);
}

int sdo_agc_init(sdo_agc *agc, const sdo_agc_params *p)
{
  memset(agc, 0, sizeof *agc);
  if (p->delay_line_size == 0 || p->delay_line_size > SDO_AGC_MAX_HIST) return 0;
  if (p->mag_history_size == 0 || p->mag_history_size > SDO_AGC_MAX_HIST) return 0;
  agc->knee        = p->threshold;
  agc->gain_slope  = p->slope_factor * 1e-2f;
  agc->fixed_gain  = 0;
  agc->hang_max    = p->hang_max;
  agc->delay_line_size  = p->delay_line_size;
  agc->mag_history_size = p->mag_history_size;
  agc->fast_alpha_rise = (float)(1.0 - exp(-1.0 / (double)p->fast_rise_t));
  agc->fast_alpha_fall = (float)(1.0 - exp(-1.0 / (double)p->fast_fall_t));
  agc->slow_alpha_rise = (float)(1.0 - exp(-1.0 / (double)p->slow_rise_t));
  agc->slow_alpha_fall = (float)(1.0 - exp(-1.0 / (double)p->slow_fall_t));
  return 1;
}

sdo_c32 sdo_agc_feed(sdo_agc *agc, sdo_c32 x)
{
  unsigned i;
  sdo_c32 xd = agc->delay_line[agc->delay_ptr];
  agc->delay_line[agc->delay_ptr] = x;
  if (++agc->delay_ptr == agc->delay_line_size) agc->delay_ptr = 0;

  /* magnitude in dB: 10 log10(|x|^2 + 1e-8) */
  float pw   = fmaf(x.re, x.re, x.im * x.im) + 1e-8f;
  float x_db = 3.01029995663981195f * sdo_log2f(pw);
  float x_db_old = agc->mag_history[agc->hist_ptr];
  agc->mag_history[agc->hist_ptr] = x_db;
  if (++agc->hist_ptr == agc->mag_history_size) agc->hist_ptr = 0;

  if (agc->peak < x_db) {
    agc->peak = x_db;
  } else if (agc->peak == x_db_old) {
    float pk = -160.0f;                                  /* SUFLOAT_MIN_REF_DB */
    for (i = 0; i < agc->mag_history_size; ++i)
      if (pk < agc->mag_history[i]) pk = agc->mag_history[i];
    agc->peak = pk;
  }

  float d = agc->peak - agc->fast_level;
  agc->fast_level = fmaf(d > 0.0f ? agc->fast_alpha_rise : agc->fast_alpha_fall, d, agc->fast_level);

  d = agc->peak - agc->slow_level;
  if (d > 0.0f) {
    agc->slow_level = fmaf(agc->slow_alpha_rise, d, agc->slow_level);
    agc->hang_n = 0;
  } else if (agc->hang_n >= agc->hang_max) {
    agc->slow_level = fmaf(agc->slow_alpha_fall, d, agc->slow_level);
  } else {
    ++agc->hang_n;
  }

  float lvl  = agc->fast_level > agc->slow_level ? agc->fast_level : agc->slow_level;
  if (lvl < agc->knee) lvl = agc->knee;
  float g_db = lvl * (agc->gain_slope - 1.0f);
  float g    = sdo_exp2f(g_db * 0.166096404744368117f) * 0.7f;   /* 10^(g_db/20) * SU_AGC_RESCALE */
  sdo_c32 y = { xd.re * g, xd.im * g };
  return y;
}

void sdo_agc_feed_bulk(sdo_agc *agc, const sdo_c32 *x, sdo_c32 *y, size_t len)
{
  size_t i;
  for (i = 0; i < len; ++i) y[i] = sdo_agc_feed(agc, x[i]);
}

/* ===================================================================================== */
/* T8: WaveSampler manual mode [REF-PINNED]  Tasks/WaveSampler.cpp:45-51, :96-175          */
/* ===================================================================================== */

void sdo_sample_manual(const sdo_c32 *data, size_t length, double symbol_count,
                       double symbol_sync, int space, sdo_c32 *out, size_t nout)
{
  double delta = (double)length / symbol_count;          /* :45 */
  double sampOffset = symbol_sync / delta;               /* :46 */
  float deltaInv = 1.f / (float)delta;                   /* :106 */
  sdo_c32 prev = { 0, 0 }, x = { 0, 0 };
  long p;
  for (p = 0; p < (long)nout; ++p) {
    double start = ((double)p - sampOffset) * delta + symbol_sync;   /* :115 */
    double end = start + delta;
    float ar = 0, ai = 0;
    long long iStart = (long long)floor(start);
    long long iEnd   = (long long)ceil(end);
    float tStart = (float)(1 - (start - (double)iStart));
    float tEnd   = (float)(1 - ((double)iEnd - end));
    long long i;
    for (i = iStart; i <= iEnd; ++i) {
      if (i >= 0 && i < (long long)length) {
        if (i == iStart)      { x.re = tStart * data[i].re; x.im = tStart * data[i].im; }
        else if (i == iEnd)   { x.re = tEnd * data[i].re;   x.im = tEnd * data[i].im; }
        else                  x = data[i];
      } else {
        x.re = 0; x.im = 0;
      }
      if (space == 0) {                                  /* AMPLITUDE: avg += x conj(x) */
        ar = ar + fmaf(x.im, x.im, x.re * x.re);
      } else {                                           /* PHASE / FREQUENCY: x conj(prev) */
        sdo_c32 d = cmul_conj(x, prev);
        ar = ar + d.re; ai = ai + d.im;
      }
      prev = x;
    }
    if (space == 0) { out[p].re = sqrtf(deltaInv * ar); out[p].im = 0; }
    else            { out[p].re = deltaInv * ar; out[p].im = deltaInv * ai; }
  }
}

/* ===================================================================================== */
/* A7 stages [UPSTREAM-RECOLLECTION]: matched filter, fixed gain, CMA equalizer           */
/* ===================================================================================== */
size_t sdo_rrc_ntaps(double sps) { return 2 * (size_t)ceil(3.0 * sps) + 1; }

void sdo_rrc_design(float *h, size_t ntaps, double sps, double beta)
{
  double *d = malloc(sizeof(double) * ntaps), sum = 0;
  size_t i;
  for (i = 0; i < ntaps; ++i) {
    double t = ((double)i - 0.5 * (double)(ntaps - 1)) / sps;          /* in symbols */
    double v;
    if (fabs(t) < 1e-12) {
      v = 1.0 - beta + 4.0 * beta / SDO_PI;
    } else if (beta > 0 && fabs(fabs(4.0 * beta * t) - 1.0) < 1e-9) {
      v = beta / sqrt(2.0) * ((1.0 + 2.0 / SDO_PI) * sin(SDO_PI / (4.0 * beta)) + (1.0 - 2.0 / SDO_PI) * cos(SDO_PI / (4.0 * beta)));
    } else {
      double a = SDO_PI * t;
      v = (sin(a * (1.0 - beta)) + 4.0 * beta * t * cos(a * (1.0 + beta))) / (a * (1.0 - 16.0 * beta * beta * t * t));
    }
    d[i] = v;
    sum += v;
  }
  for (i = 0; i < ntaps; ++i) h[i] = (float)(d[i] / sum);
  free(d);
}

void sdo_fir_feed(sdo_c32 *hist, const float *h, size_t ntaps, const sdo_c32 *x, size_t len, sdo_c32 *y)
{
  size_t m, k, hl = ntaps - 1;
  for (m = 0; m < len; ++m) {
    float yr = 0, yi = 0;
    for (k = 0; k < ntaps; ++k) {
      sdo_c32 v = (k <= m) ? x[m - k] : hist[hl + m - k];    /* sample m-k of [hist ; x] */
      yr = fmaf(h[k], v.re, yr);
      yi = fmaf(h[k], v.im, yi);
    }
    y[m] = (sdo_c32){ yr, yi };
  }
  if (hl) {                                                   /* last hl samples of [hist ; x] */
    sdo_c32 *nh = malloc(sizeof(sdo_c32) * hl);
    for (k = 0; k < hl; ++k) {
      long src = (long)k + (long)len;                         /* index into [hist ; x] */
      nh[k] = src < (long)hl ? hist[src] : x[src - (long)hl];
    }
    memcpy(hist, nh, sizeof(sdo_c32) * hl);
    free(nh);
  }
}

void sdo_scale(const sdo_c32 *x, size_t len, float g, sdo_c32 *y)
{
  size_t i;
  for (i = 0; i < len; ++i) { y[i].re = g * x[i].re; y[i].im = g * x[i].im; }
}

void sdo_cma_init(sdo_cma *q, int n, float mu)
{
  memset(q, 0, sizeof *q);
  q->n = n < 1 ? 1 : (n > SDO_CMA_MAX ? SDO_CMA_MAX : n);
  q->mu = mu;
  q->w[0].re = 1.0f;
}

sdo_c32 sdo_cma_feed(sdo_cma *q, sdo_c32 x)
{
  int i;
  float yr = 0, yi = 0;
  for (i = q->n - 1; i > 0; --i) q->d[i] = q->d[i - 1];
  q->d[0] = x;
  for (i = 0; i < q->n; ++i) {                               /* y = sum w[i] d[i] */
    yr = fmaf(q->w[i].re, q->d[i].re, yr); yr = fmaf(-q->w[i].im, q->d[i].im, yr);
    yi = fmaf(q->w[i].re, q->d[i].im, yi); yi = fmaf(q->w[i].im, q->d[i].re, yi);
  }
  if (!q->locked) {                                          /* w[i] -= mu (|y|^2 - 1) y conj(d[i]) */
    float g = fmaf(yi, yi, yr * yr) - 1.0f;
    sdo_c32 e = { yr * g, yi * g };
    for (i = 0; i < q->n; ++i) {
      sdo_c32 t = cmul_conj(e, q->d[i]);
      q->w[i].re = fmaf(-q->mu, t.re, q->w[i].re);
      q->w[i].im = fmaf(-q->mu, t.im, q->w[i].im);
    }
  }
  return (sdo_c32){ yr, yi };
}

void sdo_cma_feed_bulk(sdo_cma *q, const sdo_c32 *x, size_t len, sdo_c32 *y)
{
  size_t i;
  for (i = 0; i < len; ++i) y[i] = sdo_cma_feed(q, x[i]);
}

static inline sdo_c32 csq(sdo_c32 a)                        /* a * a, translate-product form (SPEC C) */
{
  sdo_c32 r = { fmaf(-a.im, a.im, a.re * a.re), fmaf(a.im, a.re, a.re * a.im) };
  return r;
}

void sdo_spectsrc_preproc(int kind, const sdo_c32 *x, size_t len, sdo_c32 prev0, sdo_c32 *y)
{
  size_t i;
  sdo_c32 prev = prev0;
  for (i = 0; i < len; ++i) {
    sdo_c32 v = x[i], r = { 0, 0 }, d;
    switch (kind) {
      case 1: r = v; break;
      case 2: r = cmul_conj(v, prev); break;
      case 3: d = cmul_conj(v, prev); r.re = sdo_atan2f(d.im, d.re); break;
      case 4: r.re = sdo_atan2f(v.im, v.re); break;
      case 5: r.re = v.re - prev.re; r.im = v.im - prev.im; break;
      case 6: d.re = v.re - prev.re; d.im = v.im - prev.im; r.re = sqrtf(fmaf(d.im, d.im, d.re * d.re)); break;
      case 7: r = csq(v); break;
      case 8: r = csq(csq(v)); break;
      case 9: r = csq(csq(csq(v))); break;
      default: break;
    }
    y[i] = r;
    prev = v;
  }
}

/* ===================================================================================== */
/* section 8f #3: decision space / decider / histogram / SNR estimator                      */
/* ===================================================================================== */
static float dec_value(sdo_c32 x, int mode)
{
  return mode == 0 ? sqrtf(fmaf(x.im, x.im, x.re * x.re)) : sdo_atan2f(x.im, x.re);
}

void sdo_decision_space(const sdo_c32 *x, size_t len, int mode, float *out)
{
  size_t i;
  for (i = 0; i < len; ++i)                                  /* :864-871 */
    out[i] = mode == 0 ? sqrtf(fmaf(x[i].im, x[i].im, x[i].re * x[i].re))
                       : (float)((double)sdo_atan2f(x[i].re, -x[i].im) / SDO_PI);      /* arg(j x) / PI */
}

void sdo_decide(const sdo_c32 *x, size_t len, int mode, unsigned bps, float vmin, float vmax, unsigned char *sym)
{
  size_t i;
  const int intervals = 1 << bps;
  const float d = (vmax - vmin) / (float)intervals;
  for (i = 0; i < len; ++i) {
    int s = (int)floorf((dec_value(x[i], mode) - vmin) / d);
    sym[i] = (unsigned char)(s < 0 ? 0 : (s > intervals - 1 ? intervals - 1 : s));
  }
}

void sdo_symbol_histogram(const sdo_c32 *x, size_t len, int mode, float vmin, float vmax, unsigned nbins, unsigned *hist)
{
  size_t i;
  const float d = (vmax - vmin) / (float)nbins;
  for (i = 0; i < len; ++i) {
    int b = (int)floorf((dec_value(x[i], mode) - vmin) / d);
    if (b >= 0 && b < (int)nbins) ++hist[b];
  }
}

void sdo_snr_init(sdo_snr *e, unsigned bps, float alpha)
{
  memset(e, 0, sizeof *e);
  e->sigma = 1.f / 8.f;                                      /* SNR_ESTIMATOR_DEFAULT_SIGMA */
  e->alpha = alpha;
  e->bps = bps;
  e->intervals = 1u << bps;                                  /* setBps :134-142 */
}

void sdo_snr_feed(sdo_snr *e, const unsigned *history, unsigned length, float *Hi)
{
  unsigned i, j, max = 0;
  float *gaussian = malloc(sizeof(float) * length), *Htilde = malloc(sizeof(float) * length);
  e->length = length;
  e->hx = 1.f / length;                                      /* feed :147-153 */
  for (i = 0; i < length; ++i) if (max < history[i]) max = history[i];
  if (max == 0) max = 1;
  for (i = 0; i < length; ++i) Htilde[i] = (float)history[i] / max;
  if (length > 0 && e->intervals > 0) {                      /* iterate() :83-117 */
    float delta = 0, x, term, intlen, start, skip, mx = 0;
    float sigmainv = 1.f / e->sigma, sigma3inv = sigmainv * sigmainv * sigmainv, sigma2 = e->sigma * e->sigma;
    /* recalculateModel() :30-80 */
    for (i = 0; i < length; ++i) {
      x = i * e->hx;
      if (x >= .5f) x -= 1.f;
      gaussian[i] = expf(-x * x / sigma2);
    }
    intlen = 1.f / e->intervals;
    start = .5f * intlen;
    for (i = 0; i < length; ++i) Hi[i] = 0.f;
    for (j = 0; j < e->intervals; ++j) {
      float sk = start + j * intlen;
      float t = 1.f - (sk - floorf(sk));
      unsigned skipint = (unsigned)floorf(length * sk), i1, i2;
      for (i = 0; i < length; ++i) {
        i1 = (unsigned)(length + i - skipint) % length;
        i2 = (unsigned)(length + i1 - 1) % length;
        Hi[i] += t * gaussian[i1];
        Hi[i] += (1 - t) * gaussian[i2];
      }
    }
    for (i = 0; i < length; ++i) if (Hi[i] > mx) mx = Hi[i];
    if (mx > 0.f) for (i = 0; i < length; ++i) Hi[i] /= mx;
    /* back in iterate() */
    for (i = 0; i < length; ++i) {
      x = i * e->hx;
      if (x >= .5f) x -= 1.f;
      term = 0;
      for (j = 0; j < e->intervals; ++j) { skip = start + j * intlen; term += (x - skip) * (x - skip); }
      term *= (Hi[i] - Htilde[i]) / sigma3inv;
      delta += term;
    }
    e->delta = delta / length;
    e->sigma += -e->alpha * e->delta;
    e->sqerr = 0;                                            /* calculateSquareError() :119-131 */
    for (i = 0; i < length; ++i) { float er = (Hi[i] - Htilde[i]) * (Hi[i] - Htilde[i]); e->sqerr += er * er; }
  }
  free(gaussian); free(Htilde);
}

float sdo_snr_get(const sdo_snr *e) { return 1.f / (e->intervals * e->sigma); }

void sdo_ingest_iq(int format, const void *raw, size_t n, sdo_c32 *out)
{
  size_t i;
  switch (format) {
    case 1: memcpy(out, raw, n * sizeof *out); break;
    case 2: { const unsigned char *r = raw;
      for (i = 0; i < n; ++i) { out[i].re = (float)((int)r[2 * i] - 128) * 0.0078125f; out[i].im = (float)((int)r[2 * i + 1] - 128) * 0.0078125f; } break; }
    case 3: { const signed char *r = raw;
      for (i = 0; i < n; ++i) { out[i].re = (float)r[2 * i] * 0.0078125f; out[i].im = (float)r[2 * i + 1] * 0.0078125f; } break; }
    case 4: { const short *r = raw;
      for (i = 0; i < n; ++i) { out[i].re = (float)r[2 * i] * 3.0517578125e-05f; out[i].im = (float)r[2 * i + 1] * 3.0517578125e-05f; } break; }
    default: break;
  }
}

void sdo_power_init(sdo_power *p, unsigned long long max_samples) { p->acc = p->c = 0; p->count = 0; p->max_samples = max_samples; }

size_t sdo_power_feed(sdo_power *p, const sdo_c32 *x, size_t len, sdo_c32 *out)
{
  size_t i, k = 0;
  for (i = 0; i < len; ++i) {
    const double input = (double)(x[i].re * x[i].re + x[i].im * x[i].im);      /* SU_C_REAL(x conj(x)): two binary32 products, one sum */
    const double y = input - p->c, t = p->acc + y;                             /* RMSInspector.cpp:551-556 */
    p->c = (t - p->acc) - y;
    p->acc = t;
    ++p->count;
    if (p->count >= p->max_samples) {                                          /* checkMaxSamples, :327-338 */
      out[k].re = (float)(p->acc / (double)p->count); out[k].im = 0.f; ++k;
      p->c = p->acc = 0; p->count = 0;
    }
  }
  return k;
}

float sdo_baud_nonlinear(const sdo_c32 *x, size_t n)
{
  sdo_c32 *y = malloc(n * sizeof *y);
  double *re = malloc(sizeof(double) * n), *im = malloc(sizeof(double) * n);
  float *P = malloc(sizeof(float) * n);
  size_t t;
  int k, half = (int)(n / 2), skip = (int)(0.01 * (double)n), first = -1;
  float pmax = 0.f, thr;
  double c = 0.0;
  if (skip < 4) skip = 4;
  y[0].re = y[0].im = 0.f;
  for (t = 1; t < n; ++t) {
    const float dr = x[t].re - x[t - 1].re, di = x[t].im - x[t - 1].im;
    y[t].re = fmaf(dr, dr, di * di);
    y[t].im = 0.f;
  }
  sdo_blackmann_harris_complex(y, n);
  for (t = 0; t < n; ++t) { re[t] = y[t].re; im[t] = y[t].im; }
  sdo_fft_f64(re, im, n);
  for (k = 0; k < half; ++k) { const float a = (float)re[k], b = (float)im[k]; P[k] = fmaf(a, a, b * b); }
  {
    double sum = 0;
    for (k = skip; k < half; ++k) { if (P[k] > pmax) pmax = P[k]; sum += (double)P[k]; }
    thr = (double)pmax * (double)(half - skip) >= 20.0 * sum ? 0.5f * pmax : 0.f;   /* a line, not the tallest noise bin */
  }
  for (k = skip + 1; k < half - 1 && first < 0; ++k)
    if (P[k] >= thr && P[k] >= P[k - 1] && P[k] >= P[k + 1]) first = k;
  if (thr > 0.f && first >= 0) {
    double num = 0, den = 0;
    for (k = first - 4; k <= first + 4; ++k) {
      if (k < skip || k >= half) continue;
      num += (double)P[k] * (double)k; den += (double)P[k];
    }
    c = den > 0 ? num / den : 0.0;
  }
  free(y); free(re); free(im); free(P);
  return (float)(c / (double)n);
}

float sdo_fac_first_valley(const float *R, size_t H)
{
  const float thr = 0.25f * R[0];
  size_t l;
#define SDO_S3(l_) ((R[(l_) - 1] + R[(l_)] + R[(l_) + 1]) * 0.33333334f)
  for (l = 2; l + 2 < H; ++l) {
    const float c = SDO_S3(l);
    if (c < thr && c <= SDO_S3(l + 1)) return (float)l;
  }
#undef SDO_S3
  return 0.f;
}

void sdo_source_fix(sdo_c32 *x, size_t n, int iq_reverse, float *dc, float alpha, int first)
{
  size_t i;
  if (n == 0) return;
  if (iq_reverse) for (i = 0; i < n; ++i) { float t = x[i].re; x[i].re = x[i].im; x[i].im = t; }
  if (dc) {
    double sr = 0, si = 0;
    float mr, mi;
    for (i = 0; i < n; ++i) { sr += x[i].re; si += x[i].im; }
    mr = (float)(sr / (double)n); mi = (float)(si / (double)n);
    if (first) { dc[0] = mr; dc[1] = mi; }
    else { dc[0] = dc[0] + alpha * (mr - dc[0]); dc[1] = dc[1] + alpha * (mi - dc[1]); }
    for (i = 0; i < n; ++i) { x[i].re = x[i].re - dc[0]; x[i].im = x[i].im - dc[1]; }
  }
}

#define SDO_WS_BLOCK 4096   /* SIGDIGGER_WAVESAMPLER_FEEDER_BLOCK_LENGTH, include/WaveSampler.h:28 */

/* var of sample p (Tasks/WaveSampler.cpp:240-267); products as in SPEC "element-wise" */
static float zc_var(const sdo_c32 *data, size_t p, int space, int amplitude, sdo_c32 thr, sdo_c32 ang, sdo_c32 prev)
{
  sdo_c32 x = data[p];
  if (space == 0) {
    float v, t;
    if (amplitude) { v = fmaf(x.im, x.im, x.re * x.re);   t = fmaf(thr.im, thr.im, thr.re * thr.re); }
    else           { v = fmaf(-x.im, ang.im, x.re * ang.re); t = fmaf(-thr.im, ang.im, thr.re * ang.re); }
    return v - t;
  } else if (space == 1) {
    float re = fmaf(-x.im, ang.im, x.re * ang.re), im = fmaf(x.im, ang.re, x.re * ang.im);
    return sdo_atan2f(im, re);
  } else {
    sdo_c32 u = { -x.im, x.re };                         /* SU_I * x */
    sdo_c32 d = cmul_conj(u, prev);
    return sdo_atan2f(d.im, d.re);
  }
}

size_t sdo_sample_zero_crossing(const sdo_c32 *data, size_t length, float bnor, int space, int amplitude,
                                sdo_c32 threshold, sdo_c32 zc_angle, unsigned char *out_sym, size_t nout)
{
  long p = 0, lastZc = 0;
  size_t total = 0;
  while (p < (long)length) {                               /* one work() call per iteration */
    long amount = (long)length - p;
    long i = 0;
    int last;
    sdo_c32 prev = { 0, 0 };                               /* this->prevSample, never written back */
    float prevVar = -1.0f;                                 /* this->prevVar, never written back */
    if (amount > SDO_WS_BLOCK) amount = SDO_WS_BLOCK;
    last = p + amount >= (long)length;
    while (amount--) {
      float var = zc_var(data, (size_t)p, space, amplitude, threshold, zc_angle, prev);
      if (space == 2) prev = data[p];
      if ((var > 0 || var < 0) || last) {
        if (var * prevVar < 0 || last) {
          long samples = p - lastZc;
          long symbols = (long)roundf((float)samples * bnor);
          while (symbols-- > 0 && i < SDO_WS_BLOCK) {
            if (total + (size_t)i < nout) out_sym[total + (size_t)i] = var > 0;
            ++i;
          }
          lastZc = p;
          prevVar = var;
        }
      }
      ++p;
    }
    total += (size_t)i;
  }
  return total;
}

void sdo_conj_prev(const sdo_c32 *x, size_t n, sdo_c32 prev0, sdo_c32 *y)
{
  size_t p;
  sdo_c32 prev = prev0;
  for (p = 0; p < n; ++p) { sdo_c32 cur = x[p]; y[p] = cmul_conj(cur, prev); prev = cur; }
}

/* ===================================================================================== */
/* FAC [REF-PINNED structure] Default/GenericInspector/FACTab.cpp:181-246                  */
/* ===================================================================================== */
void sdo_fac_feed(const sdo_c32 *buf, size_t n, float alpha, long view_start, long view_end,
                  float *fac, float *max, float *min)
{
  double *re = malloc(sizeof(double) * n), *im = malloc(sizeof(double) * n);
  size_t i;
  for (i = 0; i < n; ++i) { re[i] = buf[i].re; im[i] = buf[i].im; }
  sdo_fft_f64(re, im, n);                                   /* SU_FFTW(_execute(direct)) :212 */
  for (i = 0; i < n; ++i) {                                 /* bufData[i] *= conj(bufData[i]) :214-215 */
    float a = (float)re[i], b = (float)im[i];
    re[i] = (double)fmaf(b, b, a * a);
    im[i] = 0.0;
  }
  for (i = 0; i < n; ++i) im[i] = -im[i];                   /* inverse = conj(FFT(conj(.))), unnormalised :217 */
  sdo_fft_f64(re, im, n);
  for (i = 0; i < n / 2; ++i) {                             /* :219-236 */
    float a = (float)re[i], b = (float)(-im[i]);
    float v = sqrtf(fmaf(b, b, a * a));
    re[i] = (double)v;
    if (view_start <= (long)i && (long)i < view_end) {
      if (v > *max) *max = v;
      if (v < *min) *min = v;
    }
  }
  for (i = 0; i < n / 2; ++i)                               /* SU_SPLPF_FEED :238-239 */
    fac[i] += alpha * ((float)re[i] / *max - fac[i]);
  free(re); free(im);
}

/* ===================================================================================== */
/* T9: carrier centroid [REF-PINNED structure] Tasks/CarrierDetector.cpp:80-143            */
/* ===================================================================================== */

void sdo_blackmann_harris_complex(sdo_c32 *h, size_t n)
{
  size_t i;
  for (i = 0; i < n; ++i) {
    double t = 2.0 * SDO_PI * (double)i / (double)(n - 1);
    float w = (float)(0.35875 - 0.48829 * cos(t) + 0.14128 * cos(2 * t) - 0.01168 * cos(3 * t));
    h[i].re *= w; h[i].im *= w;
  }
}

/* Re(x conj(x)) as the reference's `x *= conj(x)` computes it: fl(fl(a a) + fl(b b)).  volatile: no contraction whatever
 * flags this file is built with (the -O3 -march=native build of the CPU baseline has fma instructions at hand) */
static inline float sdo_norm2_unfused(float a, float b)
{
  volatile float aa = a * a, bb = b * b;
  return aa + bb;
}

float sdo_carrier_detect(const sdo_c32 *data, size_t len, float avgRelBw, float dcNotchRelBw)
{
  size_t alloc = 1, k;
  while (alloc < len) alloc <<= 1;
  sdo_c32 *buf = calloc(alloc, sizeof *buf);
  double *re = malloc(sizeof(double) * alloc), *im = malloc(sizeof(double) * alloc);
  memcpy(buf, data, len * sizeof *buf);
  sdo_blackmann_harris_complex(buf, len);
  for (k = 0; k < alloc; ++k) { re[k] = buf[k].re; im[k] = buf[k].im; }
  sdo_fft_f64(re, im, alloc);
  for (k = 0; k < alloc; ++k) { buf[k].re = (float)re[k]; buf[k].im = (float)im[k]; }

  int i, maxNdx = 0;
  int bins = (int)((double)alloc * (double)avgRelBw) + 1;
  int delta = (bins - 1) / 2, start;
  int skipLen = (int)(.5 * (double)dcNotchRelBw * (double)alloc);
  float maxVal = 0, psd;
  float accr = 0, acci = 0;                              /* SUCOMPLEX acc: binary32, summed in bin order (:109, :130) */
  for (i = skipLen; i < (int)alloc - skipLen; ++i) {
    /* asSuComplex[i] *= conj(asSuComplex[i]) (:113): a complex product -- two rounded squares, one rounded sum, no fma */
    buf[i].re = sdo_norm2_unfused(buf[i].re, buf[i].im);
    buf[i].im = 0;
    psd = buf[i].re;
    if (psd > maxVal) { maxVal = psd; maxNdx = i; }
  }
  start = maxNdx - delta;
  for (i = 0; i < bins; ++i) {
    int j = i + start;
    float cs, sn;
    if (j < 0) j += (int)alloc;
    j %= (int)alloc;
    psd = buf[j].re;
    float nFreq = 2.f * (float)j / (float)alloc;
    sincosf((float)SDO_PI * nFreq, &sn, &cs);            /* SU_C_EXP(SU_I * pi * nFreq) = cexpf(0 + i theta) (:130) */
    accr += psd * cs;
    acci += psd * sn;
  }
  float peak = atan2f(acci, accr);                       /* SU_C_ARG (:134) */
  free(buf); free(re); free(im);
  return peak;
}

/* ===================================================================================== */
/* P2/P3: SpectrumView [REF-PINNED] Panoramic/Scanner.cpp:27-293                           */
/* ===================================================================================== */

#define SCN_DEFAULT_BIN_VALUE -200.0f
#define SCN_FREQ_RESOLUTION   1000.0
#define SCN_COUNT_MAX         5.0f
#define SCN_COUNT_RESET       1.0f

static unsigned next_pow2(unsigned n) { unsigned i = 1; while (i < n) i <<= 1; return i; }

static void specview_reset(sdo_specview *v)
{
  memset(v->psd, 0, SDO_SCANNER_SPECTRUM_SIZE * sizeof(float));
  memset(v->psdAccum, 0, SDO_SCANNER_SPECTRUM_SIZE * sizeof(float));
  memset(v->psdCount, 0, SDO_SCANNER_SPECTRUM_SIZE * sizeof(float));
}

void sdo_specview_init(sdo_specview *v, float *psd, float *accum, float *count)
{
  memset(v, 0, sizeof *v);
  v->spectrumSize = SDO_SCANNER_SPECTRUM_SIZE;
  v->fftRelBw = .5f;
  v->psd = psd; v->psdAccum = accum; v->psdCount = count;
  specview_reset(v);
}

void sdo_specview_set_range(sdo_specview *v, double fmin, double fmax)
{
  v->freqMin = fmin; v->freqMax = fmax; v->freqRange = fmax - fmin;
  v->spectrumSize = next_pow2((unsigned)(v->freqRange / SCN_FREQ_RESOLUTION));
  if (v->spectrumSize > SDO_SCANNER_SPECTRUM_SIZE)
    v->spectrumSize = SDO_SCANNER_SPECTRUM_SIZE;
  specview_reset(v);
}

void sdo_specview_interpolate(sdo_specview *v)
{
  unsigned i, j, count = 1, zero_pos = 0;
  float t = 0, left = SCN_DEFAULT_BIN_VALUE, right = SCN_DEFAULT_BIN_VALUE;
  int first = 1, inGap = 0;
  for (i = 0; i < v->spectrumSize; ++i) {
    if (!inGap) {
      if (v->psdCount[i] <= .5f) {
        inGap = 1; zero_pos = i; count = 1;
        first = i == 0;
        if (!first) left = v->psd[i - 1];
      } else {
        v->psd[i] = v->psdAccum[i] / v->psdCount[i];
        if (v->psdCount[i] > SCN_COUNT_MAX) {
          v->psdCount[i] = SCN_COUNT_RESET;
          v->psdAccum[i] = v->psd[i] * SCN_COUNT_RESET;
        }
      }
    } else {
      if (v->psdCount[i] <= .5f) {
        ++count;
      } else {
        inGap = 0;
        right = v->psd[i] = v->psdAccum[i] / v->psdCount[i];
        if (first) {
          for (j = 0; j < count; ++j) v->psd[j + zero_pos] = right;
        } else {
          for (j = 0; j < count; ++j) {
            t = (float)(j + .5f) / count;
            v->psd[j + zero_pos] = (1 - t) * left + t * right;
          }
        }
      }
    }
  }
  if (inGap)
    for (j = 0; j < count; ++j) v->psd[j + zero_pos] = left;
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void specview_feed_linear(sdo_specview *v, const float *psdData, const float *countData,
                                 size_t psdSize, double freqMin, double freqMax, int adjustSides)
{
  double inpBw, bw, freqSkip, fftCount, bins, pos, delta, srcBinW, dstBinW;
  int skip, j, k;
  inpBw = freqMax - freqMin;
  skip = adjustSides ? (int)(.5f * (1 - v->fftRelBw) * psdSize) : 0;
  freqSkip = (double)skip / psdSize * inpBw;
  bw = inpBw - 2 * freqSkip;
  fftCount = v->freqRange / bw;
  bins = v->spectrumSize / fftCount;
  srcBinW = inpBw / psdSize;
  dstBinW = v->freqRange / v->spectrumSize;
  delta = dstBinW / srcBinW;
  pos = (freqSkip + freqMin - v->freqMin) / v->freqRange;
  pos *= v->spectrumSize;
  j = pos > 0 ? (int)pos : 0;
  k = pos + bins < v->spectrumSize ? (int)(pos + bins) : (int)v->spectrumSize;
  while (j < k) {
    double freqJ = v->freqMin + dstBinW * j;
    double srcBin = (freqJ - freqMin) / srcBinW;
    int startBin = (int)srcBin, endBin = (int)(srcBin + delta), i;
    float psdAccum = 0, psdCount = 0;
    startBin = clampi(startBin, 0, (int)(psdSize - 1));
    endBin = clampi(endBin, startBin + 1, (int)psdSize);
    for (i = startBin; i < endBin; i++) {
      psdAccum += psdData[i];
      psdCount += countData != NULL ? countData[i] : 1;
    }
    if (psdCount > 0) {
      v->psdAccum[j] += psdAccum / psdCount;
      v->psdCount[j] += 1;
    }
    j++;
  }
}

static void specview_feed_histogram(sdo_specview *v, const float *psdData, size_t psdSize,
                                    double freqMin, double freqMax)
{
  double relBw = (freqMax - freqMin) / v->freqRange;
  double fStart = (freqMin - v->freqMin) / v->freqRange;
  double fEnd = (freqMax - v->freqMin) / v->freqRange;
  float t, inv = (float)(1. / psdSize), accum = 0;
  unsigned i, j;
  fStart *= v->spectrumSize; fEnd *= v->spectrumSize; relBw *= v->spectrumSize;
  j = (unsigned)fStart;
  if (j > v->spectrumSize - 1) j = v->spectrumSize - 1;
  for (i = 0; i < psdSize; ++i) accum += psdData[i];
  accum *= inv;
  if (floor(fStart) != floor(fEnd)) {
    t = (float)((fStart - floor(fStart)) / relBw);
    v->psdCount[j] += 1 - t;
    v->psdAccum[j] += (1 - t) * accum;
    if (j + 1 < v->spectrumSize) {
      v->psdCount[j + 1] += t;
      v->psdAccum[j + 1] += t * accum;
    }
  } else {
    v->psdCount[j] += 1;
    v->psdAccum[j] += accum;
  }
}

void sdo_specview_feed(sdo_specview *v, const float *psd, const float *count, size_t psdSize,
                       double freqMin, double freqMax, int adjustSides)
{
  double fftCount = (freqMax - freqMin) / v->freqRange;
  if (fftCount * v->spectrumSize >= 2)
    specview_feed_linear(v, psd, count, psdSize, freqMin, freqMax, adjustSides);
  else
    specview_feed_histogram(v, psd, psdSize, freqMin, freqMax);
  sdo_specview_interpolate(v);
}

/* ===================================================================================== */
/* bulk helpers for the test-suite (vectorised access to the D primitives)                */
/* ===================================================================================== */
void sdo_phasor_u32_bulk(const uint32_t *p, sdo_c32 *out, size_t n)
{ size_t i; for (i = 0; i < n; ++i) sdo_phasor_u32(p[i], &out[i].re, &out[i].im); }
void sdo_atan2f_bulk(const float *y, const float *x, float *out, size_t n)
{ size_t i; for (i = 0; i < n; ++i) out[i] = sdo_atan2f(y[i], x[i]); }
void sdo_log2f_bulk(const float *x, float *out, size_t n)
{ size_t i; for (i = 0; i < n; ++i) out[i] = sdo_log2f(x[i]); }
void sdo_exp2f_bulk(const float *x, float *out, size_t n)
{ size_t i; for (i = 0; i < n; ++i) out[i] = sdo_exp2f(x[i]); }

/* ===================================================================================== */
/* T10: Doppler centroid [REF-PINNED structure] Tasks/DopplerCalculator.cpp:85-175        */
/* ===================================================================================== */
/* spectrum: alloc floats (mirrored PSD, :128); res[0] = peak velocity, res[1] = sigma, res[2] = max */
void sdo_doppler_calc(const sdo_c32 *data, size_t len, float fs, double f0, float *spectrum, float *res)
{
  size_t alloc = 1, k;
  while (alloc < len) alloc <<= 1;
  if (alloc < 16) alloc = 16;
  sdo_c32 *buf = calloc(alloc, sizeof *buf);
  double *re = malloc(sizeof(double) * alloc), *im = malloc(sizeof(double) * alloc);
  memcpy(buf, data, len * sizeof *buf);
  sdo_blackmann_harris_complex(buf, len);
  for (k = 0; k < alloc; ++k) { re[k] = buf[k].re; im[k] = buf[k].im; }
  sdo_fft_f64(re, im, alloc);
  for (k = 0; k < alloc; ++k) { buf[k].re = (float)re[k]; buf[k].im = (float)im[k]; }

  int i, maxNdx = 0, bins = (int)alloc, delta = bins / 2, start;
  float maxVal = 0, psd, peak, lambda = (float)(299792458. / f0);
  float accr = 0, acci = 0;                              /* SUCOMPLEX acc (:112): binary32, bin order */
  float dispAcc = 0, totalEnergy = 0, err = 0, t, y;
  /* every sum below is built in the reference's type AND order: a binary32 running sum over 2^15 .. 2^24 bins carries a
   * rounding error of the order of sqrt(bins) ulp, so "the same number" means the same sequence of roundings */
  for (i = 0; i < bins; ++i) {
    buf[i].re = sdo_norm2_unfused(buf[i].re, buf[i].im); /* x *= conj(x) (:120) */
    buf[i].im = 0;
    psd = buf[i].re;
    if (psd > maxVal) { maxVal = psd; maxNdx = i; }
    if (spectrum) spectrum[((size_t)(bins - i) + (size_t)delta) % (size_t)bins] = psd;
    {
      volatile float vy, vt;                             /* Kahan, :131-134; volatile: must survive any optimiser */
      vy = psd - err; y = vy;
      vt = totalEnergy + y; t = vt;
      err = (t - totalEnergy) - y;
      totalEnergy = t;
    }
  }
  start = maxNdx - delta;
  /* (:157) the reference divides by the int product delta * delta, which overflows from 2^17 bins on (undefined; the
   * build here yields 0 -> sigma = inf).  The restatement keeps the meaning: delta^2 as a binary32 (exact: a power of two) */
  const float d2 = (float)delta * (float)delta;
  for (i = 0; i < bins; ++i) {
    long long j = i + start;
    float cs, sn;
    if (j < 0) j += (long long)alloc;
    j %= (long long)alloc;
    psd = buf[j].re;
    float nFreq = 2.f * (float)j / (float)alloc;
    sincosf((float)SDO_PI * nFreq, &sn, &cs);
    accr += psd * cs;
    acci += psd * sn;
    j = i;
    if (j >= delta) j -= bins;
    dispAcc += ((float)(j * j) * psd / totalEnergy) / d2;  /* j * j is an int64 product converted once (:157) */
  }
  peak = atan2f(acci, accr);
  if (peak > (float)SDO_PI) peak -= (float)(2 * SDO_PI);
  peak = fs * (peak / (float)SDO_PI) * .5f;
  res[0] = -lambda * peak;
  res[1] = fs * sqrtf(dispAcc) * .5f;
  res[2] = maxVal;
  free(buf); free(re); free(im);
}

/* ---- C2: FFT channeliser (su_specttuner semantics) [SPEC, UPSTREAM-RECOLLECTION] ------------------------------------
 * SPEC.md section C2.  Window W, hop H = W/2.  Window k = stream samples [kH, kH + W); output block k (halfsz samples)
 * = alpha .* y_k[0:halfsz] + beta .* y_{k-1}[halfsz:size], y_k = IDFT_size(h .* pick(DFT_W(window k))), y_{-1} = 0. */
void sdo_specttuner_geometry(unsigned W, double f0, double bw, double guard, sdo_st_geom *g)
{
  double actual_bw = bw * guard, k;
  unsigned min_size, size = 1;
  if (actual_bw > 2.0 * SDO_PI) actual_bw = 2.0 * SDO_PI;
  k = actual_bw / (2.0 * SDO_PI);
  min_size = (unsigned)ceil(k * (double)W - 1e-6);       /* bw * guard = 2 pi / D must give W / D bins, not one more */
  while (size < min_size) size <<= 1;
  if (size < 2) size = 2;
  if (size > W) size = W;
  g->size = size;
  g->halfsz = size / 2;
  g->width = (unsigned)ceil((double)min_size / guard);
  if (g->width > size) g->width = size;
  g->halfw = g->width >> 1;
  if (g->halfw < 1) g->halfw = 1;
  g->decimation = W / size;
  g->center = (int)(2.0 * floor(f0 / (4.0 * SDO_PI) * (double)W + 0.5));    /* even bin: a hop advances it by whole turns */
  g->center &= (int)(W - 1);
  {
    /* residual of the rounding, corrected at the output rate when `precise` */
    double f0w = fmod(f0, 2.0 * SDO_PI), ef = (double)g->center * 2.0 * SDO_PI / (double)W, lo;
    if (f0w < 0) f0w += 2.0 * SDO_PI;
    lo = f0w - ef;
    if (lo > SDO_PI) lo -= 2.0 * SDO_PI;
    if (lo < -SDO_PI) lo += 2.0 * SDO_PI;
    g->lo = lo;
    g->dphase = (uint32_t)(int64_t)llround(-lo * (double)g->decimation / (2.0 * SDO_PI) * 4294967296.0);
  }
}

/* The response is designed in binary64 and rounded to binary32; its imaginary part is rounding residue (~1e-11 of the
 * real part: the kernel is real and symmetric), so for the channeliser to be bit-pinned the binary64 transform itself is
 * part of SPEC.md C2: iterative radix-2, decimation in time, twiddle (cos, sin)(sign 2 pi k / len) from libm per
 * butterfly group, sign = -1 forward / +1 backward, unnormalised. */
static void st_fft64(double *re, double *im, size_t n, int sign)
{
  size_t i, j, len, k;
  for (i = 1, j = 0; i < n; ++i) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) { double t = re[i]; re[i] = re[j]; re[j] = t; t = im[i]; im[i] = im[j]; im[j] = t; }
  }
  for (len = 2; len <= n; len <<= 1) {
    const size_t half = len >> 1;
    for (k = 0; k < half; ++k) {
      const double ang = (double)sign * 2.0 * SDO_PI * (double)k / (double)len;
      double wr, wi;
      sincos(ang, &wi, &wr);                              /* explicitly: glibc's sincos and its sin / cos differ in rare last bits,
                                                           * and compilers disagree on merging the pair (gcc does, clang does not) */
      for (i = k; i < n; i += len) {
        const double ur = re[i], ui = im[i];
        const double vr = re[i + half] * wr - im[i + half] * wi, vi = re[i + half] * wi + im[i + half] * wr;
        re[i] = ur + vr; im[i] = ui + vi; re[i + half] = ur - vr; im[i + half] = ui - vi;
      }
    }
  }
}

void sdo_specttuner_response(unsigned W, unsigned size, unsigned halfw, sdo_c32 *hk)
{
  /* brick wall -> time domain -> centred, Blackman-Harris, back -> frequency domain; k = 1/W folded in */
  double *re = calloc(size, sizeof *re), *im = calloc(size, sizeof *im);
  const unsigned half = size / 2;
  unsigned i;
  for (i = 0; i < size; ++i) re[i] = (i < halfw || i >= size - halfw) ? 1.0 : 0.0;
  st_fft64(re, im, size, +1);
  for (i = 0; i < size; ++i) { re[i] /= (double)size; im[i] /= (double)size; }
  for (i = 0; i < half; ++i) {                            /* centre */
    double t = re[i]; re[i] = re[i + half]; re[i + half] = t;
    t = im[i]; im[i] = im[i + half]; im[i + half] = t;
  }
  for (i = 0; i < size; ++i) {
    const double t = 2.0 * SDO_PI * (double)i / (double)(size - 1);
    const double w = 0.35875 - 0.48829 * cos(t) + 0.14128 * cos(2 * t) - 0.01168 * cos(3 * t);
    re[i] *= w; im[i] *= w;
  }
  for (i = 0; i < half; ++i) {                            /* ... and back */
    double t = re[i]; re[i] = re[i + half]; re[i + half] = t;
    t = im[i]; im[i] = im[i + half]; im[i + half] = t;
  }
  st_fft64(re, im, size, -1);
  for (i = 0; i < size; ++i) {
    const int pass = i < halfw || i >= size - halfw;
    hk[i].re = pass ? (float)(re[i] / (double)W) : 0.0f;
    hk[i].im = pass ? (float)(im[i] / (double)W) : 0.0f;
  }
  free(re); free(im);
}

void sdo_specttuner_crossfade(unsigned size, float *win)
{
  unsigned i;
  for (i = 0; i < size; ++i) { const double s = sin(SDO_PI * (double)i / (double)size); win[i] = (float)(s * s); }
}

size_t sdo_specttuner_run(const sdo_c32 *x, size_t len, unsigned W, double f0, double bw, double guard, int precise,
                          sdo_c32 *out, size_t cap)
{
  sdo_st_geom g;
  const unsigned H = W / 2;
  size_t k, nwin, n = 0;
  unsigned i;
  sdo_c32 *hk, *prev;
  float *win;
  double *re, *im, *cr, *ci;
  sdo_specttuner_geometry(W, f0, bw, guard, &g);
  if (len < W) return 0;
  nwin = (len - W) / H + 1;
  hk = malloc(sizeof *hk * g.size); prev = calloc(g.size, sizeof *prev); win = malloc(sizeof *win * g.size);
  re = malloc(sizeof *re * W); im = malloc(sizeof *im * W); cr = malloc(sizeof *cr * g.size); ci = malloc(sizeof *ci * g.size);
  sdo_specttuner_response(W, g.size, g.halfw, hk);
  sdo_specttuner_crossfade(g.size, win);
  for (k = 0; k < nwin; ++k) {
    const sdo_c32 *w = x + k * H;
    for (i = 0; i < W; ++i) { re[i] = w[i].re; im[i] = w[i].im; }
    sdo_fft_f64(re, im, W);
    for (i = 0; i < g.size; ++i) {
      const unsigned idx = (i < g.halfsz ? (unsigned)g.center + i : (unsigned)g.center + i + W - g.size) & (W - 1);
      /* the spectrum is binary32 on the device; the products are binary32 too */
      const float xr = (float)re[idx], xi = (float)im[idx];
      const float yr = xr * hk[i].re - xi * hk[i].im, yi = xr * hk[i].im + xi * hk[i].re;
      cr[i] = yr; ci[i] = -(double)yi;
    }
    sdo_fft_f64(cr, ci, g.size);                          /* unnormalised backward transform */
    for (i = 0; i < g.halfsz && n < cap; ++i, ++n) {
      const float a = win[i], b = win[i + g.halfsz];
      const float yr = (float)cr[i], yi = (float)-ci[i];
      float orr = a * yr + b * prev[i + g.halfsz].re, oi = a * yi + b * prev[i + g.halfsz].im;
      if (precise) {
        float c, s;
        sdo_phasor_u32((uint32_t)n * g.dphase, &c, &s);
        { const float tr = orr * c - oi * s, ti = orr * s + oi * c; orr = tr; oi = ti; }
      }
      out[n].re = orr; out[n].im = oi;
    }
    for (i = 0; i < g.size; ++i) { prev[i].re = (float)cr[i]; prev[i].im = (float)-ci[i]; }
  }
  free(hk); free(prev); free(win); free(re); free(im); free(cr); free(ci);
  return n;
}

/* ---- C2, the binary32 statement (SPEC.md section C2, "binary32 arithmetic") ---------------------------------------------
 * The device transforms are binary32; what follows restates, operation for operation, the arithmetic SPEC.md C2 freezes
 * for them, so that HIP == oracle bit for bit and everything downstream of the channeliser (AGC, Costas, Gardner) can be
 * compared exactly on the path the analyzer and the bench run by default.  Two forms, chosen by the channel size:
 *   narrow (size 8 .. 64):  window = 64 x 64: DFT64 (8 x 8, constant twiddles) over r of x[t + 64 r], twiddle
 *                           W_4096^(t k2), DFT64 over t; response product fused; inverse = DFT_size read backwards;
 *                           cross-fade fma(beta, prev, alpha cur);
 *   wide (every other size): window = three radix-16 Stockham passes; response product and cross-fade unfused;
 *                           inverse = conj(Stockham passes(conj(.))).
 * Complex product everywhere: re = fma(a.im, -b.im, a.re b.re), im = fma(a.im, b.re, a.re b.im). */
typedef sdo_c32 cf_t;
static inline cf_t cf_add(cf_t a, cf_t b) { cf_t r = {a.re + b.re, a.im + b.im}; return r; }
static inline cf_t cf_sub(cf_t a, cf_t b) { cf_t r = {a.re - b.re, a.im - b.im}; return r; }
static inline cf_t cf_neg(cf_t a) { cf_t r = {-a.re, -a.im}; return r; }
static inline cf_t cf_scale(cf_t a, float c) { cf_t r = {a.re * c, a.im * c}; return r; }
static inline cf_t cf_add_mj(cf_t t, cf_t d) { cf_t r = {t.re + d.im, t.im - d.re}; return r; }       /* t + (-j) d */
static inline cf_t cf_sub_mj(cf_t t, cf_t d) { cf_t r = {t.re - d.im, t.im + d.re}; return r; }       /* t - (-j) d */
static inline cf_t cf_mul(cf_t a, cf_t b)
{
  cf_t r;
  r.re = fmaf(a.im, -b.im, a.re * b.re);
  r.im = fmaf(a.im, b.re, a.re * b.im);
  return r;
}
#define ST32_C8 0.70710678118654752440f

static void st32_dft4(cf_t *a0, cf_t *a1, cf_t *a2, cf_t *a3)
{
  const cf_t t0 = cf_add(*a0, *a2), t1 = cf_sub(*a0, *a2), t2 = cf_add(*a1, *a3), d = cf_sub(*a1, *a3);
  *a0 = cf_add(t0, t2); *a2 = cf_sub(t0, t2); *a1 = cf_add_mj(t1, d); *a3 = cf_sub_mj(t1, d);
}
/* e +- o W8^1 and e +- o W8^3 as fused multiply-adds */
static void st32_bfly_w8_1(cf_t e, cf_t o, cf_t *p, cf_t *m)
{
  const cf_t r = cf_add_mj(o, o);
  p->re = fmaf(r.re, ST32_C8, e.re); p->im = fmaf(r.im, ST32_C8, e.im);
  m->re = fmaf(r.re, -ST32_C8, e.re); m->im = fmaf(r.im, -ST32_C8, e.im);
}
static void st32_bfly_w8_3(cf_t e, cf_t o, cf_t *p, cf_t *m)
{
  const cf_t r = cf_add_mj(cf_neg(o), o);
  p->re = fmaf(r.re, ST32_C8, e.re); p->im = fmaf(r.im, ST32_C8, e.im);
  m->re = fmaf(r.re, -ST32_C8, e.re); m->im = fmaf(r.im, -ST32_C8, e.im);
}
static void st32_dft8(cf_t *v)
{
  cf_t e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
  st32_dft4(&e0, &e1, &e2, &e3);
  st32_dft4(&o0, &o1, &o2, &o3);
  v[0] = cf_add(e0, o0); v[4] = cf_sub(e0, o0);
  st32_bfly_w8_1(e1, o1, &v[1], &v[5]);
  v[2] = cf_add_mj(e2, o2); v[6] = cf_sub_mj(e2, o2);
  st32_bfly_w8_3(e3, o3, &v[3], &v[7]);
}
static void st32_dft16(cf_t *v)
{
  const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;
  cf_t e[8], o[8];
  int i;
  for (i = 0; i < 8; ++i) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
  st32_dft8(e);
  st32_dft8(o);
  o[1] = cf_mul(o[1], (cf_t){ c1, -s1});
  o[3] = cf_mul(o[3], (cf_t){ s1, -c1});
  o[5] = cf_mul(o[5], (cf_t){-s1, -c1});
  o[7] = cf_mul(o[7], (cf_t){-c1, -s1});
  for (i = 0; i < 8; ++i) {
    if (i == 4) { v[4] = cf_add_mj(e[4], o[4]); v[12] = cf_sub_mj(e[4], o[4]); }
    else if (i == 2) st32_bfly_w8_1(e[2], o[2], &v[2], &v[10]);
    else if (i == 6) st32_bfly_w8_3(e[6], o[6], &v[6], &v[14]);
    else { v[i] = cf_add(e[i], o[i]); v[i + 8] = cf_sub(e[i], o[i]); }
  }
}
static void st32_dftR(int R, cf_t *v)
{
  if (R == 2) { const cf_t t = v[0]; v[0] = cf_add(t, v[1]); v[1] = cf_sub(t, v[1]); }
  else if (R == 4) st32_dft4(&v[0], &v[1], &v[2], &v[3]);
  else if (R == 8) st32_dft8(v);
  else { assert(R == 16); st32_dft16(v); }
}

static struct { int ready; cf_t w64[64]; cf_t tw[4096]; float win64[64]; } st32_tab;
static void st32_init(void)
{
  int i;
  if (st32_tab.ready) return;
  for (i = 0; i < 64; ++i) {
    const double a = -2.0 * SDO_PI * (double)i / 64.0, s = sin(SDO_PI * (double)i / 64.0);
    double cr, ci;
    sincos(a, &ci, &cr);
    st32_tab.w64[i].re = (float)cr; st32_tab.w64[i].im = (float)ci;
    st32_tab.win64[i] = (float)(s * s);
  }
  for (i = 0; i < 4096; ++i) {
    const double a = -2.0 * SDO_PI * (double)i / 4096.0;
    double cr, ci;
    sincos(a, &ci, &cr);
    st32_tab.tw[i].re = (float)cr; st32_tab.tw[i].im = (float)ci;
  }
  st32_tab.ready = 1;
}
void sdo_st32_tables(sdo_c32 *w64, float *win64)
{
  st32_init();
  memcpy(w64, st32_tab.w64, sizeof st32_tab.w64);
  memcpy(win64, st32_tab.win64, sizeof st32_tab.win64);
}
/* a W_64^m: the trivial and the 45-degree powers without a product */
static cf_t st32_mul_w64(cf_t a, int m)
{
  m &= 63;
  switch (m) {
    case 0:  return a;
    case 16: return (cf_t){a.im, -a.re};
    case 32: return cf_neg(a);
    case 48: return (cf_t){-a.im, a.re};
    case 8:  return cf_scale(cf_add_mj(a, a), ST32_C8);
    case 24: return cf_scale(cf_add_mj(cf_neg(a), a), ST32_C8);
    case 40: return cf_neg(cf_scale(cf_add_mj(a, a), ST32_C8));
    case 56: return cf_neg(cf_scale(cf_add_mj(cf_neg(a), a), ST32_C8));
    default: return cf_mul(a, st32_tab.w64[m]);
  }
}
/* N = R1 R2 <= 64 points, natural order in and out: n = n1 + R1 n2, k = k2 + R2 k1 */
static void st32_dft_2f(int R1, int R2, const cf_t *in, cf_t *out)
{
  const int N = R1 * R2;
  cf_t mid[64], a[8], b[8];
  int n1, n2, k1, k2;
  for (n1 = 0; n1 < R1; ++n1) {
    for (n2 = 0; n2 < R2; ++n2) a[n2] = in[n1 + R1 * n2];
    st32_dftR(R2, a);
    for (k2 = 0; k2 < R2; ++k2) mid[n1 + R1 * k2] = st32_mul_w64(a[k2], n1 * k2 * (64 / N));
  }
  for (k2 = 0; k2 < R2; ++k2) {
    for (n1 = 0; n1 < R1; ++n1) b[n1] = mid[n1 + R1 * k2];
    st32_dftR(R1, b);
    for (k1 = 0; k1 < R1; ++k1) out[k2 + R2 * k1] = b[k1];
  }
}
static void st32_dft_reg(int log2n, const cf_t *in, cf_t *out)
{
  if (log2n == 6) st32_dft_2f(8, 8, in, out);
  else if (log2n == 5) st32_dft_2f(4, 8, in, out);
  else { memcpy(out, in, sizeof(cf_t) << log2n); st32_dftR(1 << log2n, out); }
}

/* narrow form: X = DFT_4096(win) as 64 x 64 */
void sdo_st32_forward_narrow(const sdo_c32 *win, sdo_c32 *X)
{
  static _Thread_local cf_t Bp[64][64];                   /* B[t][k2]; per thread (window ranges run side by side) */
  int t, r, l, h;
  st32_init();
  for (t = 0; t < 64; ++t) {
    cf_t in[64], A[64], lo[8], hi[8];
    for (r = 0; r < 64; ++r) in[r] = win[t + 64 * r];
    st32_dft_2f(8, 8, in, A);
    /* W_4096^(t (8h + l)) = hi[h] lo[l]; the six base powers W^(t 2^j) are table values */
    lo[1] = st32_tab.tw[(t << 0) & 4095]; lo[2] = st32_tab.tw[(t << 1) & 4095]; lo[4] = st32_tab.tw[(t << 2) & 4095];
    lo[3] = cf_mul(lo[1], lo[2]); lo[5] = cf_mul(lo[1], lo[4]); lo[6] = cf_mul(lo[2], lo[4]); lo[7] = cf_mul(lo[3], lo[4]);
    hi[1] = st32_tab.tw[(t << 3) & 4095]; hi[2] = st32_tab.tw[(t << 4) & 4095]; hi[4] = st32_tab.tw[(t << 5) & 4095];
    hi[3] = cf_mul(hi[1], hi[2]); hi[5] = cf_mul(hi[1], hi[4]); hi[6] = cf_mul(hi[2], hi[4]); hi[7] = cf_mul(hi[3], hi[4]);
    for (h = 0; h < 8; ++h)
      for (l = 0; l < 8; ++l) {
        if (h == 0) { if (l) A[l] = cf_mul(A[l], lo[l]); }
        else if (l == 0) A[8 * h] = cf_mul(A[8 * h], hi[h]);
        else A[8 * h + l] = cf_mul(A[8 * h + l], cf_mul(hi[h], lo[l]));
      }
    memcpy(Bp[t], A, sizeof A);
  }
  for (l = 0; l < 64; ++l) {
    cf_t v[64], A[64];
    for (t = 0; t < 64; ++t) v[t] = Bp[t][l];
    st32_dft_2f(8, 8, v, A);
    for (r = 0; r < 64; ++r) X[l + 64 * r] = A[r];
  }
}

/* Stockham autosort passes, radix <= 16, of fft_core.hpp's plan (MB = 4): ceil(bits / 4) passes, the first (bits % P)
 * one bit wider.  Twiddles of butterfly j in pass p: W_N^(q idx), idx = (j mod NS) N / (NS R); powers 1 and 2 are table
 * values, 4 = 2^2, 8 = 4^2, 3 = 1*2, 5 = 1*4, 6 = 2*4, 7 = 3*4, q = (q - 8) * 8 above. */
static void st32_stockham(int log2n, cf_t *d, cf_t *tmp, const cf_t *tw)
{
  const int N = 1 << log2n, P = (log2n + 3) / 4, BASE = log2n / P, EXTRA = log2n % P;
  int p, nsl = 0, j, q;
  cf_t *src = d, *dst = tmp;
  for (p = 0; p < P; ++p) {
    const int rb = BASE + (p < EXTRA ? 1 : 0), R = 1 << rb, NS = 1 << nsl;
    for (j = 0; j < N / R; ++j) {
      const int k = j & (NS - 1), j0 = ((j - k) << rb) + k;
      cf_t v[16], w[16];
      for (q = 0; q < R; ++q) v[q] = src[j + q * (N / R)];
      if (p > 0) {
        const int idx = k << (log2n - nsl - rb);
        w[1] = tw[idx & (N - 1)];
        if (R > 2) w[2] = tw[(2 * idx) & (N - 1)];
        if (R > 4) w[4] = cf_mul(w[2], w[2]);
        if (R > 8) w[8] = cf_mul(w[4], w[4]);
        if (R > 2) w[3] = cf_mul(w[1], w[2]);
        if (R > 4) { w[5] = cf_mul(w[1], w[4]); w[6] = cf_mul(w[2], w[4]); w[7] = cf_mul(w[3], w[4]); }
        if (R > 8) for (q = 9; q < 16; ++q) w[q] = cf_mul(w[q - 8], w[8]);
        for (q = 1; q < R; ++q) v[q] = cf_mul(v[q], w[q]);
      }
      st32_dftR(R, v);
      for (q = 0; q < R; ++q) dst[j0 + q * NS] = v[q];
    }
    { cf_t *t = src; src = dst; dst = t; }
    nsl += rb;
  }
  if (src != d) memcpy(d, src, sizeof(cf_t) * (size_t)N);
}

void sdo_st32_forward_wide(const sdo_c32 *win, sdo_c32 *X)
{
  static _Thread_local cf_t tmp[4096];
  st32_init();
  memcpy(X, win, sizeof(cf_t) * 4096);
  st32_stockham(12, X, tmp, st32_tab.tw);
}

/* one window of one channel: y[0 .. size) from the window's spectrum (the form follows the size) */
static void st32_channel(const cf_t *X, unsigned W, const sdo_st_geom *g, const cf_t *hk, const cf_t *tw_s, cf_t *y)
{
  const unsigned S = g->size, HS = g->halfsz;
  unsigned i, log2s = 0;
  cf_t u[4096], F[4096];
  while ((1u << log2s) < S) ++log2s;
  if (S >= 8 && S <= 64) {
    for (i = 0; i < S; ++i) {
      const unsigned idx = (i < HS ? (unsigned)g->center + i : (unsigned)g->center + i + W - S) & (W - 1);
      u[i] = cf_mul(X[idx], hk[i]);
    }
    st32_dft_reg((int)log2s, u, F);
    for (i = 0; i < S; ++i) y[i] = F[(S - i) & (S - 1)];
  } else {
    for (i = 0; i < S; ++i) {
      const unsigned idx = (i < HS ? (unsigned)g->center + i : (unsigned)g->center + i + W - S) & (W - 1);
      const cf_t x = X[idx], h = hk[i];
      u[i].re = x.re * h.re - x.im * h.im;
      u[i].im = -(x.re * h.im + x.im * h.re);
    }
    st32_stockham((int)log2s, u, F, tw_s);
    for (i = 0; i < S; ++i) { y[i].re = u[i].re; y[i].im = -u[i].im; }
  }
}

/* A bank of channels over windows [w_begin, w_end) of a stream (window w = samples [w H, w H + W)); every channel opened
 * at the start of the stream.  Channel c's output block w goes to out[c * row_stride + w * halfsz(c) ...].  w_begin > 0
 * transforms window w_begin - 1 as well (the cross-fade partner), so ranges can be computed independently.
 * Returns the number of windows the stream holds. */
size_t sdo_specttuner_bank_f32(const sdo_c32 *x, size_t len, unsigned nchan, const double *f0, const double *bw, const double *guard,
                               const int *precise, size_t w_begin, size_t w_end, sdo_c32 *out, size_t row_stride)
{
  const unsigned W = 4096, H = 2048;
  const size_t nwin = len < W ? 0 : (len - W) / H + 1;
  sdo_st_geom *g = malloc(sizeof *g * (nchan ? nchan : 1));
  cf_t **hk = calloc(nchan ? nchan : 1, sizeof *hk), **tws = calloc(nchan ? nchan : 1, sizeof *tws), **prev = calloc(nchan ? nchan : 1, sizeof *prev);
  cf_t *Xn = malloc(sizeof(cf_t) * W), *Xw = malloc(sizeof(cf_t) * W), *y = malloc(sizeof(cf_t) * W);
  int any_narrow = 0, any_wide = 0;
  unsigned c, i;
  size_t w;
  st32_init();
  if (w_end > nwin) w_end = nwin;
  for (c = 0; c < nchan; ++c) {
    sdo_specttuner_geometry(W, f0[c], bw[c], guard[c], &g[c]);
    hk[c] = malloc(sizeof(cf_t) * g[c].size);
    prev[c] = calloc(g[c].size, sizeof(cf_t));
    sdo_specttuner_response(W, g[c].size, g[c].halfw, hk[c]);
    if (g[c].size >= 8 && g[c].size <= 64) any_narrow = 1;
    else {
      any_wide = 1;
      tws[c] = malloc(sizeof(cf_t) * g[c].size);
      for (i = 0; i < g[c].size; ++i) {
        const double a = -2.0 * SDO_PI * (double)i / (double)g[c].size;
        double cr, ci;
        sincos(a, &ci, &cr);
        tws[c][i].re = (float)cr; tws[c][i].im = (float)ci;
      }
    }
  }
  for (w = w_begin > 0 ? w_begin - 1 : 0; w < w_end; ++w) {
    const int emit = w >= w_begin;
    if (any_narrow) sdo_st32_forward_narrow(x + w * H, Xn);
    if (any_wide) sdo_st32_forward_wide(x + w * H, Xw);
    for (c = 0; c < nchan; ++c) {
      const unsigned S = g[c].size, HS = g[c].halfsz;
      const int narrow = S >= 8 && S <= 64;
      st32_channel(narrow ? Xn : Xw, W, &g[c], hk[c], tws[c], y);
      if (emit) {
        cf_t *o = out + (size_t)c * row_stride + w * HS;
        for (i = 0; i < HS; ++i) {
          cf_t r;
          if (narrow) {
            const float al = st32_tab.win64[i * (64 / S)], be = st32_tab.win64[(i + HS) * (64 / S)];
            r.re = fmaf(be, prev[c][i + HS].re, y[i].re * al);
            r.im = fmaf(be, prev[c][i + HS].im, y[i].im * al);
          } else {
            const double sa = sin(SDO_PI * (double)i / (double)S), sb = sin(SDO_PI * (double)(i + HS) / (double)S);
            const float al = (float)(sa * sa), be = (float)(sb * sb);
            r.re = al * y[i].re + be * prev[c][i + HS].re;
            r.im = al * y[i].im + be * prev[c][i + HS].im;
          }
          if (precise && precise[c]) {
            float cs, sn;
            sdo_phasor_u32((uint32_t)(w * HS + i) * g[c].dphase, &cs, &sn);
            if (narrow) { const cf_t t = {fmaf(r.re, cs, -(r.im * sn)), fmaf(r.re, sn, r.im * cs)}; r = t; }
            else { const cf_t t = {r.re * cs - r.im * sn, r.re * sn + r.im * cs}; r = t; }
          }
          o[i] = r;
        }
      }
      memcpy(prev[c], y, sizeof(cf_t) * S);
    }
  }
  for (c = 0; c < nchan; ++c) { free(hk[c]); free(tws[c]); free(prev[c]); }
  free(g); free(hk); free(tws); free(prev); free(Xn); free(Xw); free(y);
  return nwin;
}

size_t sdo_specttuner_run_f32(const sdo_c32 *x, size_t len, double f0, double bw, double guard, int precise, sdo_c32 *out, size_t cap)
{
  sdo_st_geom g;
  size_t nwin;
  sdo_specttuner_geometry(4096, f0, bw, guard, &g);
  nwin = len < 4096 ? 0 : (len - 4096) / 2048 + 1;
  if (cap < nwin * g.halfsz) return 0;
  sdo_specttuner_bank_f32(x, len, 1, &f0, &bw, &guard, &precise, 0, nwin, out, 0);
  return nwin * g.halfsz;
}

/* ---- O: channel detector (su_channel_detector) [SPEC, UPSTREAM-RECOLLECTION] ----------------------------------------
 * SPEC.md section O; parameters of Suscan/AnalyzerParams.cpp:53-71. */
static int cmp_float(const void *a, const void *b) { const float x = *(const float *)a, y = *(const float *)b; return (x > y) - (x < y); }

void sdo_chandet_feed(sdo_chandet *d, const float *P)
{
  unsigned i;
  float *tmp = malloc(sizeof(float) * d->n), med;
  for (i = 0; i < d->n; ++i) d->S[i] = d->first ? P[i] : d->S[i] + d->alpha * (P[i] - d->S[i]);
  memcpy(tmp, d->S, sizeof(float) * d->n);
  qsort(tmp, d->n, sizeof(float), cmp_float);
  med = tmp[d->n / 2];
  d->N0 = d->first ? med : d->N0 + d->gamma * (med - d->N0);
  d->first = 0;
  free(tmp);
}

unsigned sdo_chandet_find(const sdo_chandet *d, sdo_chandet_record *rec, unsigned cap)
{
  const float thr = d->snr * d->N0;
  const int n = (int)d->n, half = n / 2, GAP = 2, MINW = 2;
  int j, b, t;
  unsigned k = 0;
#define AT(j) (d->S[((j) + half) & (n - 1)])
  for (j = 0; j < n; ++j) {
    int starts = 1, last = j, down = 0, width = 0;
    double sum = 0, wsum = 0;
    float peak = 0;
    if (!(AT(j) > thr)) continue;
    for (b = 1; b <= GAP + 1 && j - b >= 0; ++b) if (AT(j - b) > thr) { starts = 0; break; }
    if (!starts) continue;
    for (t = j; t < n && down <= GAP; ++t) {
      const float p = AT(t);
      if (p > thr) { last = t; down = 0; ++width; sum += (double)p; wsum += (double)p * (double)t; if (p > peak) peak = p; }
      else ++down;
    }
    if (width < MINW) continue;
    if (k < cap) { rec[k].first = j; rec[k].last = last; rec[k].width = width; rec[k].peak = peak; rec[k].sum = sum; rec[k].wsum = wsum; }
    ++k;
  }
#undef AT
  return k < cap ? k : cap;
}

/* ---- Q: the "audio" inspector [SPEC, UPSTREAM-RECOLLECTION] ------------------------------------------------------------
 * SPEC.md section Q (Default/Audio/AudioProcessor.cpp:94-169, 251-270).  One-shot over a whole channel stream. */
size_t sdo_audio_run(const sdo_c32 *x, size_t len, int mode, double efs, double bw, double fa, double cutoff, float volume,
                     sdo_c32 *out, size_t cap)
{
  const double cut = cutoff < 0.45 * fa ? cutoff : 0.45 * fa;
  const float fc = (float)(cut / efs);
  int M = (int)ceil(2.0 / (double)fc), m;
  const double ratio = efs / fa;
  const double w = SDO_PI * bw / efs;
  const uint32_t dphase = (uint32_t)(int64_t)llround((mode == 3 ? w : -w) / (2 * SDO_PI) * 4294967296.0);
  sdo_c32 *a = malloc(sizeof *a * (len ? len : 1));
  size_t i, k = 0;
  if (M < 2) M = 2;
  if (M > 128) M = 128;
  for (i = 0; i < len; ++i) {
    const sdo_c32 v = x[i];
    if (mode == 1) { a[i].re = sqrtf(v.re * v.re + v.im * v.im); a[i].im = 0; }
    else if (mode == 2) {
      const sdo_c32 p = i ? x[i - 1] : (sdo_c32){0, 0};
      a[i].re = sdo_atan2f(v.im * p.re - v.re * p.im, v.re * p.re + v.im * p.im) * 0.318309886183790671538f; a[i].im = 0;
    } else if (mode == 3 || mode == 4) {
      float c, s;
      sdo_phasor_u32((uint32_t)i * dphase, &c, &s);
      a[i].re = v.re * c - v.im * s; a[i].im = 0;
    } else a[i] = v;
  }
  for (k = 0; k < cap; ++k) {
    const double t = (double)k * ratio, fl = floor(t);
    const long long n0 = (long long)fl;
    const float frac = (float)(t - fl), w0 = 3.14159265358979323846f / (float)(M + 1);
    float accx = 0, accy = 0, norm = 0, sc;
    if (n0 + M + 1 > (long long)len - 1) break;
    for (m = -M; m <= M + 1; ++m) {
      const float u = (float)m - frac, arg = 6.28318530717958647692f * fc * u;
      const float sinc = fabsf(arg) < 1e-6f ? 1.0f : sinf(arg) / arg;
      const float g = sinc * (0.5f + 0.5f * cosf(w0 * u));
      const long long j = n0 + m;
      const sdo_c32 v = j >= 0 ? a[j] : (sdo_c32){0, 0};
      accx += v.re * g; accy += v.im * g; norm += g;
    }
    sc = volume / norm;
    out[k].re = accx * sc; out[k].im = accy * sc;
  }
  free(a);
  return k;
}
