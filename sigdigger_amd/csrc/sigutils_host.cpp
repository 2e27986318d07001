// sigutils_host.cpp -- the per-sample libsigutils calls SigDigger's offline Tasks make, served by the product on the HOST
// (include/sigutils/{ncqo,pll,agc,clock,iir,taps}.h):
//
//   su_ncqo_*            Tasks/CarrierXlator.cpp:36-37,57-60
//   su_pll_*             Tasks/PLLSyncTask.cpp:36,53-56,84
//   su_costas_*          Tasks/CostasRecoveryTask.cpp:41,58-61,89
//   su_agc_*             Tasks/AGCTask.cpp:41-53,70-73,101
//   su_clock_detector_*  Tasks/WaveSampler.cpp:60-65,88,192-205
//   su_iir_rrc_init / su_iir_filt_*   Tasks/WaveSampler.cpp:74,92 (compiled out of the reference by default)
//   su_taps_apply_blackmann_harris_complex   Tasks/CarrierDetector.cpp:87-89, Tasks/DopplerCalculator.cpp:92
//
// Why host code in a GPU library: the reference embeds these states BY VALUE in its task objects and calls the functions
// once per sample from `while (amount--)` loops -- nothing a device can serve one call at a time -- and the north star asks
// that those Tasks link unchanged.  So the calls are here, with the reference's signatures, and they run the SAME fixed
// sequences of binary32 operations as the device kernels (SPEC.md sections D - H; the primitives are the shared source
// csrc/sd_math.hpp, compiled for both sides): a Task linked against this library produces, bit for bit, what the block
// entry points of include/sigdigger_amd.h (suamd_*_bank_feed, suamd_xlate_bulk) produce on the GPU for the same
// samples.  The GPU win for a Task is the block call; this file is what makes the unpatched Task link.
//
// Built with -ffp-contract=off like every bit-pinned translation unit: only the fma calls written below fuse.
#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/sigutils/types.h"
#include "../../include/sigutils/ncqo.h"
#include "../../include/sigutils/pll.h"
#include "../../include/sigutils/agc.h"
#include "../../include/sigutils/clock.h"
#include "../../include/sigutils/iir.h"
#include "../../include/sigutils/taps.h"
#include "design.hpp"
#include "sd_math.hpp"

void suamd_set_error(const char *fmt, ...);                // capi.hip

namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr float kPiF = 3.14159265358979323846f, kTwoPiF = 6.28318530717958647692f;

inline sd::c32 to_c32(SUCOMPLEX x) { return sd::c32{x.real(), x.imag()}; }
inline SUCOMPLEX from_c32(sd::c32 x) { return SUCOMPLEX(x.re, x.im); }

}  // namespace

extern "C" {

// ---- NCO (SPEC.md C, "Translate"): phase(n) = phase + n dphase, 2^32 per turn ----------------------------------------
SUAMD_API void su_ncqo_init(su_ncqo_t *ncqo, SUFLOAT fnor)
{
  ncqo->phase = 0;
  ncqo->n = 0;
  ncqo->dphase = suamd_fnor_to_dphase((double)fnor);
}

SUAMD_API void su_ncqo_set_phase(su_ncqo_t *ncqo, SUFLOAT phi)
{
  const uint32_t now = (uint32_t)(int64_t)std::llround((double)phi / (2.0 * kPi) * 4294967296.0);
  ncqo->phase = now - (uint32_t)ncqo->n * ncqo->dphase;    // the phase of the NEXT read
}

SUAMD_API void su_ncqo_set_freq(su_ncqo_t *ncqo, SUFLOAT fnor)
{
  ncqo->phase += (uint32_t)ncqo->n * ncqo->dphase;
  ncqo->n = 0;
  ncqo->dphase = suamd_fnor_to_dphase((double)fnor);
}

SUAMD_API SUCOMPLEX su_ncqo_read(su_ncqo_t *ncqo)
{
  float c, s;
  sd::phasor_u32(ncqo->phase + (uint32_t)ncqo->n * ncqo->dphase, c, s);
  ++ncqo->n;
  return SUCOMPLEX(c, s);
}

// ---- PLL (SPEC.md F) -----------------------------------------------------------------------------------------------------
SUAMD_API SUBOOL su_pll_init(su_pll_t *pll, SUFLOAT fhint, SUFLOAT fc)
{
  if (!pll) return SU_FALSE;
  const double w = kPi * (double)fc;
  const double dinv = 1.0 / (1.0 + 2.0 * 0.707 * w + w * w);
  pll->phase = 0;
  pll->alpha = (float)(4.0 * w * w * dinv);
  pll->beta = (float)(4.0 * 0.707 * w * dinv);
  pll->omega = (float)(kPi * (double)fhint);
  return SU_TRUE;
}

SUAMD_API SUCOMPLEX su_pll_track(su_pll_t *pll, SUCOMPLEX xin)
{
  const sd::c32 x = to_c32(xin);
  float c, s;
  sd::phasor_u32(pll->phase, c, s);
  const sd::c32 out = sd::cmul_conj(x, sd::c32{c, s});
  float err = sd::atan2_(x.im, x.re) - sd::phase_to_rad(pll->phase);
  if (err > kPiF) err -= kTwoPiF;
  if (err < -kPiF) err += kTwoPiF;
  const float dphi = sd::fma_(pll->beta, err, pll->omega);
  pll->omega = sd::fma_(pll->alpha, err, pll->omega);
  pll->phase += (uint32_t)sd::rad_to_dphase(dphi);
  return from_c32(out);
}

SUAMD_API void su_pll_finalize(su_pll_t *) {}

// ---- Costas loop (SPEC.md E) ---------------------------------------------------------------------------------------------
SUAMD_API SUBOOL su_costas_init(su_costas_t *costas, enum sigutils_costas_kind kind, SUFLOAT fhint, SUFLOAT arm_bw,
                                unsigned int arm_order, SUFLOAT loop_bw)
{
  if (!costas) return SU_FALSE;
  if (kind != SU_COSTAS_KIND_BPSK && kind != SU_COSTAS_KIND_QPSK && kind != SU_COSTAS_KIND_8PSK) {
    suamd_set_error("su_costas_init: unsupported kind %d", (int)kind);
    return SU_FALSE;
  }
  if (arm_order == 0) arm_order = 1;
  if (arm_order > SU_COSTAS_MAX_ARM_ORDER) { suamd_set_error("su_costas_init: arm_order %u unsupported (<= %d)", arm_order, SU_COSTAS_MAX_ARM_ORDER); return SU_FALSE; }
  std::memset(costas, 0, sizeof *costas);
  costas->kind = (int)kind;
  costas->a = (float)(kPi * (double)loop_bw);
  costas->b = 0.5f * costas->a * costas->a;
  costas->gain = 1.0f;
  costas->omega = (float)(kPi * (double)fhint);
  costas->order = (int)arm_order - 1;
  sdk_design::butter_lp(costas->order, (double)arm_bw, costas->fb, costas->fa);
  return SU_TRUE;
}

SUAMD_API void su_costas_set_loop_gain(su_costas_t *costas, SUFLOAT gain) { if (costas) costas->gain = gain; }

SUAMD_API SUCOMPLEX su_costas_feed(su_costas_t *costas, SUCOMPLEX xin)
{
  su_costas_t &k = *costas;
  float c, s;
  sd::phasor_u32(k.phase, c, s);
  const sd::c32 m = sd::cmul_conj(to_c32(xin), sd::c32{c, s});
  // arm filter: direct form I, the history first (oldest term first), the new sample last
  float tr = 0.0f, ti = 0.0f;
  for (int i = k.order; i >= 1; --i) { tr = sd::fma_(k.fb[i], k.xh[i][0], tr); ti = sd::fma_(k.fb[i], k.xh[i][1], ti); }
  for (int i = k.order; i >= 1; --i) { tr = sd::fma_(-k.fa[i], k.yh[i][0], tr); ti = sd::fma_(-k.fa[i], k.yh[i][1], ti); }
  sd::c32 z{sd::fma_(k.fb[0], m.re, tr), sd::fma_(k.fb[0], m.im, ti)};
  for (int i = k.order; i >= 2; --i) {
    k.xh[i][0] = k.xh[i - 1][0]; k.xh[i][1] = k.xh[i - 1][1];
    k.yh[i][0] = k.yh[i - 1][0]; k.yh[i][1] = k.yh[i - 1][1];
  }
  if (k.order >= 1) { k.xh[1][0] = m.re; k.xh[1][1] = m.im; k.yh[1][0] = z.re; k.yh[1][1] = z.im; }
  z.re = k.gain * z.re;
  z.im = k.gain * z.im;
  float e;
  const float si = sd::sgn(z.re), sq = sd::sgn(z.im);
  if (k.kind == SU_COSTAS_KIND_BPSK) e = z.re * z.im;
  else if (k.kind == SU_COSTAS_KIND_QPSK) e = si * z.im - sq * z.re;
  else if (__builtin_fabsf(z.re) >= __builtin_fabsf(z.im)) e = si * z.im - (sq * z.re) * 0.41421356237309504880f;
  else e = (si * z.im) * 0.41421356237309504880f - sq * z.re;
  const float dphi = sd::fma_(k.a, e, k.omega);
  k.omega = sd::fma_(k.b, e, k.omega);
  k.phase += (uint32_t)sd::rad_to_dphase(dphi);
  return from_c32(z);
}

SUAMD_API void su_costas_finalize(su_costas_t *) {}

// ---- AGC (SPEC.md H) -------------------------------------------------------------------------------------------------------
SUAMD_API SUBOOL su_agc_init(su_agc_t *agc, const struct su_agc_params *p)
{
  if (!agc || !p) return SU_FALSE;
  if (p->delay_line_size == 0 || p->delay_line_size > SU_AGC_MAX_HISTORY || p->mag_history_size == 0 || p->mag_history_size > SU_AGC_MAX_HISTORY) {
    suamd_set_error("su_agc_init: delay_line_size / mag_history_size must be in 1..%d", SU_AGC_MAX_HISTORY);
    return SU_FALSE;
  }
  std::memset(agc, 0, sizeof *agc);
  agc->knee = p->threshold;
  agc->gain_slope = p->slope_factor * 1e-2f;
  agc->hang_max = p->hang_max;
  agc->delay_line_size = p->delay_line_size;
  agc->mag_history_size = p->mag_history_size;
  agc->fast_alpha_rise = (float)(1.0 - std::exp(-1.0 / (double)p->fast_rise_t));
  agc->fast_alpha_fall = (float)(1.0 - std::exp(-1.0 / (double)p->fast_fall_t));
  agc->slow_alpha_rise = (float)(1.0 - std::exp(-1.0 / (double)p->slow_rise_t));
  agc->slow_alpha_fall = (float)(1.0 - std::exp(-1.0 / (double)p->slow_fall_t));
  return SU_TRUE;
}

SUAMD_API SUCOMPLEX su_agc_feed(su_agc_t *agc, SUCOMPLEX xin)
{
  su_agc_t &g = *agc;
  const float xr = xin.real(), xi = xin.imag();
  // the sample that leaves the delay line is the one the gain is applied to
  const float dr = g.delay_line[g.delay_ptr][0], di = g.delay_line[g.delay_ptr][1];
  g.delay_line[g.delay_ptr][0] = xr; g.delay_line[g.delay_ptr][1] = xi;
  if (++g.delay_ptr == g.delay_line_size) g.delay_ptr = 0;
  // magnitude in dB and its maximum over the last mag_history_size samples (the history starts at 0 dB)
  const float x_db = 3.01029995663981195f * sd::log2_(sd::fma_(xr, xr, xi * xi) + 1e-8f);
  g.mag_history[g.hist_ptr] = x_db;
  if (++g.hist_ptr == g.mag_history_size) g.hist_ptr = 0;
  float peak = g.mag_history[0];
  for (unsigned i = 1; i < g.mag_history_size; ++i) if (peak < g.mag_history[i]) peak = g.mag_history[i];
  // level trackers
  float d = peak - g.fast_level;
  g.fast_level = sd::fma_(d > 0.0f ? g.fast_alpha_rise : g.fast_alpha_fall, d, g.fast_level);
  d = peak - g.slow_level;
  if (d > 0.0f) { g.slow_level = sd::fma_(g.slow_alpha_rise, d, g.slow_level); g.hang_n = 0; }
  else if (g.hang_n >= g.hang_max) g.slow_level = sd::fma_(g.slow_alpha_fall, d, g.slow_level);
  else ++g.hang_n;
  float lvl = g.fast_level > g.slow_level ? g.fast_level : g.slow_level;
  if (lvl < g.knee) lvl = g.knee;
  const float g_db = lvl * (g.gain_slope - 1.0f);
  const float gain = sd::exp2_(g_db * 0.166096404744368117f) * 0.7f;
  return SUCOMPLEX(dr * gain, di * gain);
}

SUAMD_API void su_agc_finalize(su_agc_t *) {}

// ---- Gardner clock detector (SPEC.md G) ----------------------------------------------------------------------------------
SUAMD_API SUBOOL su_clock_detector_init(su_clock_detector_t *cd, SUFLOAT loop_gain, SUFLOAT bhint, SUSCOUNT bufsiz)
{
  if (!cd || !(bhint > 0.0f)) return -1;                      // compared with -1: Tasks/WaveSampler.cpp:60-65
  std::memset(cd, 0, sizeof *cd);
  cd->alpha = 2e-1f;
  cd->beta = 1.2e-4f;
  cd->gain = loop_gain;
  cd->phi = 0.25f;
  cd->bnor = bhint;
  cd->bmin = 0.5f * bhint;
  cd->bmax = bhint > 0.5f ? 1.0f : 2.0f * bhint;
  // at most one symbol per sample fed, and a caller feeds up to bufsiz samples between two reads of up to bufsiz
  cd->size = 2 * (bufsiz ? bufsiz : 1) + 16;
  cd->buf = static_cast<SUCOMPLEX *>(std::malloc(cd->size * sizeof(SUCOMPLEX)));
  if (!cd->buf) { suamd_set_error("su_clock_detector_init: out of memory"); return -1; }
  return SU_TRUE;
}

SUAMD_API void su_clock_detector_set_baud(su_clock_detector_t *cd, SUFLOAT bnor)
{
  if (!cd || !(bnor > 0.0f)) return;
  cd->bnor = bnor; cd->bmin = 0.5f * bnor; cd->bmax = bnor > 0.5f ? 1.0f : 2.0f * bnor;
}

SUAMD_API void su_clock_detector_feed(su_clock_detector_t *cd, SUCOMPLEX xin)
{
  su_clock_detector_t &k = *cd;
  const float vr = xin.real(), vi = xin.imag();
  k.phi = k.phi + k.bnor;
  if (k.phi >= 0.5f) {
    // the half-symbol instant lay mu samples before this one
    const float mu = (k.phi - 0.5f) / k.bnor;
    const float qr = sd::fma_(mu, k.prev[0] - vr, vr), qi = sd::fma_(mu, k.prev[1] - vi, vi);
    k.phi = k.phi - 0.5f;
    k.halfcycle = !k.halfcycle;
    if (!k.halfcycle) {                                        // a full cycle: a symbol
      k.x2[0] = k.x0[0]; k.x2[1] = k.x0[1];
      k.x0[0] = qr; k.x0[1] = qi;
      const float er = k.x0[0] - k.x2[0], ei = k.x0[1] - k.x2[1];
      const float e = k.gain * sd::fma_(k.x1[1], ei, k.x1[0] * er);
      k.phi = sd::fma_(k.alpha, e, k.phi);
      float b = sd::fma_(k.beta, e, k.bnor);
      if (b < k.bmin) b = k.bmin;
      if (b > k.bmax) b = k.bmax;
      k.bnor = b;
      if (k.avail == k.size) {                                 // never read: keep the newest
        std::memmove(static_cast<void *>(k.buf), k.buf + 1, (k.size - 1) * sizeof(SUCOMPLEX));
        --k.avail;
      }
      k.buf[k.avail++] = SUCOMPLEX(qr, qi);
    } else {
      k.x1[0] = qr; k.x1[1] = qi;
    }
  }
  k.prev[0] = vr; k.prev[1] = vi;
}

SUAMD_API SUSDIFF su_clock_detector_read(su_clock_detector_t *cd, SUCOMPLEX *buf, size_t size)
{
  if (!cd || !buf) return -1;
  const size_t n = cd->avail < size ? (size_t)cd->avail : size;
  std::memcpy(static_cast<void *>(buf), cd->buf, n * sizeof(SUCOMPLEX));
  std::memmove(static_cast<void *>(cd->buf), cd->buf + n, ((size_t)cd->avail - n) * sizeof(SUCOMPLEX));
  cd->avail -= n;
  return (SUSDIFF)n;
}

SUAMD_API void su_clock_detector_finalize(su_clock_detector_t *cd)
{
  if (!cd) return;
  std::free(cd->buf);
  cd->buf = nullptr; cd->size = cd->avail = 0;
}

// ---- matched filter (SPEC.md I): root-raised-cosine FIR, k ascending, one fma per component ------------------------
SUAMD_API SUBOOL su_iir_rrc_init(su_iir_filt_t *filt, SUSCOUNT n, SUFLOAT T, SUFLOAT beta)
{
  if (!filt || n == 0 || !(T > 0.0f) || !(beta >= 0.0f) || beta > 1.0f) { suamd_set_error("su_iir_rrc_init: bad parameters"); return SU_FALSE; }
  SUSCOUNT taps = (SUSCOUNT)std::ceil((double)n * (double)T) + 1;
  if (!(taps & 1)) ++taps;
  if (taps > (1u << 20)) { suamd_set_error("su_iir_rrc_init: %llu taps", (unsigned long long)taps); return SU_FALSE; }
  filt->n = taps;
  filt->h = static_cast<SUFLOAT *>(std::malloc(taps * sizeof(SUFLOAT)));
  filt->d = static_cast<SUCOMPLEX *>(std::calloc(taps, sizeof(SUCOMPLEX)));
  if (!filt->h || !filt->d) { su_iir_filt_finalize(filt); suamd_set_error("su_iir_rrc_init: out of memory"); return SU_FALSE; }
  suamd_rrc_design(filt->h, (unsigned)taps, (double)T, (double)beta);
  return SU_TRUE;
}

SUAMD_API SUCOMPLEX su_iir_filt_feed(su_iir_filt_t *filt, SUCOMPLEX x)
{
  if (!filt || !filt->n) return x;
  std::memmove(static_cast<void *>(filt->d + 1), filt->d, (size_t)(filt->n - 1) * sizeof(SUCOMPLEX));
  filt->d[0] = x;
  float yr = 0.0f, yi = 0.0f;
  for (SUSCOUNT k = 0; k < filt->n; ++k) { yr = sd::fma_(filt->h[k], filt->d[k].real(), yr); yi = sd::fma_(filt->h[k], filt->d[k].imag(), yi); }
  return SUCOMPLEX(yr, yi);
}

SUAMD_API void su_iir_filt_finalize(su_iir_filt_t *filt)
{
  if (!filt) return;
  std::free(filt->h); std::free(filt->d);
  filt->h = nullptr; filt->d = nullptr; filt->n = 0;
}

// ---- window ------------------------------------------------------------------------------------------------------------------
SUAMD_API void su_taps_apply_blackmann_harris_complex(SUCOMPLEX *h, SUSCOUNT size)
{
  for (SUSCOUNT i = 0; i < size; ++i) {
    const double t = 2.0 * kPi * (double)i / (double)(size - 1);
    const float w = (float)(0.35875 - 0.48829 * std::cos(t) + 0.14128 * std::cos(2 * t) - 0.01168 * std::cos(3 * t));
    h[i] = SUCOMPLEX(h[i].real() * w, h[i].imag() * w);
  }
}

SUAMD_API void su_taps_apply_blackmann_harris(SUFLOAT *h, SUSCOUNT size)
{
  for (SUSCOUNT i = 0; i < size; ++i) {
    const double t = 2.0 * kPi * (double)i / (double)(size - 1);
    h[i] *= (float)(0.35875 - 0.48829 * std::cos(t) + 0.14128 * std::cos(2 * t) - 0.01168 * std::cos(3 * t));
  }
}

}  // extern "C"
