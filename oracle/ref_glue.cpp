/*
 * oracle/ref_glue.cpp -- TEST INFRASTRUCTURE ONLY (built by oracle/Makefile.ref into oracle/_ref/libsdref.so).
 *
 * Drives the REFERENCE'S OWN translation units -- compiled from where they lie under /root/reference, never copied --
 * so that tests/test_ref_pin.py can pin oracle/sdo.c to them:
 *
 *   ref_*()   extern "C" entry points that construct the reference's classes (SigDigger::Averager, Suscan::PSDMessage,
 *             the Tasks/ work loops, Panoramic SpectrumView, SNREstimator) and run them on caller-supplied arrays;
 *   su_*()    the per-sample libsigutils calls the Tasks make (libsigutils is absent) are NOT here: they resolve in the
 *             product library (csrc/sigutils_host.cpp), so a Task run through ref_*() is the reference's unchanged loop on
 *             the product's per-sample implementation; tests/test_ref_pin.py compares that with the oracle's restatement
 *             bit for bit -- two independent implementations of SPEC.md D - H.  The upstream primitive itself stays
 *             "parity unpinned";
 *   fftwf_*   FFTW3f's five calls (absent) on the oracle's binary64 FFT, rounded to binary32.
 *
 * Nothing under sigdigger_amd/ or include/ refers to this file.
 */
#include <QCoreApplication>
#include <QObject>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>

#include <sigutils/types.h>
#include <sigutils/ncqo.h>
#include <sigutils/pll.h>
#include <sigutils/agc.h>
#include <sigutils/clock.h>
#include <sigutils/iir.h>
#include <sigutils/taps.h>
#include <fftw3.h>
#include <sdo.h>

#include <Averager.h>
#include <Suscan/Messages/PSDMessage.h>
#include <QuadDemodTask.h>
#include <DelayedConjTask.h>
#include <HistogramFeeder.h>
#include <WaveSampler.h>
#include <CarrierDetector.h>
#include <DopplerCalculator.h>
#include <CarrierXlator.h>
#include <AGCTask.h>
#include <CostasRecoveryTask.h>
#include <PLLSyncTask.h>
#include <LPFTask.h>
#include <Scanner.h>
#include <SNREstimator.h>
#include <Decider.h>

static inline sdo_c32 to_sdo(SUCOMPLEX x) { sdo_c32 r = { x.real(), x.imag() }; return r; }
static inline SUCOMPLEX from_sdo(sdo_c32 x) { return SUCOMPLEX(x.re, x.im); }

/* ======================= FFTW3f (absent; SigDigger links it itself) on the oracle's FFT ======================= */
/* The per-sample libsigutils calls the Tasks make (su_ncqo_*, su_pll_*, su_costas_*, su_agc_*, su_clock_detector_*,
 * su_taps_apply_blackmann_harris_complex) are NOT defined here any more: the Tasks' objects resolve them in the PRODUCT
 * library (sigdigger_amd/csrc/sigutils_host.cpp; the link is --no-undefined), through the product's own headers
 * include/sigutils/{ncqo,pll,agc,clock,iir,taps}.h -- north_star: "Tasks/ ... link unchanged". */
extern "C" {

/* ---- FFTW3f ---- */
struct refshim_fftwf_plan_s { int n; fftwf_complex *in, *out; int sign; };
fftwf_plan fftwf_plan_dft_1d(int n, fftwf_complex *in, fftwf_complex *out, int sign, unsigned)
{
  if (n <= 0 || (n & (n - 1))) return nullptr;
  auto *p = new refshim_fftwf_plan_s{ n, in, out, sign };
  return p;
}
void fftwf_execute(const fftwf_plan p)
{
  std::vector<double> re(p->n), im(p->n);
  for (int i = 0; i < p->n; ++i) { re[i] = p->in[i][0]; im[i] = p->sign == FFTW_FORWARD ? p->in[i][1] : -p->in[i][1]; }
  sdo_fft_f64(re.data(), im.data(), (size_t)p->n);
  for (int i = 0; i < p->n; ++i) { p->out[i][0] = (float)re[i]; p->out[i][1] = (float)(p->sign == FFTW_FORWARD ? im[i] : -im[i]); }
}
void  fftwf_destroy_plan(fftwf_plan p) { delete p; }
void *fftwf_malloc(size_t n) { return std::malloc(n); }
void  fftwf_free(void *p) { std::free(p); }

}  // extern "C"

/* SuWidgets' Decider (absent): SPEC.md section K via the oracle */
void Decider::decide(const SUCOMPLEX *data, Symbol *symbols, size_t len) const
{
  sdo_decide(reinterpret_cast<const sdo_c32 *>(data), len, mode == MODULUS ? 0 : 1, bps, min, max, symbols);
}

/* ======================= the reference's own loops ======================= */
static void ensure_app()
{
  static int argc = 1;
  static char arg0[] = "sdref";
  static char *argv[] = { arg0, nullptr };
  if (!QCoreApplication::instance()) new QCoreApplication(argc, argv);
}

static struct suscan_analyzer_psd_msg *make_psd_msg(const float *frame, size_t n)
{
  auto *m = static_cast<suscan_analyzer_psd_msg *>(std::calloc(1, sizeof(suscan_analyzer_psd_msg)));
  m->psd_size = n;
  m->psd_data = static_cast<SUFLOAT *>(std::malloc(n * sizeof(SUFLOAT)));
  std::memcpy(m->psd_data, frame, n * sizeof(float));
  return m;
}

extern "C" {

/* Suscan::PSDMessage::PSDMessage (Suscan/Messages/PSDMessage.cpp:26-39) on one linear frame -> out[n] */
void ref_psd_message(const float *frame, size_t n, float *out)
{
  Suscan::PSDMessage m(make_psd_msg(frame, n));           /* disposed through suscan_analyzer_dispose_message */
  std::memcpy(out, m.get(), n * sizeof(float));
}

/* SigDigger::Averager::feed (Misc/Averager.cpp:25-50) over nframes frames that went through the PSDMessage ctor first
 * (the order of UIMediator::feedPSD); sizes[f] = frame length (a change re-initialises); out = get() after the last */
size_t ref_averager(const float *frames, const size_t *sizes, size_t nframes, float alpha, float *out)
{
  SigDigger::Averager avg;
  avg.setAlpha(alpha);
  size_t off = 0;
  for (size_t f = 0; f < nframes; ++f) {
    Suscan::PSDMessage m(make_psd_msg(frames + off, sizes[f]));
    avg.feed(m);
    off += sizes[f];
  }
  std::memcpy(out, avg.get(), avg.size() * sizeof(float));
  return avg.size();
}

void ref_quad_demod(const SUCOMPLEX *x, SUCOMPLEX *y, size_t len)
{
  ensure_app();
  QuadDemodTask t(x, y, len);
  while (t.work()) {}
}

void ref_delayed_conj(const SUCOMPLEX *x, SUCOMPLEX *y, size_t len, size_t delay)
{
  ensure_app();
  DelayedConjTask t(x, y, len, delay);
  while (t.work()) {}
}

void ref_carrier_xlate(const SUCOMPLEX *x, SUCOMPLEX *y, size_t len, float rel_freq, float phase)
{
  ensure_app();
  SigDigger::CarrierXlator t(x, y, len, rel_freq, phase);
  while (t.work()) {}
}

void ref_agc_task(const SUCOMPLEX *x, SUCOMPLEX *y, size_t len, float tau)
{
  ensure_app();
  AGCTask t(x, y, len, tau);
  while (t.work()) {}
}

void ref_costas_task(const SUCOMPLEX *x, SUCOMPLEX *y, size_t len, int kind, float tau, float loop_bw)
{
  ensure_app();
  CostasRecoveryTask t(x, y, len, tau, loop_bw, (enum sigutils_costas_kind)kind);
  while (t.work()) {}
}

void ref_pll_task(const SUCOMPLEX *x, SUCOMPLEX *y, size_t len, float cutoff)
{
  ensure_app();
  PLLSyncTask t(x, y, len, cutoff);
  while (t.work()) {}
}

/* LPFTask (Tasks/LPFTask.cpp:44-124), unchanged, on the PRODUCT's su_specttuner_* (the GPU channeliser): needs a device */
int ref_lpf_task(const SUCOMPLEX *x, SUCOMPLEX *y, size_t len, float bw)
{
  ensure_app();
  try {
    LPFTask t(x, y, len, bw);
    while (t.work()) {}
  } catch (Suscan::Exception &e) {
    std::fprintf(stderr, "ref_lpf_task: %s\n", e.what());
    return 0;
  }
  return 1;
}

/* HistogramFeeder::work (Tasks/HistogramFeeder.cpp:35-87): space 0 amplitude, 1 phase, 2 frequency; returns count */
size_t ref_histogram_feeder(const SUCOMPLEX *x, size_t len, int space, float *out)
{
  ensure_app();
  SigDigger::SamplingProperties props{};
  props.space = (SigDigger::SamplingSpace)space;
  props.data = x; props.length = len;
  SigDigger::HistogramFeeder t(props);
  size_t n = 0;
  QObject::connect(&t, &SigDigger::HistogramFeeder::data, [&](const float *d, unsigned int q) {
    std::memcpy(out + n, d, q * sizeof(float)); n += q;
  });
  while (t.work()) {}
  return n;
}

/* WaveSampler (Tasks/WaveSampler.cpp:32-333): sync 0 MANUAL / 1 GARDNER / 2 ZERO_CROSSING.  Returns the number of
 * samples / symbols delivered through data(WaveSampleSet), concatenated */
size_t ref_wave_sampler(const SUCOMPLEX *x, size_t len, int sync, int space, double fs, double rate, double loop_gain,
                        double symbol_count, size_t symbol_sync, int amplitude, float thr_re, float thr_im,
                        float zc_re, float zc_im, int dec_mode, unsigned dec_bps, float dec_min, float dec_max,
                        SUCOMPLEX *out, unsigned char *out_sym, size_t cap)
{
  ensure_app();
  SigDigger::SamplingProperties props{};
  props.sync = (SigDigger::SamplingClockSync)sync;
  props.space = (SigDigger::SamplingSpace)space;
  props.fs = fs; props.rate = rate; props.loopGain = loop_gain;
  props.amplitude = amplitude != 0;
  props.threshold = SUCOMPLEX(thr_re, thr_im);
  props.zeroCrossingAngle = SUCOMPLEX(zc_re, zc_im);
  props.data = x; props.length = len;
  props.symbolSync = symbol_sync; props.symbolCount = symbol_count;
  Decider dec;
  dec.setDecisionMode(dec_mode == 0 ? Decider::MODULUS : Decider::ARGUMENT);
  dec.setBps(dec_bps); dec.setMinimum(dec_min); dec.setMaximum(dec_max);
  SigDigger::WaveSampler t(props, &dec);
  size_t n = 0;
  QObject::connect(&t, &SigDigger::WaveSampler::data, [&](SigDigger::WaveSampleSet set) {
    for (size_t i = 0; i < set.len && n < cap; ++i, ++n) { out[n] = set.block[i]; out_sym[n] = set.symbols[i]; }
  });
  while (t.work()) {}
  return n;
}

float ref_carrier_detector(const SUCOMPLEX *x, size_t len, double avg_rel_bw, double dc_notch_rel_bw)
{
  ensure_app();
  SigDigger::CarrierDetector t(x, len, avg_rel_bw, dc_notch_rel_bw);
  while (t.work()) {}
  return t.getPeak();
}

/* DopplerCalculator (Tasks/DopplerCalculator.cpp:52-182): res = {peak, sigma}; spectrum = the taken psd (complex) */
size_t ref_doppler(const SUCOMPLEX *x, size_t len, float fs, double f0, SUCOMPLEX *spectrum, size_t cap, float *res)
{
  ensure_app();
  SigDigger::DopplerCalculator t(f0, x, len, fs);
  while (t.work()) {}
  res[0] = t.getPeak(); res[1] = t.getSigma();
  std::vector<SUCOMPLEX> psd = t.takeSpectrum();
  const size_t n = psd.size() < cap ? psd.size() : cap;
  std::memcpy(static_cast<void *>(spectrum), psd.data(), n * sizeof(SUCOMPLEX));
  return psd.size();
}

/* SpectrumView (Panoramic/Scanner.cpp:27-293) */
void *ref_specview_new(void) { return new SigDigger::SpectrumView(); }
void  ref_specview_destroy(void *v) { delete static_cast<SigDigger::SpectrumView *>(v); }
void  ref_specview_set_range(void *v, double fmin, double fmax) { static_cast<SigDigger::SpectrumView *>(v)->setRange(fmin, fmax); }
void  ref_specview_feed(void *v, const float *psd, const float *count, size_t n, double fmin, double fmax, int adjust)
{
  static_cast<SigDigger::SpectrumView *>(v)->feed(psd, count, n, fmin, fmax, adjust != 0);
}
void  ref_specview_feed_center(void *v, const float *psd, const float *count, size_t n, double center, int adjust)
{
  static_cast<SigDigger::SpectrumView *>(v)->feed(psd, count, n, center, adjust != 0);
}
void  ref_specview_interpolate(void *v) { static_cast<SigDigger::SpectrumView *>(v)->interpolate(); }
void  ref_specview_set_fft(void *v, double fft_bandwidth, float rel_bw)
{
  auto *s = static_cast<SigDigger::SpectrumView *>(v);
  s->fftBandwidth = fft_bandwidth; s->fftRelBw = rel_bw;
}
void  ref_specview_get(void *v, float *psd, float *accum, float *count)
{
  auto *s = static_cast<SigDigger::SpectrumView *>(v);
  std::memcpy(psd, s->psd, sizeof s->psd); std::memcpy(accum, s->psdAccum, sizeof s->psdAccum);
  std::memcpy(count, s->psdCount, sizeof s->psdCount);
}

/* SNREstimator (Misc/SNREstimator.cpp:30-169): nfeeds calls of feed(history); out = {sigma, snr, mse}; model = Hi */
size_t ref_snr_estimator(const unsigned *history, unsigned length, unsigned bps, float alpha, unsigned nfeeds,
                         float *out, float *model)
{
  SigDigger::SNREstimator e;
  e.setBps(bps);
  e.setAlpha(alpha);
  std::vector<unsigned int> h(history, history + length);
  for (unsigned k = 0; k < nfeeds; ++k) e.feed(h);
  out[0] = e.getSigma(); out[1] = e.getSNR(); out[2] = e.getMSE();
  const std::vector<float> &m = e.getModel();
  std::memcpy(model, m.data(), m.size() * sizeof(float));
  return m.size();
}

}  // extern "C"
