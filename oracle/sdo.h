/*
 * oracle/sdo.h -- CPU ORACLE for the SigDigger DSP hot path.   *** TEST INFRASTRUCTURE ONLY ***
 *
 * This is a plain-C restatement of the algorithms on the north-star path.  It is
 * used ONLY as the checker by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.  Nothing under sigdigger_amd/ may include, link or call it.
 *
 * PARITY STATUS: two classes.  (1) [REF-PINNED] functions restate arithmetic that IS present in
 * /root/reference, line by line, and since round 2 are checked against the reference's OWN compiled
 * code: oracle/Makefile.ref builds the cited translation units from where they lie (g++ -O2 + moc,
 * outputs only into oracle/_ref/) and tests/test_ref_pin.py compares this file with them on seeded
 * inputs.  (2) "parity unpinned" versus upstream sigutils/suscan for the rest: the reference tree
 * (the SigDigger GUI) contains no golden vectors and the DSP libraries (sigutils, suscan, FFTW3f;
 * unpinned master, Scripts/dist-common.sh:331-333) are absent;
 * functions marked
 * [SPEC] follow the semantics frozen in SPEC.md (seeded from SURVEY.md Appendix C)
 * and are validated from first principles in tests/ (numpy.fft, scipy.signal,
 * libm, lock/convergence tests).
 *
 * Deterministic arithmetic: every [SPEC] function of the inspector chain is a
 * fixed sequence of IEEE-754 binary32 operations (add, mul, fma, div, compare,
 * int<->float conversion).  Build with -ffp-contract=off (see Makefile) so that
 * only the fmaf() calls written below fuse.  The HIP kernels implement the same
 * sequences, hence bit-exact comparison is possible for that part of the path.
 */
#ifndef SDO_H
#define SDO_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } sdo_c32;       /* == SUCOMPLEX (interleaved f32) */

/* ---- D: deterministic math primitives (SPEC.md section D) ------------------------- */
void  sdo_phasor_u32(uint32_t phase, float *c, float *s);   /* e^{j 2 pi phase / 2^32} */
float sdo_atan2f(float y, float x);
float sdo_log2f(float x);                                   /* x > 0 */
float sdo_exp2f(float x);
uint32_t sdo_fnor_to_dphase(double fnor);                   /* round(fnor * 2^31) mod 2^32 */

/* ---- A3/A4/A9: reference-owned PSD post-processing [REF-PINNED] ------------------- */
void sdo_psd_shift_db(float *psd, size_t n);                /* Suscan/Messages/PSDMessage.cpp:26-39 */
/* Misc/Averager.cpp:25-50.  last/bufsiz are the caller-held state. Returns 1 if it
 * (re)initialised (copy) instead of blending. */
int  sdo_averager_feed(float *last, size_t *bufsiz, const float *x, size_t n, float alpha);
void sdo_inspector_spectrum_db_shift(float *data, size_t len); /* GenericInspector.cpp:231-254 */

/* ---- A2: windowed FFT power spectrum [SPEC] --------------------------------------- */
enum { SDO_WIN_NONE = 0, SDO_WIN_HAMMING, SDO_WIN_HANN, SDO_WIN_FLAT_TOP, SDO_WIN_BLACKMANN_HARRIS };
void sdo_window(int type, float *w, size_t n);
/* double-precision radix-2 FFT, in place, n power of two */
void sdo_fft_f64(double *re, double *im, size_t n);
/* out[o][i] = (1/navg) * sum_{f<navg} |FFT(w .* x[(o*navg+f)*hop ...])[i]|^2 * scale,
 * natural FFT order (DC at index 0); nout = nframes / navg outputs */
void sdo_psd_frames(const sdo_c32 *x, size_t nframes, size_t n, size_t hop,
                    const float *window, size_t navg, float scale, float *out);

/* ---- T1/K4: NCO translate [SPEC] -------------------------------------------------- */
/* y[i] = x[i] * e^{j 2 pi (p0 + (n0+i) dp) / 2^32}; Tasks/CarrierXlator.cpp:36-37,57-60 */
void sdo_xlate_bulk(const sdo_c32 *x, sdo_c32 *y, size_t len, uint32_t p0, uint32_t dp, uint64_t n0);

/* ---- K4+K5: batched translate + decimating low-pass [SPEC] ------------------------ */
void sdo_lpf_design(float *h, size_t ntaps, double fc_nor);  /* hamming-windowed sinc, unit DC gain */
/* g[k] = h[k] * phasor(-(k*dp)) for k<ntaps, zero for ntaps<=k<ntaps_padded */
void sdo_chan_modulate_taps(const float *h, size_t ntaps, uint32_t dp, sdo_c32 *g);
/* One channel. hist = the ntaps-1 input samples preceding x[0] (oldest first).
 * Produces outputs for every absolute input index n = n0+i with n % D == 0:
 *   acc = sum_{k=0}^{ntaps-1} g[k] * xx[n-k]   (k ascending, 4 fmaf per tap)
 *   y   = acc * phasor(p0 + n*dp)
 * returns number of outputs written. */
size_t sdo_chan_feed(const sdo_c32 *hist, const sdo_c32 *x, size_t len, uint64_t n0,
                     const sdo_c32 *g, size_t ntaps, uint32_t D, uint32_t p0, uint32_t dp,
                     sdo_c32 *y);

/* ---- T5/T7/T11: element-wise demodulators [REF-PINNED structure, SPEC arg()] -------- */
/* Tasks/QuadDemodTask.cpp:44-60.  prev = sample before x[0]; first!=0 => dest[0]=0 */
void sdo_quad_demod(const sdo_c32 *x, sdo_c32 *y, size_t len, sdo_c32 prev, int first);
/* Tasks/DelayedConjTask.cpp:70-84 (delay line expressed as direct indexing) */
void sdo_delayed_conj(const sdo_c32 *x, sdo_c32 *y, size_t len, size_t delay);
/* Tasks/HistogramFeeder.cpp:45-66; space: 0 amplitude, 1 phase, 2 frequency. returns count */
size_t sdo_histogram_feed(const sdo_c32 *x, size_t len, int space, float *out);

/* ---- K6/K7: carrier recovery loops [SPEC] ------------------------------------------ */
#define SDO_IIR_MAX_ORDER 4
typedef struct {
  int   order;                              /* number of poles (arm_order - 1)        */
  float b[SDO_IIR_MAX_ORDER + 1];
  float a[SDO_IIR_MAX_ORDER + 1];           /* a[0] = 1                                */
  sdo_c32 xh[SDO_IIR_MAX_ORDER + 1];        /* xh[i] = x[n-i], i>=1                    */
  sdo_c32 yh[SDO_IIR_MAX_ORDER + 1];
} sdo_iir;
void sdo_butter_lp(int order, double fc_nor, float *b, float *a);   /* bilinear Butterworth */

enum { SDO_COSTAS_NONE = 0, SDO_COSTAS_BPSK, SDO_COSTAS_QPSK, SDO_COSTAS_8PSK };
typedef struct {
  int      kind;
  uint32_t phase;                           /* NCO phase, 2^32 per turn               */
  float    omega;                           /* NCO frequency, rad/sample              */
  float    a, b;                            /* loop gains                             */
  float    gain;
  sdo_iir  af;                              /* arm filter                             */
} sdo_costas;
/* su_costas_init(c, kind, fhint, arm_bw, arm_order, loop_bw): Tasks/CostasRecoveryTask.cpp:41 */
int  sdo_costas_init(sdo_costas *c, int kind, float fhint, float arm_bw, unsigned arm_order, float loop_bw);
sdo_c32 sdo_costas_feed(sdo_costas *c, sdo_c32 x);       /* Tasks/CostasRecoveryTask.cpp:58-61 */
void sdo_costas_feed_bulk(sdo_costas *c, const sdo_c32 *x, sdo_c32 *y, size_t len);

typedef struct {
  uint32_t phase;
  float    omega;
  float    alpha, beta;
} sdo_pll;
int  sdo_pll_init(sdo_pll *p, float fhint, float fc);     /* Tasks/PLLSyncTask.cpp:36 */
sdo_c32 sdo_pll_track(sdo_pll *p, sdo_c32 x);             /* Tasks/PLLSyncTask.cpp:53-56 */
void sdo_pll_track_bulk(sdo_pll *p, const sdo_c32 *x, sdo_c32 *y, size_t len);

/* ---- K8: Gardner clock recovery [SPEC] --------------------------------------------- */
typedef struct {
  float alpha, beta, gain;
  float phi, bnor, bmin, bmax;
  int   halfcycle;
  sdo_c32 prev, x0, x1, x2;
} sdo_clock;
/* su_clock_detector_init(cd, loop_gain, bhint, bufsiz): Tasks/WaveSampler.cpp:60-65 */
int    sdo_clock_init(sdo_clock *cd, float loop_gain, float bhint);
/* feeds len samples, appends recovered symbols to out; returns number appended */
size_t sdo_clock_feed_bulk(sdo_clock *cd, const sdo_c32 *x, size_t len, sdo_c32 *out);

/* ---- K9: AGC [SPEC] ------------------------------------------------------------------ */
#define SDO_AGC_MAX_HIST 64
typedef struct {
  float threshold, slope_factor;
  unsigned hang_max, delay_line_size, mag_history_size;
  float fast_rise_t, fast_fall_t, slow_rise_t, slow_fall_t;
} sdo_agc_params;
extern const sdo_agc_params sdo_agc_params_default;       /* su_agc_params_INITIALIZER */
typedef struct {
  float knee, gain_slope, fixed_gain;
  float fast_alpha_rise, fast_alpha_fall, slow_alpha_rise, slow_alpha_fall;
  unsigned hang_max, hang_n, delay_line_size, mag_history_size;
  unsigned delay_ptr, hist_ptr;
  float peak, fast_level, slow_level;
  sdo_c32 delay_line[SDO_AGC_MAX_HIST];
  float   mag_history[SDO_AGC_MAX_HIST];
} sdo_agc;
int  sdo_agc_init(sdo_agc *agc, const sdo_agc_params *p);  /* Tasks/AGCTask.cpp:41-53 */
sdo_c32 sdo_agc_feed(sdo_agc *agc, sdo_c32 x);            /* Tasks/AGCTask.cpp:70-73 */
void sdo_agc_feed_bulk(sdo_agc *agc, const sdo_c32 *x, sdo_c32 *y, size_t len);
/* AGCTask's time constants from tau: Tasks/AGCTask.cpp:22-28,43-47 */
void sdo_agc_params_from_tau(sdo_agc_params *p, float tau);

/* ---- T8: WaveSampler manual / zero-crossing [REF-PINNED] ---------------------------- */
/* Tasks/WaveSampler.cpp:96-175. space: 0 amplitude, 1 phase, 2 frequency.
 * Computes symbols [0, symbol_count). */
void sdo_sample_manual(const sdo_c32 *data, size_t length, double symbol_count,
                       double symbol_sync, int space, sdo_c32 *out, size_t nout);

/* Tasks/WaveSampler.cpp:215-292 (ZERO_CROSSING), all work() calls of one capture concatenated.
 * The reference's block structure is part of the result and is kept literally: blocks of 4096
 * input samples; prevVar (= -1) and prevSample (= 0) are members that sampleZeroCrossing() never
 * writes back, so every block restarts from them; `last` holds for every sample of the final block;
 * at most 4096 symbols per block; lastZc persists.  space as above; out_sym[i] = (var > 0).
 * Returns the number of symbols (<= capacity `nout`, which must be >= 4096 * ceil(length/4096)). */
size_t sdo_sample_zero_crossing(const sdo_c32 *data, size_t length, float bnor, int space, int amplitude,
                                sdo_c32 threshold, sdo_c32 zc_angle, unsigned char *out_sym, size_t nout);
/* Tasks/WaveSampler.cpp:188-196 (GARDNER, FREQUENCY space): y[p] = x[p] conj(x[p-1]), x[-1] = prev0 */
void sdo_conj_prev(const sdo_c32 *x, size_t n, sdo_c32 prev0, sdo_c32 *y);

/* ---- A7: inspector stages behind the rest of the config vocabulary [UPSTREAM-RECOLLECTION] ------- */
/* mf.type = MANUAL, mf.roll-off (Default/GenericInspector/InspectorCtl/MfControl.cpp:56-78):
 * root-raised-cosine taps, n = 2*ceil(3*sps)+1 (six symbols), unit DC gain, double precision */
size_t sdo_rrc_ntaps(double sps);
void   sdo_rrc_design(float *h, size_t ntaps, double sps, double rolloff);
/* real-tap FIR at the channel rate: y[m] = sum_k h[k] x[m-k], k ascending fma chain; hist = the
 * ntaps-1 samples before x (updated) */
void   sdo_fir_feed(sdo_c32 *hist, const float *h, size_t ntaps, const sdo_c32 *x, size_t len, sdo_c32 *y);
/* agc.enabled = false, agc.gain (InspectorCtl/GainControl.cpp:51-60): y = g x */
void   sdo_scale(const sdo_c32 *x, size_t len, float g, sdo_c32 *y);
/* clock.type = MANUAL, clock.phase (InspectorCtl/ClockRecovery.cpp:59-93): the Gardner detector with
 * loop gain 0 (fixed baud) and initial phase 0.5 * phase -> sdo_clock_init(cd, 0, bnor); cd->phi = 0.5f * phase */
/* equalizer.type = CMA, equalizer.rate, equalizer.locked (InspectorCtl/EqualizerControl.cpp:56-75) */
#define SDO_CMA_MAX 16
typedef struct { int n; float mu; int locked; sdo_c32 w[SDO_CMA_MAX], d[SDO_CMA_MAX]; } sdo_cma;
void   sdo_cma_init(sdo_cma *q, int n, float mu);
sdo_c32 sdo_cma_feed(sdo_cma *q, sdo_c32 x);
void   sdo_cma_feed_bulk(sdo_cma *q, const sdo_c32 *x, size_t len, sdo_c32 *y);

/* ---- section 8f #2: inspector spectrum sources [UPSTREAM-RECOLLECTION of libsuscan's spectsrc list] --- */
/* the per-sample transform in front of the inspector PSD; prev0 = last sample of the previous block.
 * kind: 1 psd (identity), 2 cyclo (x conj(prev)), 3 fmspect (arg(x conj(prev)), 0), 4 pmspect (arg x, 0),
 * 5 timediff (x - prev), 6 abstimediff (|x - prev|, 0), 7 exp_2 (x^2), 8 exp_4, 9 exp_8 */
#define SDO_SPECTSRC_COUNT 9
void sdo_spectsrc_preproc(int kind, const sdo_c32 *x, size_t len, sdo_c32 prev0, sdo_c32 *y);

/* ---- section 8f #3: decision space, decider, symbol histogram, SNR estimator ------------------------- */
/* decision space delivered by InspectorUI::feed (Default/GenericInspector/InspectorUI.cpp:863-873) [REF-PINNED]:
 * mode 0 MODULUS |x|, mode 1 ARGUMENT arg(j x)/pi */
void sdo_decision_space(const sdo_c32 *x, size_t len, int mode, float *out);
/* Decider (SuWidgets, absent) [UPSTREAM-RECOLLECTION]; range per InspectorUI.cpp:229-250 (MODULUS [0,1],
 * ARGUMENT [-pi,pi]): v = |x| or arg(x); sym = clamp(floor((v - vmin) / ((vmax - vmin) / 2^bps)), 0, 2^bps - 1) */
void sdo_decide(const sdo_c32 *x, size_t len, int mode, unsigned bps, float vmin, float vmax, unsigned char *sym);
/* history of the Histogram widget [UPSTREAM-RECOLLECTION]: counts of v over nbins equal bins of [vmin, vmax) */
void sdo_symbol_histogram(const sdo_c32 *x, size_t len, int mode, float vmin, float vmax, unsigned nbins, unsigned *hist);
/* SNREstimator (Misc/SNREstimator.cpp:30-169, include/SNREstimator.h) [REF-PINNED] */
typedef struct { float sigma, alpha, hx, delta, sqerr; unsigned bps, intervals, length; } sdo_snr;
void  sdo_snr_init(sdo_snr *e, unsigned bps, float alpha);            /* setBps + setAlpha; sigma = 1/8 */
void  sdo_snr_feed(sdo_snr *e, const unsigned *history, unsigned length, float *model /* Hi, length */);
float sdo_snr_get(const sdo_snr *e);                                  /* getSNR(): 1 / (intervals * sigma) */

/* ---- ingest (section 8f #1): file-source sample formats -> SUCOMPLEX ------------------------------ */
/* format 1 f32, 2 u8 (v-128)/128, 3 s8 v/128, 4 s16 v/32768 [UPSTREAM-RECOLLECTION: libsndfile norm] */
void sdo_ingest_iq(int format, const void *raw, size_t nsamples, sdo_c32 *out);
/* source conditioning (SPEC.md section L) [UPSTREAM-RECOLLECTION: the arithmetic is libsuscan's]: in place, I/Q swap,
 * then the tracked DC level dc[2] (NULL: none) follows the block mean -- dc = first ? mean : dc + alpha (mean - dc) --
 * and is subtracted.  Setters: Suscan/Analyzer.cpp:240-256 */
void sdo_source_fix(sdo_c32 *x, size_t nsamples, int iq_reverse, float *dc, float alpha, int first);

/* ---- T9: carrier centroid [REF-PINNED structure] ------------------------------------ */
/* Tasks/CarrierDetector.cpp:80-143. returns peak in rad/sample */
float sdo_carrier_detect(const sdo_c32 *data, size_t len, float avg_rel_bw, float dc_notch_rel_bw);
void  sdo_blackmann_harris_complex(sdo_c32 *h, size_t n);
/* ---- T10: Doppler centroid [REF-PINNED structure] Tasks/DopplerCalculator.cpp:85-175 --- */
void  sdo_doppler_calc(const sdo_c32 *data, size_t len, float fs, double f0, float *spectrum, float *res);

/* ---- section 8f #4: fast autocorrelation, FACTab::feed [REF-PINNED structure] -------------------- */
/* Default/GenericInspector/FACTab.cpp:181-246: one full buffer of n = 2^k samples -> FFT -> x conj(x) ->
 * inverse FFT (unnormalised) -> |.| of the first half; running max / min over [view_start, view_end);
 * fac[i] += alpha (|.|/max - fac[i]).  fac: n/2 floats, *max / *min: running extrema (init -inf / +inf). */
void sdo_fac_feed(const sdo_c32 *buf, size_t n, float alpha, long view_start, long view_end,
                  float *fac, float *max, float *min);

/* ---- "power" inspector class [REF-PINNED: the raw-mode loop of RMSInspector::samplesMessage] ------------------ */
/* Default/RMSInspector/RMSInspector.cpp:538-562 (Kahan sum of Re(x conj x) in binary64), :327-338 (mean at count >= N) */
typedef struct { double acc, c; unsigned long long count, max_samples; } sdo_power;
void   sdo_power_init(sdo_power *p, unsigned long long max_samples);
size_t sdo_power_feed(sdo_power *p, const sdo_c32 *x, size_t len, sdo_c32 *out /* >= (count + len) / N */);

/* ---- section 8f #2: baud estimators (SPEC.md section M) [UPSTREAM-RECOLLECTION: libsuscan's estimators] --- */
/* nonlinear: y[n] = |x[n] - x[n-1]|^2 (y[0] = 0), Blackman-Harris, FFT; the lowest local maximum of |Y|^2 in
 * [max(4, n/100), n/2) that reaches half of the strongest one -- provided that one stands 20x above the mean level,
 * else there is no estimate (0) --, power centroid over +-4 bins; returns centroid / n.
 * n = 2^k. */
float sdo_baud_nonlinear(const sdo_c32 *x, size_t n);
/* first valley of the 3-tap smoothed autocorrelation below fac[0] / 4; returns the lag in samples (0: none) */
float sdo_fac_first_valley(const float *fac, size_t n_half);

/* ---- P2/P3: SpectrumView [REF-PINNED] ----------------------------------------------- */
#define SDO_SCANNER_SPECTRUM_SIZE 65536
typedef struct {
  double freqMin, freqMax, freqRange;
  unsigned spectrumSize;
  double fftBandwidth;
  float  fftRelBw;
  float *psd, *psdAccum, *psdCount;          /* SDO_SCANNER_SPECTRUM_SIZE each */
} sdo_specview;
void sdo_specview_init(sdo_specview *v, float *psd, float *accum, float *count);
void sdo_specview_set_range(sdo_specview *v, double fmin, double fmax);  /* Scanner.cpp:41-54 */
void sdo_specview_feed(sdo_specview *v, const float *psd, const float *count, size_t psdSize,
                       double freqMin, double freqMax, int adjustSides);  /* Scanner.cpp:239-256 */
void sdo_specview_interpolate(sdo_specview *v);                          /* Scanner.cpp:56-116 */

/* ---- C2: FFT channeliser, su_specttuner semantics (SPEC.md section C2) [UPSTREAM-RECOLLECTION] -------------------------
 * Tasks/LPFTask.cpp:52-69,83-87 (f0, bw, guard in angular units; guard = 2 pi / bw: no decimation) */
typedef struct { unsigned size, halfsz, width, halfw, decimation; int center; double lo; uint32_t dphase; } sdo_st_geom;
void   sdo_specttuner_geometry(unsigned W, double f0, double bw, double guard, sdo_st_geom *g);
void   sdo_specttuner_response(unsigned W, unsigned size, unsigned halfw, sdo_c32 *hk);   /* k h[i], k = 1/W */
void   sdo_specttuner_crossfade(unsigned size, float *win);                              /* sin^2(pi i / size) */
/* one channel over a whole stream: windows k = 0 .. (len - W)/(W/2), halfsz outputs each; returns the count */
size_t sdo_specttuner_run(const sdo_c32 *x, size_t len, unsigned W, double f0, double bw, double guard, int precise,
                          sdo_c32 *out, size_t cap);

/* The binary32 statement of the same channeliser (SPEC.md C2 "binary32 arithmetic"): operation for operation what the
 * device kernels compute (csrc/specttuner_wave.hip for sizes 8 .. 64, csrc/specttuner.hip for the others), so that the
 * default channeliser of the analyzer and of the bench is compared bit for bit.  W = 4096 only. */
void   sdo_st32_tables(sdo_c32 *w64 /*[64]: W_64^m*/, float *win64 /*[64]: sin^2(pi i / 64)*/);   /* the kernels' literal tables */
void   sdo_st32_forward_narrow(const sdo_c32 *win, sdo_c32 *X);    /* DFT_4096 as 64 x 64 (8 x 8 register transforms) */
void   sdo_st32_forward_wide(const sdo_c32 *win, sdo_c32 *X);      /* DFT_4096 as three radix-16 Stockham passes */
size_t sdo_specttuner_bank_f32(const sdo_c32 *x, size_t len, unsigned nchan, const double *f0, const double *bw, const double *guard,
                               const int *precise, size_t w_begin, size_t w_end, sdo_c32 *out, size_t row_stride);
size_t sdo_specttuner_run_f32(const sdo_c32 *x, size_t len, double f0, double bw, double guard, int precise, sdo_c32 *out, size_t cap);

/* ---- O: channel detector, su_channel_detector (SPEC.md section O) [UPSTREAM-RECOLLECTION] ---------------------------- */
typedef struct { unsigned n; float alpha, gamma, snr; int first; float N0; float *S; } sdo_chandet;   /* S: n floats, caller-owned */
typedef struct { int first, last, width; float peak; double sum, wsum; } sdo_chandet_record;
void     sdo_chandet_feed(sdo_chandet *d, const float *P);                          /* P: linear power, natural FFT order */
unsigned sdo_chandet_find(const sdo_chandet *d, sdo_chandet_record *rec, unsigned cap);   /* ordered by first bin */

/* ---- Q: the "audio" inspector class (SPEC.md section Q) [UPSTREAM-RECOLLECTION] ----------------------------------------
 * mode 1 AM, 2 FM, 3 USB, 4 LSB, 5 RAW; efs: channel rate; fa: audio rate; returns the number of output samples */
size_t sdo_audio_run(const sdo_c32 *x, size_t len, int mode, double efs, double bw, double fa, double cutoff, float volume,
                     sdo_c32 *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* SDO_H */
