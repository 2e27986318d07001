// kernels.hpp -- internal launch interface between the C-ABI layer (capi.hip) and the
// gfx950 kernels.  Not installed; the public boundary is include/sigdigger_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

namespace sdk {

// ---- kernel timer (capi.hip): measurement aid behind suamd_kernel_timing() --------------------------------------
// While it is on, the launches that go through launch_timed() carry a start / stop event pair bound to the dispatch
// itself (hipExtLaunchKernelGGL): the elapsed time of the pair is the kernel's own duration, as rocprofv3 reports it,
// not the distance of two stream events around the launch (which adds the queue's gaps, 3-6 us per launch).
bool timing_on();
void timing_pair(const char *name, hipEvent_t *start, hipEvent_t *stop);
// The pair's START is a marker the runtime enqueues in front of the kernel's packet, its stop is the kernel's own end: if
// the host is held up between writing the two packets while the queue is empty (a page fault, the allocator, a descheduled
// thread), the marker is stamped at once and the kernel starts a millisecond later -- an event pair of a 77 us kernel
// then reads 0.94 ms (round 6: one such sample in 20 turned roofline.frac 0.44 into 0.28; rocprofv3, which reads the
// dispatch's own timestamps, never shows such a launch).  So a timed launch is preceded by a GATE: a one-lane kernel that
// waits (bounded: 20 ms) for a word in pinned host memory which the host sets once both packets are in the queue.
// timing_gate() launches it on `st` and returns the word (nullptr: no gate, the launch is timed as before).
volatile unsigned *timing_gate(hipStream_t st);
template <class K, class... A>
inline void launch_timed(const char *name, K kern, dim3 grid, dim3 block, size_t lds, hipStream_t st, A... args)
{
  if (timing_on()) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    timing_pair(name, &e0, &e1);
    volatile unsigned *gate = timing_gate(st);
    hipExtLaunchKernelGGL(kern, grid, block, lds, st, e0, e1, 0, args...);
    if (gate) __atomic_store_n(const_cast<unsigned *>(gate), 1u, __ATOMIC_RELEASE);
  } else hipLaunchKernelGGL(kern, grid, block, lds, st, args...);
}

// A batch of nchan sample rows: element (c, m) lives at base[c*cs + m*ms] (units: complex samples).
//   channel-major  [c][m]: cs = row pitch, ms = 1   (what consumers of sample batches want)
//   time-major     [m][c]: cs = 1, ms = nchan pitch (what one-lane-per-channel kernels want:
//                                                    a wavefront's access is one contiguous 512 B)
struct View { long long cs, ms; };

// ---- psd.hip ----
// partial: scratch for split-frame accumulation, >= nout * psd_split(nout, navg, log2n) * N floats (may be
// nullptr when psd_split() == 1)
int        psd_split(long long nout, int navg, int log2n);
void       psd_split_target(int target);      // this thread's next psd_split() calls aim at `target` workgroups (0: the defaults)
hipError_t psd_frames(int log2n, const void *x, long long hop, int navg, const float *window,
                      const void *tw, float scale, int mode, float *out, long long nout, float *partial,
                      hipStream_t st);
hipError_t psd_frames_32k(const void *x, long long hop, int navg, const float *window, const void *tw, const void *tw2,
                          float scale, int mode, float *out, long long nout, float *partial, hipStream_t st);
hipError_t psd_shift_db(float *psd, long long n, long long nframes, hipStream_t st);
hipError_t averager_feed(float *last, const float *x, long long n, float alpha, int blend, hipStream_t st);
hipError_t insp_spectrum_db_shift(float *data, long long len, long long nspec, hipStream_t st);

// ---- chan.hip ----
hipError_t xlate_bulk(const void *x, void *y, long long len, uint32_t p0, uint32_t dp, uint64_t n0, hipStream_t st);
// g: float4 [nchan][ntaps] (re, re, -im, im); g2 (may be null): the same taps as compact (re, im) pairs, [nchan][ntaps] + 64 spare entries
hipError_t chan_modulate_taps(const float *h, int ntaps, const uint32_t *dphase, int nchan, void *g, void *g2, hipStream_t st);
struct ChanFeedArgs {
  const void *x;        // input block (device), len samples
  const void *hist;     // ntaps-1 samples preceding x[0] (device)
  void       *hist_next; // receives the history for the next block (ping-pong with hist)
  long long   len;
  uint64_t    n0;       // absolute index of x[0]
  const void *g;        // float4 [nchan][ntaps] modulated taps (re, re, -im, im)
  const void *g2;       // float2 [nchan][ntaps] (+ 64 spare) the same taps as (re, im): the stream kernel's scalar loads; may be null
  const uint32_t *dphase;   // [nchan]
  const uint32_t *phase0;   // [nchan]
  int         ntaps, nchan;
  uint32_t    D;
  uint64_t    m_first;  // first output index (n = m*D)
  long long   n_out;
  void       *y;        // element (c, m) at y[c*yv.cs + m*yv.ms]
  View        yv;
  int         exclusive = 0;   // the caller has the device to itself while this feed runs (suamd_chanbank_set_exclusive)
};
hipError_t chan_feed(const ChanFeedArgs &a, hipStream_t st);
// chan_stream.hip: few channels as a stream (LDS-DMA ring); false = not this kernel's shape (nothing launched)
bool chan_stream_feed(const ChanFeedArgs &a, const void *g2, hipStream_t st, hipError_t *err);
// tiling of one feed (chan.hip plans it)
struct FirGeom {
  int D, ntaps, nchan;
  int MT;            // outputs per workgroup tile (multiple of 64)
  int KD;            // ceil((ntaps-1)/D)*D : window starts KD samples before the tile's first output
  int span;          // samples staged = MT*D + KD   (window index i <-> absolute n = nbase + i)
  int PAD;           // LDS index of window sample i = i + (i / D) * PAD, with D + PAD odd
  int lds_samples;   // padded window size in samples
};
// many 1-channel banks (their own taps, carrier, decimation, state) fed in ONE launch, grid.y = bank:
// chan_gang_plan fills an item per bank, the caller puts the table on the device, chan_gang_feed launches
struct ChanGangItem { ChanFeedArgs a; FirGeom ge; unsigned ntiles; unsigned lds; int sparse; };
hipError_t chan_gang_plan(const ChanFeedArgs &a, ChanGangItem *item);
hipError_t chan_gang_feed(const ChanGangItem *d_items, int n, unsigned max_tiles, unsigned max_lds, hipStream_t st);
hipError_t chan_update_hist(void *hist_next, const void *hist, const void *x, long long len, int ntaps, hipStream_t st);

// ---- specttuner.hip: FFT channeliser (SPEC.md section C2) ----
struct StChan {                 // one channel of a size group
  int center;                   // centre bin (even) of the W-point spectrum
  int hsel;                     // which response table hk[hsel][size]
  int row;                      // output row: element m of the feed at y[row*cs + m*ms]
  int precise;                  // residual NCO on
  uint32_t dphase;              // its step per OUTPUT sample
  unsigned long long n_open;    // the group's output counter when the channel was opened (its NCO starts there)
};
struct StArgs {
  const void *x;                // len samples (device); the virtual stream is hist ++ x when have_hist
  const void *hist;             // the W/2 samples before x[0]
  void *hist_out;               // wavefront kernel, optional: receives the last W/2 samples of x (the next feed's history)
  int have_hist;
  long long nwin;               // windows in this feed: window w = virtual samples [w W/2, w W/2 + W)
  int run;                      // windows per workgroup
  const void *tw_w, *tw_s;      // W- and size-point twiddle tables (complex)
  const StChan *chans; int nchan;
  const void *hk;               // [nsel][size] complex: k h[i] (zero outside the pass band)
  unsigned long long *tstamp;   // STW_TSTAMP builds only: [run][window of the run][16] s_memtime stamps
  int y32;                      // wavefront kernel: every byte offset (row * cs + time * ms) * 8 of this feed's outputs is below 2^31
  int hk_uniform;               // every channel of the launch has hsel == chans[0].hsel (wavefront kernel: response kept in LDS)
  int nsel;                     // response tables in hk (distinct pass-band widths of the launch's channels)
  int seam_polls;               // wavefront kernel: polls (~0.25 us each) a run waits for its successor's payload before it transforms the seam window itself
  unsigned epoch;               // wavefront kernel: the flag value of this launch (a host counter per size group, never 0)
  void *handoff; unsigned *flags;   // wavefront kernel: [runs * blocks][16 KiB] seam payload, one flag per run (holds the epoch of the launch that published it)
  const void *hkt;              // the same per channel, transposed for the wavefront kernel: [ceil(nchan/64)][size][64]
  const float *win;             // [size]: sin^2(pi i / size)
  const void *prev_in; void *prev_out;   // [nchan][size/2]: second half of the last block of the previous feed (ping-pong)
  unsigned long long n0;        // outputs per channel emitted by earlier feeds (phase of the residual NCO)
  void *y; View yv;
  const void *const *rows;      // when set: channel `row`'s samples of this feed start at rows[row] (unit time stride), y / yv.cs unused
  int any_precise;              // two-wavefront kernel: 0 promises that no channel of the launch is precise, 2 that all are (instantiations with one form of the channel stage only), 1: some
};
// channels one workgroup serves side by side for inverse transforms of 2^log2s points
int st_channels_per_group(int log2s);
int st_plan_run(int log2s, int nchan, long long nwin);      // windows per workgroup that fill one round of the chip
hipError_t specttuner_feed(int log2w, int log2s, const StArgs &a, hipStream_t st);
// specttuner_wave.hip: one wavefront per window, sizes 8..64 (W = 4096); channels one wavefront serves
int stw_channels_per_wave(int log2s);
hipError_t specttuner_feed_wave(int log2s, const StArgs &a, hipStream_t st);
// specttuner_pair.hip: two wavefronts per window; channels of 8 .. 64 bins, a.nsel <= stp_max_responses() (the tables live
// in LDS), a.run >= 2 (same results bit for bit)
int stp_max_responses();
hipError_t specttuner_feed_pair(int log2s, const StArgs &a, hipStream_t st);

// ---- chandet.hip: su_channel_detector (SPEC.md section O) ----
struct ChanDetRecord { int first, last, width; float peak; double sum, wsum; };   // bins in frequency order (0 = -fs/2)
// n = 2^k <= 16384 bins, linear power in natural FFT order
hipError_t chandet_feed(float *S, const float *P, int n, float alpha, float gamma, int first, float *N0, hipStream_t st);
hipError_t chandet_find(const float *S, int n, const float *N0, float snr, ChanDetRecord *rec, unsigned *count, unsigned cap,
                        hipStream_t st);

// ---- loops.hip ----
hipError_t quad_demod_batch(const void *x, View xv, void *y, View yv, int nchan, long long len,
                            const void *prev, int first, void *prev_out, hipStream_t st);
hipError_t delayed_conj_bulk(const void *x, void *y, long long len, long long delay, hipStream_t st);
hipError_t histogram_feed_bulk(const void *x, long long len, int space, float *out, hipStream_t st);
hipError_t sample_manual_bulk(const void *data, long long length, double symbol_count, double symbol_sync, int space,
                              void *out, long long nout, hipStream_t st);
hipError_t conj_prev_bulk(const void *x, void *y, long long len, float prev_re, float prev_im, hipStream_t st);
// WaveSampler ZERO_CROSSING stages (blocks of 4096 input samples)
hipError_t zc_var(const void *data, long long length, int space, int amplitude, float thr_re, float thr_im,
                  float ang_re, float ang_im, float *var, hipStream_t st);
hipError_t zc_scan(const float *var, long long length, long long nblocks, long long *last_pos, hipStream_t st);
hipError_t zc_emit(const float *var, long long length, long long nblocks, float bnor, const long long *last_pos,
                   unsigned char *seg, unsigned *count, hipStream_t st);
hipError_t zc_compact(const unsigned char *seg, const unsigned *count, const unsigned long long *offset,
                      unsigned char *out, long long nblocks, hipStream_t st);

struct CostasState {            // SoA over channels, all device pointers
  uint32_t *phase; float *omega;
  float *xh; float *yh;         // [4][2][nchan] : history index, re/im, channel
};
struct CostasParams { int kind; int order; float a, b, gain; float fb[5]; float fa[5]; };
hipError_t costas_feed(const CostasParams &p, const CostasState &s, int nchan, const void *x, View xv,
                       void *y, View yv, long long len, hipStream_t st);

struct PllState { uint32_t *phase; float *omega; };
hipError_t pll_feed(float alpha, float beta, const PllState &s, int nchan, const void *x, View xv,
                    void *y, View yv, long long len, hipStream_t st);

struct ClockState {             // all [nchan]
  float *phi, *bnor; int *halfcycle; float *prev, *x0, *x1, *x2;   // complex ones: [2][nchan]
};
struct ClockParams { float alpha, beta, gain, bmin, bmax; };
hipError_t clock_feed(const ClockParams &p, const ClockState &s, int nchan, const void *x, View xv,
                      long long len, void *sym, long long sym_stride, uint32_t *count, hipStream_t st);

// ---- gangs.hip: many 1-channel banks with their OWN parameters, one lane each, in one launch -------------
// (the live analyzer's inspectors differ in loop bandwidth, baud, decimation ... and cannot share a bank;
// a gang runs their recurrences side by side like a bank does).  Rows are contiguous (unit time stride),
// lengths may differ per item.  items: device array.
struct CostasGangItem { CostasParams p; CostasState s; const void *x; void *y; long long len; };
struct ClockGangItem { ClockParams p; ClockState s; const void *x; long long len; void *sym; uint32_t *count;
                       int steps;      // > 0: the item's group of 64 runs round by round (clock_ring) with this many advance steps; 0: clock_stream_tm
                       int uniform; }; // every item of the group has the same parameters
struct PllGangItem { float alpha, beta; PllState s; const void *x; void *y; long long len; };
// symbol rows: the length is *count when count != nullptr (the clock gang's device-side counts), else fixed_len
struct CmaGangItem { float mu; int locked; void *w; void *dl; const void *x; void *y; const uint32_t *count; long long fixed_len; };
// The recurrence gangs work on time-major slabs (tm[group][m][lane], `slab` elements per group of 64 items, at least
// (ceil(maxlen / 64) + 1) * 64 * 64): rows_tm_gather fills them from the items' rows, rows_tm_scatter writes them back.
// item k of the device table sits at d_items + k * item_bytes with its row pointer / length (long long) at the offsets.
// optional group table: group g = items [first, first + count) (at most 64), e.g. one loop type each; without it
// group g = items [64 g, 64 g + 64)
struct GangGroup { int first, count, kind, order, gain1; };   // gain1: every item's loop gain is exactly 1 (the multiply is skipped)
hipError_t rows_tm_gather(const void *d_items, int item_bytes, int off_ptr, int off_len, int n, const GangGroup *d_groups, int ngroups,
                          int elem_bytes, void *tm, long long slab, long long maxlen, hipStream_t st);
hipError_t rows_tm_scatter(const void *d_items, int item_bytes, int off_ptr, int off_len, int n, const GangGroup *d_groups, int ngroups,
                           int elem_bytes, const void *tm, long long slab, long long maxlen, hipStream_t st);
hipError_t pll_gang(const PllGangItem *d_items, int n, void *tm, long long slab, hipStream_t st);
hipError_t cma_gang(const CmaGangItem *d_items, int n, int ntaps, hipStream_t st);
hipError_t costas_gang(const CostasGangItem *d_items, const GangGroup *d_groups, int ngroups, void *tm, long long slab, hipStream_t st);
hipError_t clock_gang(const ClockGangItem *d_items, int n, void *tm, long long slab, hipStream_t st);
// The same gangs on rows that ARE columns of a time-major slab already (the live analyzer's inspectors of the FFT filter
// bank: sample m of an item sits `pitch` elements behind sample m - 1, items side by side in a row of the slab): no gather,
// no scatter -- lane j streams its column in place, byte offset (item pointer - base) from a wave-uniform base.  `in` /
// `out`: an address at or below every item's x / y with all of them less than 4 GiB above it.  The slabs must be readable
// one tile (64 rows) beyond the longest item.
struct GangSlab { const void *in; void *out; long long pitch_in, pitch_out; };
hipError_t costas_gang_slab(const CostasGangItem *d_items, const GangGroup *d_groups, int ngroups, GangSlab io, hipStream_t st);
hipError_t pll_gang_slab(const PllGangItem *d_items, int n, GangSlab io, hipStream_t st);
hipError_t clock_gang_slab(const ClockGangItem *d_items, int n, GangSlab io, hipStream_t st);

struct AgcParams {
  float knee, gain_slope;
  float fast_alpha_rise, fast_alpha_fall, slow_alpha_rise, slow_alpha_fall;
  unsigned hang_max, delay_line_size, mag_history_size;
};
struct AgcState {               // device
  float *delay_line;            // [delay_line_size][2][nchan]: the last inputs, oldest first
  float *mag_history;           // [mag_history_size-1][nchan]: the last magnitudes (dB), oldest first
  unsigned *hang_n;             // [nchan]
  float *fast_level, *slow_level;
};
// |x|^2 -> dB and its sliding maximum (parallel), level tracking (one lane per channel), gain
// (parallel), state carry.  scratch: >= 2 * len * nchan floats.
hipError_t agc_feed(const AgcParams &p, const AgcState &s, int nchan, const void *x, View xv,
                    void *y, View yv, long long len, float *scratch, hipStream_t st);
// the same in three steps, so that the level trackers of many 1-channel banks can run as one gang
hipError_t agc_feed_pre(const AgcParams &p, const AgcState &s, int nchan, const void *x, View xv, long long len,
                        float *scratch, hipStream_t st);
hipError_t agc_feed_post(const AgcParams &p, const AgcState &s, int nchan, const void *x, View xv, void *y, View yv,
                         long long len, float *scratch, hipStream_t st);
// the level trackers between the two (the one recurrence of the AGC: one lane per channel)
hipError_t agc_feed_level(const AgcParams &p, const AgcState &s, int nchan, long long len, float *scratch, hipStream_t st);
struct AgcGangItem { AgcParams p; AgcState s; float *peak; long long len; };
// gain on the delayed input for outputs [m0, m1) of many 1-channel banks (rows contiguous): one launch, grid.y = item
struct AgcApplyItem { AgcParams p; const float *delay_line; const void *x; void *y; const float *lvl; long long m0, m1; };
hipError_t agc_apply_items(const AgcApplyItem *d_items, int n, long long max_span, hipStream_t st);
hipError_t agc_state_update(const AgcParams &p, const AgcState &s, int nchan, const void *x, View xv, long long len,
                            const float *db, hipStream_t st);
hipError_t agc_level_gang(const AgcGangItem *d_items, int n, void *tm, long long slab, hipStream_t st);
// steps (1)+(2) and (5) of agc_feed for many 1-channel banks with contiguous rows: one launch each
struct AgcPreItem { const void *x; const float *hist; float *db, *peak; long long len; int H; };
struct AgcStateItem { float *delay_line, *hist; const void *x; const float *db; long long len; int delay, H;
                      long long xs, dbs; };   // element strides of x and db (1: contiguous rows; a slab's pitch: columns)
hipError_t agc_pre_items(const AgcPreItem *d_items, int n, long long max_len, hipStream_t st);
hipError_t agc_state_items(const AgcStateItem *d_items, int n, hipStream_t st);
// the same steps for items that are columns of a time-major slab (the live analyzer's inspectors of the FFT filter bank):
// samples in column `lane` of x (pitch px), magnitudes / peaks / levels in column `lane` of two work slabs of pitch pw,
// outputs [m0, m1) to column lane_y of y.  One launch each for all items; tiles of 64 items x 64 / 128 time steps.
struct AgcSlabItem { AgcParams p; AgcState s; int lane, lane_y; long long len, m0, m1; };
// (halo: the longest magnitude history of the items, mag_history_size - 1)
hipError_t agc_pre_slab(const AgcSlabItem *d_items, int n, const void *x, long long px, float *db, float *peak, long long pw,
                        long long max_len, int halo, hipStream_t st);
hipError_t agc_apply_slab(const AgcSlabItem *d_items, int n, const void *x, long long px, void *y, long long py, const float *lvl,
                          long long pw, long long mlo, long long mhi, hipStream_t st);
hipError_t agc_level_gang_slab(const AgcGangItem *d_items, int n, GangSlab io, hipStream_t st);

// rows -> landing zones (host-mapped or device), grid.x = item: n = count ? *count : fixed samples of src go to dst,
// n to *count_out; count (a device counter the producer accumulates into) is cleared for the next block
struct DeliverItem { const void *src; void *dst; uint32_t *count; unsigned fixed; uint32_t *count_out;
                     unsigned stride; };   // element stride of src (1: a contiguous row; a slab's pitch: a column)
hipError_t rows_deliver(const DeliverItem *d_items, int n, hipStream_t st);

// ---- specview.hip ----
struct SpecViewLinear {          // geometry of one frame, computed on the host in double precision
  double viewFreqMin, dstBinW, freqMin, srcBinW, delta;
  int j0, k, psdSize;
};
struct SpecViewHist { int psdSize; float inv, t; unsigned j, spectrumSize; int split; };
hipError_t specview_feed_linear(const SpecViewLinear &g, const float *psd, const float *count, float *accum,
                                float *cnt, hipStream_t st);
hipError_t specview_feed_hist(const SpecViewHist &g, const float *psd, float *accum, float *cnt, hipStream_t st);
// reset_scratch: 1024 x 64-bit device words (the count-cap resets of one call, one mask per 64 bins)
hipError_t specview_interpolate(float *psd, float *accum, float *cnt, int n, unsigned long long *reset_scratch, hipStream_t st);
// nframes linear-mode feeds (each followed by the count-cap reset of interpolate()) in one launch
hipError_t specview_sweep_linear(const SpecViewLinear *d_geom, int nframes, const float *frames, long long frame_stride,
                                 const float *cnt_before, float *accum, float *cnt, int n, hipStream_t st);

// ---- stages.hip ----
hipError_t rows_scale(const void *x, View xv, void *y, View yv, int nchan, long long len, float g, hipStream_t st);
hipError_t rows_xlate(const void *x, View xv, void *y, View yv, int nchan, long long len, const uint32_t *dphase,
                      const uint32_t *phase0, uint64_t n0, hipStream_t st);
hipError_t decision_space(const void *x, long long len, int mode, float *out, hipStream_t st);
hipError_t decide(const void *x, long long len, int mode, int intervals, float vmin, float d, unsigned char *sym, hipStream_t st);
hipError_t symbol_histogram(const void *x, long long len, int mode, float vmin, float d, int nbins, unsigned *hist, hipStream_t st);
// state: {sigma, delta, sqerr}; length <= 4096
hipError_t snr_feed(const unsigned *history, int length, int intervals, float alpha, float *state, float *model, hipStream_t st);
hipError_t spectsrc_preproc(int kind, const void *x, long long len, float prev_re, float prev_im, const void *prev_dev, void *y, hipStream_t st);
// hist / hist_next: [ntaps-1][nchan] complex (time-major), ping-pong
hipError_t rows_fir(const void *x, View xv, void *y, View yv, int nchan, long long len, const float *h, int ntaps,
                    const void *hist, void *hist_next, hipStream_t st);
// w, dl: [n][nchan] complex; rows x / y channel-major with the given strides; count == nullptr -> fixed_len each
hipError_t cma_feed(int n, float mu, int locked, void *w, void *dl, int nchan, const void *x, long long x_stride,
                    const uint32_t *count, long long fixed_len, void *y, long long y_stride, hipStream_t st);

// ---- ingest.hip ----
// format: 1 float32, 2 unsigned 8, 3 signed 8, 4 signed 16 (interleaved I/Q) -> SUCOMPLEX
hipError_t ingest_iq(int format, const void *raw, long long nsamp, void *out, hipStream_t st);
// "power" inspector class: mean of |x|^2 over consecutive windows of N samples; window j of a feed covers samples
// [j N - cnt, (j + 1) N - cnt) (cnt = samples carried in acc_in), out[j] = (mean, 0); the unfinished tail goes to acc_out
hipError_t power_integrate(const void *x, long long len, long long N, long long cnt, const double *acc_in, double *acc_out,
                           void *out, hipStream_t st);
// baud estimators (SPEC.md section M): y[n] = (|x[n] - x[n-1]|^2, 0), y[0] = 0  -- its spectrum has a line at the baud;
// first valley of a fast autocorrelation (fac: n_half floats) below a quarter of fac[0] -> out[0] = lag (0: none), out[1] = 1 / lag
hipError_t baud_nl_transform(const void *x, long long n, void *y, hipStream_t st);
hipError_t fac_first_valley(const float *fac, int n_half, float *out, hipStream_t st);
// the baud line in the transform X (n points) of the transformed block: the LOWEST local maximum of |X|^2 in [skip, n/2)
// that reaches half of the strongest one (a comb of harmonics has no strongest tooth worth trusting), then the power
// centroid over +-4 bins.  res[0] = centroid bin (0: none), value[0] = centroid / n
hipError_t baud_line(const void *X, int n, int skip, double *res, float *value, hipStream_t st);
hipError_t carrier_norm(const float *res, float *value, hipStream_t st);
// source conditioning in front of the path: I/Q swap, then removal of a tracked DC level (dc: device float[2],
// or nullptr).  The level follows the block means: dc = first ? mean : dc + alpha (mean - dc), and the block is
// corrected with the updated level.  partial: device scratch of 2 * 256 floats.
hipError_t source_fix(void *x, long long nsamp, int iq_reverse, float *dc, float alpha, int first, float *partial, hipStream_t st);

// ---- fft.hip ----
// batch: that many transforms side by side (n elements apart in both buffers), one launch per pass for all of them
hipError_t fft_forward(void *a, void *b, int log2n, void **result, hipStream_t st, int batch = 1);
hipError_t psd_frames_large(int log2n, const void *x, long long hop, int navg, const float *window, float scale,
                            int mode, float *out, long long nout, void *a, void *b, float *acc, int batch, hipStream_t st);
// ---- psd_large.hip ---- frames beyond the LDS in two trips through HBM (four-step transform)
int        psd_large_log2n2(int log2n);
int        psd_large_chunk(int navg);
hipError_t psd_frames_large2(int log2n, const void *x, long long hop, int navg, const float *window, const void *tw_n,
                             const void *tw_row, float scale, int mode, float *out, long long nout, void *a, float *P, int pring,
                             int batch, hipStream_t st);
hipError_t fac_feed(void *a, void *b, int log2n, float alpha, long long view_start, long long view_end, float *absbuf,
                    float *fac, unsigned *mx, unsigned *mn, hipStream_t st);
hipError_t window_pad(const void *data, long long len, long long alloc, void *buf, hipStream_t st);
hipError_t spectrum_centroid(void *buf, long long alloc, long long lo, long long hi, float *mirror, long long bins,
                             long long delta, int with_dispersion, float *blk_max, long long *blk_idx,
                             int nblk, float *res, hipStream_t st);

}  // namespace sdk
