/*
 * include/sigdigger_amd.h -- C ABI of libsigdigger_amd.so (MI355X / gfx950).
 *
 * Drop-in boundary for the SigDigger DSP hot path: every entry point
 * below replaces one block-at-a-time loop (or one libsigutils/libsuscan call) that the
 * reference makes, and cites it.  Signatures are plain C: pointers, sizes, scalars.
 *
 *   d_*    : DEVICE pointers (HBM).  Complex data is SUCOMPLEX = interleaved float32 I/Q,
 *            exactly the layout the reference passes around (include/Suscan/Messages/SamplesMessage.h:33-58).
 *   stream : a hipStream_t passed as void* (NULL = the null stream).  All work is enqueued
 *            asynchronously on it; nothing here synchronises unless stated.
 *   return : SUBOOL (SU_TRUE / SU_FALSE) or a pointer (NULL on failure), the reference's own
 *            convention (include/Suscan/Compat.h:28-36 wraps every call in SU_ATTEMPT).
 *            suamd_last_error() returns the reason (thread-local).
 *
 * There is NO CPU fallback: without a usable gfx950 device every constructor fails.
 */
#ifndef SIGDIGGER_AMD_H
#define SIGDIGGER_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SUAMD_API __attribute__((visibility("default")))

typedef float    SUFLOAT;
typedef double   SUFREQ;
typedef uint64_t SUSCOUNT;
typedef int64_t  SUSDIFF;
typedef int      SUBOOL;
#ifdef SUAMD_COMPLEX_IS_SUCOMPLEX                    /* <sigutils/types.h> came first: one sample type   */
typedef SUCOMPLEX suamd_complex;
#else
typedef struct { SUFLOAT re, im; } suamd_complex;   /* layout-identical to SUCOMPLEX */
#endif
#ifndef SU_TRUE
#  define SU_TRUE  1
#  define SU_FALSE 0
#endif

/* A batch of `nchan` sample rows: element (channel c, time m) lives at
 * base[c*chan_stride + m*time_stride], strides in complex samples.
 *   channel-major [c][m]: { row_pitch, 1 }  -- the layout of per-inspector sample batches
 *                                               (struct suscan_analyzer_sample_batch_msg)
 *   time-major    [m][c]: { 1, nchan_pitch } -- the layout the one-lane-per-channel loops stream
 *                                               at full speed (a wavefront reads 512 contiguous bytes) */
typedef struct { SUSCOUNT chan_stride, time_stride; } suamd_view;

typedef struct suamd_ctx         suamd_ctx_t;
typedef struct suamd_psd         suamd_psd_t;
typedef struct suamd_chanbank    suamd_chanbank_t;
typedef struct suamd_costas_bank suamd_costas_bank_t;
typedef struct suamd_pll_bank    suamd_pll_bank_t;
typedef struct suamd_clock_bank  suamd_clock_bank_t;
typedef struct suamd_agc_bank    suamd_agc_bank_t;

/* ------------------------------------------------------------------------------------ */
/* library / context                                                                    */
/* ------------------------------------------------------------------------------------ */
SUAMD_API const char *suamd_last_error(void);
SUAMD_API const char *suamd_version(void);
/* Measurement aid (bench.py's roofline leg; no counterpart in the reference).  While enabled, every launch of the
 * channeliser and PSD kernels ("stw_kernel", "st_kernel", "chan_fir_kernel", "psd_kernel", "psd_reduce_kernel") carries
 * an event pair bound to the dispatch itself; suamd_kernel_timing_read waits for the launches of `kernel` (NULL: all of
 * them) recorded since the last read and returns the sum / min / max of their own durations in milliseconds -- the figure
 * rocprofv3 --kernel-trace reports, without the queue gaps two stream events around a launch would add.  Process-wide. */
SUAMD_API void   suamd_kernel_timing(SUBOOL enable);
SUAMD_API SUBOOL suamd_kernel_timing_read(const char *kernel, double *sum_ms, double *min_ms, double *max_ms, unsigned *launches);
/* Tuning (no counterpart in the reference): every knob that changes HOW the library computes -- launch plans, kernel
 * choices, schedules; never the result -- lives in one struct (csrc/tuning.hpp lists the fields with their environment
 * names, ranges and meaning).  The SUAMD_* environment is read ONCE, at first use; these calls change a field afterwards
 * (what the A / B parity tests do).  `name` is the field's name or its environment variable's.  suamd_tuning_set returns
 * SU_FALSE for an unknown name or a value outside the field's range; suamd_tuning_describe walks the table (SU_FALSE past
 * its end); suamd_tuning_reset goes back to defaults + environment.  Process-wide; objects made under a plan keep it where
 * the field says so. */
SUAMD_API SUBOOL suamd_tuning_set(const char *name, long long value);
SUAMD_API SUBOOL suamd_tuning_get(const char *name, long long *value);
SUAMD_API SUBOOL suamd_tuning_describe(unsigned index, const char **name, const char **env, long long *def, long long *lo, long long *hi, const char **doc);
SUAMD_API void   suamd_tuning_reset(void);
/* Replaces suscan_sigutils_init + su_lib_gen_wisdom for this path (Suscan/Library.cpp:97,
 * App/Loader.cpp:46): binds a GPU and builds the shared tables. */
SUAMD_API suamd_ctx_t *suamd_ctx_new(int device_ordinal);
SUAMD_API void         suamd_ctx_destroy(suamd_ctx_t *ctx);
SUAMD_API int          suamd_ctx_device(const suamd_ctx_t *ctx);
/* Streams confined to a set of compute units (no counterpart in the reference; the live path and bench.py use it to keep
 * the one-wavefront recurrences -- su_costas / su_clock_detector / su_agc level trackers, milliseconds per launch -- off
 * the CUs the transform kernels plan their single round of workgroups for).  `cu_mask` has `nwords` 32-bit words, bit i
 * of the whole = compute unit i in the driver's numbering, which deals consecutive bits to consecutive XCDs (bit i is
 * on XCD i mod 8 of an MI355X: suamd_probe_placement shows it); at least one bit must be set.  The stream gets a hardware
 * queue of its own and is destroyed with suamd_stream_destroy.  hipExtStreamCreateWithCUMask takes no flags: the stream has
 * the DEFAULT flags, i.e. it synchronises with the legacy null stream like any hipStreamCreate stream -- a caller that
 * wants its masked streams to run beside other work must not issue that work on the null stream (PyTorch's default stream
 * on ROCm IS the null stream: sigdigger_amd/pipeline.py keeps every kernel of a partitioned pipeline on streams of its
 * own).  NULL on failure. */
SUAMD_API unsigned suamd_ctx_cu_count(const suamd_ctx_t *ctx);
SUAMD_API void    *suamd_stream_new_cu_mask(suamd_ctx_t *ctx, const uint32_t *cu_mask, unsigned nwords);
SUAMD_API SUBOOL   suamd_stream_destroy(suamd_ctx_t *ctx, void *stream);
/* Where `nblocks` one-wavefront workgroups launched on `stream` ran: h_where[b] = xcc_id << 16 | se_id << 8 | cu_id of
 * workgroup b (host array; the call synchronises the stream).  Every workgroup spins for `spin_ticks` of the shader clock
 * so that a launch larger than the stream's CUs really spreads over all of them. */
SUAMD_API SUBOOL   suamd_probe_placement(suamd_ctx_t *ctx, void *stream, unsigned nblocks, unsigned spin_ticks, uint32_t *h_where);

/* ------------------------------------------------------------------------------------ */
/* A2-A4, A9: main-spectrum PSD                                                          */
/* ------------------------------------------------------------------------------------ */
/* enum sigutils_channel_detector_window (include/Suscan/AnalyzerParams.h:37-43) */
enum suamd_window {
  SUAMD_WINDOW_NONE = 0, SUAMD_WINDOW_HAMMING, SUAMD_WINDOW_HANN,
  SUAMD_WINDOW_FLAT_TOP, SUAMD_WINDOW_BLACKMANN_HARRIS
};
enum suamd_psd_mode {
  SUAMD_PSD_LINEAR    = 0,  /* what struct suscan_analyzer_psd_msg::psd_data carries: linear
                               power, natural FFT order (Suscan/Messages/PSDMessage.cpp:29-38
                               proves it by shifting + taking dB itself)                   */
  SUAMD_PSD_DB_SHIFTED = 1  /* PSDMessage ctor fused in: fftshift + SU_POWER_DB             */
};
/* Plan for detector_params.window_size / .window (Suscan/AnalyzerParams.cpp:53-71).
 * window_size: power of two, 512..1048576 (FFTWidget's 2^9..2^20, Default/FFT/FFTWidget.cpp:350-351);
 * up to 16384 the frame stays in LDS, larger frames go pass by pass through HBM. */
SUAMD_API suamd_psd_t *suamd_psd_new(suamd_ctx_t *ctx, unsigned window_size, int window_type);
SUAMD_API void         suamd_psd_destroy(suamd_psd_t *psd);
/* Launch plan of the in-LDS sizes (512 .. 16384 points): when a feed has fewer outputs than workgroups the chip holds,
 * the navg frames of an output are split over several workgroups (partial sums, one more short launch) -- aiming at
 * `workgroups` per launch.  0 restores the default: 1024 up to 4096 points, 256 above (one workgroup per CU: what a launch
 * that shares the chip with other streams' long-running kernels can count on).  512 is for 8192-point frames on a chip the
 * launch has to itself (two workgroups per CU fit).  The result does not depend on it beyond the summation order. */
SUAMD_API SUBOOL       suamd_psd_set_split_target(suamd_psd_t *psd, unsigned workgroups);
/* For o < nframes/navg:
 *   d_out[o*N + i] = scale/navg * sum_{f<navg} |FFT_N(window .* d_x[(o*navg+f)*hop ...])[i]|^2
 * (mode LINEAR), or its fftshift + 10*log10(. + 1e-8) (mode DB_SHIFTED). */
SUAMD_API SUBOOL suamd_psd_feed(suamd_psd_t *psd, const suamd_complex *d_x, SUSCOUNT nframes,
                                SUSCOUNT hop, unsigned navg, SUFLOAT scale, int mode,
                                SUFLOAT *d_out, void *stream);
/* PSDMessage::PSDMessage loop (Suscan/Messages/PSDMessage.cpp:29-38), nframes frames of n floats in place */
SUAMD_API SUBOOL suamd_psd_shift_db(suamd_ctx_t *ctx, SUFLOAT *d_psd, SUSCOUNT n, SUSCOUNT nframes, void *stream);
/* Averager::feed blend branch (Misc/Averager.cpp:44-47): last[i] += alpha*(x[i]-last[i]);
 * blend == SU_FALSE is the memcpy branch (:48). */
SUAMD_API SUBOOL suamd_averager_feed(suamd_ctx_t *ctx, SUFLOAT *d_last, const SUFLOAT *d_x, SUSCOUNT n,
                                     SUFLOAT alpha, SUBOOL blend, void *stream);
/* GenericInspector::inspectorMessage SPECTRUM case (Default/GenericInspector/GenericInspector.cpp:231-247) */
SUAMD_API SUBOOL suamd_inspector_spectrum_db_shift(suamd_ctx_t *ctx, SUFLOAT *d_data, SUSCOUNT len,
                                                   SUSCOUNT nspectra, void *stream);

/* ------------------------------------------------------------------------------------ */
/* A6: the "audio" inspector class                                                       */
/* ------------------------------------------------------------------------------------ */
/* What libsuscan's audio inspector does behind AudioProcessor (Default/Audio/AudioProcessor.cpp:94-169, 251-270):
 * demodulate the channel samples (audio.demodulator: 1 AM, 2 FM, 3 USB, 4 LSB, 5 RAW), low-pass at audio.cutoff and
 * resample from the channel rate to audio.sample-rate, times audio.volume, muted while the channel power is below
 * audio.squelch-level when audio.squelch is set (SPEC.md section Q).  The consumer plays the real part
 * (Audio/AudioPlayback.cpp:603-604). */
typedef struct suamd_audio suamd_audio_t;
SUAMD_API suamd_audio_t *suamd_audio_new(suamd_ctx_t *ctx, SUFLOAT equiv_fs, SUFLOAT bandwidth);
SUAMD_API void     suamd_audio_destroy(suamd_audio_t *au);
SUAMD_API SUBOOL   suamd_audio_configure(suamd_audio_t *au, int demodulator, SUFLOAT sample_rate, SUFLOAT cutoff, SUFLOAT volume,
                                         SUBOOL squelch, SUFLOAT squelch_level);
SUAMD_API SUSCOUNT suamd_audio_output_count(const suamd_audio_t *au, SUSCOUNT len);    /* of the next feed of len samples */
SUAMD_API SUBOOL   suamd_audio_feed(suamd_audio_t *au, const suamd_complex *d_x, SUSCOUNT len, suamd_complex *d_out,
                                    SUSCOUNT *n_out, void *stream);

/* ------------------------------------------------------------------------------------ */
/* N1: channel detector (su_channel_detector)                                            */
/* ------------------------------------------------------------------------------------ */
/* What libsuscan runs on the analyzer's spectrum to announce channels (SUSCAN_ANALYZER_MESSAGE_TYPE_CHANNEL,
 * Suscan/Analyzer.cpp:75-98) with struct sigutils_channel_detector_params' alpha (spectrum smoothing), gamma (noise
 * floor smoothing), snr (threshold as a linear power ratio over the floor) -- Suscan/AnalyzerParams.cpp:53-71; beta (signal
 * level smoothing) smooths the reported peak level.  SPEC.md section O: smoothed spectrum, floor = smoothed median, runs
 * of bins above snr * floor (gaps of up to two bins bridged, single-bin runs dropped). */
typedef struct suamd_chandet suamd_chandet_t;
/* Hz relative to the spectrum's centre; snr, S0, N0 in dB.  beta (the detector's signal-level smoothing): a channel
 * that contains the centre of a channel of the previous list continues it -- S0 <- S0_prev + beta (S0_now - S0_prev) in
 * dB, snr = S0 - N0 with it, age <- age_prev + 1 -- any other starts at S0_now with age 0 (SPEC.md section O).
 * beta <= 0 or >= 1: no smoothing. */
struct suamd_channel { SUFREQ fc, f_lo, f_hi; SUFLOAT bw, snr, S0, N0; unsigned age; };
SUAMD_API suamd_chandet_t *suamd_chandet_new(suamd_ctx_t *ctx, unsigned n /* power of two, 512 .. 16384 */,
                                             SUFLOAT alpha, SUFLOAT beta, SUFLOAT gamma, SUFLOAT snr);
SUAMD_API void   suamd_chandet_destroy(suamd_chandet_t *det);
/* one frame of n floats: linear power, natural FFT order (what suamd_psd_feed writes in mode LINEAR) */
SUAMD_API SUBOOL suamd_chandet_feed(suamd_chandet_t *det, const SUFLOAT *d_psd, void *stream);
/* the channels of the smoothed spectrum, ordered by frequency: at most cap; returns the count (-1 on error).
 * Synchronises the stream. */
SUAMD_API int    suamd_chandet_channels(suamd_chandet_t *det, SUFLOAT samp_rate, struct suamd_channel *out, unsigned cap, void *stream);
/* the same in two steps for callers that must not wait: _find enqueues detection + the copy into the detector's pinned
 * landing zone `slot` (0 / 1); _collect reads it after the caller has synchronised with the stream */
SUAMD_API SUBOOL suamd_chandet_find(suamd_chandet_t *det, int slot, void *stream);
SUAMD_API int    suamd_chandet_collect(suamd_chandet_t *det, int slot, SUFLOAT samp_rate, struct suamd_channel *out, unsigned cap);
SUAMD_API SUFLOAT suamd_chandet_noise_floor(suamd_chandet_t *det, void *stream);    /* linear; synchronises */

/* ------------------------------------------------------------------------------------ */
/* T2 / N2: FFT channeliser (su_specttuner)                                              */
/* ------------------------------------------------------------------------------------ */
/* The device side of su_specttuner_new / _open_channel / _feed_bulk (Tasks/LPFTask.cpp:52-69,83-87; the channeliser
 * behind every suscan inspector): windows of `window_size` samples advancing by half a window, ONE forward FFT per
 * window shared by all channels, per channel a bin pick around its (even) centre bin, its frequency response, an
 * inverse FFT at the decimated rate window_size / size and a sin^2 cross-fade with the previous window (SPEC.md C2).
 * include/sigutils/specttuner.h is the sigutils-named host front end of the same object. */
typedef struct suamd_specttuner suamd_specttuner_t;
SUAMD_API suamd_specttuner_t *suamd_specttuner_new(suamd_ctx_t *ctx, unsigned window_size /* 4096 */);
SUAMD_API void   suamd_specttuner_destroy(suamd_specttuner_t *st);
/* f0, bw: angular frequency (rad / sample); guard >= 1 sizes the channel for bw * guard (guard = 2 pi / bw: no
 * decimation, LPFTask.cpp:65); precise: the rounding of f0 to an even bin is corrected by an NCO at the output rate.
 * Returns the channel index (>= 0) or -1.  Channels opened / closed between feeds keep the others' state. */
SUAMD_API int    suamd_specttuner_open_channel(suamd_specttuner_t *st, double f0, double bw, double guard, SUBOOL precise);
SUAMD_API SUBOOL suamd_specttuner_close_channel(suamd_specttuner_t *st, int channel);
SUAMD_API unsigned suamd_specttuner_channel_size(const suamd_specttuner_t *st, int channel);        /* bins = inverse FFT size */
SUAMD_API unsigned suamd_specttuner_channel_decimation(const suamd_specttuner_t *st, int channel);  /* window_size / size */
/* Feeds `len` samples (device), len a multiple of window_size / 2.  The first feed needs a whole window before it
 * yields anything (len / (W/2) - 1 blocks); later feeds yield len / (W/2) blocks of size / 2 samples per channel.
 * Channel c's samples of this feed go to d_y[c*view.chan_stride + m*view.time_stride], m = 0 ..; counts[c] (host,
 * may be NULL) receives how many.  Any split of a stream into feeds gives the same samples. */
SUAMD_API SUBOOL suamd_specttuner_feed(suamd_specttuner_t *st, const suamd_complex *d_x, SUSCOUNT len,
                                       suamd_complex *d_y, suamd_view view, SUSCOUNT *counts, void *stream);
/* the same with one row per channel anywhere in device memory: d_rows (a DEVICE array indexed by channel) holds where
 * each channel's samples of this feed start (contiguous in time) */
SUAMD_API SUBOOL suamd_specttuner_feed_rows(suamd_specttuner_t *st, const suamd_complex *d_x, SUSCOUNT len,
                                            suamd_complex *const *d_rows, SUSCOUNT *counts, void *stream);
/* the same when the caller knows that every row it names starts inside [d_base, d_base + span_bytes) with span_bytes well
 * below 2 GiB (rows carved from one arena, or allocated together): the narrow-channel kernels then address the outputs
 * with 32-bit offsets from d_base -- buffer stores with a wave-uniform descriptor instead of a 64-bit address per lane
 * (31.7 -> 29.3 us per 4 Mi x 64 block, and the 8- / 16-bin banks run on the two-wavefront kernel too).  A promise the
 * library cannot check: a row outside the range is written where the offset arithmetic puts it. */
SUAMD_API SUBOOL suamd_specttuner_feed_rows_near(suamd_specttuner_t *st, const suamd_complex *d_x, SUSCOUNT len,
                                                 suamd_complex *const *d_rows, const void *d_base, size_t span_bytes,
                                                 SUSCOUNT *counts, void *stream);
/* Both at once: channels of at most `view_max_size` bins (the narrow sizes, whose kernels store one channel per lane) go
 * through the view -- a time-major slab {1, pitch} keeps those stores contiguous --, wider ones to their rows as in
 * _feed_rows_near (d_base NULL: as in _feed_rows).  The live analyzer's feed: narrow inspectors are columns of one slab,
 * wide ones own their rows. */
SUAMD_API SUBOOL suamd_specttuner_feed_mixed(suamd_specttuner_t *st, const suamd_complex *d_x, SUSCOUNT len,
                                             suamd_complex *d_y, suamd_view view, unsigned view_max_size,
                                             suamd_complex *const *d_rows, const void *d_base, size_t span_bytes,
                                             SUSCOUNT *counts, void *stream);
/* windows per workgroup run (default 3): a run re-transforms the window before it */
SUAMD_API SUBOOL suamd_specttuner_set_run(suamd_specttuner_t *st, unsigned run);
/* How many of the chip's 1024 window slots (4 windows per CU: the LDS) a launch of the narrow-channel kernels may plan
 * for; the run length of a feed is ceil(windows / slots).  Default 768 (0 restores it): a launch must fit ONE round, and
 * kernels of other streams (the recurrences of earlier blocks) hold a few slots for milliseconds -- a workgroup that
 * finds none waits for a whole run of another.  1024 is for a tuner that has the device to itself (an offline
 * LPFTask-style pass): 78 instead of 88 us per 16 Mi x 64 block. */
SUAMD_API SUBOOL suamd_specttuner_set_slots(suamd_specttuner_t *st, unsigned slots);
/* Entries a `counts` array passed to suamd_specttuner_feed / _feed_rows must hold: one per slot of the tuner's channel
 * table (slots of closed channels are reused but the table never shrinks), i.e. the highest index ever returned by
 * _open_channel plus one.  A feed zeroes every entry and fills those of the open channels. */
SUAMD_API unsigned suamd_specttuner_channel_capacity(const suamd_specttuner_t *st);
/* Forgets the stream position: the half-window history and every channel's cross-fade partner (the next feed starts
 * like the first one: a whole window before anything comes out, y_{-1} = 0).  For a seek in the source, or when
 * feeding resumes after a gap.  Channels stay open; their output counters (the `precise` NCO's phase) run on. */
SUAMD_API SUBOOL suamd_specttuner_reset(suamd_specttuner_t *st, void *stream);
/* The design of a channel without a device (host arithmetic only, SPEC.md C2): geometry and the binary32 response
 * k h[i] the kernels multiply the picked bins with.  geom[0..5] = size, halfsz, halfw, decimation, center, dphase;
 * hk (may be NULL) receives 2 * size floats (re, im interleaved).  Returns SU_FALSE on bad parameters. */
SUAMD_API SUBOOL suamd_specttuner_design(unsigned window_size, double f0, double bw, double guard, uint32_t geom[6], SUFLOAT *hk);

/* ------------------------------------------------------------------------------------ */
/* T1 / K4: NCO carrier translate                                                        */
/* ------------------------------------------------------------------------------------ */
/* su_ncqo_init(-relFreq) + su_ncqo_set_phase(-phase) + per-sample su_ncqo_read loop
 * (Tasks/CarrierXlator.cpp:36-37,57-60).  Phase is a 32-bit accumulator (2^32 per turn):
 *   y[i] = x[i] * exp(j 2 pi (phase0 + (n0+i)*dphase) / 2^32)
 * suamd_fnor_to_dphase(fnor) converts a normalised frequency (2f/fs) to dphase. */
SUAMD_API uint32_t suamd_fnor_to_dphase(double fnor);
SUAMD_API SUBOOL suamd_xlate_bulk(suamd_ctx_t *ctx, const suamd_complex *d_x, suamd_complex *d_y,
                                  SUSCOUNT len, uint32_t phase0, uint32_t dphase, SUSCOUNT n0, void *stream);

/* ------------------------------------------------------------------------------------ */
/* K4+K5: bank of inspector channels: translate + low-pass + decimate                    */
/* ------------------------------------------------------------------------------------ */
/* One bank = the set of channels opened with suscan_analyzer_open_ex_async
 * (Suscan/Analyzer.cpp:459-484) that share a decimation; replaces the su_specttuner
 * channel path (Tasks/LPFTask.cpp:52-69,83-87) with the 255-tap polyphase FIR the north
 * star prescribes.  fnor[c] = channel centre (2 fc / fs); taps = real low-pass prototype. */
SUAMD_API void   suamd_lpf_design(SUFLOAT *taps, unsigned ntaps, double fc_nor);
SUAMD_API suamd_chanbank_t *suamd_chanbank_new(suamd_ctx_t *ctx, unsigned nchan, const double *fnor,
                                               unsigned decimation, const SUFLOAT *taps, unsigned ntaps);
SUAMD_API void   suamd_chanbank_destroy(suamd_chanbank_t *bank);
/* Number of outputs per channel the next feed of len samples will produce. */
SUAMD_API SUSCOUNT suamd_chanbank_output_count(const suamd_chanbank_t *bank, SUSCOUNT len);
/* Feeds len input samples (shared by all channels); writes output m < *n_out of channel c at
 * d_y[c*yv.chan_stride + m*yv.time_stride].  State (history, sample clock) carries to the next
 * call, so a stream may be fed block by block. */
SUAMD_API SUBOOL suamd_chanbank_feed(suamd_chanbank_t *bank, const suamd_complex *d_x, SUSCOUNT len,
                                     suamd_complex *d_y, suamd_view yv, SUSCOUNT *n_out, void *stream);
SUAMD_API SUBOOL suamd_chanbank_reset(suamd_chanbank_t *bank, void *stream);
/* Launch plan only (same samples): the caller promises that nothing else runs on the device while this bank's feeds do
 * (an offline pass; the stream pipeline's transform window).  Long feeds of the one- / two-channel stream kernel then run
 * as one persistent workgroup per CU, which needs every CU's whole LDS and register file -- beside long-running kernels of
 * other streams that shape takes up to 1.6 x longer, which is why it is not the default. */
SUAMD_API SUBOOL suamd_chanbank_set_exclusive(suamd_chanbank_t *bank, SUBOOL exclusive);
/* Many 1-channel banks (each its own centre frequency, decimation, taps and stream position) fed the SAME wideband
 * block in one launch -- the analyzer's inspectors, which all channelise the block the source just delivered
 * (suscan's inspector scheduler, Suscan/Analyzer.h:137-168).  d_y[i]: contiguous output row of bank i, n_out[i]
 * (may be NULL) its number of output samples.  Identical to n suamd_chanbank_feed calls. */
SUAMD_API SUBOOL suamd_chanbank_gang_feed(suamd_ctx_t *ctx, suamd_chanbank_t *const *banks, unsigned n,
                                          const suamd_complex *d_x, SUSCOUNT len, suamd_complex *const *d_y,
                                          SUSCOUNT *n_out, void *stream);

/* ------------------------------------------------------------------------------------ */
/* T5 / T7 / T11: element-wise demodulators (batched: nchan rows of len samples)         */
/* ------------------------------------------------------------------------------------ */
/* QuadDemodTask::work loop (Tasks/QuadDemodTask.cpp:44-60).  d_prev[c] (may be NULL when
 * first) = sample preceding row c; first => dest[0] = 0.  If d_prev_out != NULL the last
 * sample of each row is stored there for the next block. */
SUAMD_API SUBOOL suamd_quad_demod_batch(suamd_ctx_t *ctx, const suamd_complex *d_x, suamd_view xv,
                                        suamd_complex *d_y, suamd_view yv, unsigned nchan, SUSCOUNT len,
                                        const suamd_complex *d_prev, SUBOOL first,
                                        suamd_complex *d_prev_out, void *stream);
/* DelayedConjTask::work loop (Tasks/DelayedConjTask.cpp:70-84), whole capture at once */
SUAMD_API SUBOOL suamd_delayed_conj_bulk(suamd_ctx_t *ctx, const suamd_complex *d_x, suamd_complex *d_y,
                                         SUSCOUNT len, SUSCOUNT delay, void *stream);
/* HistogramFeeder::work loops (Tasks/HistogramFeeder.cpp:45-66); space 0 amplitude,
 * 1 phase, 2 frequency (len-1 outputs) */
SUAMD_API SUBOOL suamd_histogram_feed_bulk(suamd_ctx_t *ctx, const suamd_complex *d_x, SUSCOUNT len,
                                           int space, SUFLOAT *d_out, void *stream);

/* WaveSampler::sampleManual (Tasks/WaveSampler.cpp:96-175; delta / sampOffset of its ctor :45-46):
 * fractional-boundary boxcar per symbol over the whole capture.  space: 0 AMPLITUDE (rms),
 * 1 PHASE, 2 FREQUENCY (sum of x conj(prev)); symbols [0, nout) are written (complex, as
 * WaveSampleSet::block) */
SUAMD_API SUBOOL suamd_sample_manual_bulk(suamd_ctx_t *ctx, const suamd_complex *d_data, SUSCOUNT length,
                                          double symbol_count, SUSCOUNT symbol_sync, int space,
                                          suamd_complex *d_out, SUSCOUNT nout, void *stream);

/* Sample formats of the file source (Default/SourceConfig/FileSourcePage.cpp:80-104, same order as
 * enum suscan_source_format there; WAV / SigMF are containers around one of the raw payloads).
 * d_out[i] = (I, Q) of interleaved raw sample i: u8 (v-128)/128, s8 v/128, s16 v/32768, f32 as is.
 * d_raw 16-byte aligned.  This is the ingest step in front of the path:
 * the host ships 2-4 B/sample over PCIe, the GPU expands to SUCOMPLEX. */
enum suamd_sample_format {
  SUAMD_FORMAT_RAW_FLOAT32 = 1, SUAMD_FORMAT_RAW_UNSIGNED8 = 2, SUAMD_FORMAT_RAW_SIGNED8 = 3, SUAMD_FORMAT_RAW_SIGNED16 = 4
};
SUAMD_API SUBOOL   suamd_ingest_iq(suamd_ctx_t *ctx, int format, const void *d_raw, SUSCOUNT nsamples,
                                   suamd_complex *d_out, void *stream);
SUAMD_API unsigned suamd_format_bytes_per_sample(int format);   /* per complex sample; 0 = unknown */
/* Source conditioning in front of the path (Suscan::Analyzer::setIQReverse / setDCRemove, Suscan/Analyzer.cpp:
 * 240-256; the arithmetic lives in libsuscan's source worker -- absent -- and is frozen in SPEC.md section L):
 * in place, x <- swap(x) if iq_reverse, then x <- x - dc if d_dc != NULL, where the level d_dc[0..1] (device)
 * follows the block means of the swapped signal: dc = first ? mean : dc + alpha (mean - dc), updated BEFORE
 * the block is corrected. */
SUAMD_API SUBOOL suamd_source_fix(suamd_ctx_t *ctx, suamd_complex *d_x, SUSCOUNT nsamples, SUBOOL iq_reverse,
                                  SUFLOAT *d_dc, SUFLOAT alpha, SUBOOL first, void *stream);

/* WaveSampler::sampleZeroCrossing (Tasks/WaveSampler.cpp:215-292), all work() calls of one capture:
 * run lengths between sign changes of `var` -> round(samples * bnor) symbols of value (var > 0).
 * var: space 0 AMPLITUDE = Re(x conj x) - Re(thr conj thr) if `amplitude`, else Re(x angle) - Re(thr angle);
 * 1 PHASE = arg(x angle); 2 FREQUENCY = arg(j x conj(prev)).  The reference's block structure is kept
 * (4096-sample blocks restarting from prevVar = -1 / prevSample = 0, `last` for the final block,
 * <= 4096 symbols per block).  d_symbols (bytes, WaveSampleSet::symbols) must hold
 * 4096 * ceil(length / 4096) entries; returns the number written, -1 on error.  Synchronises. */
SUAMD_API SUSDIFF suamd_sample_zero_crossing_bulk(suamd_ctx_t *ctx, const suamd_complex *d_data, SUSCOUNT length,
                                                  SUFLOAT bnor, int space, SUBOOL amplitude, SUFLOAT thr_re,
                                                  SUFLOAT thr_im, SUFLOAT ang_re, SUFLOAT ang_im,
                                                  unsigned char *d_symbols, SUSCOUNT capacity, void *stream);
/* WaveSampler::sampleGardner in FREQUENCY space (Tasks/WaveSampler.cpp:188-196): d_y[p] = d_x[p] conj(d_x[p-1]),
 * d_x[-1] = prev (this->prevSample); the result is what su_clock_detector_feed sees -> suamd_clock_bank_feed */
SUAMD_API SUBOOL suamd_conj_prev_bulk(suamd_ctx_t *ctx, const suamd_complex *d_x, suamd_complex *d_y, SUSCOUNT len,
                                      SUFLOAT prev_re, SUFLOAT prev_im, void *stream);

/* ------------------------------------------------------------------------------------ */
/* K6-K9: per-channel recurrences, one lane per channel                                  */
/* ------------------------------------------------------------------------------------ */
/* enum sigutils_costas_kind (Components/TimeWindow.cpp:1948-1960) */
enum suamd_costas_kind { SUAMD_COSTAS_NONE = 0, SUAMD_COSTAS_BPSK, SUAMD_COSTAS_QPSK, SUAMD_COSTAS_8PSK };

/* su_costas_init(&c, kind, fhint, arm_bw, arm_order, loop_bw) for nchan identical loops
 * (Tasks/CostasRecoveryTask.cpp:41) */
SUAMD_API suamd_costas_bank_t *suamd_costas_bank_new(suamd_ctx_t *ctx, unsigned nchan, int kind, SUFLOAT fhint,
                                                     SUFLOAT arm_bw, unsigned arm_order, SUFLOAT loop_bw);
SUAMD_API void   suamd_costas_bank_destroy(suamd_costas_bank_t *b);
/* the `while (amount--) dest[p] = su_costas_feed(&costas, origin[p])` loop
 * (Tasks/CostasRecoveryTask.cpp:58-61) for every row c */
SUAMD_API SUBOOL suamd_costas_bank_feed(suamd_costas_bank_t *b, const suamd_complex *d_x, suamd_view xv,
                                        suamd_complex *d_y, suamd_view yv, SUSCOUNT len, void *stream);
/* copies omega[c] (rad/sample) and phase[c] (2^32/turn) to host; synchronises the stream */
SUAMD_API SUBOOL suamd_costas_bank_get_state(suamd_costas_bank_t *b, SUFLOAT *omega, uint32_t *phase, void *stream);

/* su_pll_init(&pll, fhint, fc) / su_pll_track loop (Tasks/PLLSyncTask.cpp:36,53-56) */
SUAMD_API suamd_pll_bank_t *suamd_pll_bank_new(suamd_ctx_t *ctx, unsigned nchan, SUFLOAT fhint, SUFLOAT fc);
SUAMD_API void   suamd_pll_bank_destroy(suamd_pll_bank_t *b);
SUAMD_API SUBOOL suamd_pll_bank_feed(suamd_pll_bank_t *b, const suamd_complex *d_x, suamd_view xv,
                                     suamd_complex *d_y, suamd_view yv, SUSCOUNT len, void *stream);
SUAMD_API SUBOOL suamd_pll_bank_get_state(suamd_pll_bank_t *b, SUFLOAT *omega, uint32_t *phase, void *stream);

/* su_clock_detector_init(&cd, loop_gain, bhint, bufsiz) + feed/read loops
 * (Tasks/WaveSampler.cpp:60-65,177-213).  Symbols of row c are appended at
 * d_sym[c*sym_stride + d_count[c]++]; d_count (uint32 per channel) is zeroed by the caller
 * (or carried over to keep appending). */
SUAMD_API suamd_clock_bank_t *suamd_clock_bank_new(suamd_ctx_t *ctx, unsigned nchan, SUFLOAT loop_gain, SUFLOAT bhint);
SUAMD_API void   suamd_clock_bank_destroy(suamd_clock_bank_t *b);
SUAMD_API SUBOOL suamd_clock_bank_feed(suamd_clock_bank_t *b, const suamd_complex *d_x, suamd_view xv,
                                       SUSCOUNT len, suamd_complex *d_sym, SUSCOUNT sym_stride,
                                       uint32_t *d_count, void *stream);
SUAMD_API SUBOOL suamd_clock_bank_get_state(suamd_clock_bank_t *b, SUFLOAT *bnor, SUFLOAT *phi, void *stream);
/* clock.type = MANUAL with clock.phase (InspectorCtl/ClockRecovery.cpp:59-93): a bank made with
 * loop_gain = 0 keeps its baud; this sets the sampling phase accumulator of every channel
 * (0.5 * clock.phase; Gardner's start value is 0.25).  Synchronises. */
SUAMD_API SUBOOL suamd_clock_bank_set_phase(suamd_clock_bank_t *b, SUFLOAT phi, void *stream);

/* ------------------------------------------------------------------------------------ */
/* section 8f #2: inspector spectrum sources (INSPECTOR/SPECTRUM messages,                 */
/* Suscan/Analyzer.cpp:539-547, Default/GenericInspector/GenericInspector.cpp:231-254).            */
/* Source ids are 1-based (0 = none, RMSInspector.cpp:721-727): psd, cyclo (x conj(prev)), fmspect */
/* (arg(x conj(prev))), pmspect (arg x), timediff (x - prev), abstimediff (|x - prev|), exp_2/4/8  */
/* (x^2, x^4, x^8).  This is the per-sample transform; the spectrum itself is suamd_psd_feed on    */
/* the result (linear power, natural order: the consumer takes dB and rotates, A9).                 */
/* ------------------------------------------------------------------------------------ */
SUAMD_API unsigned    suamd_spectsrc_count(void);
SUAMD_API const char *suamd_spectsrc_name(unsigned id);
SUAMD_API SUBOOL      suamd_spectsrc_preproc(suamd_ctx_t *ctx, unsigned id, const suamd_complex *d_x, SUSCOUNT len,
                                             SUFLOAT prev_re, SUFLOAT prev_im, suamd_complex *d_y, void *stream);
/* the same with the sample before the block taken from device memory (NULL: zero) -- no host round trip between blocks */
SUAMD_API SUBOOL      suamd_spectsrc_preproc_from(suamd_ctx_t *ctx, unsigned id, const suamd_complex *d_x, SUSCOUNT len,
                                                  const suamd_complex *d_prev, suamd_complex *d_y, void *stream);

/* ------------------------------------------------------------------------------------ */
/* section 8f #3: symbol decision and SNR analytics on the device (1 B/symbol to the host)  */
/* ------------------------------------------------------------------------------------ */
enum suamd_decision_mode { SUAMD_DECIDER_MODULUS = 0, SUAMD_DECIDER_ARGUMENT = 1 };   /* InspectorUI.cpp:229-250 */
/* the decision-space floats InspectorUI::feed forwards (Default/GenericInspector/InspectorUI.cpp:863-873):
 * MODULUS |x|, ARGUMENT arg(j x) / pi */
SUAMD_API SUBOOL suamd_decision_space(suamd_ctx_t *ctx, const suamd_complex *d_x, SUSCOUNT len, int mode,
                                      SUFLOAT *d_out, void *stream);
/* Decider::feed: v = |x| (range [0, 1]) or arg x (range [-pi, pi]); symbol = the interval of 2^bps equal ones
 * v falls into, clamped (SuWidgets is absent: semantics frozen in SPEC.md section K) */
SUAMD_API SUBOOL suamd_decide(suamd_ctx_t *ctx, const suamd_complex *d_x, SUSCOUNT len, int mode, unsigned bps,
                              SUFLOAT vmin, SUFLOAT vmax, unsigned char *d_sym, void *stream);
/* the Histogram widget's history: counts of v over nbins equal bins of [vmin, vmax), added to d_hist */
SUAMD_API SUBOOL suamd_symbol_histogram(suamd_ctx_t *ctx, const suamd_complex *d_x, SUSCOUNT len, int mode,
                                        SUFLOAT vmin, SUFLOAT vmax, unsigned nbins, unsigned *d_hist, void *stream);
/* SNREstimator (Misc/SNREstimator.cpp:30-169): setBps + setAlpha at creation, feed(history) = one model
 * recalculation + gradient step on sigma; get: sigma, getSNR() = 1 / (2^bps sigma), sum of squared errors^2 */
typedef struct suamd_snr_estimator suamd_snr_estimator_t;
SUAMD_API suamd_snr_estimator_t *suamd_snr_estimator_new(suamd_ctx_t *ctx, unsigned bps, SUFLOAT alpha);
SUAMD_API void     suamd_snr_estimator_destroy(suamd_snr_estimator_t *e);
SUAMD_API SUBOOL   suamd_snr_estimator_feed(suamd_snr_estimator_t *e, const unsigned *d_history, unsigned length, void *stream);
SUAMD_API SUBOOL   suamd_snr_estimator_get(suamd_snr_estimator_t *e, SUFLOAT *sigma, SUFLOAT *snr, SUFLOAT *sqerr, void *stream);
SUAMD_API SUFLOAT *suamd_snr_estimator_model(suamd_snr_estimator_t *e);     /* device pointer: getModel(), `length` floats */

/* ------------------------------------------------------------------------------------ */
/* section 8f #4: fast autocorrelation of the inspector's sample stream                    */
/* FACTab::feed (Default/GenericInspector/FACTab.cpp:181-246): per full buffer of `size` samples   */
/* FFT -> x conj(x) -> inverse FFT -> |.| of the first half; running max / min over               */
/* [view_start, view_end); fac[i] += alpha (|.|/max - fac[i]).                                      */
/* ------------------------------------------------------------------------------------ */
typedef struct suamd_fac suamd_fac_t;
SUAMD_API suamd_fac_t *suamd_fac_new(suamd_ctx_t *ctx, unsigned size /* 2^k, 16..2^24 */, SUFLOAT alpha);
SUAMD_API void     suamd_fac_destroy(suamd_fac_t *f);
SUAMD_API void     suamd_fac_set_alpha(suamd_fac_t *f, SUFLOAT alpha);
SUAMD_API SUBOOL   suamd_fac_reset(suamd_fac_t *f, void *stream);           /* resizeFAC / setSampleRate: fac = 0, max / min reset */
/* d_data: nbuffers consecutive buffers of `size` samples (the caller keeps the partial-buffer remainder) */
SUAMD_API SUBOOL   suamd_fac_feed(suamd_fac_t *f, const suamd_complex *d_data, SUSCOUNT nbuffers, SUSDIFF view_start,
                                  SUSDIFF view_end, void *stream);
SUAMD_API SUFLOAT *suamd_fac_array(suamd_fac_t *f);                          /* device pointer, size/2 floats */
SUAMD_API SUBOOL   suamd_fac_get_range(suamd_fac_t *f, SUFLOAT *min, SUFLOAT *max, void *stream);

/* ------------------------------------------------------------------------------------ */
/* A7: stages behind the rest of the inspector config vocabulary                         */
/* (Default/GenericInspector/InspectorCtl/{GainControl,AfcControl,MfControl,Equalizer-   */
/* Control}.cpp; semantics frozen in SPEC.md section I -- libsuscan is absent)           */
/* ------------------------------------------------------------------------------------ */
/* agc.enabled = false: rows scaled by the fixed agc.gain */
SUAMD_API SUBOOL suamd_rows_scale(suamd_ctx_t *ctx, const suamd_complex *d_x, suamd_view xv, suamd_complex *d_y,
                                  suamd_view yv, unsigned nchan, SUSCOUNT len, SUFLOAT gain, void *stream);
/* afc.costas-order = 0 (manual) with afc.offset: a free-running NCO per channel,
 * y_c[n] = x_c[n] * exp(j pi fnor_c n), n counted across feeds; pass fnor = -2 offset / equiv_fs */
typedef struct suamd_nco_bank suamd_nco_bank_t;
SUAMD_API suamd_nco_bank_t *suamd_nco_bank_new(suamd_ctx_t *ctx, unsigned nchan, const double *fnor);
SUAMD_API void   suamd_nco_bank_destroy(suamd_nco_bank_t *b);
SUAMD_API SUBOOL suamd_nco_bank_feed(suamd_nco_bank_t *b, const suamd_complex *d_x, suamd_view xv, suamd_complex *d_y,
                                     suamd_view yv, SUSCOUNT len, void *stream);
/* mf.type = MANUAL with mf.roll-off: root-raised-cosine taps spanning six symbols
 * (suamd_rrc_ntaps(sps) = 2 ceil(3 sps) + 1, unit DC gain), and a real-tap FIR at the channel rate for a
 * bank of rows (history carried across feeds; d_y must not alias d_x) */
SUAMD_API unsigned suamd_rrc_ntaps(double sps);
SUAMD_API void     suamd_rrc_design(SUFLOAT *taps, unsigned ntaps, double sps, double rolloff);
typedef struct suamd_fir_bank suamd_fir_bank_t;
SUAMD_API suamd_fir_bank_t *suamd_fir_bank_new(suamd_ctx_t *ctx, unsigned nchan, const SUFLOAT *taps, unsigned ntaps);
SUAMD_API void   suamd_fir_bank_destroy(suamd_fir_bank_t *b);
SUAMD_API SUBOOL suamd_fir_bank_feed(suamd_fir_bank_t *b, const suamd_complex *d_x, suamd_view xv, suamd_complex *d_y,
                                     suamd_view yv, SUSCOUNT len, void *stream);
/* equalizer.type = CMA with equalizer.rate / equalizer.locked: constant-modulus equalizer of `ntaps`
 * (1..16) complex weights per channel at the symbol rate, w[0] = 1 initially.  Rows are channel-major;
 * channel c consumes d_count[c] symbols (the clock bank's counts) or fixed_len when d_count is NULL. */
typedef struct suamd_cma_bank suamd_cma_bank_t;
SUAMD_API suamd_cma_bank_t *suamd_cma_bank_new(suamd_ctx_t *ctx, unsigned nchan, unsigned ntaps, SUFLOAT rate);
SUAMD_API void   suamd_cma_bank_destroy(suamd_cma_bank_t *b);
SUAMD_API void   suamd_cma_bank_set_locked(suamd_cma_bank_t *b, SUBOOL locked);
SUAMD_API void   suamd_cma_bank_set_rate(suamd_cma_bank_t *b, SUFLOAT rate);
SUAMD_API SUBOOL suamd_cma_bank_feed(suamd_cma_bank_t *b, const suamd_complex *d_x, SUSCOUNT x_stride,
                                     const uint32_t *d_count, SUSCOUNT fixed_len, suamd_complex *d_y,
                                     SUSCOUNT y_stride, void *stream);
SUAMD_API SUBOOL suamd_cma_bank_get_weights(suamd_cma_bank_t *b, suamd_complex *weights /* [ntaps][nchan] */, void *stream);

/* ------------------------------------------------------------------------------------ */
/* Gangs: n one-channel banks -- each with its OWN parameters, state, row and length -- run side by     */
/* side, one lane each, in one launch per loop type.  This is how the live analyzer serves many            */
/* inspectors that cannot share a bank (different loop bandwidths, bauds, decimations): the serial       */
/* recurrences of all of them cost the time of one.  Rows are contiguous; results are bit-identical to    */
/* feeding every bank on its own.  Gang calls of one context go on one stream (or are ordered otherwise). */
/* ------------------------------------------------------------------------------------ */
SUAMD_API SUBOOL suamd_costas_gang_feed(suamd_ctx_t *ctx, suamd_costas_bank_t *const *banks, unsigned n,
                                        const suamd_complex *const *d_x, suamd_complex *const *d_y,
                                        const SUSCOUNT *len, void *stream);
SUAMD_API SUBOOL suamd_agc_gang_feed(suamd_ctx_t *ctx, suamd_agc_bank_t *const *banks, unsigned n,
                                     const suamd_complex *const *d_x, suamd_complex *const *d_y,
                                     const SUSCOUNT *len, void *stream);
SUAMD_API SUBOOL suamd_pll_gang_feed(suamd_ctx_t *ctx, suamd_pll_bank_t *const *banks, unsigned n,
                                     const suamd_complex *const *d_x, suamd_complex *const *d_y,
                                     const SUSCOUNT *len, void *stream);
/* symbol rows: item i consumes *d_count[i] symbols (d_count may be NULL: then fixed_len[i]); in place allowed */
SUAMD_API SUBOOL suamd_cma_gang_feed(suamd_ctx_t *ctx, suamd_cma_bank_t *const *banks, unsigned n,
                                     const suamd_complex *const *d_x, const uint32_t *const *d_count,
                                     const SUSCOUNT *fixed_len, suamd_complex *const *d_y, void *stream);
/* suamd_agc_gang_feed in its four steps, for callers that pipeline sub-ranges [m0, m1) of a block through the level
 * trackers and the stages behind them: pre (whole block) -> { level, apply } per sub-range in order -> finish */
SUAMD_API SUBOOL suamd_agc_gang_pre(suamd_ctx_t *ctx, suamd_agc_bank_t *const *banks, unsigned n,
                                    const suamd_complex *const *d_x, const SUSCOUNT *len, void *stream);
SUAMD_API SUBOOL suamd_agc_gang_level(suamd_ctx_t *ctx, suamd_agc_bank_t *const *banks, unsigned n, const SUSCOUNT *len,
                                      const SUSCOUNT *m0, const SUSCOUNT *m1, void *stream);
SUAMD_API SUBOOL suamd_agc_gang_apply(suamd_ctx_t *ctx, suamd_agc_bank_t *const *banks, unsigned n,
                                      const suamd_complex *const *d_x, suamd_complex *const *d_y, const SUSCOUNT *len,
                                      const SUSCOUNT *m0, const SUSCOUNT *m1, void *stream);
SUAMD_API SUBOOL suamd_agc_gang_finish(suamd_ctx_t *ctx, suamd_agc_bank_t *const *banks, unsigned n,
                                       const suamd_complex *const *d_x, const SUSCOUNT *len, void *stream);
/* Results of many rows to their landing zones in one launch (the analyzer's message hand-off,
 * Suscan/Messages/SamplesMessage.h: one batch per inspector per block): row i delivers n_i = *d_count[i] samples
 * (d_count[i] NULL, or d_count NULL: fixed_len[i]) of d_src[i] to dst[i] and n_i to *count_out[i]; a device counter
 * is cleared afterwards, ready for the next suamd_clock_gang_feed.  dst / count_out may be device memory or
 * host memory mapped into the device (hipHostMalloc): then no copy is left for the host to issue. */
SUAMD_API SUBOOL suamd_rows_deliver(suamd_ctx_t *ctx, unsigned n, const suamd_complex *const *d_src,
                                    uint32_t *const *d_count, const SUSCOUNT *fixed_len,
                                    suamd_complex *const *dst, uint32_t *const *count_out, void *stream);
SUAMD_API SUBOOL suamd_clock_gang_feed(suamd_ctx_t *ctx, suamd_clock_bank_t *const *banks, unsigned n,
                                       const suamd_complex *const *d_x, const SUSCOUNT *len,
                                       suamd_complex *const *d_sym, uint32_t *const *d_count, void *stream);

/* ------------------------------------------------------------------------------------ */
/* Gangs on time-major slabs.  The same one-channel banks when their rows are COLUMNS of one slab per side     */
/* already -- what suamd_specttuner_feed leaves with the view {1, pitch}: sample m of item i at               */
/* d_x[i][m * x_pitch], the items side by side in a slab row.  Nothing is gathered or scattered: a            */
/* recurrence wavefront streams the slab where it lies (512 contiguous bytes per time step for 64 adjacent     */
/* columns), the feed-forward AGC steps work in tiles of 64 columns.  This is how the live analyzer serves     */
/* the inspectors of its FFT filter bank (Suscan/Analyzer.cpp:411-484: independent handles, one shared         */
/* baseband).  Results are bit-identical to the row forms above.  The row starts of a side must lie within     */
/* 4 GiB of the lowest one, and a slab must be readable 64 rows beyond its longest item.                       */
/* ------------------------------------------------------------------------------------ */
SUAMD_API SUBOOL suamd_costas_gang_feed_slab(suamd_ctx_t *ctx, suamd_costas_bank_t *const *banks, unsigned n,
                                             const suamd_complex *const *d_x, SUSCOUNT x_pitch,
                                             suamd_complex *const *d_y, SUSCOUNT y_pitch, const SUSCOUNT *len, void *stream);
SUAMD_API SUBOOL suamd_pll_gang_feed_slab(suamd_ctx_t *ctx, suamd_pll_bank_t *const *banks, unsigned n,
                                          const suamd_complex *const *d_x, SUSCOUNT x_pitch,
                                          suamd_complex *const *d_y, SUSCOUNT y_pitch, const SUSCOUNT *len, void *stream);
/* (symbols still leave as contiguous rows d_sym[i], counted in *d_count[i]: they are few) */
SUAMD_API SUBOOL suamd_clock_gang_feed_slab(suamd_ctx_t *ctx, suamd_clock_bank_t *const *banks, unsigned n,
                                            const suamd_complex *const *d_x, SUSCOUNT x_pitch, const SUSCOUNT *len,
                                            suamd_complex *const *d_sym, uint32_t *const *d_count, void *stream);
/* The AGC's four steps (suamd_agc_gang_pre / _level / _apply / _finish).  Item i is column d_x[i] - d_x_slab (< x_pitch)
 * of the input slab; its magnitudes and levels live in the same column of two work slabs the caller lends for the
 * block: d_work holds 2 * work_rows * x_pitch floats, work_rows >= every len[i].  _apply writes column
 * d_y[i] - d_y_slab of the output slab.  mag_history_size and delay_line_size <= 64. */
SUAMD_API SUBOOL suamd_agc_gang_pre_slab(suamd_ctx_t *ctx, suamd_agc_bank_t *const *banks, unsigned n,
                                         const suamd_complex *d_x_slab, SUSCOUNT x_pitch, const suamd_complex *const *d_x,
                                         const SUSCOUNT *len, float *d_work, SUSCOUNT work_rows, void *stream);
SUAMD_API SUBOOL suamd_agc_gang_level_slab(suamd_ctx_t *ctx, suamd_agc_bank_t *const *banks, unsigned n,
                                           const suamd_complex *d_x_slab, SUSCOUNT x_pitch, const suamd_complex *const *d_x,
                                           const SUSCOUNT *len, const SUSCOUNT *m0, const SUSCOUNT *m1,
                                           float *d_work, SUSCOUNT work_rows, void *stream);
SUAMD_API SUBOOL suamd_agc_gang_apply_slab(suamd_ctx_t *ctx, suamd_agc_bank_t *const *banks, unsigned n,
                                           const suamd_complex *d_x_slab, SUSCOUNT x_pitch, const suamd_complex *const *d_x,
                                           suamd_complex *d_y_slab, SUSCOUNT y_pitch, suamd_complex *const *d_y,
                                           const SUSCOUNT *len, const SUSCOUNT *m0, const SUSCOUNT *m1,
                                           float *d_work, SUSCOUNT work_rows, void *stream);
SUAMD_API SUBOOL suamd_agc_gang_finish_slab(suamd_ctx_t *ctx, suamd_agc_bank_t *const *banks, unsigned n,
                                            const suamd_complex *d_x_slab, SUSCOUNT x_pitch, const suamd_complex *const *d_x,
                                            const SUSCOUNT *len, float *d_work, SUSCOUNT work_rows, void *stream);
/* suamd_rows_deliver for sources with an element stride each (src_stride NULL: all 1): a slab column hands its batch over */
SUAMD_API SUBOOL suamd_rows_deliver_strided(suamd_ctx_t *ctx, unsigned n, const suamd_complex *const *d_src,
                                            const SUSCOUNT *src_stride, uint32_t *const *d_count, const SUSCOUNT *fixed_len,
                                            suamd_complex *const *dst, uint32_t *const *count_out, void *stream);

/* struct su_agc_params (Tasks/AGCTask.cpp:41-47) + su_agc_params_INITIALIZER defaults */
struct suamd_agc_params {
  SUFLOAT  threshold;
  SUFLOAT  slope_factor;
  unsigned hang_max;
  unsigned delay_line_size;
  unsigned mag_history_size;
  SUFLOAT  fast_rise_t, fast_fall_t, slow_rise_t, slow_fall_t;
};
#define suamd_agc_params_INITIALIZER { -100.f, 6.f, 100, 20, 20, 2.f, 4.f, 20.f, 40.f }
/* AGCTask ctor time constants (Tasks/AGCTask.cpp:22-28,43-47) */
SUAMD_API void   suamd_agc_params_from_tau(struct suamd_agc_params *p, SUFLOAT tau);
SUAMD_API suamd_agc_bank_t *suamd_agc_bank_new(suamd_ctx_t *ctx, unsigned nchan, const struct suamd_agc_params *p);
SUAMD_API void   suamd_agc_bank_destroy(suamd_agc_bank_t *b);
/* `dest[p] = su_agc_feed(&agc, origin[p])` loop (Tasks/AGCTask.cpp:70-73) */
SUAMD_API SUBOOL suamd_agc_bank_feed(suamd_agc_bank_t *b, const suamd_complex *d_x, suamd_view xv,
                                     suamd_complex *d_y, suamd_view yv, SUSCOUNT len, void *stream);
/* The same on two streams: the level trackers (the AGC's one recurrence: one wavefront per 64 channels, milliseconds per
 * launch) on `stream_level`, its feed-forward kernels (|x|^2 in dB, sliding maximum, gain on the delayed input, state carry:
 * many workgroups, microseconds) on `stream_wide`, ordered by device-side event waits.  For callers that confine their
 * recurrence streams to a few compute units (suamd_stream_new_cu_mask): the feed-forward kernels must not be confined
 * with them.  The input is read and the output complete on `stream_wide`.  Same samples bit for bit. */
SUAMD_API SUBOOL suamd_agc_bank_feed_split(suamd_agc_bank_t *b, const suamd_complex *d_x, suamd_view xv,
                                           suamd_complex *d_y, suamd_view yv, SUSCOUNT len, void *stream_level, void *stream_wide);

/* ------------------------------------------------------------------------------------ */
/* T9 / T10: whole-capture FFT tasks                                                     */
/* ------------------------------------------------------------------------------------ */
/* Forward FFT of n = 2^log2n complex points (16 <= n <= 2^24), out of place: the
 * SU_FFTW(_plan_dft_1d)(n, buf, buf, FFTW_FORWARD, ...) + _execute pair of
 * Tasks/CarrierDetector.cpp:58-75,94.  d_work: scratch of n complex values. */
SUAMD_API SUBOOL suamd_fft_forward_bulk(suamd_ctx_t *ctx, const suamd_complex *d_in, suamd_complex *d_out,
                                        suamd_complex *d_work, unsigned log2n, void *stream);
/* The "power" inspector class (A6; consumer: RMSInspector, Default/RMSInspector/RMSInspector.cpp:255-338, 538-562, config
 * key power.integrate-samples :415): every window of N consecutive channel samples yields one output sample
 * (mean of Re(x conj x), 0).  The arithmetic is the one the reference runs itself in raw mode -- binary64 sum of the
 * binary32 powers over the count (Kahan there, a fixed-order tree here).  Windows span feeds; set_integrate starts over. */
typedef struct suamd_power_bank suamd_power_bank_t;
SUAMD_API suamd_power_bank_t *suamd_power_bank_new(suamd_ctx_t *ctx, SUSCOUNT integrate_samples);
SUAMD_API void     suamd_power_bank_destroy(suamd_power_bank_t *bank);
SUAMD_API SUBOOL   suamd_power_bank_set_integrate(suamd_power_bank_t *bank, SUSCOUNT integrate_samples, void *stream);
SUAMD_API SUSCOUNT suamd_power_bank_output_count(const suamd_power_bank_t *bank, SUSCOUNT len);
SUAMD_API SUBOOL   suamd_power_bank_feed(suamd_power_bank_t *bank, const suamd_complex *d_x, SUSCOUNT len, suamd_complex *d_out,
                                         SUSCOUNT *n_out, void *stream);

/* A5: the frame time-to-live rule in front of the averager (UIMediator::feedPSD, UIMediator/SpectrumMediator.cpp:35-85,
 * 127): a PSD frame whose age (now - rt_time), less the running estimate of the intrinsic delivery delay, exceeds
 * ttl_ms is dropped -- unless the source has just looped.  Host arithmetic on time stamps (binary64), state in the
 * caller's struct; the delay estimate is SU_SPLPF_FEED with SU_SPLPF_ALPHA(10) (recollection: 1 - exp(-1/10)) after a
 * calibration phase (SIGDIGGER_UI_MEDIATOR_PSD_CAL_LEN = 10, include/UIMediator.h:36). */
typedef struct { double rt_delta_real; unsigned rt_calibrations; SUBOOL have_rt_delta; } suamd_psd_ttl_t;
#define suamd_psd_ttl_INITIALIZER { 0.0, 0, SU_FALSE }
/* SU_TRUE = feed the frame to the averager / spectrum (not expired, or looped) */
SUAMD_API SUBOOL suamd_psd_ttl_accept(suamd_psd_ttl_t *state, double now_s, double rt_time_s, double ttl_ms, SUBOOL looped);

/* Capture export: ExportSamplesTask (Tasks/ExportSamplesTask.cpp:41-284) for a capture that
 * lives in HBM -- streamed to the host a pinned chunk at a time and written in the reference's formats:
 *   "raw"  interleaved float32 I/Q                       (:250-270, libsndfile RAW | FLOAT)
 *   "wav"  RIFF/WAVE, IEEE float, 2 channels, (int) fs   (:226-248, libsndfile WAV | FLOAT; fmt + fact + data chunks)
 *   "m"    exportToMatlab's text, character for character (:41-68: 6 significant digits, "re + imi, ")
 *   "mat"  MAT-file level 5 with sampleRate (1x1), deltaT (1x1), X (2 x N: rows I, Q), single precision (:164-206;
 *          libsigutils' su_mat_file is absent: layout per the published format, readable by scipy.io.loadmat)
 * Any other format string is refused with the reference's message.  Synchronises the stream. */
SUAMD_API SUBOOL suamd_export_capture(suamd_ctx_t *ctx, const char *path, const char *format, const suamd_complex *d_data,
                                      SUSCOUNT len, SUFLOAT fs, void *stream);

/* Baud estimators behind the inspectors' ESTIMATOR messages (Suscan/Analyzer.cpp:549-565,
 * InspectorUI::updateEstimator, Default/GenericInspector/InspectorUI.cpp:1003-1015; the estimators themselves are
 * libsuscan's -- absent -- and are frozen in SPEC.md section M).  Both work on the first `size` samples of a block:
 *   NONLINEAR: y[n] = |x[n] - x[n-1]|^2 has spectral lines at the baud and its harmonics; Blackman-Harris, FFT, the
 *              LOWEST local maximum of |Y|^2 outside a 1 % DC notch that reaches half of the strongest one (which must
 *              stand 20x above the mean level: otherwise no estimate), 9-bin power centroid;
 *   FAC:       fast autocorrelation (suamd_fac, alpha 0.25) -> first valley of the 3-tap smoothed curve below a
 *              quarter of lag 0; baud = 1 / lag (whole samples).
 * feed() only enqueues (results land in host-mapped memory); get() is valid after the stream was synchronised and
 * returns the normalised baud (symbols per sample, x equiv_fs = Hz), 0 = no estimate.  Blocks shorter than `size`
 * leave the estimate as it is. */
typedef struct suamd_baud_estimator suamd_baud_estimator_t;
/* kind 2, "carrier": the channel's residual carrier offset in cycles per sample, (-1/2, 1/2] -- the reference's own
 * CarrierDetector computation (Tasks/CarrierDetector.cpp:99-137: window, transform, strongest bin, power-weighted phasor sum
 * around it) with avgRelBw = 1/2 and no DC notch, on the first `size` samples of a block; x equiv_fs = what afc.offset takes
 * (SPEC.md section M).  Same object, same calls as the two baud estimators. */
enum suamd_baud_estimator_kind { SUAMD_BAUD_ESTIMATOR_FAC = 0, SUAMD_BAUD_ESTIMATOR_NONLINEAR = 1, SUAMD_ESTIMATOR_CARRIER = 2 };
SUAMD_API suamd_baud_estimator_t *suamd_baud_estimator_new(suamd_ctx_t *ctx, int kind, unsigned size /* 2^k, 512..2^20 */);
SUAMD_API void     suamd_baud_estimator_destroy(suamd_baud_estimator_t *e);
SUAMD_API unsigned suamd_baud_estimator_size(const suamd_baud_estimator_t *e);
SUAMD_API SUBOOL   suamd_baud_estimator_feed(suamd_baud_estimator_t *e, const suamd_complex *d_x, SUSCOUNT len, void *stream);
SUAMD_API SUFLOAT  suamd_baud_estimator_get(const suamd_baud_estimator_t *e);
/* the same with the estimate landing in the caller's own (pinned) host location when the stream gets there: for
 * callers with several blocks in flight.  len must cover the analysis window. */
SUAMD_API SUBOOL   suamd_baud_estimator_feed_to(suamd_baud_estimator_t *e, const suamd_complex *d_x, SUSCOUNT len,
                                                SUFLOAT *h_value, void *stream);

/* CarrierDetector::work, all states (Tasks/CarrierDetector.cpp:49-147): zero-pad to a power of two,
 * Blackman-Harris over len, FFT, |X|^2 arg-max outside the DC notch, circular centroid over
 * alloc*avg_rel_bw + 1 bins.  *peak = carrier in rad/sample.  Synchronises the stream. */
SUAMD_API SUBOOL suamd_carrier_detect(suamd_ctx_t *ctx, const suamd_complex *d_data, SUSCOUNT len,
                                      SUFLOAT avg_rel_bw, SUFLOAT dc_notch_rel_bw, SUFLOAT *peak, void *stream);
/* DopplerCalculator::work (Tasks/DopplerCalculator.cpp:52-182): same front end; mirrored power
 * spectrum into d_spectrum (may be NULL; *alloc floats), velocity centroid *peak (m/s), its
 * spread *sigma (Hz-domain std dev as the reference computes it) and the spectrum maximum. */
SUAMD_API SUSCOUNT suamd_doppler_alloc_size(SUSCOUNT len);
SUAMD_API SUBOOL suamd_doppler_calc(suamd_ctx_t *ctx, const suamd_complex *d_data, SUSCOUNT len, SUFLOAT fs,
                                    SUFREQ f0, SUFLOAT *d_spectrum, SUFLOAT *peak, SUFLOAT *sigma, SUFLOAT *max,
                                    void *stream);

/* ------------------------------------------------------------------------------------ */
/* P2 / P3: panoramic scanner SpectrumView                                               */
/* ------------------------------------------------------------------------------------ */
/* struct SpectrumView (include/Scanner.h:63-110): psd / psdAccum / psdCount of up to 65536 bins
 * resident on the GPU.  set_range = SpectrumView::setRange (Panoramic/Scanner.cpp:41-54), feed =
 * SpectrumView::feed (:239-256: feedLinearMode :118-185 or feedHistogramMode :187-237, then
 * interpolate :56-116).  d_psd: one PSD frame as delivered by PSDMessage (shifted, dB). */
typedef struct suamd_specview suamd_specview_t;
#define SUAMD_SCANNER_SPECTRUM_SIZE 65536
SUAMD_API suamd_specview_t *suamd_specview_new(suamd_ctx_t *ctx);
SUAMD_API void     suamd_specview_destroy(suamd_specview_t *v);
SUAMD_API SUBOOL   suamd_specview_set_range(suamd_specview_t *v, SUFREQ freq_min, SUFREQ freq_max, void *stream);
SUAMD_API void     suamd_specview_set_fft(suamd_specview_t *v, SUFREQ fft_bandwidth, SUFLOAT fft_rel_bw);
SUAMD_API unsigned suamd_specview_spectrum_size(const suamd_specview_t *v);
/* d_count may be NULL (every source bin counts 1); adjust_sides crops (1 - fftRelBw)/2 per side */
SUAMD_API SUBOOL   suamd_specview_feed(suamd_specview_t *v, const SUFLOAT *d_psd, const SUFLOAT *d_count,
                                       SUSCOUNT psd_size, SUFREQ freq_min, SUFREQ freq_max, SUBOOL adjust_sides,
                                       void *stream);
/* nframes consecutive frames d_psd[f*psd_size ...] centred on center[f] (host array), each spanning
 * fft_bandwidth: Scanner::onPSDMessage (Panoramic/Scanner.cpp:503-523) for a whole sweep */
SUAMD_API SUBOOL   suamd_specview_feed_sweep(suamd_specview_t *v, const SUFLOAT *d_psd, SUSCOUNT psd_size,
                                             SUSCOUNT nframes, const SUFREQ *center, SUBOOL adjust_sides, void *stream);
/* device pointers to the view's arrays (valid until destroy): 0 psd, 1 psdAccum, 2 psdCount */
SUAMD_API SUFLOAT *suamd_specview_array(suamd_specview_t *v, int which);

#ifdef __cplusplus
}
#endif
#endif /* SIGDIGGER_AMD_H */
