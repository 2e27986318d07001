// tuning.hpp -- every knob that changes HOW the library computes (launch plans, kernel choices, schedules; never WHAT: each
// setting gives the same bits), in ONE documented struct.  Rounds 1-5 grew ~40 getenv("SUAMD_*") calls through the
// translation units, some read on every feed; now the environment is read ONCE, at the first use of sdk::tuning(), into
// this struct, and the C ABI (suamd_tuning_set / _get / _describe, include/sigdigger_amd.h) changes a field afterwards --
// what the A / B tests use.  -1 / 0 mean "the library decides" unless a field says otherwise.
//
// Not here: deployment configuration of the live analyzer (SUAMD_DEVICES, SUAMD_ANALYZER_BCAST, SUAMD_RCCL_LIB,
// SUAMD_ANALYZER_CHANNELISER ...: read once by suscan_analyzer_new) and the instrumented build's
// switches (-DSUAMD_INSTRUMENT: phase clocks, wrong-result timing experiments -- never in the shipped library).
#pragma once

// X(field, "ENV_NAME", default, min, max, "what it does")
#define SUAMD_TUNING_FIELDS(X)                                                                                                              \
  /* FFT channeliser (specttuner_host.cpp, specttuner*.hip) */                                                                              \
  X(st_run,          "SUAMD_ST_RUN",          0, 0, 4096,    "windows per workgroup of a channeliser launch; 0: planned from the block and the slot budget") \
  X(st_kernel,       "SUAMD_ST_KERNEL",       0, 0, 2,       "narrow-channel kernel: 0 two wavefronts per window (stp_kernel) where it applies, 1 one wavefront (stw_kernel), 2 the workgroup kernel (st_kernel) for every size; env: pair / wave / wg") \
  X(st_seam_polls,   "SUAMD_ST_SEAM_POLLS",   -1, -1, 1 << 20, "polls a run spends on its successor's seam payload before it transforms the seam window itself; -1: the default (256), 0: always self-transform") \
  X(st_y32,          "SUAMD_ST_Y32",          1, 0, 1,       "0: 64-bit output addressing everywhere (debug)") \
  X(st_slots,        "SUAMD_ST_SLOTS",        0, 0, 4096,    "window slots a launch is planned for; 0: 768 (3/4 of the chip's 1024) unless suamd_specttuner_set_slots says otherwise") \
  X(st_seam,         "SUAMD_ST_SEAM",         -1, -1, 1,     "wide channels: 1 always / 0 never hand the seam over through st_seam_kernel instead of a warm-up window; -1: banks of one channel group") \
  X(st_variant,      "SUAMD_ST_VARIANT",      0, 0, 8,       "st_kernel instantiation override (measurements)") \
  X(st_ngl,          "SUAMD_ST_NGL",          0, 0, 4,       "channel groups a wide-channel workgroup serves per forward transform; 0: two where the bank has more than one") \
  X(st_pair_sep,     "SUAMD_ST_PAIR_SEP",     1, 0, 1,       "stp_kernel: 0 never gives the forward swaps 16 KiB of LDS of their own") \
  X(st_row_stage,    "SUAMD_ST_ROW_STAGE",    1, 0, 1,       "stp_kernel, per-channel rows (the live analyzer): 0 stores lane by lane instead of through the LDS transposition") \
  /* main spectrum (psd.hip, psd_large.hip, capi.hip) */                                                                                    \
  X(psd_large,       "SUAMD_PSD_LARGE",       -1, -1, 1,     "frames above 16384 points: 0 round 2's radix-16 passes through HBM, 1 the two-trip transform for 32768 points too, -1 by size (32768: one trip; above: two); env: passes / twotrip") \
  X(psd_large_points, "SUAMD_PSD_LARGE_POINTS", 27, 15, 30,  "log2 of the points one batch of the two-trip PSD keeps in its intermediate") \
  X(psd_large_batch, "SUAMD_PSD_LARGE_BATCH", 0, 0, 1 << 20, "frames per batch of the two-trip PSD; 0: from psd_large_points (tests: awkward batch boundaries)") \
  X(psd_stream,      "SUAMD_PSD_STREAM",      -1, -1, 1,     "psd_kernel: 1 / 0 force / forbid the streaming (many frames per workgroup) form") \
  X(psd_split_target, "SUAMD_PSD_SPLIT_TARGET", 0, 0, 1 << 16, "workgroups a PSD launch that averages many frames into few outputs is split into; 0: one per CU unless suamd_psd_set_split_target says otherwise") \
  X(psd_min_frames,  "SUAMD_PSD_MIN_FRAMES",  0, 0, 1 << 16, "fewest frames a split workgroup takes; 0: the default (2)") \
  X(psd_large_n2,    "SUAMD_PSD_LARGE_N2",    0, 0, 1 << 12, "two-trip PSD: row length override (measurements)") \
  /* translate + FIR bank (chan.hip, chan_stream.hip) */                                                                                    \
  X(fir_smem_taps,   "SUAMD_FIR_SMEM_TAPS",   0, 0, 2,       "chan_fir_kernel: 1 / 2 force taps from scalar loads / LDS") \
  X(fir_nout,        "SUAMD_FIR_NOUT",        0, 0, 8,       "chan_fir_kernel: outputs per lane override") \
  X(fir_tc4,         "SUAMD_FIR_TC4",         0, 0, 1,       "chan_fir_kernel: 1 four-channel tile") \
  X(fir_stream,      "SUAMD_FIR_STREAM",      -1, -1, 2,     "banks of one or two channels: 0 chan_fir_kernel, 1 chan_pair_kernel, -1 by shape") \
  X(fir_pair_nw,     "SUAMD_FIR_PAIR_NW",     0, 0, 8,       "chan_pair_kernel: wavefronts per workgroup (1 / 2 / 4 / 8; 8 = persistent streams of 1024-output tiles); 0: by shape") \
  X(fir_pair_tpw,    "SUAMD_FIR_PAIR_TPW",    0, 0, 1 << 20, "chan_pair_kernel: tiles per persistent workgroup; 0: planned") \
  /* recurrences (loops.hip) */                                                                                                             \
  X(serial_xcd,      "SUAMD_SERIAL_XCD",      1, 0, 1,       "0: the one-wavefront recurrence launches do not spread AGC / Costas / clock over three XCDs") \
  X(clock_mode,      "SUAMD_CLOCK_MODE",      -1, -1, 2,     "bank clock recovery schedule: 0 lock step, 1 crossing by crossing (divergent), 2 round by round (clock_ring); -1: 2 up to 24 samples per half cycle, else 0") \
  /* live analyzer (analyzer.cpp): run-time behaviour of the worker, not arithmetic */                                                      \
  X(analyzer_debug,  "SUAMD_ANALYZER_DEBUG",  0, 0, 1,       "1: the worker narrates its blocks on stderr") \
  X(analyzer_trace,  "SUAMD_ANALYZER_TRACE",  0, 0, 1,       "1: one block at a time with hipEvent stamps per stage (a timeline on stderr)") \
  X(analyzer_poison_rows, "SUAMD_ANALYZER_POISON_ROWS", 0, 0, 1, "1: inspector rows start as NaNs (debug)") \
  X(analyzer_subranges, "SUAMD_ANALYZER_SUBRANGES", 0, 0, 8, "sub-ranges a block takes through the serial stages; 0: 4, or 2 beyond 128 inspectors") \
  X(analyzer_pipeline, "SUAMD_ANALYZER_PIPELINE", 1, 0, 1,   "0: one block in flight instead of two") \
  X(analyzer_slab,   "SUAMD_ANALYZER_SLAB",   1, 0, 1,       "FFT filter bank: 0 gives every inspector rows of its own (rounds 1-5) instead of columns of one time-major slab for the narrow (<= 64-bin) channels") \
  X(analyzer_stage_priority, "SUAMD_ANALYZER_STAGE_PRIORITY", 99, -2, 99, "priority of the three recurrence streams (HIP: lower is higher); 99: the device's highest; -2 (env: off): default-priority streams")

namespace sdk {

struct Tuning {
#define SUAMD_TUNING_DECL(f, env, def, lo, hi, doc) long long f = def;
  SUAMD_TUNING_FIELDS(SUAMD_TUNING_DECL)
#undef SUAMD_TUNING_DECL
};

struct TuningField { const char *name, *env; long long def, lo, hi; const char *doc; long long Tuning::*member; };

// the process-wide instance: the first call reads the environment (once); later reads cost a load
Tuning &tuning();
const TuningField *tuning_fields(unsigned *count);
bool tuning_set(const char *name, long long value);          // false: no such field, or value outside [lo, hi]
bool tuning_get(const char *name, long long *value);
void tuning_reset();                                         // back to defaults + environment

}  // namespace sdk
