/*
 * include/suscan_amd.h -- live-path C ABI: the subset of libsuscan that SigDigger's
 * Suscan::Analyzer wrapper calls (Suscan/Analyzer.cpp:63-639, Suscan/MQ.cpp:24-44), served by
 * libsigdigger_amd.so with the PSD and the inspector chains running on the MI355X.
 *
 * Function names, argument order and the message / ownership protocol are the reference's
 * (SURVEY.md section 8b, Appendix A/B); struct LAYOUTS cannot be checked against upstream headers
 * (absent from /root/reference), so compatibility is source level: recompile the wrapper against
 * this header.  Every message returned by suscan_analyzer_read() is an ordinary malloc'd host
 * object owned by the caller (consumers mutate payloads in place: PSDMessage.cpp:34-38) and must
 * be released with exactly one suscan_analyzer_dispose_message().
 */
#ifndef SUSCAN_AMD_H
#define SUSCAN_AMD_H

#include <stdint.h>
#include <sys/time.h>
#include "sigdigger_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

#ifndef SUHANDLE_DEFINED
#  define SUHANDLE_DEFINED
typedef int32_t SUHANDLE;
#endif

/* ---- message queue (Suscan/MQ.cpp:28-43) --------------------------------------------------- */
struct suscan_mq { void *impl; };
SUAMD_API SUBOOL suscan_mq_init(struct suscan_mq *mq);
SUAMD_API void   suscan_mq_finalize(struct suscan_mq *mq);
SUAMD_API void  *suscan_mq_read(struct suscan_mq *mq, uint32_t *type);          /* blocks */
SUAMD_API SUBOOL suscan_mq_poll(struct suscan_mq *mq, uint32_t *type, void **msg);
SUAMD_API SUBOOL suscan_mq_write(struct suscan_mq *mq, uint32_t type, void *msg);

/* ---- message types switched on in Suscan/Analyzer.cpp:75-98, :330-375 ----------------------- */
#define SUSCAN_ANALYZER_MESSAGE_TYPE_SOURCE_INFO 0x0
#define SUSCAN_ANALYZER_MESSAGE_TYPE_SOURCE_INIT 0x1
#define SUSCAN_ANALYZER_MESSAGE_TYPE_CHANNEL     0x2
#define SUSCAN_ANALYZER_MESSAGE_TYPE_EOS         0x3
#define SUSCAN_ANALYZER_MESSAGE_TYPE_READ_ERROR  0x4
#define SUSCAN_ANALYZER_MESSAGE_TYPE_INTERNAL    0x5
#define SUSCAN_ANALYZER_MESSAGE_TYPE_SAMPLES_LOST 0x6
#define SUSCAN_ANALYZER_MESSAGE_TYPE_INSPECTOR   0x7
#define SUSCAN_ANALYZER_MESSAGE_TYPE_PSD         0x8
#define SUSCAN_ANALYZER_MESSAGE_TYPE_SAMPLES     0x9
#define SUSCAN_ANALYZER_MESSAGE_TYPE_THROTTLE    0xa
#define SUSCAN_ANALYZER_MESSAGE_TYPE_PARAMS      0xb
#define SUSCAN_WORKER_MSG_TYPE_HALT              0xffffffffu

#define SUSCAN_ANALYZER_INIT_SUCCESS  0
#define SUSCAN_ANALYZER_INIT_PROGRESS 1
#define SUSCAN_ANALYZER_INIT_FAILURE  -1

/* ---- analyzer parameters (Suscan/AnalyzerParams.cpp:27-71) ----------------------------------- */
enum suscan_analyzer_mode { SUSCAN_ANALYZER_MODE_CHANNEL = 0, SUSCAN_ANALYZER_MODE_WIDE_SPECTRUM = 1 };
/* include/Suscan/AnalyzerParams.h:37-43; same values as enum suamd_window */
enum sigutils_channel_detector_window {
  SU_CHANNEL_DETECTOR_WINDOW_NONE = 0, SU_CHANNEL_DETECTOR_WINDOW_HAMMING = 1, SU_CHANNEL_DETECTOR_WINDOW_HANN = 2,
  SU_CHANNEL_DETECTOR_WINDOW_FLAT_TOP = 3, SU_CHANNEL_DETECTOR_WINDOW_BLACKMANN_HARRIS = 4
};
struct sigutils_channel_detector_params {
  SUFLOAT alpha, beta, gamma, snr;           /* spectrum / signal-level / noise-level averaging, SNR threshold */
  SUSCOUNT window_size;
  enum sigutils_channel_detector_window window;
};
struct suscan_analyzer_params {
  enum suscan_analyzer_mode mode;
  struct sigutils_channel_detector_params detector_params;
  SUFLOAT channel_update_int;
  SUFLOAT psd_update_int;
  SUFREQ  min_freq, max_freq;
};
#define suscan_analyzer_params_INITIALIZER \
  { SUSCAN_ANALYZER_MODE_CHANNEL, { 1e-2f, 1e-3f, .5f, 2.f, 4096, SU_CHANNEL_DETECTOR_WINDOW_BLACKMANN_HARRIS }, .1f, .04f, -1, -1 }

/* struct sigutils_channel (Suscan/Analyzer.cpp:417-424) */
struct sigutils_channel { SUFREQ fc, f_lo, f_hi; SUFLOAT bw, snr, S0, N0; SUFREQ ft; uint32_t age, present; };
#define sigutils_channel_INITIALIZER { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }

/* ---- source configuration: the subset needed to stand up a file / tone-generator source ------ */
typedef struct suscan_source_config suscan_source_config_t;
/* Default/SourceConfig/FileSourcePage.cpp:80-104; AUTO resolves by file extension.  Compact
 * formats cross PCIe as they are and are expanded on the GPU (suamd_ingest_iq) */
enum suscan_source_format {
  SUSCAN_SOURCE_FORMAT_AUTO = 0, SUSCAN_SOURCE_FORMAT_RAW_FLOAT32 = 1, SUSCAN_SOURCE_FORMAT_RAW_UNSIGNED8 = 2,
  SUSCAN_SOURCE_FORMAT_RAW_SIGNED8 = 3, SUSCAN_SOURCE_FORMAT_RAW_SIGNED16 = 4, SUSCAN_SOURCE_FORMAT_WAV = 5,
  SUSCAN_SOURCE_FORMAT_SIGMF = 6
};
SUAMD_API suscan_source_config_t *suscan_source_config_new(const char *type, enum suscan_source_format fmt);
SUAMD_API void   suscan_source_config_destroy(suscan_source_config_t *cfg);
SUAMD_API void   suscan_source_config_set_samp_rate(suscan_source_config_t *cfg, unsigned int rate);
SUAMD_API void   suscan_source_config_set_freq(suscan_source_config_t *cfg, SUFREQ freq);
SUAMD_API SUBOOL suscan_source_config_set_path(suscan_source_config_t *cfg, const char *path);
SUAMD_API void   suscan_source_config_set_loop(suscan_source_config_t *cfg, SUBOOL loop);
SUAMD_API SUBOOL suscan_source_config_set_param(suscan_source_config_t *cfg, const char *key, const char *val);
/* the rest of what Suscan::Source::Config calls on a source config (Suscan/Source.cpp:27-645).  A file / generator
 * source keeps every value; what it cannot act on (LNB, gains, antenna, ppm ...) is recorded and read back.  Device
 * specs, XML (de)serialisation and metadata guessing belong to libsuscan's control plane and are not served. */
SUAMD_API suscan_source_config_t *suscan_source_config_clone(const suscan_source_config_t *cfg);
SUAMD_API const char *suscan_source_config_get_label(const suscan_source_config_t *cfg);
SUAMD_API SUBOOL suscan_source_config_set_label(suscan_source_config_t *cfg, const char *label);
SUAMD_API const char *suscan_source_config_get_type(const suscan_source_config_t *cfg);
SUAMD_API enum suscan_source_format suscan_source_config_get_format(const suscan_source_config_t *cfg);
SUAMD_API void   suscan_source_config_set_type_format(suscan_source_config_t *cfg, const char *type, enum suscan_source_format fmt);
SUAMD_API const char *suscan_source_config_get_path(const suscan_source_config_t *cfg);      /* NULL: none */
SUAMD_API SUFREQ suscan_source_config_get_freq(const suscan_source_config_t *cfg);
SUAMD_API SUFREQ suscan_source_config_get_lnb_freq(const suscan_source_config_t *cfg);
SUAMD_API void   suscan_source_config_set_lnb_freq(suscan_source_config_t *cfg, SUFREQ lnb);
SUAMD_API unsigned int suscan_source_config_get_samp_rate(const suscan_source_config_t *cfg);
SUAMD_API unsigned int suscan_source_config_get_average(const suscan_source_config_t *cfg);  /* decimation, >= 1 */
SUAMD_API SUBOOL suscan_source_config_set_average(suscan_source_config_t *cfg, unsigned int average);
SUAMD_API SUFLOAT suscan_source_config_get_bandwidth(const suscan_source_config_t *cfg);
SUAMD_API void   suscan_source_config_set_bandwidth(suscan_source_config_t *cfg, SUFLOAT bw);
SUAMD_API SUFLOAT suscan_source_config_get_ppm(const suscan_source_config_t *cfg);
SUAMD_API void   suscan_source_config_set_ppm(suscan_source_config_t *cfg, SUFLOAT ppm);
SUAMD_API SUBOOL suscan_source_config_get_loop(const suscan_source_config_t *cfg);
SUAMD_API SUBOOL suscan_source_config_get_dc_remove(const suscan_source_config_t *cfg);
SUAMD_API void   suscan_source_config_set_dc_remove(suscan_source_config_t *cfg, SUBOOL dc_remove);
SUAMD_API SUBOOL suscan_source_config_get_iq_balance(const suscan_source_config_t *cfg);
SUAMD_API void   suscan_source_config_set_iq_balance(suscan_source_config_t *cfg, SUBOOL iq_balance);
SUAMD_API void   suscan_source_config_get_start_time(const suscan_source_config_t *cfg, struct timeval *tv);
SUAMD_API void   suscan_source_config_set_start_time(suscan_source_config_t *cfg, struct timeval tv);
/* start time + file length / sample rate; SU_FALSE when the length is unknown (generator, unreadable file) */
SUAMD_API SUBOOL suscan_source_config_get_end_time(const suscan_source_config_t *cfg, struct timeval *tv);
SUAMD_API SUBOOL suscan_source_config_file_is_valid(const suscan_source_config_t *cfg);
SUAMD_API SUBOOL suscan_source_config_is_real_time(const suscan_source_config_t *cfg);       /* never: file / generator */
SUAMD_API SUBOOL suscan_source_config_is_seekable(const suscan_source_config_t *cfg);
SUAMD_API SUBOOL suscan_source_config_get_freq_limits(const suscan_source_config_t *cfg, SUFREQ *min, SUFREQ *max);
SUAMD_API const char *suscan_source_config_get_antenna(const suscan_source_config_t *cfg);   /* NULL: none */
SUAMD_API SUBOOL suscan_source_config_set_antenna(suscan_source_config_t *cfg, const char *antenna);
SUAMD_API SUFLOAT suscan_source_config_get_gain(const suscan_source_config_t *cfg, const char *name);
SUAMD_API SUBOOL suscan_source_config_set_gain(suscan_source_config_t *cfg, const char *name, SUFLOAT value);
SUAMD_API const char *suscan_source_config_get_param(const suscan_source_config_t *cfg, const char *key);   /* NULL: unset */
SUAMD_API void   suscan_source_config_clear_params(suscan_source_config_t *cfg);
SUAMD_API SUBOOL suscan_source_config_walk_params(const suscan_source_config_t *cfg,
                     SUBOOL (*func)(const suscan_source_config_t *cfg, const char *key, const char *value, void *userdata),
                     void *userdata);

/* fields read by Suscan::AnalyzerSourceInfo (include/Suscan/Analyzer.h:113-254).  A SOURCE_INFO message and the
 * loaned pointer of suscan_analyzer_get_source_info() own their strings / lists; copies are deep
 * (suscan_source_info_init_copy) */
struct suscan_source_gain_info { char *name; SUFLOAT min, max, step, value; };
struct suscan_source_info {
  uint64_t permissions;
  SUSCOUNT source_samp_rate, effective_samp_rate;
  SUFLOAT  measured_samp_rate;
  SUFREQ   frequency, freq_min, freq_max, lnb;
  SUFLOAT  bandwidth, ppm;
  char    *antenna;                          /* NULL: none selected ("N/A", Analyzer.h:173-178) */
  SUBOOL   dc_remove, iq_reverse, agc, seekable;
  SUBOOL   replay;
  SUSCOUNT history_length;
  struct timeval source_start, source_end;
  struct suscan_source_gain_info **gain_list; unsigned int gain_count;      /* a file source has no gains ... */
  char   **antenna_list;                      unsigned int antenna_count;   /* ... and no antennas            */
};
SUAMD_API void   suscan_source_info_init(struct suscan_source_info *info);
SUAMD_API SUBOOL suscan_source_info_init_copy(struct suscan_source_info *dst, const struct suscan_source_info *src);
SUAMD_API void   suscan_source_info_finalize(struct suscan_source_info *info);

/* ---- inspector configuration (Suscan/Config.cpp; key vocabulary: Default/GenericInspector/InspectorCtl) */
enum suscan_field_type { SUSCAN_FIELD_TYPE_STRING, SUSCAN_FIELD_TYPE_INTEGER, SUSCAN_FIELD_TYPE_FLOAT,
                         SUSCAN_FIELD_TYPE_FILE, SUSCAN_FIELD_TYPE_BOOLEAN };
struct suscan_field { enum suscan_field_type type; SUBOOL optional; char *name; char *desc; };
struct suscan_field_value {
  SUBOOL set;
  const struct suscan_field *field;
  union { uint64_t as_int; SUFLOAT as_float; SUBOOL as_bool; };
  char *as_string;
};
typedef struct suscan_config_desc { char *global_name; struct suscan_field **field_list; unsigned field_count; }
  suscan_config_desc_t;
typedef struct suscan_config { const suscan_config_desc_t *desc; struct suscan_field_value **values; }
  suscan_config_t;
SUAMD_API const suscan_config_desc_t *suscan_inspector_config_desc(const char *class_name);  /* "psk", "fsk", "ask", "raw", "power" */
SUAMD_API suscan_config_t *suscan_config_new(const suscan_config_desc_t *desc);
SUAMD_API suscan_config_t *suscan_config_dup(const suscan_config_t *cfg);
SUAMD_API void   suscan_config_destroy(suscan_config_t *cfg);
SUAMD_API struct suscan_field_value *suscan_config_get_value(const suscan_config_t *cfg, const char *name);
SUAMD_API SUBOOL suscan_config_set_integer(suscan_config_t *cfg, const char *name, uint64_t v);
SUAMD_API SUBOOL suscan_config_set_float(suscan_config_t *cfg, const char *name, SUFLOAT v);
SUAMD_API SUBOOL suscan_config_set_bool(suscan_config_t *cfg, const char *name, SUBOOL v);
SUAMD_API SUBOOL suscan_config_set_string(suscan_config_t *cfg, const char *name, const char *v);
SUAMD_API SUBOOL suscan_config_desc_has_prefix(const suscan_config_desc_t *desc, const char *prefix);

/* ---- messages (fields SigDigger dereferences: SURVEY.md Appendix B) --------------------------- */
struct suscan_analyzer_psd_msg {
  int64_t  fc;
  uint32_t inspector_id;
  struct timeval timestamp, rt_time;
  SUBOOL   looped;
  SUSCOUNT history_size;
  SUFLOAT  samp_rate, measured_samp_rate, N0;
  SUSCOUNT psd_size;
  SUFLOAT *psd_data;                          /* linear power, natural FFT order (DC at 0) */
};
struct suscan_analyzer_sample_batch_msg {
  uint32_t inspector_id;
  suamd_complex *samples;                     /* SUCOMPLEX */
  SUSCOUNT sample_count;
};
enum suscan_analyzer_inspector_msgkind {
  SUSCAN_ANALYZER_INSPECTOR_MSGKIND_OPEN, SUSCAN_ANALYZER_INSPECTOR_MSGKIND_SET_ID,
  SUSCAN_ANALYZER_INSPECTOR_MSGKIND_GET_CONFIG, SUSCAN_ANALYZER_INSPECTOR_MSGKIND_SET_CONFIG,
  SUSCAN_ANALYZER_INSPECTOR_MSGKIND_ESTIMATOR, SUSCAN_ANALYZER_INSPECTOR_MSGKIND_SPECTRUM,
  SUSCAN_ANALYZER_INSPECTOR_MSGKIND_RESET_EQUALIZER, SUSCAN_ANALYZER_INSPECTOR_MSGKIND_CLOSE,
  SUSCAN_ANALYZER_INSPECTOR_MSGKIND_SET_FREQ, SUSCAN_ANALYZER_INSPECTOR_MSGKIND_SET_BANDWIDTH,
  SUSCAN_ANALYZER_INSPECTOR_MSGKIND_SET_WATERMARK, SUSCAN_ANALYZER_INSPECTOR_MSGKIND_WRONG_HANDLE,
  SUSCAN_ANALYZER_INSPECTOR_MSGKIND_WRONG_OBJECT, SUSCAN_ANALYZER_INSPECTOR_MSGKIND_INVALID_ARGUMENT,
  SUSCAN_ANALYZER_INSPECTOR_MSGKIND_WRONG_KIND, SUSCAN_ANALYZER_INSPECTOR_MSGKIND_INVALID_CHANNEL,
  SUSCAN_ANALYZER_INSPECTOR_MSGKIND_ORBIT_REPORT, SUSCAN_ANALYZER_INSPECTOR_MSGKIND_SET_TLE,
  SUSCAN_ANALYZER_INSPECTOR_MSGKIND_SIGNAL
};
/* sgdp4's xyz_t and the report of kind ORBIT_REPORT (include/Suscan/Messages/InspectorMessage.h:33-78) */
#ifndef SUAMD_XYZ_T_DEFINED
#  define SUAMD_XYZ_T_DEFINED
typedef struct xyz {
  union { double x, lon, azimuth; };
  union { double y, lat, elevation; };
  union { double z, height, distance; };
} xyz_t;
#endif
struct suscan_orbit_report { struct timeval rx_time; xyz_t satpos; SUFLOAT freq_corr; double vlos_vel; };
/* su_channel_detector's result at channel_update_int (Suscan/Analyzer.cpp:75-98 disposes it, ChannelMessage.cpp:26) */
struct suscan_analyzer_channel_msg { struct sigutils_channel **channel_list; unsigned int channel_count; };
struct suscan_analyzer_inspector_msg {
  enum suscan_analyzer_inspector_msgkind kind;
  uint32_t inspector_id;
  uint32_t req_id;
  SUHANDLE handle;
  int      status;
  char    *class_name;
  struct sigutils_channel channel;
  suscan_config_t *config;                    /* borrowed by readers: they suscan_config_dup it */
  SUBOOL   precise;
  uint32_t fs;
  SUFLOAT  equiv_fs, bandwidth, lo;
  unsigned spectsrc_count; char **spectsrc_list;
  unsigned estimator_count; char **estimator_list;
  uint32_t spectsrc_id, estimator_id;
  SUFLOAT *spectrum_data; SUSCOUNT spectrum_size; SUSCOUNT samp_rate;
  SUSCOUNT watermark;
  SUBOOL   enabled;                           /* ESTIMATOR: estimator switched on; SET_TLE: correction enabled */
  SUFLOAT  value;                             /* ESTIMATOR: the estimate, in the unit of the config field it feeds (Hz / baud) */
  struct suscan_orbit_report orbit_report;    /* ORBIT_REPORT (InspectorMessage.cpp:65) */
  SUBOOL   tle_enable;                        /* SET_TLE (InspectorMessage.cpp:230) */
  char    *signal_name;                       /* SIGNAL (InspectorMessage.cpp:239): never NULL in a delivered message */
  double   signal_value;                      /* SIGNAL (InspectorMessage.cpp:248) */
};
struct suscan_analyzer_status_msg { int code; char *err_msg; };

/* ---- analyzer (Suscan/Analyzer.cpp:111-639) --------------------------------------------------- */
typedef struct suscan_analyzer suscan_analyzer_t;
SUAMD_API suscan_analyzer_t *suscan_analyzer_new(const struct suscan_analyzer_params *params,
                                                 suscan_source_config_t *config, struct suscan_mq *mq);
SUAMD_API void   suscan_analyzer_destroy(suscan_analyzer_t *analyzer);
SUAMD_API void  *suscan_analyzer_read(suscan_analyzer_t *analyzer, uint32_t *type);
SUAMD_API void   suscan_analyzer_dispose_message(uint32_t type, void *ptr);
SUAMD_API void   suscan_analyzer_req_halt(suscan_analyzer_t *analyzer);
SUAMD_API SUBOOL suscan_analyzer_set_params_async(suscan_analyzer_t *analyzer,
                                                  const struct suscan_analyzer_params *params, uint32_t req_id);
SUAMD_API SUBOOL suscan_analyzer_set_throttle_async(suscan_analyzer_t *analyzer, SUSCOUNT samp_rate, uint32_t req_id);
SUAMD_API unsigned int suscan_analyzer_get_samp_rate(const suscan_analyzer_t *analyzer);
SUAMD_API SUFLOAT suscan_analyzer_get_measured_samp_rate(const suscan_analyzer_t *analyzer);
SUAMD_API struct suscan_source_info *suscan_analyzer_get_source_info(const suscan_analyzer_t *analyzer);  /* loaned */
SUAMD_API SUBOOL suscan_analyzer_open_async(suscan_analyzer_t *analyzer, const char *class_name,
                                            const struct sigutils_channel *channel, uint32_t req_id);
SUAMD_API SUBOOL suscan_analyzer_open_ex_async(suscan_analyzer_t *analyzer, const char *class_name,
                                               const struct sigutils_channel *channel, SUBOOL precise,
                                               SUHANDLE parent, uint32_t req_id);
SUAMD_API SUBOOL suscan_analyzer_close_async(suscan_analyzer_t *analyzer, SUHANDLE handle, uint32_t req_id);
SUAMD_API SUBOOL suscan_analyzer_set_inspector_id_async(suscan_analyzer_t *analyzer, SUHANDLE handle,
                                                        uint32_t inspector_id, uint32_t req_id);
SUAMD_API SUBOOL suscan_analyzer_set_inspector_config_async(suscan_analyzer_t *analyzer, SUHANDLE handle,
                                                            const suscan_config_t *config, uint32_t req_id);
SUAMD_API SUBOOL suscan_analyzer_set_inspector_watermark_async(suscan_analyzer_t *analyzer, SUHANDLE handle,
                                                               SUSCOUNT watermark, uint32_t req_id);
/* InspectorMessage's view of the source / estimator registries (Suscan/Messages/InspectorMessage.cpp:44-61:
 * it reads ->desc of the spectrum source class and ->desc / ->field of the estimator class; NULL = unknown
 * name, then the name itself is shown). */
struct suscan_spectsrc_class  { const char *name; const char *desc; };
struct suscan_estimator_class { const char *name; const char *desc; const char *field; };
SUAMD_API const struct suscan_spectsrc_class  *suscan_spectsrc_class_lookup(const char *name);
SUAMD_API const struct suscan_estimator_class *suscan_estimator_class_lookup(const char *name);

/* Analyzer::setSpectrumSource (Suscan/Analyzer.cpp:539-547): 0 = none, k = spectsrc_list[k-1] of the OPEN
 * message; afterwards every block yields an INSPECTOR message of kind SPECTRUM with spectrum_data
 * (linear power, natural order), spectrum_size, samp_rate = equiv_fs and spectsrc_id */
SUAMD_API SUBOOL suscan_analyzer_inspector_set_spectrum_async(suscan_analyzer_t *analyzer, SUHANDLE handle,
                                                              uint32_t spectsrc_id, uint32_t req_id);
SUAMD_API SUBOOL suscan_analyzer_set_inspector_freq_overridable(suscan_analyzer_t *analyzer, SUHANDLE handle,
                                                                SUFREQ freq);
SUAMD_API SUBOOL suscan_analyzer_set_inspector_bandwidth_overridable(suscan_analyzer_t *analyzer, SUHANDLE handle,
                                                                     SUFREQ bw);
/* Analyzer::setInspectorEnabled (Suscan/Analyzer.cpp:549-565): estimator_id indexes estimator_list of the OPEN
 * message; while enabled, INSPECTOR messages of kind ESTIMATOR carry {estimator_id, enabled, value} (value 0 = no
 * estimate yet, which is how InspectorUI::updateEstimator reads it, InspectorUI.cpp:1003-1015) */
SUAMD_API SUBOOL suscan_analyzer_inspector_estimator_cmd_async(suscan_analyzer_t *analyzer, SUHANDLE handle,
                                                               uint32_t estimator_id, SUBOOL enabled, uint32_t req_id);
/* Analyzer::setInspectorDopplerCorrection / disableDopplerCorrection (Suscan/Analyzer.cpp:567-591).  Orbit propagation
 * is outside this path: tle == NULL (disable) is acknowledged with a SET_TLE message, anything else with
 * INVALID_ARGUMENT */
typedef struct orbit orbit_t;
SUAMD_API SUBOOL suscan_analyzer_inspector_set_tle_async(suscan_analyzer_t *analyzer, SUHANDLE handle,
                                                         const orbit_t *tle, uint32_t req_id);

/* ---- baseband filters (Suscan/Analyzer.cpp:127-142; Default/Source/SourceWidget.cpp:1155-1171) -------------
 * Called on the analyzer's worker thread with every block of SUCOMPLEX samples before anything else sees it
 * (lower prio first; plain registration = prio 0 in call order); what they write is what the PSD and the
 * inspectors get; returning SU_FALSE stops the analyzer with a READ_ERROR.  `offset` = samples delivered so far. */
typedef SUBOOL (*suscan_analyzer_baseband_filter_func_t)(void *privdata, suscan_analyzer_t *analyzer,
                                                         suamd_complex *samples, SUSCOUNT length, SUSCOUNT offset);
SUAMD_API SUBOOL suscan_analyzer_register_baseband_filter(suscan_analyzer_t *analyzer,
                                                          suscan_analyzer_baseband_filter_func_t func, void *privdata);
SUAMD_API SUBOOL suscan_analyzer_register_baseband_filter_with_prio(suscan_analyzer_t *analyzer,
                                                                    suscan_analyzer_baseband_filter_func_t func,
                                                                    void *privdata, int64_t prio);

/* ---- source controls (Suscan/Analyzer.cpp:144-281).  A file / generator source has no tuner: frequency, LNB,
 * bandwidth, ppm, gain, antenna and AGC settings are recorded and reflected in the source info (a SOURCE_INFO
 * message follows every change); I/Q reversal and DC removal act on the samples (suamd_source_fix); seek moves the
 * file position; history / replay are accepted (the file is its own history). */
SUAMD_API SUBOOL suscan_analyzer_set_freq(suscan_analyzer_t *analyzer, SUFREQ freq, SUFREQ lnb);
SUAMD_API SUBOOL suscan_analyzer_set_gain(suscan_analyzer_t *analyzer, const char *name, SUFLOAT value);
SUAMD_API SUBOOL suscan_analyzer_set_antenna(suscan_analyzer_t *analyzer, const char *name);
SUAMD_API SUBOOL suscan_analyzer_set_bw(suscan_analyzer_t *analyzer, SUFLOAT bw);
SUAMD_API SUBOOL suscan_analyzer_set_ppm(suscan_analyzer_t *analyzer, SUFLOAT ppm);
SUAMD_API SUBOOL suscan_analyzer_set_agc(suscan_analyzer_t *analyzer, SUBOOL enabled);
SUAMD_API SUBOOL suscan_analyzer_set_dc_remove(suscan_analyzer_t *analyzer, SUBOOL remove);
SUAMD_API SUBOOL suscan_analyzer_set_iq_reverse(suscan_analyzer_t *analyzer, SUBOOL reverse);
SUAMD_API SUBOOL suscan_analyzer_seek(suscan_analyzer_t *analyzer, const struct timeval *pos);
SUAMD_API SUBOOL suscan_analyzer_set_history_size(suscan_analyzer_t *analyzer, SUSCOUNT size);
SUAMD_API SUBOOL suscan_analyzer_replay(suscan_analyzer_t *analyzer, SUBOOL replay);
SUAMD_API void   suscan_analyzer_get_source_time(const suscan_analyzer_t *analyzer, struct timeval *tv);

/* ---- wide-spectrum (panoramic) controls (Suscan/Analyzer.cpp:177-195, 258-281; Panoramic/Scanner.cpp:295-370).
 * Valid in SUSCAN_ANALYZER_MODE_WIDE_SPECTRUM only (SU_FALSE otherwise).  A file source has no tuner: the capture is the
 * sweep -- every block is one dwell -- and the PSD frames are labelled with the frequencies the sweep strategy visits
 * (SPEC.md section P): step = rel_bandwidth * samp_rate over [hop min, hop max] (the analyzer parameters' min_freq /
 * max_freq until set_hop_range is called); PROGRESSIVE walks the slots in order, STOCHASTIC draws them (DISCRETE: slot
 * centres, CONTINUOUS: anywhere); min == max is the Scanner's noHop case.  Scanner::onPSDMessage (Panoramic/Scanner.cpp:
 * 503-523) feeds each frame to its SpectrumView at that frequency. */
enum suscan_analyzer_sweep_strategy { SUSCAN_ANALYZER_SWEEP_STRATEGY_STOCHASTIC = 0, SUSCAN_ANALYZER_SWEEP_STRATEGY_PROGRESSIVE = 1 };
enum suscan_analyzer_spectrum_partitioning { SUSCAN_ANALYZER_SPECTRUM_PARTITIONING_DISCRETE = 0, SUSCAN_ANALYZER_SPECTRUM_PARTITIONING_CONTINUOUS = 1 };
SUAMD_API SUBOOL suscan_analyzer_set_sweep_stratrgy(suscan_analyzer_t *analyzer, enum suscan_analyzer_sweep_strategy strategy);  /* sic */
SUAMD_API SUBOOL suscan_analyzer_set_spectrum_partitioning(suscan_analyzer_t *analyzer, enum suscan_analyzer_spectrum_partitioning p);
SUAMD_API SUBOOL suscan_analyzer_set_hop_range(suscan_analyzer_t *analyzer, SUFREQ min, SUFREQ max);
SUAMD_API SUBOOL suscan_analyzer_set_rel_bandwidth(suscan_analyzer_t *analyzer, SUFLOAT rel_bw);
SUAMD_API SUBOOL suscan_analyzer_set_buffering_size(suscan_analyzer_t *analyzer, SUSCOUNT size);

#ifdef __cplusplus
}
#endif
#endif /* SUSCAN_AMD_H */
