/*
 * include/sigutils/specttuner.h -- <sigutils/specttuner.h> (include/LPFTask.h:23): the FFT channeliser behind
 * Tasks/LPFTask.cpp:52-69,83-87,104-107,123 and behind every inspector channel of the live analyzer, served by
 * libsigdigger_amd.so on the GPU (csrc/specttuner.hip; SPEC.md section C2).
 *
 * Same calls, same callback contract: su_specttuner_feed_bulk() consumes host samples; every time half a window of
 * new samples has arrived the 50 %-overlapped forward FFT of the window runs ONCE for all open channels, each channel
 * picks its bins, applies its response, goes back to the time domain at its decimated rate and cross-fades with the
 * previous window; on_data() is then called on the feeding thread with window_size / (2 * decimation) samples in
 * host memory that stay valid until the next feed (Tasks/LPFTask.cpp:32).  guard = 2 pi / bw gives decimation 1
 * (LPFTask.cpp:65 "ensures no decimation").
 */
#ifndef SIGDIGGER_AMD_SIGUTILS_SPECTTUNER_H
#define SIGDIGGER_AMD_SIGUTILS_SPECTTUNER_H

#include "types.h"

#ifdef __cplusplus
extern "C" {
#endif

struct sigutils_specttuner_params {
  SUSCOUNT window_size;                      /* 4096 (su_specttuner's default, the only size the kernels are built for):
                                              * su_specttuner_new returns NULL for anything else, suamd_last_error() says why */
  SUBOOL   early_windowing;                  /* accepted for source compatibility; the response is applied per channel */
};
#define sigutils_specttuner_params_INITIALIZER { 4096, SU_TRUE }

struct sigutils_specttuner_channel;
struct sigutils_specttuner_channel_params {
  SUFLOAT f0;                                /* centre, angular frequency (rad / sample); taken modulo 2 pi (must be finite) */
  SUFLOAT delta_f;                           /* accepted, unused */
  SUFLOAT bw;                                /* bandwidth, angular frequency, > 0 (bw * guard is clipped to 2 pi); else NULL */
  SUFLOAT guard;                             /* relative guard band, >= 1: the channel is sized for bw * guard */
  SUBOOL  precise;                           /* correct the residual of the centre-bin rounding with an NCO */
  void   *privdata;
  SUBOOL (*on_data)(const struct sigutils_specttuner_channel *channel, void *privdata,
                    const SUCOMPLEX *data, SUSCOUNT size);
};
#define sigutils_specttuner_channel_params_INITIALIZER { 0, 0, 0, 1, SU_FALSE, NULL, NULL }

typedef struct sigutils_specttuner         su_specttuner_t;
typedef struct sigutils_specttuner_channel su_specttuner_channel_t;

SUAMD_API su_specttuner_t *su_specttuner_new(const struct sigutils_specttuner_params *params);
SUAMD_API void   su_specttuner_destroy(su_specttuner_t *st);              /* closes every open channel */
SUAMD_API su_specttuner_channel_t *su_specttuner_open_channel(su_specttuner_t *st,
                                                              const struct sigutils_specttuner_channel_params *params);
SUAMD_API SUBOOL su_specttuner_close_channel(su_specttuner_t *st, su_specttuner_channel_t *channel);
/* consumes all `size` samples (the upstream call may stop at a window boundary and is looped by its own bulk
 * wrapper; here the loop is inside) */
SUAMD_API SUBOOL su_specttuner_feed_bulk(su_specttuner_t *st, const SUCOMPLEX *buf, SUSCOUNT size);
SUAMD_API SUFLOAT  su_specttuner_channel_get_decimation(const su_specttuner_channel_t *channel);
SUAMD_API SUFLOAT  su_specttuner_channel_get_bw(const su_specttuner_channel_t *channel);
SUAMD_API SUFLOAT  su_specttuner_channel_get_f0(const su_specttuner_channel_t *channel);
SUAMD_API unsigned su_specttuner_channel_get_size(const su_specttuner_channel_t *channel);   /* bins = IFFT size */

#ifdef __cplusplus
}
#endif
#endif
