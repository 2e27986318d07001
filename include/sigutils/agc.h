/*
 * include/sigutils/agc.h -- <sigutils/agc.h> (include/AGCTask.h:23): su_agc_t behind Tasks/AGCTask.cpp:41-53,70-73,101,
 * served by libsigdigger_amd.so (csrc/sigutils_host.cpp): host code, one sample per call, state by value
 * (include/AGCTask.h:39; no heap: su_agc_finalize has nothing to free).  SPEC.md section H; the block form on the GPU is
 * suamd_agc_bank_*.
 */
#ifndef SIGDIGGER_AMD_SIGUTILS_AGC_H
#define SIGDIGGER_AMD_SIGUTILS_AGC_H
#include "types.h"
#ifdef __cplusplus
extern "C" {
#endif

struct su_agc_params {
  SUFLOAT threshold, slope_factor;
  unsigned int hang_max, delay_line_size, mag_history_size;      /* sizes 1 .. 64 */
  SUFLOAT fast_rise_t, fast_fall_t, slow_rise_t, slow_fall_t;    /* time constants in samples */
};
#define su_agc_params_INITIALIZER { -100, 6, 100, 20, 20, 2, 4, 20, 40 }

#define SU_AGC_MAX_HISTORY 64
typedef struct sigutils_agc {
  SUFLOAT knee, gain_slope;
  SUFLOAT fast_alpha_rise, fast_alpha_fall, slow_alpha_rise, slow_alpha_fall;
  unsigned int hang_max, hang_n, delay_line_size, mag_history_size, delay_ptr, hist_ptr;
  SUFLOAT fast_level, slow_level;
  SUFLOAT delay_line[SU_AGC_MAX_HISTORY][2];
  SUFLOAT mag_history[SU_AGC_MAX_HISTORY];
} su_agc_t;
#define su_agc_INITIALIZER { 0 }

SUAMD_API SUBOOL    su_agc_init(su_agc_t *agc, const struct su_agc_params *params);
SUAMD_API SUCOMPLEX su_agc_feed(su_agc_t *agc, SUCOMPLEX x);
SUAMD_API void      su_agc_finalize(su_agc_t *agc);

#ifdef __cplusplus
}
#endif
#endif
