/* refshim: <sigutils/agc.h> (absent): su_agc_t, served by oracle/ref_glue.cpp over oracle/sdo.c */
#ifndef REFSHIM_SIGUTILS_AGC_H
#define REFSHIM_SIGUTILS_AGC_H
#include <sigutils/types.h>
#include <sdo.h>
#ifdef __cplusplus
extern "C" {
#endif
struct su_agc_params {
  SUFLOAT threshold, slope_factor;
  unsigned int hang_max, delay_line_size, mag_history_size;
  SUFLOAT fast_rise_t, fast_fall_t, slow_rise_t, slow_fall_t;
};
/* = sdo_agc_params_default (SPEC.md section I) */
#define su_agc_params_INITIALIZER { -100, 6, 100, 20, 20, 2, 4, 20, 40 }
typedef struct sigutils_agc { sdo_agc impl; } su_agc_t;
#define su_agc_INITIALIZER { }
SUBOOL su_agc_init(su_agc_t *agc, const struct su_agc_params *params);
SUCOMPLEX su_agc_feed(su_agc_t *agc, SUCOMPLEX x);
void su_agc_finalize(su_agc_t *agc);
#ifdef __cplusplus
}
#endif
#endif
