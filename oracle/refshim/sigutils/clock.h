/* refshim: <sigutils/clock.h> (absent): su_clock_detector_t, served by oracle/ref_glue.cpp over oracle/sdo.c */
#ifndef REFSHIM_SIGUTILS_CLOCK_H
#define REFSHIM_SIGUTILS_CLOCK_H
#include <sigutils/types.h>
#include <sdo.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct sigutils_clock_detector { sdo_clock impl; SUCOMPLEX *buf; SUSCOUNT size, avail; } su_clock_detector_t;
#define su_clock_detector_INITIALIZER { }
SUBOOL su_clock_detector_init(su_clock_detector_t *cd, SUFLOAT loop_gain, SUFLOAT bhint, SUSCOUNT bufsiz);
void   su_clock_detector_feed(su_clock_detector_t *cd, SUCOMPLEX x);
SUSDIFF su_clock_detector_read(su_clock_detector_t *cd, SUCOMPLEX *buf, size_t size);
void   su_clock_detector_finalize(su_clock_detector_t *cd);
#ifdef __cplusplus
}
#endif
#endif
