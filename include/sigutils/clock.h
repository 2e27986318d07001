/*
 * include/sigutils/clock.h -- <sigutils/clock.h> (include/WaveSampler.h:25): the Gardner clock detector behind
 * Tasks/WaveSampler.cpp:60-65,88,192-205, served by libsigdigger_amd.so (csrc/sigutils_host.cpp): host code, one
 * sample per call, state by value (include/WaveSampler.h:44) plus the symbol stream su_clock_detector_read drains.
 * SPEC.md section G; the block form on the GPU is suamd_clock_bank_*.  su_clock_detector_init is compared against -1 by
 * the reference (Tasks/WaveSampler.cpp:60-65).
 */
#ifndef SIGDIGGER_AMD_SIGUTILS_CLOCK_H
#define SIGDIGGER_AMD_SIGUTILS_CLOCK_H
#include "types.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct sigutils_clock_detector {
  SUFLOAT alpha, beta, gain, phi, bnor, bmin, bmax;
  int     halfcycle;
  SUFLOAT prev[2], x0[2], x1[2], x2[2];
  SUCOMPLEX *buf;                               /* symbols not yet read: [0, avail) */
  SUSCOUNT size, avail;
} su_clock_detector_t;
#define su_clock_detector_INITIALIZER { 0, 0, 0, 0, 0, 0, 0, 0, {0}, {0}, {0}, {0}, NULL, 0, 0 }

SUAMD_API SUBOOL  su_clock_detector_init(su_clock_detector_t *cd, SUFLOAT loop_gain, SUFLOAT bhint, SUSCOUNT bufsiz);  /* -1 on failure */
SUAMD_API void    su_clock_detector_feed(su_clock_detector_t *cd, SUCOMPLEX x);
SUAMD_API SUSDIFF su_clock_detector_read(su_clock_detector_t *cd, SUCOMPLEX *buf, size_t size);
SUAMD_API void    su_clock_detector_set_baud(su_clock_detector_t *cd, SUFLOAT bnor);
SUAMD_API void    su_clock_detector_finalize(su_clock_detector_t *cd);

#ifdef __cplusplus
}
#endif
#endif
