/*
 * include/sigutils/taps.h -- <sigutils/taps.h> (Tasks/CarrierDetector.cpp:20,87-89, Tasks/DopplerCalculator.cpp:20,92):
 * the Blackman-Harris window applied in place to a complex buffer, served by libsigdigger_amd.so
 * (csrc/sigutils_host.cpp; host code -- the callers hold the buffer in host memory and hand it to FFTW next).  The
 * whole-capture GPU forms of those two tasks are suamd_carrier_detect / suamd_doppler_calc.
 */
#ifndef SIGDIGGER_AMD_SIGUTILS_TAPS_H
#define SIGDIGGER_AMD_SIGUTILS_TAPS_H
#include "types.h"
#ifdef __cplusplus
extern "C" {
#endif

/* h[i] *= (float)(0.35875 - 0.48829 cos t + 0.14128 cos 2t - 0.01168 cos 3t), t = 2 pi i / (size - 1) (binary64 window) */
SUAMD_API void su_taps_apply_blackmann_harris_complex(SUCOMPLEX *h, SUSCOUNT size);
SUAMD_API void su_taps_apply_blackmann_harris(SUFLOAT *h, SUSCOUNT size);

#ifdef __cplusplus
}
#endif
#endif
