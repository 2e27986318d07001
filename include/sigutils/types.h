/*
 * include/sigutils/types.h -- the <sigutils/types.h> SigDigger includes (include/Suscan/Compat.h:25,
 * Misc/Averager.cpp:21, include/QuadDemodTask.h:23 ...), as served by libsigdigger_amd.so.
 *
 * Scalar types, the SUCOMPLEX sample type and the arithmetic macros the reference's own loops are written in
 * (SURVEY.md Appendix A "Macros used").  libsigutils is absent from /root/reference, so the macro bodies follow
 * SURVEY.md Appendix C / section 8c (what the reference's usage proves: SU_LOG is log10, normalised frequency is
 * 2f/fs, angular = pi * normalised, SU_SPLPF_FEED is y += alpha (x - y)); SPEC.md section A freezes the epsilon of
 * SU_POWER_DB.  Included before sigdigger_amd.h it makes suamd_complex the same type as SUCOMPLEX, so the
 * reference's wrappers (SamplesMessage.h:44, SourceWidget.cpp:1155) see the pointer types they expect.
 */
#ifndef SIGDIGGER_AMD_SIGUTILS_TYPES_H
#define SIGDIGGER_AMD_SIGUTILS_TYPES_H

#include <stdint.h>
#include <stddef.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
#  include <complex>
#  define SUCOMPLEX std::complex<float>
#  define SU_C_REAL(c)  ((c).real())
#  define SU_C_IMAG(c)  ((c).imag())
#  define SU_C_ABS(c)   std::abs(c)
#  define SU_C_ARG(c)   std::arg(c)
#  define SU_C_EXP(c)   std::exp(c)
#  define SU_C_CONJ(c)  std::conj(c)
#  define SU_I          std::complex<float>(0, 1)
#else
#  include <complex.h>
#  define SUCOMPLEX float _Complex
#  define SU_C_REAL(c)  crealf(c)
#  define SU_C_IMAG(c)  cimagf(c)
#  define SU_C_ABS(c)   cabsf(c)
#  define SU_C_ARG(c)   cargf(c)
#  define SU_C_EXP(c)   cexpf(c)
#  define SU_C_CONJ(c)  conjf(c)
#  define SU_I          I
#endif
#define SUAMD_COMPLEX_IS_SUCOMPLEX 1

#include "../sigdigger_amd.h"   /* SUFLOAT, SUFREQ, SUSCOUNT, SUSDIFF, SUBOOL, SU_TRUE / SU_FALSE */

typedef double   SUDOUBLE;
typedef uint32_t SUBITS;
#ifndef SUHANDLE_DEFINED
#  define SUHANDLE_DEFINED
typedef int32_t  SUHANDLE;
#endif

#ifndef PI
#  define PI 3.141592653589793238462643
#endif
#define SU_ASFLOAT(x)   ((SUFLOAT) (x))
#define SU_ADDSFX(x)    x##f
#define SU_SQRT(x)      sqrtf(x)
#define SU_FLOOR(x)     floorf(x)
#define SU_CEIL(x)      ceilf(x)
#define SU_ROUND(x)     roundf(x)
#define SU_ABS(x)       fabsf(x)
#define SU_LOG(x)       log10f(x)
#define SU_LN(x)        logf(x)
#define SU_EXP(x)       expf(x)
#define SU_POW(x, y)    powf(x, y)
#define SU_COS(x)       cosf(x)
#define SU_SIN(x)       sinf(x)
#define SU_MIN(a, b)    ((a) < (b) ? (a) : (b))
#define SU_MAX(a, b)    ((a) > (b) ? (a) : (b))
#define SU_SGN(x)       ((x) < 0 ? -1 : ((x) > 0 ? 1 : 0))

/* power / magnitude <-> dB (Suscan/Messages/PSDMessage.cpp:36; GenericInspector.cpp:237; Appendix C) */
#define SU_POWER_DB_RAW(p)  (10 * SU_LOG(p))
#define SU_POWER_DB(p)      SU_POWER_DB_RAW((p) + SU_ASFLOAT(1e-8))
#define SU_DB_RAW(p)        (20 * SU_LOG(p))
#define SU_DB(p)            SU_DB_RAW((p) + SU_ASFLOAT(1e-8))
#define SU_POWER_MAG_RAW(d) SU_POW(10.f, SU_ASFLOAT(.1) * (d))
#define SU_MAG_RAW(d)       SU_POW(10.f, SU_ASFLOAT(.05) * (d))

/* single-pole low pass (Misc/Averager.cpp:46 analogue; UIMediator/SpectrumMediator.cpp:66-74) */
#define SU_SPLPF_ALPHA(tau)      (1.f - SU_EXP(-1.f / (tau)))
#define SU_SPLPF_FEED(y, x, a)   y += (a) * ((x) - (y))

/* frequency conventions (Components/TimeWindow.cpp:1556,1568,1667-1669; Tasks/LPFTask.cpp:64;
 * Tasks/WaveSampler.cpp:48) */
#define SU_ABS2NORM_FREQ(fs, f)  (2 * SU_ASFLOAT(f) / SU_ASFLOAT(fs))
#define SU_NORM2ABS_FREQ(fs, f)  (SU_ASFLOAT(fs) * SU_ASFLOAT(f) * SU_ASFLOAT(.5))
#define SU_NORM2ANG_FREQ(f)      (PI * SU_ASFLOAT(f))
#define SU_ANG2NORM_FREQ(w)      (SU_ASFLOAT(w) / PI)
#define SU_ABS2NORM_BAUD(fs, b)  (SU_ASFLOAT(b) / SU_ASFLOAT(fs))
#define SU_NORM2ABS_BAUD(fs, b)  (SU_ASFLOAT(b) * SU_ASFLOAT(fs))
#define SU_DEG2RAD(d)            ((d) * (PI / 180.))
#define SU_RAD2DEG(r)            ((r) * (180. / PI))

/* FFTW3f through SU_FFTW(_plan_dft_1d) ... (Tasks/CarrierDetector.cpp:43-94, include/CarrierDetector.h:38-39) */
#if defined(__has_include)
#  if __has_include(<fftw3.h>)
#    include <fftw3.h>
#  endif
#endif
#define SU_FFTW(method) fftwf ## method

#ifndef STRINGIFY
#  define _SU_STRINGIFY(x) #x
#  define STRINGIFY(x) _SU_STRINGIFY(x)
#endif

/* log + error-path helpers used by the reference's C++ (Tasks/WaveSampler.cpp:60-65, Suscan/Logger.cpp) */
#ifndef SU_LOG_DOMAIN
#  define SU_LOG_DOMAIN __FILE__
#endif
#define SU_ERROR(...)   (fprintf(stderr, "(e) " __VA_ARGS__))
#define SU_WARNING(...) (fprintf(stderr, "(!) " __VA_ARGS__))
#define SU_INFO(...)    (fprintf(stderr, "(i) " __VA_ARGS__))
#define SU_TRYCATCH(expr, action)                                              \
  if (!(expr)) {                                                               \
    SU_ERROR("exception in \"%s\" (%s:%d)\n", STRINGIFY(expr), __FILE__, __LINE__); \
    action;                                                                    \
  }

#endif /* SIGDIGGER_AMD_SIGUTILS_TYPES_H */
