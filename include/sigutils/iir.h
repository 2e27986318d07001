/*
 * include/sigutils/iir.h -- <sigutils/iir.h> (include/WaveSampler.h:26): su_iir_filt_t, the matched filter member of
 * WaveSampler (include/WaveSampler.h:50; its use is compiled out of the reference unless SIGDIGGER_WAVESAMPLER_USE_MF,
 * Tasks/WaveSampler.cpp:55-93).  Served by libsigdigger_amd.so (csrc/sigutils_host.cpp): a root-raised-cosine FIR with
 * the taps of SPEC.md section I ("matched filter"), one sample per call; the block form on the GPU is suamd_fir_bank_*.
 */
#ifndef SIGDIGGER_AMD_SIGUTILS_IIR_H
#define SIGDIGGER_AMD_SIGUTILS_IIR_H
#include "types.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct sigutils_iir_filt {
  SUSCOUNT  n;                                  /* taps */
  SUFLOAT  *h;                                  /* [n] */
  SUCOMPLEX *d;                                 /* delay line, d[0] newest */
} su_iir_filt_t;
#define su_iir_filt_INITIALIZER { 0, NULL, NULL }

/* n symbol periods of T samples each, roll-off beta: n T + 1 taps (odd), unit DC gain */
SUAMD_API SUBOOL    su_iir_rrc_init(su_iir_filt_t *filt, SUSCOUNT n, SUFLOAT T, SUFLOAT beta);
SUAMD_API SUCOMPLEX su_iir_filt_feed(su_iir_filt_t *filt, SUCOMPLEX x);
SUAMD_API void      su_iir_filt_finalize(su_iir_filt_t *filt);

#ifdef __cplusplus
}
#endif
#endif
