/* refshim: <sigutils/iir.h> (absent): only named by include/WaveSampler.h:50 (the matched filter is compiled out,
 * Tasks/WaveSampler.cpp:55-93) */
#ifndef REFSHIM_SIGUTILS_IIR_H
#define REFSHIM_SIGUTILS_IIR_H
#include <sigutils/types.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct sigutils_iir_filt { void *impl; } su_iir_filt_t;
#define su_iir_filt_INITIALIZER { NULL }
SUBOOL su_iir_rrc_init(su_iir_filt_t *filt, SUSCOUNT n, SUFLOAT T, SUFLOAT beta);
void   su_iir_filt_finalize(su_iir_filt_t *filt);
#ifdef __cplusplus
}
#endif
#endif
