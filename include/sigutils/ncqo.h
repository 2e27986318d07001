/*
 * include/sigutils/ncqo.h -- <sigutils/ncqo.h> (include/CarrierXlator.h:25): the numerically controlled oscillator behind
 * Tasks/CarrierXlator.cpp:36-37,57-60, served by libsigdigger_amd.so (csrc/sigutils_host.cpp).
 *
 * The reference embeds the state BY VALUE in its task objects (include/CarrierXlator.h:39) and calls su_ncqo_read once
 * per sample from a tight loop, so this is host code: the same binary32 operations as the device kernels
 * (SPEC.md sections C / D1, shared source csrc/sd_math.hpp), one sample at a time.  The block-at-a-time GPU form of
 * the same loop is suamd_xlate_bulk (include/sigdigger_amd.h).
 */
#ifndef SIGDIGGER_AMD_SIGUTILS_NCQO_H
#define SIGDIGGER_AMD_SIGUTILS_NCQO_H
#include "types.h"
#ifdef __cplusplus
extern "C" {
#endif

/* phase(n) = phase + n * dphase, 2^32 per turn: closed form, so that a block kernel and this loop agree bit for bit */
typedef struct sigutils_ncqo { uint32_t phase, dphase; uint64_t n; } su_ncqo_t;
#define su_ncqo_INITIALIZER { 0, 0, 0 }

SUAMD_API void      su_ncqo_init(su_ncqo_t *ncqo, SUFLOAT fnor);        /* fnor = 2 f / fs; dphase = llrint(fnor 2^31) */
SUAMD_API void      su_ncqo_set_phase(su_ncqo_t *ncqo, SUFLOAT phi);    /* radians */
SUAMD_API void      su_ncqo_set_freq(su_ncqo_t *ncqo, SUFLOAT fnor);    /* keeps the current phase */
SUAMD_API SUCOMPLEX su_ncqo_read(su_ncqo_t *ncqo);                      /* e^{j phase(n)}, then n += 1 */

#ifdef __cplusplus
}
#endif
#endif
