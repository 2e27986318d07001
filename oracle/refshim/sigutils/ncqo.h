/* refshim: <sigutils/ncqo.h> (absent).  The state is the oracle's (SPEC.md section B); oracle/ref_glue.cpp serves it. */
#ifndef REFSHIM_SIGUTILS_NCQO_H
#define REFSHIM_SIGUTILS_NCQO_H
#include <sigutils/types.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct sigutils_ncqo { uint32_t phase, dphase; uint64_t n; } su_ncqo_t;
#define su_ncqo_INITIALIZER { 0, 0, 0 }
void su_ncqo_init(su_ncqo_t *ncqo, SUFLOAT fnor);
void su_ncqo_set_phase(su_ncqo_t *ncqo, SUFLOAT phi);
SUCOMPLEX su_ncqo_read(su_ncqo_t *ncqo);
#ifdef __cplusplus
}
#endif
#endif
