/* refshim: <sigutils/pll.h> (absent): su_pll_t and su_costas_t, served by oracle/ref_glue.cpp over oracle/sdo.c */
#ifndef REFSHIM_SIGUTILS_PLL_H
#define REFSHIM_SIGUTILS_PLL_H
#include <sigutils/types.h>
#include <sigutils/ncqo.h>
#include <sdo.h>
#ifdef __cplusplus
extern "C" {
#endif
enum sigutils_costas_kind { SU_COSTAS_KIND_NONE, SU_COSTAS_KIND_BPSK, SU_COSTAS_KIND_QPSK, SU_COSTAS_KIND_8PSK };
typedef struct sigutils_pll { sdo_pll impl; } su_pll_t;
typedef struct sigutils_costas { sdo_costas impl; } su_costas_t;
#define su_pll_INITIALIZER    { }
#define su_costas_INITIALIZER { }
SUBOOL su_pll_init(su_pll_t *pll, SUFLOAT fhint, SUFLOAT fc);
SUCOMPLEX su_pll_track(su_pll_t *pll, SUCOMPLEX x);
void su_pll_finalize(su_pll_t *pll);
SUBOOL su_costas_init(su_costas_t *costas, enum sigutils_costas_kind kind, SUFLOAT fhint, SUFLOAT arm_bw,
                      unsigned int arm_order, SUFLOAT loop_bw);
SUCOMPLEX su_costas_feed(su_costas_t *costas, SUCOMPLEX x);
void su_costas_finalize(su_costas_t *costas);
#ifdef __cplusplus
}
#endif
#endif
