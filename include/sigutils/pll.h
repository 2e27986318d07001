/*
 * include/sigutils/pll.h -- <sigutils/pll.h> (include/PLLSyncTask.h:23, include/CostasRecoveryTask.h:23): su_pll_t and
 * su_costas_t behind Tasks/PLLSyncTask.cpp:36,53-56,84 and Tasks/CostasRecoveryTask.cpp:41,58-61,89, served by
 * libsigdigger_amd.so (csrc/sigutils_host.cpp): host code, one sample per call, state by value inside the task object
 * (include/PLLSyncTask.h:39, include/CostasRecoveryTask.h:39).  SPEC.md sections E / F; the same binary32 operation
 * sequence as costas_kernel / pll_kernel (csrc/loops.hip), whose block forms are suamd_costas_bank_* / suamd_pll_bank_*.
 */
#ifndef SIGDIGGER_AMD_SIGUTILS_PLL_H
#define SIGDIGGER_AMD_SIGUTILS_PLL_H
#include "types.h"
#include "ncqo.h"
#ifdef __cplusplus
extern "C" {
#endif

enum sigutils_costas_kind { SU_COSTAS_KIND_NONE, SU_COSTAS_KIND_BPSK, SU_COSTAS_KIND_QPSK, SU_COSTAS_KIND_8PSK };

typedef struct sigutils_pll {
  uint32_t phase;                               /* 2^32 per turn */
  SUFLOAT  omega, alpha, beta;
} su_pll_t;
#define su_pll_INITIALIZER { 0, 0, 0, 0 }

#define SU_COSTAS_MAX_ARM_ORDER 5               /* arm filter = Butterworth low-pass of order arm_order - 1 <= 4 */
typedef struct sigutils_costas {
  int      kind;
  uint32_t phase;
  SUFLOAT  omega, a, b, gain;
  int      order;                               /* of the arm filter */
  SUFLOAT  fb[SU_COSTAS_MAX_ARM_ORDER], fa[SU_COSTAS_MAX_ARM_ORDER];
  SUFLOAT  xh[SU_COSTAS_MAX_ARM_ORDER][2], yh[SU_COSTAS_MAX_ARM_ORDER][2];   /* [i]: i samples ago (re, im); [0] unused */
} su_costas_t;
#define su_costas_INITIALIZER { 0, 0, 0, 0, 0, 0, 0, {0}, {0}, {{0}}, {{0}} }

SUAMD_API SUBOOL    su_pll_init(su_pll_t *pll, SUFLOAT fhint, SUFLOAT fc);
SUAMD_API SUCOMPLEX su_pll_track(su_pll_t *pll, SUCOMPLEX x);
SUAMD_API void      su_pll_finalize(su_pll_t *pll);

SUAMD_API SUBOOL    su_costas_init(su_costas_t *costas, enum sigutils_costas_kind kind, SUFLOAT fhint, SUFLOAT arm_bw,
                                   unsigned int arm_order, SUFLOAT loop_bw);
SUAMD_API SUCOMPLEX su_costas_feed(su_costas_t *costas, SUCOMPLEX x);
SUAMD_API void      su_costas_set_loop_gain(su_costas_t *costas, SUFLOAT gain);
SUAMD_API void      su_costas_finalize(su_costas_t *costas);

#ifdef __cplusplus
}
#endif
#endif
