/* refshim: <sigutils/taps.h> (absent): the Blackman-Harris window of Tasks/CarrierDetector.cpp:87-89 */
#ifndef REFSHIM_SIGUTILS_TAPS_H
#define REFSHIM_SIGUTILS_TAPS_H
#include <sigutils/types.h>
#ifdef __cplusplus
extern "C" {
#endif
void su_taps_apply_blackmann_harris_complex(SUCOMPLEX *h, SUSCOUNT size);
#ifdef __cplusplus
}
#endif
#endif
