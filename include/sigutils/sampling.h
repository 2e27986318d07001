/* include/sigutils/sampling.h -- <sigutils/sampling.h> (Tasks/WaveSampler.cpp:22, Tasks/DopplerCalculator.cpp:21): the
 * frequency / baud normalisation macros (SU_ABS2NORM_FREQ, SU_NORM2ABS_FREQ, SU_ANG2NORM_FREQ, SU_ABS2NORM_BAUD ...).
 * They live in types.h here. */
#ifndef SIGDIGGER_AMD_SIGUTILS_SAMPLING_H
#define SIGDIGGER_AMD_SIGUTILS_SAMPLING_H
#include "types.h"
#endif
