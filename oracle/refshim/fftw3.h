/* refshim: <fftw3.h> (FFTW3f, absent): the five calls of Tasks/CarrierDetector.cpp:43-46,58-75,94 and
 * Tasks/DopplerCalculator.cpp, served by oracle/ref_glue.cpp with the oracle's double-precision FFT
 * (sdo_fft_f64) rounded to binary32 -- at least as accurate as FFTW3f's own single-precision result. */
#ifndef REFSHIM_FFTW3_H
#define REFSHIM_FFTW3_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef float fftwf_complex[2];
typedef struct refshim_fftwf_plan_s *fftwf_plan;
#define FFTW_FORWARD  (-1)
#define FFTW_BACKWARD (+1)
#define FFTW_ESTIMATE (1U << 6)
#define FFTW_MEASURE  (0U)
fftwf_plan fftwf_plan_dft_1d(int n, fftwf_complex *in, fftwf_complex *out, int sign, unsigned flags);
void  fftwf_execute(const fftwf_plan p);
void  fftwf_destroy_plan(fftwf_plan p);
void *fftwf_malloc(size_t n);
void  fftwf_free(void *p);
#ifdef __cplusplus
}
#endif
#endif
