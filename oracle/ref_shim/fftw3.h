/* fftw3f stand-in. FFTW is a third-party dependency of the reference
 * (CMakeLists.txt:52-59,107-114 pins 3.3.10 for the built-in fallback) and is not
 * installed here. The five entry points the reference calls (acquire.c:315-320,
 * 194, 254, 375-378) are backed by oracle/cpu_fft.c, the same deterministic
 * float FFT the C restatement uses. */
#pragma once
#include <complex.h>
typedef float complex fftwf_complex;
typedef struct oracle_fft_plan *fftwf_plan;
#define FFTW_FORWARD (-1)
#define FFTW_ESTIMATE (1U << 6)
fftwf_complex *fftwf_alloc_complex(size_t n);
void fftwf_free(void *p);
fftwf_plan fftwf_plan_dft_1d(int n, fftwf_complex *in, fftwf_complex *out, int sign, unsigned flags);
void fftwf_execute(const fftwf_plan p);
void fftwf_destroy_plan(fftwf_plan p);
