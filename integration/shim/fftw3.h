/* fftw3.h stand-in for the DROP-IN build: types only.  The reference's src/acquire.h (included by input.h) names fftwf_plan /
 * fftwf_complex in acquire_t, whose layout the drop-in keeps; the drop-in itself never calls FFTW (src/acquire.c is not part of
 * it -- the FFT runs on the GPU), so nothing here is defined or linked.  A maintainer has the real header installed. */
#pragma once
#include <complex.h>
typedef float complex fftwf_complex;
typedef struct fftwf_plan_s *fftwf_plan;
