/* fftw3f entry points for the oracle/_ref build (CHECKER ONLY), backed by oracle/cpu_fft.c.  The librtlsdr no-ops live in
 * integration/shim/rtlsdr_stubs.c. */
#include <stdlib.h>
#include <string.h>
#include "fftw3.h"
#include "../cpu_fft.h"

struct oracle_fft_plan { int n; float complex *in, *out; };

fftwf_complex *fftwf_alloc_complex(size_t n) { return aligned_alloc(64, ((n * sizeof(fftwf_complex) + 63) / 64) * 64); }
void fftwf_free(void *p) { free(p); }
fftwf_plan fftwf_plan_dft_1d(int n, fftwf_complex *in, fftwf_complex *out, int sign, unsigned flags)
{
    (void)sign; (void)flags;
    struct oracle_fft_plan *p = malloc(sizeof(*p));
    p->n = n; p->in = in; p->out = out;
    return p;
}
void fftwf_execute(const fftwf_plan p) { oracle_fft_forward(p->n, (const float *)p->in, (float *)p->out); }
void fftwf_destroy_plan(fftwf_plan p) { free(p); }
