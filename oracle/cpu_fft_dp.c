/* TEST INFRASTRUCTURE (oracle) -- a SECOND stand-in for fftw3f behind the same interface as cpu_fft.c: the transform evaluated in double
 * precision (radix-2 decimation in time, twiddles from cos / sin in double) and rounded to float once at the end -- every output within half an ulp
 * of the exact DFT of the float inputs, i.e. "some other FFT library" as far as the caller can tell.  Linked into oracle/_ref/libnrsc5_ref_sse_dp.so
 * (oracle/Makefile): the UNMODIFIED reference on top of a different FFT, to measure what the reference does against ITSELF when only its FFT changes
 * (tools/cpu_cfo_lock_sweep.py --self; DESIGN.md (c) limit 2).  Never part of the product. */
#include <math.h>
#include <stdlib.h>
#include "cpu_fft.h"

#define MAXN 4096

void oracle_fft_forward(int n, const float *in, float *out)
{
    int lg = 0;
    while ((1 << lg) < n) lg++;
    if ((1 << lg) != n || n > MAXN) abort();
    static double re[MAXN], im[MAXN];
    double xr[MAXN], xi[MAXN];
    (void)re; (void)im;
    for (int k = 0; k < n; k++) {                              /* bit-reversed load */
        unsigned r = 0;
        for (int b = 0; b < lg; b++) r |= ((unsigned)(k >> b) & 1u) << (lg - 1 - b);
        xr[r] = (double)in[2 * k]; xi[r] = (double)in[2 * k + 1];
    }
    for (int len = 2; len <= n; len <<= 1) {
        const int half = len >> 1;
        for (int j = 0; j < half; j++) {
            const double a = -2.0 * M_PI * (double)j / (double)len, wr = cos(a), wi = sin(a);
            for (int i = j; i < n; i += len) {
                const double tr = wr * xr[i + half] - wi * xi[i + half], ti = wr * xi[i + half] + wi * xr[i + half];
                xr[i + half] = xr[i] - tr; xi[i + half] = xi[i] - ti;
                xr[i] += tr; xi[i] += ti;
            }
        }
    }
    for (int k = 0; k < n; k++) { out[2 * k] = (float)xr[k]; out[2 * k + 1] = (float)xi[k]; }
}
