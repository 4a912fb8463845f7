/* TEST INFRASTRUCTURE (oracle) -- see cpu_fft.h.
 * Stockham autosort, radix-4 passes + one radix-2 pass when log2(n) is odd.
 * Twiddles are evaluated in double precision and rounded once to float. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "cpu_fft.h"

#define MAXN 4096

typedef struct { float re, im; } cf;

static cf *tw_cache[13];
static pthread_mutex_t tw_lock = PTHREAD_MUTEX_INITIALIZER;

static const cf *twiddles(int n, int lg)
{
    pthread_mutex_lock(&tw_lock);
    if (!tw_cache[lg]) {
        cf *w = malloc(sizeof(cf) * n);
        for (int k = 0; k < n; k++) {
            double a = -2.0 * M_PI * (double)k / (double)n;
            w[k].re = (float)cos(a);
            w[k].im = (float)sin(a);
        }
        tw_cache[lg] = w;
    }
    pthread_mutex_unlock(&tw_lock);
    return tw_cache[lg];
}

static inline cf cmul(cf a, cf b) { cf r = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re }; return r; }
static inline cf cadd(cf a, cf b) { cf r = { a.re + b.re, a.im + b.im }; return r; }
static inline cf csub(cf a, cf b) { cf r = { a.re - b.re, a.im - b.im }; return r; }
static inline cf cmulj(cf a) { cf r = { -a.im, a.re }; return r; }   /* j*a */

void oracle_fft_forward(int n, const float *in, float *out)
{
    int lg = 0;
    while ((1 << lg) < n) lg++;
    if ((1 << lg) != n || n > MAXN) abort();
    const cf *w = twiddles(n, lg);
    cf bufa[MAXN], bufb[MAXN];
    const cf *x = (const cf *)in;
    cf *y = bufa;
    int len = n, s = 1;

    while (len >= 4) {
        const int q4 = len / 4;
        for (int p = 0; p < q4; p++) {
            const cf w1 = w[p * s], w2 = w[2 * p * s], w3 = w[3 * p * s];
            for (int q = 0; q < s; q++) {
                const cf a = x[q + s * p], b = x[q + s * (p + q4)];
                const cf c = x[q + s * (p + 2 * q4)], d = x[q + s * (p + 3 * q4)];
                const cf apc = cadd(a, c), amc = csub(a, c), bpd = cadd(b, d), jbmd = cmulj(csub(b, d));
                y[q + s * (4 * p + 0)] = cadd(apc, bpd);
                y[q + s * (4 * p + 1)] = cmul(w1, csub(amc, jbmd));
                y[q + s * (4 * p + 2)] = cmul(w2, csub(apc, bpd));
                y[q + s * (4 * p + 3)] = cmul(w3, cadd(amc, jbmd));
            }
        }
        x = y;
        y = (y == bufa) ? bufb : bufa;
        len /= 4; s *= 4;
    }
    if (len == 2) {
        for (int q = 0; q < s; q++) {
            const cf a = x[q], b = x[q + s];
            y[q] = cadd(a, b);
            y[q + s] = csub(a, b);
        }
        x = y;
    }
    memcpy(out, x, sizeof(cf) * n);
}
