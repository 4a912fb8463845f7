// Short transcendental forms for the serial chains of the sync kernels (Costas loops: 32 dependent sincos + atan2 per
// reference carrier and block, sync.c:90-130).  The device's libm (ocml) spends ~100 instructions per call on argument
// reduction and ulp-exact polynomials; here the argument is reduced exactly in double (3 instructions), sine / cosine come
// from the transcendental unit (v_sin_f32 / v_cos_f32, input in revolutions) and the arc tangent from a 4-term polynomial
// after the tan(pi/8) reduction.  Measured on the MI355X against double precision (tools/probe/math_probe.hip):
// |error| <= 5e-7 in sin / cos for |x| <= 2000 rad, <= 3e-7 rad in atan2 -- the reference's own float32 chain (glibc)
// is not reproduced to the last ulp by ANY other libm either; SURVEY 8c pins these quantities at 1e-4.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace nrsc5 {

__device__ __forceinline__ void fast_sincos(float x, float &s, float &c)
{
#ifdef HIPEMU
    s = sinf(x); c = cosf(x);
#elif defined(NRSC5HIP_ACCURATE_TRIG)                           // diagnostic build (tools/gpu_cfo_batch.py --accurate): double-precision libm, rounded once
    s = (float)sin((double)x); c = (float)cos((double)x);
#else
    const double t = (double)x * 0.15915494309189533577;       // revolutions
    const float f = (float)(t - rint(t));                      // [-0.5, 0.5], exact to float precision for any float x
    s = __builtin_amdgcn_sinf(f);
    c = __builtin_amdgcn_cosf(f);
#endif
}

// x already inside [-pi, pi] (or a few turns): no double needed
__device__ __forceinline__ void fast_sincos_reduced(float x, float &s, float &c)
{
#ifdef HIPEMU
    s = sinf(x); c = cosf(x);
#elif defined(NRSC5HIP_ACCURATE_TRIG)
    s = (float)sin((double)x); c = (float)cos((double)x);
#else
    const float f = x * 0.15915494309189533577f;
    s = __builtin_amdgcn_sinf(f);
    c = __builtin_amdgcn_cosf(f);
#endif
}

__device__ __forceinline__ float fast_atan2(float y, float x)
{
#ifdef HIPEMU
    return atan2f(y, x);
#elif defined(NRSC5HIP_ACCURATE_TRIG)
    return (float)atan2((double)y, (double)x);
#else
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    float a = mn * __builtin_amdgcn_rcpf(mx);                  // [0, 1]; 0 / 0 -> NaN, handled below
    if (mx == 0.0f) a = 0.0f;
    float y0 = 0.0f, z = a;
    if (a > 0.41421356237309503f) { y0 = 0.78539816339744831f; z = (a - 1.0f) * __builtin_amdgcn_rcpf(a + 1.0f); }
    const float w = z * z;
    float p = __builtin_fmaf(8.05374449538e-2f, w, -1.38776856032e-1f);
    p = __builtin_fmaf(p, w, 1.99777106478e-1f);
    p = __builtin_fmaf(p, w, -3.33329491539e-1f);
    float r = y0 + __builtin_fmaf(p * w, z, z);
    if (ay > ax) r = 1.57079632679489662f - r;
    if (x < 0.0f) r = 3.14159265358979324f - r;
    return copysignf(r, y);
#endif
}

// Double-precision cosine / sine / arc tangent of SMALL arguments by their Taylor series (Horner), for the NCO step of the next
// block (prepare_block: the single lane that runs it sits at the end of the block-step chain; the device libm's three calls were
// ~5 k of the sync kernel's ~11 k "finish" cycles).  |x| <= 0.25: the series are cut where the next term is below 1e-20 of the
// result, the evaluation error is a few ulps of a double -- the callers round the results to float or use them as a phase
// increment whose own uncertainty is twelve orders of magnitude larger.
__device__ __forceinline__ void small_cos_sin(double x, double &c, double &s)
{
    const double w = x * x;
    // cos: 1 - w/2! + w^2/4! - ... + w^9/18!
    double pc = 1.0 / 6402373705728000.0;
    pc = pc * w - 1.0 / 20922789888000.0;
    pc = pc * w + 1.0 / 87178291200.0;
    pc = pc * w - 1.0 / 479001600.0;
    pc = pc * w + 1.0 / 3628800.0;
    pc = pc * w - 1.0 / 40320.0;
    pc = pc * w + 1.0 / 720.0;
    pc = pc * w - 1.0 / 24.0;
    pc = pc * w + 0.5;
    c = 1.0 - w * pc;
    // sin: x (1 - w/3! + w^2/5! - ... - w^9/19!)
    double ps = -1.0 / 121645100408832000.0;
    ps = ps * w + 1.0 / 355687428096000.0;
    ps = ps * w - 1.0 / 1307674368000.0;
    ps = ps * w + 1.0 / 6227020800.0;
    ps = ps * w - 1.0 / 39916800.0;
    ps = ps * w + 1.0 / 362880.0;
    ps = ps * w - 1.0 / 5040.0;
    ps = ps * w + 1.0 / 120.0;
    ps = ps * w - 1.0 / 6.0;
    s = x + x * (w * ps);
}
__device__ __forceinline__ double small_atan(double t)         // |t| <= 0.26: t (1 - w/3 + w^2/5 - ... + w^14/29), w = t^2
{
    const double u = -(t * t);
    double q = 1.0 / 29.0;
#pragma unroll
    for (int k = 13; k >= 0; k--) q = q * u + 1.0 / (double)(2 * k + 1);
    return t * q;
}

}  // namespace nrsc5
