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

// cargf as the reference's libm computes it.  The coarse carrier angle of an acquisition block (acquire.c:153) and the AM receiver's carrier / equaliser phases go through
// atan2f ONCE per block and their last bit matters: one ulp of the coarse angle is a phase ramp of 1.6e-5 rad across the block that runs the CFO search -- the size of the
// oscillator's rounding drift (DESIGN.md (c) limit 2) -- and OCML's atan2f differs from glibc's in the last bit for 16 % of arguments.
// fdlibm's float arc tangent (s_atanf.c / e_atan2f.c, as glibc 2.35 ships them for x86-64: no FMA variant exists for these two), restated: float operations only,
// each rounded once -- the device reproduces glibc's atan2f bit for bit (tests/test_ref_atan2f.py: 2e7 arguments of four distributions against this container's libm, 0 mismatches; 1e8 when it was written).
__device__ inline float ref_atanf(float x)
{
    const float atanhi[4] = { 4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f };
    const float atanlo[4] = { 5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f };
    const float aT[11] = { 3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                           6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f };
    unsigned hxu; __builtin_memcpy(&hxu, &x, 4);
    const int hx = (int)hxu, ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {                                    // |x| >= 2^25 (0x50800000 = 2^34 in older sources; glibc: 2^25)
        if (ix > 0x7f800000) return x + x;                     // NaN
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {                                     // |x| < 0.4375
        if (ix < 0x31000000) return x;                         // |x| < 2^-29
        id = -1;
    } else {
        x = __builtin_fabsf(x);
        if (ix < 0x3f980000) {                                 // |x| < 1.1875
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }      // 7/16 <= |x| < 11/16
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }                              // 11/16 <= |x| < 19/16
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }      // |x| < 2.4375
            else { id = 3; x = -1.0f / x; }
        }
    }
    const float z = x * x, w = z * z;
    const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return hx < 0 ? -r : r;
}
__device__ inline float ref_atan2f(float y, float x)
{
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    unsigned hxu, hyu; __builtin_memcpy(&hxu, &x, 4); __builtin_memcpy(&hyu, &y, 4);
    const int hx = (int)hxu, hy = (int)hyu, ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;      // NaN
    if (hx == 0x3f800000) return ref_atanf(y);                  // x = 1
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);          // 2 * sign(x) + sign(y)
    if (iy == 0) { switch (m) { case 0: case 1: return y; case 2: return pi + tiny; default: return -pi - tiny; } }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) { switch (m) { case 0: return pi_o_4 + tiny; case 1: return -pi_o_4 - tiny; case 2: return 3.0f * pi_o_4 + tiny; default: return -3.0f * pi_o_4 - tiny; } }
        switch (m) { case 0: return 0.0f; case 1: return -0.0f; case 2: return pi + tiny; default: return -pi - tiny; }
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = ref_atanf(__builtin_fabsf(y / x));
    switch (m) {
    case 0: return z;
    case 1: return -z;
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

// cexpf(I y) as the reference's libm computes it: glibc's cexpf (s_cexp_template.c) is expf(+-0) = 1 times sincosf(y), and sincosf is the single-precision
// routine of glibc >= 2.28 (sysdeps/ieee754/flt-32/s_sincosf.c + s_sincosf.h, from Arm's optimized routines): reduction and a degree-7 / degree-8 polynomial pair in
// DOUBLE, rounded to float once -- a 0.56-ulp function that neither OCML's sincosf, nor v_sin_f32 / v_cos_f32, nor a correctly rounded double sine reproduces to the
// last bit (1.3 % of arguments differ).  Restated here operation for operation, INCLUDING which multiply-adds are fused: on every x86-64 host with FMA + AVX2 glibc's
// ifunc picks `__sincosf_fma` (the same source built with -mfma -mavx2, contraction on), and in that build (Ubuntu glibc 2.35-0ubuntu3.x, the image of this container and
// of the GPU box; read from its disassembly) EVERY a + b * c of the polynomial and the `x - n * hpi` of the fast reduction is one fused operation, the products x * s,
// x * x, x2 * x, x2 * x2, x3 * x2, x4 * x2 and x * hpi_inv are plain.  tests/test_ref_sincosf.py compares with this container's libm on 1e8 arguments.
// (A host WITHOUT FMA runs the unfused build and differs from this in the last bit for a fraction of a percent of arguments -- the UNMODIFIED reference is not
// bit-reproducible from such a host to an FMA host either.)
__device__ inline void ref_sincosf(float y, float &sinp, float &cosp)
{
    // __sincosf_table[0]; table [1] is the same with c0..c4 negated (a negated fma chain rounds to the negated result)
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10, C4 = 0x1.99343027bf8c3p-16;
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    unsigned yi; __builtin_memcpy(&yi, &y, 4);
    const unsigned top = (yi >> 20) & 0x7ff;                   // abstop12
    double x = (double)y, sgn = 1.0;
    int n = 0, q = 0;                                          // n: quadrant (parity swaps the polynomials), q: quadrant incl. the sign of a large argument
    if (top < 0x3f4) {                                         // |y| < pi/4
        if (top < 0x398) { sinp = y; cosp = 1.0f; return; }    // |y| < 2^-12
    } else if (top < 0x42f) {                                  // |y| < 120: reduce_fast, hpi_inv prescaled by 2^24, truncating conversion
        const double r = x * 0x1.45F306DC9C883p+23;
        n = ((int)r + 0x800000) >> 24;
        x = __builtin_fma(-(double)n, 0x1.921FB54442D18p0, x);
        q = n;
    } else if (top < 0x7f8) {                                  // reduce_large: 4/pi to 192 bits, a 32 x 96 -> 128 bit product, exact 2.62 fixed point
        const unsigned inv_pio4[24] = { 0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529, 0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1,
                                        0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041 };
        // arr[0], arr[4], arr[8] of the reference = the 96-bit window of 4/pi's bit string that starts 3 bytes in front of byte `idx`.  For |y| < 2^24 (idx <= 3: every
        // phase a Costas loop can reach) the window comes out of two 64-bit constants by shifts -- read from the table (which the compiler keeps in constant memory:
        // three dependent per-lane loads on the serial chain of every large-argument call) it was half of an exact Costas step's time in the CFO search.
        const unsigned idx = (yi >> 26) & 15;
        unsigned a0, a4, a8;
        if (idx <= 3) {
            const unsigned long long hi = 0x000000a2f9836e4eull, lo = 0x441529fc2757d1f5ull;
            const unsigned sh = 8u * idx;
            const unsigned long long H = sh ? (hi << sh) | (lo >> (64u - sh)) : hi, L = lo << sh;
            a0 = (unsigned)(H >> 32); a4 = (unsigned)H; a8 = (unsigned)(L >> 32);
        } else {
            const unsigned *arr = &inv_pio4[idx];
            a0 = arr[0]; a4 = arr[4]; a8 = arr[8];
        }
        const int shift = (yi >> 23) & 7;
        unsigned xi = (yi & 0xffffff) | 0x800000;
        xi <<= shift;
        unsigned long long res0 = (unsigned)(xi * a0);
        const unsigned long long res1 = (unsigned long long)xi * a4, res2 = (unsigned long long)xi * a8;
        res0 = (res2 >> 32) | (res0 << 32);
        res0 += res1;
        const unsigned long long nn = (res0 + (1ULL << 61)) >> 62;
        res0 -= nn << 62;
        x = (double)(long long)res0 * 0x1.921FB54442D18p-62;
        n = (int)nn;
        q = n + (int)(yi >> 31);
    } else {                                                   // inf / NaN
        sinp = cosp = y - y; return;
    }
    if (((q + 1) & 2) != 0) sgn = -1.0;                         // sign[q & 3] = { 1, -1, -1, 1 }
    const double cs = (q & 2) ? -1.0 : 1.0;                     // table [1]
    const double x2 = x * x, X = x * sgn;
    const double x3 = x2 * X, x4 = x2 * x2;
    const double s1 = __builtin_fma(x2, S3, S2), c2 = __builtin_fma(x2, C4 * cs, C3 * cs);
    const double x5 = x2 * x3, x6 = x2 * x4;
    const double c1 = __builtin_fma(x2, C1 * cs, C0 * cs);
    const double s = __builtin_fma(x3, S1, X), c = __builtin_fma(x4, C2 * cs, c1);
    const float sv = (float)__builtin_fma(s1, x5, s), cv = (float)__builtin_fma(c2, x6, c);
    if (n & 1) { sinp = cv; cosp = sv; } else { sinp = sv; cosp = cv; }
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
