// Half-band 2:1 decimator evaluated straight from the caller's cu8 capture (engine option batch_zero_copy): replaces
// decimate_samples + halfband_q15_execute / dotprod_halfband_4 (input.c:52-69, firdecim_q15.c:137-165) for streams whose
// capture is resident in HBM, so that no decimated copy is ever written.
//
// Exact in float32, and here is why.  The reference forms Q15 samples a = (x - 127) * 64 from the bytes x and computes
//     acc = (int16)(acc + (((a[2i] + a[14 - 2i]) * t_i) >> 15)),  i = 0..3;   y = (int16)(acc + a[7]).
// With x' = x - 127 (|x'| <= 128) and s = x'_{2i} + x'_{14-2i}:  ((64 s) t_i) >> 15 = floor(s t_i / 512); s has 9 bits and
// t_i 15, so s * (t_i / 512) is exact in float32 and so is its floor.  |y| <= 1.556 * 8192 + 8192 < 32768: the int16
// accumulator never wraps for 8-bit input, and sums of five such integers are exact in float32 too.
// (pinned: exhaustive check of the division step and a 200 000-window comparison with the integer code in
//  tests/test_halfband_float.py; the integer kernel k_decimate_fm_cu8 stays the streaming seam's K1.)
#pragma once
#include "nrsc5_dev.h"

namespace nrsc5 {

struct HbTaps { float t0, t1, t2, t3; };                        // t_i / 512, pair (a[2i], a[14 - 2i])

__device__ __forceinline__ HbTaps hb_taps(const int16_t *hb_q15)
{
    HbTaps t;
    t.t0 = (float)hb_q15[0] * (1.0f / 512.0f); t.t1 = (float)hb_q15[1] * (1.0f / 512.0f);
    t.t2 = (float)hb_q15[2] * (1.0f / 512.0f); t.t3 = (float)hb_q15[3] * (1.0f / 512.0f);
    return t;
}

// dword d of the capture = raw complex samples 2d (low half: I, Q bytes) and 2d + 1 (high half); before the start of the
// stream the decimator's history is zero, i.e. the byte value 127
__device__ __forceinline__ uint32_t hb_raw_dword(const uint32_t *rw, long long d) { return d >= 0 ? rw[d] : 0x7f7f7f7fu; }
__device__ __forceinline__ float2 hb_even(uint32_t w) { return make_float2((float)(w & 0xffu) - 127.0f, (float)((w >> 8) & 0xffu) - 127.0f); }
__device__ __forceinline__ float2 hb_odd(uint32_t w) { return make_float2((float)((w >> 16) & 0xffu) - 127.0f, (float)(w >> 24) - 127.0f); }

// decimated sample from its eight even raw samples e[0..7] (raw 2a-14, 2a-12, .., 2a) and the centre sample o (raw 2a-7),
// all as x' = byte - 127; result = the Q15 integers of the reference, held in floats
__device__ __forceinline__ float2 hb_output(const float2 *e, float2 o, const HbTaps &t)
{
    float2 acc = make_float2(64.0f * o.x, 64.0f * o.y);
    acc.x += floorf((e[0].x + e[7].x) * t.t0); acc.y += floorf((e[0].y + e[7].y) * t.t0);
    acc.x += floorf((e[1].x + e[6].x) * t.t1); acc.y += floorf((e[1].y + e[6].y) * t.t1);
    acc.x += floorf((e[2].x + e[5].x) * t.t2); acc.y += floorf((e[2].y + e[5].y) * t.t2);
    acc.x += floorf((e[3].x + e[4].x) * t.t3); acc.y += floorf((e[3].y + e[4].y) * t.t3);
    return acc;
}

// one decimated sample a of a raw capture as Q15 integers (acquisition window; not the hot path)
__device__ inline c16 hb_sample_q15(const uint8_t *raw, long long a, const HbTaps &t)
{
    const uint32_t *rw = (const uint32_t *)raw;
    float2 e[8]; float2 o = make_float2(0.0f, 0.0f);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t w = hb_raw_dword(rw, a - 7 + k);
        e[k] = hb_even(w);
        if (k == 3) o = hb_odd(w);                              // raw sample 2a - 7 = odd half of dword a - 4
    }
    const float2 y = hb_output(e, o, t);
    c16 r; r.r = (int16_t)(int)y.x; r.i = (int16_t)(int)y.y;
    return r;
}

// x / 32767.0f, correctly rounded, without the divider: q0 = x r, one Newton correction through the exact residual
// (equal to the IEEE quotient for every int16 x: tests/test_halfband_float.py).  cq15_to_cf / _conj, defines.h:106-111.
__device__ __forceinline__ float q15_to_float(float x)
{
    const float r = 1.0f / 32767.0f;
    const float q0 = x * r;
    const float e = __builtin_fmaf(-q0, 32767.0f, x);
    return __builtin_fmaf(e, r, q0);
}

}  // namespace nrsc5
