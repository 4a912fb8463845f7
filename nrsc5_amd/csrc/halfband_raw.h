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

// ---- the symbol kernel's form: one fused multiply-add per product ---------------------------------------------------
// With the float32 rounding mode set to round-toward-minus-infinity, acc' = fma(s, t_i / 512, acc) IS acc + floor(s t_i / 512)
// as long as acc is an integer in [2^23, 2^24) (ulp 1): the product is exact inside the fma and the single rounding drops its
// fraction downwards.  The accumulator therefore starts at HB_BIAS = 1.5 * 2^23 (+ the centre sample) and HB_BIAS is taken
// off at the end; |y| < 2^15 keeps it in range.  Both components of a complex sample ride one v_pk_fma_f32.
// Everything else the decimator does between hb_round_down() and hb_round_nearest() is exact in any rounding mode (byte ->
// float conversions, sums of small integers, the bias subtraction), and the Q15 -> float division that is NOT is left to the
// reader of the tile (k_mixfft's sample()), so the compiler has nothing rounding-sensitive to move into the region.
constexpr float HB_BIAS = 12582912.0f;

#ifdef HIPEMU
struct hb_v2 { float x, y; };
__device__ __forceinline__ hb_v2 hb_make(float x, float y) { hb_v2 r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ hb_v2 hb_add(hb_v2 a, hb_v2 b) { return hb_make(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ void hb_round_down() {}
__device__ __forceinline__ void hb_round_nearest() {}
__device__ __forceinline__ float hb_byte(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xffu); }
// two outputs' four products each (interleaved on the device so that no fma waits for its predecessor)
__device__ __forceinline__ void hb_fma4x2(hb_v2 &a, hb_v2 &b, const hb_v2 *pa, const hb_v2 *pb, const hb_v2 *t)
{
    for (int i = 0; i < 4; i++) {
        a.x += floorf(pa[i].x * t[i].x); a.y += floorf(pa[i].y * t[i].y);
        b.x += floorf(pb[i].x * t[i].x); b.y += floorf(pb[i].y * t[i].y);
    }
}
__device__ __forceinline__ void hb_fma4(hb_v2 &a, const hb_v2 *pa, const hb_v2 *t)
{
    for (int i = 0; i < 4; i++) { a.x += floorf(pa[i].x * t[i].x); a.y += floorf(pa[i].y * t[i].y); }
}
__device__ __forceinline__ void hb_fma4x2_s(hb_v2 &a, hb_v2 &b, const hb_v2 *pa, const hb_v2 *pb, const hb_v2 *t) { hb_fma4x2(a, b, pa, pb, t); }
__device__ __forceinline__ void hb_fma4_s(hb_v2 &a, const hb_v2 *pa, const hb_v2 *t) { hb_fma4(a, pa, t); }
#else
typedef float hb_v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ hb_v2 hb_make(float x, float y) { hb_v2 r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ hb_v2 hb_add(hb_v2 a, hb_v2 b) { return a + b; }
// MODE[1:0] = single-precision rounding: 0 nearest-even, 2 toward -inf.  The asm statements are volatile (ordered among
// themselves) and the sched_barrier keeps the machine scheduler from moving anything else across the switch.
__device__ __forceinline__ void hb_round_down()
{
    // fenced on BOTH sides: with the fence behind it only, the scheduler moved the switch itself up across ~150 instructions of the code in front of it --
    // the NCO's v_sin / v_cos among them, which then ran rounding down and made the zero-copy batch differ from the streaming seam in the last bit
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2\n\ts_nop 1" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void hb_round_nearest()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\ts_nop 1" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ float hb_byte(uint32_t w, int k)
{
    float f;
    switch (k) {
    case 0: asm("v_cvt_f32_ubyte0_e32 %0, %1" : "=v"(f) : "v"(w)); break;
    case 1: asm("v_cvt_f32_ubyte1_e32 %0, %1" : "=v"(f) : "v"(w)); break;
    case 2: asm("v_cvt_f32_ubyte2_e32 %0, %1" : "=v"(f) : "v"(w)); break;
    default: asm("v_cvt_f32_ubyte3_e32 %0, %1" : "=v"(f) : "v"(w)); break;
    }
    return f;
}
__device__ __forceinline__ void hb_fma4x2(hb_v2 &a, hb_v2 &b, const hb_v2 *pa, const hb_v2 *pb, const hb_v2 *t)
{
    asm volatile("v_pk_fma_f32 %0, %2, %10, %0\n\t"
                 "v_pk_fma_f32 %1, %6, %10, %1\n\t"
                 "v_pk_fma_f32 %0, %3, %11, %0\n\t"
                 "v_pk_fma_f32 %1, %7, %11, %1\n\t"
                 "v_pk_fma_f32 %0, %4, %12, %0\n\t"
                 "v_pk_fma_f32 %1, %8, %12, %1\n\t"
                 "v_pk_fma_f32 %0, %5, %13, %0\n\t"
                 "v_pk_fma_f32 %1, %9, %13, %1"
                 : "+v"(a), "+v"(b)
                 : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]), "v"(pb[0]), "v"(pb[1]), "v"(pb[2]), "v"(pb[3]),
                   "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]));
}
// the same with the taps in scalar register pairs (wave-uniform; one SGPR source per instruction is what the constant bus allows):
// eight VGPRs less for the 64-VGPR kernel
__device__ __forceinline__ void hb_fma4x2_s(hb_v2 &a, hb_v2 &b, const hb_v2 *pa, const hb_v2 *pb, const hb_v2 *t)
{
    asm volatile("v_pk_fma_f32 %0, %2, %10, %0\n\t"
                 "v_pk_fma_f32 %1, %6, %10, %1\n\t"
                 "v_pk_fma_f32 %0, %3, %11, %0\n\t"
                 "v_pk_fma_f32 %1, %7, %11, %1\n\t"
                 "v_pk_fma_f32 %0, %4, %12, %0\n\t"
                 "v_pk_fma_f32 %1, %8, %12, %1\n\t"
                 "v_pk_fma_f32 %0, %5, %13, %0\n\t"
                 "v_pk_fma_f32 %1, %9, %13, %1"
                 : "+v"(a), "+v"(b)
                 : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]), "v"(pb[0]), "v"(pb[1]), "v"(pb[2]), "v"(pb[3]),
                   "s"(t[0]), "s"(t[1]), "s"(t[2]), "s"(t[3]));
}
__device__ __forceinline__ void hb_fma4_s(hb_v2 &a, const hb_v2 *pa, const hb_v2 *t)
{
    asm volatile("v_pk_fma_f32 %0, %1, %5, %0\n\t"
                 "v_pk_fma_f32 %0, %2, %6, %0\n\t"
                 "v_pk_fma_f32 %0, %3, %7, %0\n\t"
                 "v_pk_fma_f32 %0, %4, %8, %0"
                 : "+v"(a)
                 : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]), "s"(t[0]), "s"(t[1]), "s"(t[2]), "s"(t[3]));
}
__device__ __forceinline__ void hb_fma4(hb_v2 &a, const hb_v2 *pa, const hb_v2 *t)
{
    asm volatile("v_pk_fma_f32 %0, %1, %5, %0\n\t"
                 "v_pk_fma_f32 %0, %2, %6, %0\n\t"
                 "v_pk_fma_f32 %0, %3, %7, %0\n\t"
                 "v_pk_fma_f32 %0, %4, %8, %0"
                 : "+v"(a)
                 : "v"(pa[0]), "v"(pa[1]), "v"(pa[2]), "v"(pa[3]), "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]));
}
#endif

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
