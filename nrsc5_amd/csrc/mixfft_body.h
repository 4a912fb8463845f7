// The symbol transform of one workgroup (mix, CP fold, half-band, FFT-2048, live-bin cut) as device code shared by the kernels of k_mixfft.hip and the dataflow
// kernel at the end of k_sync.hip (k_flow): everything here is inline / a template.  Moved out of k_mixfft.hip in round 6, unchanged.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "wave_ops.h"
#include "fastmath.h"
#include "halfband_raw.h"
#include "prepare_block.h"
#include "flow_ops.h"

namespace nrsc5 {
#ifdef NRSC5HIP_MIXFFT_NOLOAD
constexpr bool DIAG_NOLOAD = true;
#else
constexpr bool DIAG_NOLOAD = false;
#endif

__device__ inline int stream_of(const int *ids, int idx) { return ids ? ids[idx] : idx; }

// Complex values as a native 2-vector: the compiler then keeps them in aligned register pairs and every complex add / subtract /
// scale is ONE packed instruction (v_pk_add_f32 / v_pk_mul_f32, the swaps and sign flips of a complex product riding in op_sel /
// neg modifiers) -- written on a struct of two floats the same arithmetic came out with 18 % of the kernel's VALU instructions
// being v_mov_b32 that only built register pairs (16-point DFT + twiddles: 336 -> 240 VALU instructions).  Same IEEE operations
// in the same order either way (no contraction): the bits do not change.  The CPU emulator build keeps the struct.
#ifdef HIPEMU
struct cf { float x, y; };
__device__ inline cf cf_make(float x, float y) { cf r; r.x = x; r.y = y; return r; }
__device__ inline cf cadd(cf a, cf b) { return cf_make(a.x + b.x, a.y + b.y); }
__device__ inline cf csub(cf a, cf b) { return cf_make(a.x - b.x, a.y - b.y); }
__device__ inline cf emul(cf a, cf b) { return cf_make(a.x * b.x, a.y * b.y); }                 // element by element
__device__ inline cf cmul(cf a, cf b) { return cf_make(a.x * b.x - a.y * b.y, a.y * b.x + a.x * b.y); }
__device__ inline cf cmul_k(cf a, cf b) { return cmul(a, b); }
__device__ inline cf mul_mj(cf a) { return cf_make(a.y, -a.x); }                                // a * (-j)
__device__ inline cf cf_swap(cf a) { return cf_make(a.y, a.x); }
__device__ inline cf cf_neg_x(cf a) { return cf_make(-a.x, a.y); }
#else
typedef float cf __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cf cf_make(float x, float y) { cf r; r.x = x; r.y = y; return r; }
__device__ __forceinline__ cf cadd(cf a, cf b) { return a + b; }
__device__ __forceinline__ cf csub(cf a, cf b) { return a - b; }
__device__ __forceinline__ cf emul(cf a, cf b) { return a * b; }
// (a.x b.x - a.y b.y, a.y b.x + a.x b.y): the second product joins through fma(t, (-1, 1), p) = p -+ t, rounded once like the
// subtraction / addition it replaces (t * +-1 is exact) -- 3 packed instructions; written as p + (-t.x, t.y) the compiler negated
// BOTH halves and moved one back
__device__ __forceinline__ cf cmul3(cf a, cf b) { const cf t = a.yx * b.yy, sg = {-1.0f, 1.0f}; return __builtin_elementwise_fma(t, sg, a * b.xx); }
// ... and in TWO: p = (a.x b.x, a.y b.x), then (-a.y b.y + p.x, a.x b.y + p.y) as ONE packed fma whose operand swaps and the sign of the low half ride in
// op_sel / neg_lo.  The second product is no longer rounded before it is added (a fused multiply-add on each component): within half an ulp of the
// three-instruction form, far inside the 1e-4 the float path is held to.  81 complex products per work-item and symbol.
__device__ __forceinline__ cf cmul(cf a, cf b)
{
#ifdef NRSC5HIP_CMUL_UNFUSED                                       // diagnostic build (python -m nrsc5_amd.build --cmul-unfused): every product rounded before it is added, as the CPU twin does
    return cmul3(a, b);
#endif
    cf p;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(p) : "v"(a), "v"(b));
    return p;
}
// b a compile-time constant (or wave-uniform): a scalar register pair
__device__ __forceinline__ cf cmul_k(cf a, cf b)
{
#ifdef NRSC5HIP_CMUL_UNFUSED
    return cmul3(a, b);
#endif
    cf p;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "v"(a), "s"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(p) : "v"(a), "s"(b));
    return p;
}
__device__ __forceinline__ cf mul_mj(cf a) { return cf_make(a.y, -a.x); }
__device__ __forceinline__ cf cf_swap(cf a) { return a.yx; }
__device__ __forceinline__ cf cf_neg_x(cf a) { return cf_make(-a.x, a.y); }
#endif
__device__ inline cf cf_of(float2 a) { return cf_make(a.x, a.y); }
// Unit phasor (cos x, sin x) for x in [-pi, pi].  v_sin / v_cos are transcendental-unit instructions: a VALU instruction reading their result needs one
// wait state, which the compiler inserts for its own instructions but not in front of inline assembly -- and cmul IS inline assembly (its first use
// right behind v_cos read a stale cosine: test_gpu_symbol_kernel_256_lanes).  The results therefore leave through a statement that owns the wait state.
__device__ __forceinline__ cf unit_phasor(float x)
{
    float sn, cs; fast_sincos_reduced(x, sn, cs);
#ifndef HIPEMU
    asm("s_nop 0" : "+v"(sn), "+v"(cs));
#endif
    return cf_make(cs, sn);
}

// forward 4-point DFT in place, natural order out
__device__ inline void dft4(cf &a0, cf &a1, cf &a2, cf &a3)
{
    const cf t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = mul_mj(csub(a1, a3));
    a0 = cadd(t0, t2); a1 = cadd(t1, t3); a2 = csub(t0, t2); a3 = csub(t1, t3);
}

// forward 8-point DFT, natural order in v[0..7] -> natural order out
__device__ inline void dft8(cf *v)
{
    cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    cf o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    dft4(e0, e1, e2, e3);
    dft4(o0, o1, o2, o3);
    const float c = 0.70710678118654752440f;
    o1 = emul(cadd(o1, mul_mj(o1)), cf_make(c, c));                  // * W8^1 = c(1 - j): (c (x + y), c (y - x))
    o2 = mul_mj(o2);                                                  // * W8^2 = -j
    o3 = emul(cadd(cf_swap(o3), cf_neg_x(o3)), cf_make(c, -c));      // * W8^3 = -c(1 + j): (c (y - x), -c (x + y))
    v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
    v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
    v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
    v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}

// forward 16-point DFT in place; output X[a + 4b] lands in v[4a + b]
__device__ inline void dft16(cf *v)
{
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) dft4(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);   // v[4k1+n2] = Y[k1][n2]
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, c2 = 0.70710678118654752440f;
    // W16^m, m = n2*k1
    v[5]  = cmul_k(v[5],  cf_make(c1, -s1));    // m=1
    v[6]  = cmul_k(v[6],  cf_make(c2, -c2));    // m=2
    v[7]  = cmul_k(v[7],  cf_make(s1, -c1));    // m=3
    v[9]  = cmul_k(v[9],  cf_make(c2, -c2));    // m=2
    v[10] = mul_mj(v[10]);                        // m=4
    v[11] = cmul_k(v[11], cf_make(-c2, -c2));   // m=6
    v[13] = cmul_k(v[13], cf_make(s1, -c1));    // m=3
    v[14] = cmul_k(v[14], cf_make(-c2, -c2));   // m=6
    v[15] = cmul_k(v[15], cf_make(-c1, s1));    // m=9
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) dft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
}

// dft16 reduced to the six outputs a work-item of the symbol kernel can ever store: bins k1 + 8 k2 + 128 m with
// m = 2, 3, 4 (upper sideband) and 11, 12, 13 (lower sideband) -- the other ten lie outside sync.c:785-789's 2 x 267 live bins
// for every (k1, k2).  Same first stage and twiddles as dft16; of the second stage's 4-point transforms only the wanted
// outputs: v[8] = X[2], v[12] = X[3], v[1] = X[4], v[14] = X[11], v[3] = X[12], v[7] = X[13].
__device__ inline void dft16_live(cf *v)
{
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) dft4(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, c2 = 0.70710678118654752440f;
    v[5]  = cmul_k(v[5],  cf_make(c1, -s1));
    v[6]  = cmul_k(v[6],  cf_make(c2, -c2));
    v[7]  = cmul_k(v[7],  cf_make(s1, -c1));
    v[9]  = cmul_k(v[9],  cf_make(c2, -c2));
    v[10] = mul_mj(v[10]);
    v[11] = cmul_k(v[11], cf_make(-c2, -c2));
    v[13] = cmul_k(v[13], cf_make(s1, -c1));
    v[14] = cmul_k(v[14], cf_make(-c2, -c2));
    v[15] = cmul_k(v[15], cf_make(-c1, s1));
    {   // a = 0: outputs b = 1 (X[4]) and b = 3 (X[12])
        const cf t1 = csub(v[0], v[2]), t3 = mul_mj(csub(v[1], v[3]));
        v[1] = cadd(t1, t3); v[3] = csub(t1, t3);
    }
    {   // a = 1: b = 3 (X[13])
        const cf t1 = csub(v[4], v[6]), t3 = mul_mj(csub(v[5], v[7]));
        v[7] = csub(t1, t3);
    }
    v[8] = cadd(cadd(v[8], v[10]), cadd(v[9], v[11]));         // a = 2: b = 0 (X[2])
    {   // a = 3: b = 0 (X[3]) and b = 2 (X[11])
        const cf t0 = cadd(v[12], v[14]), t2 = cadd(v[13], v[15]);
        v[12] = cadd(t0, t2); v[14] = csub(t0, t2);
    }
}

// (block size as a constant: read as blockDim.x it is two DEPENDENT global loads -- implicit-argument pointer, then the dispatch packet -- in
// front of everything that uses it; profiles/r04_mixfft_phases.txt)
template <int NT> struct StageBTwiddles {
    static_assert(256 % NT == 0, "whole rounds");
    float2 v[256 / NT];
    __device__ __forceinline__ void load(const float2 *tw)      // issued with the kernel's first burst of loads ...
    {
#pragma unroll
        for (int k = 0; k < 256 / NT; k++) v[k] = tw[8 * ((int)threadIdx.x + NT * k)];
    }
    __device__ __forceinline__ void park(cf *twB) const         // ... written to LDS once the capture loads are under way
    {
#pragma unroll
        for (int k = 0; k < 256 / NT; k++) twB[(int)threadIdx.x + NT * k] = cf_of(v[k]);
    }
};
template <int NT> __device__ __forceinline__ void fft_stage_b_twiddles(cf *twB, const float2 *tw)
{
    StageBTwiddles<NT> t; t.load(tw); t.park(twB);
}

constexpr int PITCH_A = 272;   // floats2 per k1 row (256 + 16: rows of one half-wave land on disjoint banks)
constexpr int PITCH_B = 17;    // per r2 row inside a k1 row of the second layout (16 + 1)

// 2048-point forward FFT by a 128-lane workgroup.
//  in : x[0..7]  = samples r + 256*n1 for r = tid,       n1 = 0..7
//       x[8..15] = samples r + 256*n1 for r = tid + 128
//  out: x[4a+b]  = bin  k1 + 8*k2 + 128*(a + 4b)   with k1 = tid >> 4, k2 = tid & 15
//  LIVE: only x[1], x[3], x[7], x[8], x[12], x[14] are produced (dft16_live)
// twB = W256^m, m = 0..255 (= tw[8 m]) in LDS, filled by fft_stage_b_twiddles before the first barrier the caller passes: the second
// exchange's twiddles are the same 256 values for every workgroup, and as gathers from the global table they were 15 load
// instructions of 16 cache lines each per wave on top of the first exchange's 14
// ta (PRELOADED): the work-item's fourteen stage-A twiddles, loaded by the caller BEFORE its last barrier -- issued behind it (where they are
// used) they were an L2 round trip at the head of every FFT
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// after_a: called once stage A has written its LDS tile (its twiddles are dead, x is about to be re-read): the persistent forms issue the
// NEXT symbol's capture loads there
template <bool LIVE, bool PRELOADED = false, typename AfterA = NoHook>
__device__ inline void fft2048_wg(cf *x, cf *lds, const float2 *twA, const cf *twB, const cf *ta = nullptr, AfterA after_a = AfterA())
{
    const int tid = threadIdx.x & 127;                         // (two symbols may share a 256-lane workgroup: k_mixfft's NPAR)
    // stage A: two radix-8 butterflies, twiddle W2048^(k1*r), scatter to [k1][r]
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int r = tid + 128 * h;
        dft8(x + 8 * h);
#pragma unroll
        for (int k1 = 0; k1 < 8; k1++) {
            cf v = x[8 * h + k1];
            if (k1) v = cmul(v, PRELOADED ? ta[7 * h + k1 - 1] : cf_of(twA[(k1 - 1) * 256 + r]));   // = twiddle[(k1 r) & 2047], consecutive work-items consecutive entries
            lds[k1 * PITCH_A + r] = v;
        }
    }
    after_a();
    __syncthreads();
    // stage B: lane (k1, r2): 16-point DFT over r1 of [k1][r2 + 16 r1], twiddle W256^(r2*k2)
    {
        const int k1 = tid >> 4, r2 = tid & 15;
#pragma unroll
        for (int r1 = 0; r1 < 16; r1++) x[r1] = lds[k1 * PITCH_A + r2 + 16 * r1];
        dft16(x);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const int k2 = (i >> 2) + 4 * (i & 3);
            cf v = x[i];
            if (r2) v = cmul(v, twB[(r2 * k2) & 255]);
            lds[k1 * PITCH_A + r2 * PITCH_B + k2] = v;
        }
    }
    __syncthreads();
    // stage C: lane (k1, k2): 16-point DFT over r2
    {
        const int k1 = tid >> 4, k2 = tid & 15;
#pragma unroll
        for (int r2 = 0; r2 < 16; r2++) x[r2] = lds[k1 * PITCH_A + r2 * PITCH_B + k2];
        if (LIVE) dft16_live(x); else dft16(x);
    }
}

// Zero-copy batch: the symbol's 2160 decimated samples straight from the cu8 capture.  Work-item t produces the
// contiguous outputs 17 t .. 17 t + 16 from 24 consecutive dwords (each raw sample is unpacked once and feeds up to eight
// outputs), converts them as cq15_to_cf_conj does and parks them in the FFT's LDS tile; the callers then pick their
// strided 17 samples from there.
//
// Even raw samples E[k] (dword a0 + 17 t - 7 + k, low half) pair up as (E[i + j], E[i + 7 - j]) for output i: one index of
// every pair is even and one odd, so the -127 offsets of both (x' = byte - 127) are applied as -254 to the even-indexed E
// only; the centre sample's -127 * 64 goes into the accumulator's start value.  The tile receives the Q15 INTEGERS
// (imaginary part negated: the FM receiver's spectrum flip); the Q15 -> float scale 1 / 32767 rides on the NCO phasor they are mixed with.
// the 24 consecutive dwords of the capture work-item t needs for the symbol whose first decimated sample is a0: six dwordx4
// loads, issued and NOT waited for here -- the caller overlaps them with the previous symbol's FFT
__device__ __forceinline__ void raw_symbol_load(const uint8_t *raw, long long a0, uint32_t (&W)[24], int tid)
{
    const int m0 = 17 * tid;
    const int nout = min(17, SYM_N - m0);                      // 17 for work-items 0..126, 1 for the last
    const uint32_t *rw = (const uint32_t *)raw;
    const long long d0 = a0 + m0 - 7;
    if (a0 >= 7) {                                             // block-uniform: all but a stream's very first symbol
#ifdef HIPEMU
        struct u32x4 { uint32_t x, y, z, w; };
        typedef u32x4 u32x4_dw;
        const uint32_t *gw = rw;
#else
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef u32x4 u32x4_dw __attribute__((aligned(4)));
        const __attribute__((address_space(1))) uint32_t *gw = (const __attribute__((address_space(1))) uint32_t *)rw;   // captures live in HBM: global_load, not flat
#endif
        // Six loads, no branch between them: predicated on "the last work-item stops at the symbol's end" they came out as two loads, a wait
        // for BOTH, then four more -- the capture's HBM latency paid twice per workgroup (profiles/r04_mixfft_phases.txt).  The last work-item
        // (one real output, 8 dwords) re-reads its second quad for k >= 2: inside the symbol, and what it computes from them lands in the tile's
        // unused tail.
        const int kmax = nout + 7 > 20 ? 5 : 1;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const u32x4 v = *(const u32x4_dw *)(gw + d0 + 4 * (k < kmax ? k : kmax));
            // (each of these loads touches 34 cache lines per wave -- 64 lanes 68 bytes apart; reading the wave's 4.4 KB once, 16
            // consecutive bytes per lane, and handing the dwords out through LDS was measured: no difference, 15.2 ms either way)
            W[4 * k] = v.x; W[4 * k + 1] = v.y; W[4 * k + 2] = v.z; W[4 * k + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 24; k++) W[k] = (k < nout + 7) ? hb_raw_dword(rw, d0 + k) : 0x7f7f7f7fu;
    }
}

__device__ inline void raw_symbol_halfband(const uint32_t (&W)[24], cf *tile, const HbTaps &taps, int tid)
{
    const int m0 = 17 * tid;
    const hb_v2 T[4] = {hb_make(taps.t0, taps.t0), hb_make(taps.t1, taps.t1), hb_make(taps.t2, taps.t2), hb_make(taps.t3, taps.t3)};
    const hb_v2 off = hb_make(-254.0f, -254.0f);
    hb_v2 E[24];
#pragma unroll
    for (int k = 0; k < 24; k++) {
        E[k] = hb_make(hb_byte(W[k], 0), hb_byte(W[k], 1));
        if (!(k & 1)) E[k] = hb_add(E[k], off);
    }
    auto start = [&](int i) -> hb_v2 {                          // HB_BIAS + 64 (o - 127), o = raw sample 2(m0 + i) - 7: exact
        const float c = HB_BIAS - 127.0f * 64.0f;
        return hb_make(__builtin_fmaf(hb_byte(W[i + 3], 2), 64.0f, c), __builtin_fmaf(hb_byte(W[i + 3], 3), 64.0f, c));
    };
    auto pairs = [&](int i, hb_v2 *p) {
#pragma unroll
        for (int j = 0; j < 4; j++) p[j] = hb_add(E[i + j], E[i + 7 - j]);
    };
    auto park = [&](int i, hb_v2 acc) {
        tile[m0 + i] = cf_make(acc.x - HB_BIAS, HB_BIAS - acc.y);      // the last work-item's spare outputs land in the tile's unused tail
    };
    hb_round_down();
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        hb_v2 pa[4], pb[4];
        pairs(i, pa); pairs(i + 1, pb);
        hb_v2 a = start(i), b = start(i + 1);
        hb_fma4x2(a, b, pa, pb, T);
        park(i, a); park(i + 1, b);
    }
    {
        hb_v2 pa[4];
        pairs(16, pa);
        hb_v2 a = start(16);
        hb_fma4(a, pa, T);
        park(16, a);
    }
    hb_round_nearest();
}

// RAW: the stream reads its cu8 capture in place (zero-copy batch) -- else its samples come from the Q15 FIFO.  Block-uniform, so
// the two forms are separate instantiations rather than a test per sample.
// A uniform pointer the optimiser cannot see through: loads through it are NOT hoisted out of the symbol loop (the 14 stage-A
// twiddles and the pulse-shape values of a work-item are loop-invariant; kept in registers across the loop they cost 32 VGPRs
// and the fourth wave per SIMD)
template <typename T> __device__ __forceinline__ const T *per_symbol(const T *p)
{
#ifndef HIPEMU
    asm volatile("" : "+s"(p));
#endif
    return p;
}

// a value every lane holds alike, moved to scalar registers (addresses and block parameters derived from it then cost no VGPRs)
template <typename T> __device__ __forceinline__ T uniform64(T v)
{
#ifndef HIPEMU
    static_assert(sizeof(T) == 8, "two dwords");
    unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(u >> 32));
    u = ((unsigned long long)hi << 32) | lo;
    return __builtin_bit_cast(T, u);
#else
    return v;
#endif
}

// what a workgroup needs of the block's bookkeeping: from the stream state (k_prepare or the previous k_sync wrote it) or, in the fast
// streaming seam, computed here from the state the sync kernel will commit it to (prepare_values, prepare_block.h)
struct SymParams { long long a00; double dtheta, theta; int active; double growth; int nco_mode; };

// The amplitude of the reference's oscillator at sample j of a symbol, (1 + g)^j, to first order (j g <= 1.3e-4, the second-order term 8e-9 is below
// float resolution): the work-item's start phasor (sample tid) times 1 + tid g, its STEP-sample step times 1 + STEP g.  Two double-precision
// fmas and two packed multiplies per work-item and symbol.
__device__ __forceinline__ float nco_ramp(double g, int n) { return (float)(1.0 + (double)n * g); }

// ---- exact-oscillator mode (StreamState::nco_mode; DESIGN.md (c) limit 2) --------------------------------------------------------------
// The block where a freshly reset stream runs its CFO search (detect_cfo, sync.c:292-337) is the one place where the last bits of the FFT's
// INPUT decide what the receiver does next: the search runs Costas loops over bins that hold no carrier, and whatever differs by 1e-5 rad
// is amplified into a different loop state.  For such blocks the symbol kernel takes the oscillator from a table k_nco_exact filled with
// the reference's own recurrence and mixes operation for operation as acquire.c:237-252 does -- the FFT's input is then the reference's, bit
// for bit; what remains is the transform's own rounding (2e-7 of the largest bin: measured harmless, tests/test_oracle_fft_independence.py).
__device__ __forceinline__ cf nco_tab_phasor(const DevBuffers &db, int s, int sym, int j)
{
    const float2 v = db.nco_tab[((size_t)s * NSYM + sym) * SYM_N + j];
    return cf_make(v.x, v.y);
}
// phase * cq15_to_cf_conj(sample) (defines.h:111, acquire.c:241): q holds the Q15 integers (re, -im); the divisions and the four products / two
// sums of the float complex multiplication each rounded as gcc -O3 compiles them (no contraction: -ffp-contract=off here too)
__device__ __forceinline__ cf mix_exact(cf ph, cf q)
{
    const float a = ph.x, b = ph.y;
    const float c = q.x / 32767.0f, dd = q.y / 32767.0f;
    const float ac = a * c, bd = b * dd, ad = a * dd, bc = b * c;
    return cf_make(ac - bd, ad + bc);
}

// diagnostic build only (-DNRSC5HIP_MIXFFT_PHASES, tools/gpu_mixfft_phases.py): shader cycles of wave 0 of stream 0's workgroups between the
// marks, accumulated in db.sync_phase_cycles[8..15]; the release kernel carries none of this
#ifdef NRSC5HIP_MIXFFT_PHASES
#define MIX_MARK_BEGIN long long mix_t0 = (long long)clock64()
#define MIX_MARK(i, wait) do { if (wait) { __builtin_amdgcn_s_waitcnt(0); } if (db.sync_phase_cycles && s == 0 && threadIdx.x == 0) { const long long now = (long long)clock64(); \
    atomicAdd((unsigned long long *)&db.sync_phase_cycles[8 + (i)], (unsigned long long)(now - mix_t0)); mix_t0 = now; } } while (0)
#else
#define MIX_MARK_BEGIN do { } while (0)
#define MIX_MARK(i, wait) do { } while (0)
#endif

// The loads of a workgroup's prologue that depend on nothing but the kernel arguments -- half-band taps, the stage-B twiddles, the two
// pulse-shape values of a work-item -- issued in ONE burst beside the stream-state loads, before anything is waited for.  As the code stood
// (each where it is used) a workgroup began with six DEPENDENT trips to memory: state, capture pointer, twiddle loop (one trip per iteration),
// block size (two), capture; profiles/r04_mixfft_phases.txt.
template <int NT> struct SymPrologue {
    StageBTwiddles<NT> twb;
    HbTaps taps;
    float w0, w1;                                              // shape[tid] (head of the symbol), shape[2048 + tid] (its cyclic extension; tid < CP_N)
    __device__ __forceinline__ void load(const DevTables &tb, int tid)
    {
        twb.load(tb.twiddle);
        taps = hb_taps(tb.hb_q15);
        w0 = tb.shape[tid];
        w1 = tb.shape[min(FFT_N + tid, SYM_N - 1)];            // (no branch: work-items >= CP_N never use theirs)
    }
};

// FLOW (k_flow.hip): the bins leave WRITE-THROUGH (sc1 stores: straight to memory, dropped from this XCD's L2) -- the block step that consumes them runs as another
// workgroup of the same launch, possibly on another XCD, and is released by a counter, not by a launch boundary.  wg: the workgroup's index among the stream's
// NSYM / (SPW * NPAR) symbol workgroups (blockIdx.x of k_mixfft).
template <bool RAW, int SPW, int NPAR, bool EXACT = false, bool FLOW = false>
__device__ __forceinline__ void mixfft_symbols(const DevTables &tb, const DevBuffers &db, const uint8_t *raw, const SymParams &sp, int s, cf *lds, cf *twB, const SymPrologue<128 * NPAR> &pro, const int wg)
{
    // SPW consecutive symbols of one stream per workgroup: the stage-B twiddles, the half-band taps and the NCO step are set up
    // once, and symbol n + 1's capture loads (24 dwords per work-item) are in flight while symbol n goes through mix and FFT --
    // with one symbol per workgroup every workgroup began its life waiting for HBM with nothing else to do (17 % VALU-busy).
    MIX_MARK_BEGIN;
    const int sym0 = (int)(wg * NPAR + (threadIdx.x >> 7)) * SPW;
    const long long a00 = sp.a00;                              // first sample of symbol 0 in the decimated stream
    uint32_t W[24];
    if (RAW && !DIAG_NOLOAD) raw_symbol_load(raw, a00 + (long long)sym0 * SYM_N, W, threadIdx.x & 127);      // first thing once the position is known
    if (RAW && DIAG_NOLOAD) { for (int k = 0; k < 24; k++) W[k] = 0x7f7f7f7fu + (uint32_t)k * 0x01010101u * (threadIdx.x & 3); }   // DIAGNOSTIC (NRSC5HIP_MIXFFT_NOLOAD build): the kernel without its capture loads
    pro.twb.park(twB);                                         // first read two barriers from here
    const double dth = sp.dtheta;
    const HbTaps taps = pro.taps;
    double a1 = 128.0 * dth;
    a1 -= 2 * M_PI * rint(a1 * (1.0 / (2 * M_PI)));
    cf stp;
    {
        const cf u = unit_phasor((float)a1);
        float sn = u.y, cs = u.x;
#ifndef HIPEMU
        if (SPW > 1) {                                         // the same value in every lane: held in a scalar register pair across the symbol loop
            sn = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sn)));
            cs = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, cs)));
        }
#endif
        const float g1 = nco_ramp(sp.growth, 128);             // wave-uniform: the step carries the ramp of 128 samples
        stp = cf_make(cs * g1, sn * g1);
    }
#pragma unroll 1
    for (int i = 0; i < SPW; i++) {
        const int sym = sym0 + i;
        const long long a0 = a00 + (long long)sym * SYM_N;
        int tid = threadIdx.x & 127;
#ifndef HIPEMU
        if (SPW > 1) asm volatile("" : "+v"(tid));                 // addresses derived from it are recomputed per symbol, not held across the loop
#endif
        const float2 *twA = SPW > 1 ? per_symbol(tb.twiddle_a) : tb.twiddle_a;
        const float *shape = SPW > 1 ? per_symbol(tb.shape) : tb.shape;
        // NCO phasor of sample j = tid + 128 q (q = 0..16): one accurate evaluation at q = 0 and one of the
        // 128-sample step, then a 16-step complex recurrence (error ~1e-6, far inside the float pipeline's own)
        double a0p = sp.theta + (double)sym * SYM_N * dth + (double)tid * dth;
        a0p -= 2 * M_PI * rint(a0p * (1.0 / (2 * M_PI)));
        cf ph;
        // (reduced to [-pi, pi] in double above.)  The phasor carries the Q15 -> float scale 1 / 32767 (cq15_to_cf, defines.h:106-111) through
        // its recurrence: the samples enter the mix as the integers they are -- three packed instructions per sample less than dividing each one
        // as the reference does, and within an ulp of it
        {
            const float g0 = nco_ramp(sp.growth, tid) * (1.0f / 32767.0f);   // (the symbol starts renormalised: amplitude 1 at sample 0)
            ph = emul(unit_phasor((float)a0p), cf_make(g0, g0));
        }
        MIX_MARK(1, 1);                                            // set-up + the capture loads' latency
        if (RAW) {
            raw_symbol_halfband(W, lds, taps, tid);
            __syncthreads();
        }
        MIX_MARK(2, 0);                                            // half-band + barrier
        const c16 *win = db.q15 + (size_t)s * db.q15_cap + a0;     // FIFO path (streaming seam, cs16 input)

        auto sample = [&](int j) -> cf {
            if (RAW) return lds[j];                                // the tile holds Q15 integers, conjugated
            const c16 s16 = win[j];
            return cf_make((float)s16.r, -(float)s16.i);           // cq15_to_cf_conj, defines.h:111, less its scale (carried by the phasor)
        };
        const float w0 = pro.w0, w1 = pro.w1;
        (void)shape;
        cf ta[14];                                                 // in flight while the mix runs
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int k1 = 1; k1 < 8; k1++) ta[7 * h + k1 - 1] = cf_of(twA[(k1 - 1) * 256 + tid + 128 * h]);
        cf x[16];
        if (EXACT) {                                               // the reference's oscillator from the table, its mix operation for operation
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int h = q & 1, n1 = q >> 1;
                const int j = tid + 128 * q;
                cf m = mix_exact(nco_tab_phasor(db, s, sym, j), sample(j));
                if (q == 0 && tid < CP_N) m = emul(m, cf_make(w0, w0));      // shape[j] * sample (acquire.c:243)
                x[8 * h + n1] = m;
            }
            if (tid < CP_N) {
                const int j = FFT_N + tid;
                const cf m = mix_exact(nco_tab_phasor(db, s, sym, j), sample(j));
                x[0] = cadd(x[0], emul(cf_make(w1, w1), m));       // fftin[j - 2048] += shape[j] * sample (acquire.c:247)
            }
        } else {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int h = q & 1, n1 = q >> 1;
            const int j = tid + 128 * q;
            cf m = cmul(ph, sample(j));
            if (q == 0 && tid < CP_N) m = emul(m, cf_make(w0, w0));
            x[8 * h + n1] = m;
            ph = cmul(ph, stp);
        }
        if (tid < CP_N) {                                          // fold the cyclic extension back (acquire.c:246-247)
            const int j = FFT_N + tid;
            const cf m = cmul(ph, sample(j));                      // ph = phasor of sample tid + 2048
            x[0] = cadd(x[0], emul(cf_make(w1, w1), m));
        }
        }
        if (RAW) __syncthreads();                                  // every work-item has its samples: the tile becomes the FFT's
        MIX_MARK(3, 0);                                            // NCO, mix, fold + barrier

        // (the next symbol's samples: issued after stage A -- right after the half-band, beside the fourteen stage-A twiddles and the sixteen points,
        // they pushed the persistent forms over 128 VGPRs)
        auto next_loads = [&]() { if (RAW && i + 1 < SPW) raw_symbol_load(raw, a0 + SYM_N, W, tid); };
        fft2048_wg<true, true>(x, lds, twA, twB, ta, next_loads);
        if (SPW > 1) __syncthreads();                              // stage C has read the tile: the next symbol may park its samples there
        MIX_MARK(4, 0);                                            // the FFT (three barriers)

        // fftshift (bin 1024 = DC) and the live-bin cut: x[4a + b] = bin kbase + 128 (a + 4 b); after the shift the work-item's
        // six candidates sit at kbase + 128 m', m' = 10, 11, 12 (upper sideband, bins 1304 .. 1570) and 3, 4, 5 (lower, 478 .. 744)
        // (Gathering the six outputs in the idle LDS tile and storing 534 consecutive values instead -- two more barriers -- was measured:
        // no gain for the kernel, 34.6 -> 35.3 ms for the pass.)
        cf *out = (cf *)(db.bins + ((size_t)s * NSYM + sym) * LIVE_N);
        const int kbase = (tid >> 4) + 8 * (tid & 15);
        static_assert(LB0 == 478 && UB0 == 1304 && UB1 == 1570 && LIVE_HALF == 267, "the six-output cut below is laid out for these edges");
        auto put = [&](int k, cf v) __attribute__((always_inline)) { if (FLOW) flow_store_through(&out[k], v); else out[k] = v; };
        if (kbase >= LB0 - 384) put(kbase + 384 - LB0, x[14]);                     // m' = 3  (X[11])
        put(kbase + 512 - LB0, x[3]);                                              // m' = 4  (X[12])
        if (kbase + 640 < LB0 + LIVE_HALF) put(kbase + 640 - LB0, x[7]);           // m' = 5  (X[13])
        if (kbase + 1280 >= UB0) put(LIVE_HALF + kbase + 1280 - UB0, x[8]);        // m' = 10 (X[2])
        put(LIVE_HALF + kbase + 1408 - UB0, x[12]);                                // m' = 11 (X[3])
        if (kbase + 1536 <= UB1) put(LIVE_HALF + kbase + 1536 - UB0, x[1]);        // m' = 12 (X[4])
        MIX_MARK(5, 1);                                            // the stores, waited for
    }
}

// (Held to 96 VGPRs for a fifth wave per SIMD -- amdgpu_waves_per_eu(5, 5), 7 dwords spilled -- the kernel was measured SLOWER,
// 17.2 vs 16.1 ms per pass, and the decode waves beside it lose their room: k_p1_forward 12 -> 21 ms of device time.)
#ifndef HIPEMU
#define MIXFFT_OCCUPANCY __attribute__((amdgpu_waves_per_eu(4, 4)))     // 128 VGPRs: four waves per SIMD, eight workgroups per CU (the LDS limit)
#else
#define MIXFFT_OCCUPANCY
#endif
// NPAR = 2: two symbols of the stream side by side in one 256-lane workgroup (each half its own tile; the stage-B twiddle table, the
// dispatch and the wave launch shared) -- half as many workgroups per launch at the same waves per SIMD
// the workgroup's LDS as one struct (round 6: shared with the dataflow kernel, k_flow.hip)
template <int NPAR> struct MixLds {
    alignas(16) cf lds_all[NPAR * 8 * PITCH_A];
    cf twB[256];
    SymParams sh_sp;
};

// FLOW: `flow_sp` holds the block's parameters (handed over by the previous block step of the same launch, or read from the stream state by the caller for the
// launch's first step) -- the stream state itself is not read for them
template <int SPW, int NPAR, bool FLOW = false>
__device__ __forceinline__ void mixfft_wg(uint8_t *lds_base, const DevTables &tb, const DevBuffers &db, const int s, const int wg, int local_prepare, const SymParams *flow_sp = nullptr)
{
    MixLds<NPAR> &L = *reinterpret_cast<MixLds<NPAR> *>(lds_base);
#ifdef NRSC5HIP_MIXFFT_PHASES
    const long long mix_entry = (long long)clock64();
#endif
    const StreamState &st = db.state[s];
    // first burst: everything that needs no stream state, then the state itself -- all in flight before the first wait
    SymPrologue<128 * NPAR> pro;
    pro.load(tb, threadIdx.x & 127);
    const uint8_t *raw = st.raw;                               // (read here, not behind the test of `active`: one trip to memory, not two)
    SymParams sp;
    if (FLOW) {
        sp = *flow_sp;
    } else if (local_prepare) {                                // block-uniform (fast streaming seam: no k_prepare launch in front of this kernel)
        SymParams &sh_sp = L.sh_sp;
        if (threadIdx.x == 0) {
            const Prepared p = prepare_values(st, false);
            sh_sp.active = p.active; sh_sp.a00 = (st.rd - st.base) + p.samperr; sh_sp.dtheta = p.dtheta; sh_sp.theta = p.theta; sh_sp.growth = p.growth; sh_sp.nco_mode = 0;   // (the fused seam runs FINE blocks only: closed form)
        }
        __syncthreads();
        sp = sh_sp;
    } else {
        sp.active = st.active; sp.a00 = (st.rd - st.base) + st.samperr_cur; sp.dtheta = st.dtheta; sp.theta = st.theta; sp.growth = st.growth; sp.nco_mode = st.nco_mode;
    }
    sp.active = wave_uniform(sp.active); sp.a00 = uniform64(sp.a00); sp.dtheta = uniform64(sp.dtheta); sp.theta = uniform64(sp.theta); sp.growth = uniform64(sp.growth); sp.nco_mode = wave_uniform(sp.nco_mode);   // scalar registers
    if (!sp.active) return;                                    // block-uniform
#ifdef NRSC5HIP_MIXFFT_PHASES
    if (db.sync_phase_cycles && s == 0 && threadIdx.x == 0) atomicAdd((unsigned long long *)&db.sync_phase_cycles[8], (unsigned long long)((long long)clock64() - mix_entry));
#endif
    cf *lds_all = L.lds_all;
    cf *lds = lds_all + (threadIdx.x >> 7) * (8 * PITCH_A);
    static_assert(sizeof(cf) == sizeof(float2), "a complex value is two floats either way");
    static_assert(8 * PITCH_A >= 17 * 128 && 17 * 127 < SYM_N, "17 decimated samples per work-item fit in the FFT tile");
    static_assert(NSYM % (SPW * NPAR) == 0, "whole workgroups per block");
    cf *twB = L.twB;
    if (FLOW) {                                                // dataflow grid: zero-copy FINE streams on the closed-form oscillator only (the caller checks)
        mixfft_symbols<true, SPW, NPAR, false, true>(tb, db, raw, sp, s, lds, twB, pro, wg);
        return;
    }
    if (sp.nco_mode) {                                         // block-uniform, rare (a freshly reset stream's first blocks): its own instantiation
        if (raw) mixfft_symbols<true, SPW, NPAR, true>(tb, db, raw, sp, s, lds, twB, pro, wg);
        else mixfft_symbols<false, SPW, NPAR, true>(tb, db, raw, sp, s, lds, twB, pro, wg);
        return;
    }
    if (raw) mixfft_symbols<true, SPW, NPAR>(tb, db, raw, sp, s, lds, twB, pro, wg);
    else mixfft_symbols<false, SPW, NPAR>(tb, db, raw, sp, s, lds, twB, pro, wg);
}


}  // namespace nrsc5
