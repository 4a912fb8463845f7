// K3 + K4 -- OFDM symbol extraction for gfx950: timing pick, Q15 -> float (conjugated), NCO mix,
// raised-cosine cyclic-prefix fold (acquire.c:160-168, 237-252) and the 2048-point forward FFT that
// the reference delegates to fftw3f (acquire.c:254, 315-320) + fftshift (defines.h:123), fused.
// Only the 2 x 267 bins sync.c:785-789 keeps are written back: 8.6 KB in, 4.3 KB out per symbol.
//
// One workgroup (128 lanes = 2 waves) per (stream, symbol).  FFT 2048 = 8 x 16 x 16, each lane
// carries 16 complex points in registers; two LDS exchanges (17.4 KB, padded pitch 272 / 17 so
// every ds_read/ds_write of a pass is bank-conflict-free).
//
// The reference advances the NCO with a sequential complex recurrence (69120 dependent steps per
// block, acquire.c:250), renormalised at the end of every symbol (acquire.c:252).  Here the phase of
// sample j is evaluated in closed form, theta_sym + j * dtheta, in double precision: lanes are
// independent and the result is within ~1e-5 rad of the recurrence (whose own rounding drift is of
// that order).  What the recurrence does deterministically is reproduced: dtheta is the angle of the
// ROUNDED increment (cosf, sinf) the reference multiplies by, and the phasor carries that increment's
// length -- |inc| = 1 + g, |g| <= 6e-8, so the reference's amplitude runs as (1 + g)^j ~ 1 + j g
// across a symbol, up to 1.3e-4 at sample 2159 (nco_ramp below; DESIGN.md (c) limit 2: without the
// ramp 2 % of the locks after a CFO search deviated in loop-internal state, with it 0.5 %).
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "wave_ops.h"
#include "fastmath.h"
#include "halfband_raw.h"
#include "prepare_block.h"

#include "mixfft_body.h"

namespace nrsc5 {

template <int SPW, int NPAR>
__global__ __launch_bounds__(128 * NPAR) MIXFFT_OCCUPANCY void k_mixfft(DevTables tb, DevBuffers db, const int *ids, int local_prepare)
{
    wave_set_priority_high();                                  // block-step chain = critical path; decode waves run at priority 0
    const int s = wave_uniform(stream_of(ids, blockIdx.y));
    __shared__ __attribute__((aligned(16))) uint8_t lds[sizeof(MixLds<NPAR>)];
    mixfft_wg<SPW, NPAR>(lds, tb, db, s, (int)blockIdx.x, local_prepare);
}

// =====================================================================================================================
// The 256-lane form (knob value mixfft_syms = 32): FFT 2048 = 8 x 8 x 8 x 4, EIGHT complex points per work-item, so that the kernel
// fits 64 VGPRs and eight waves per SIMD stay resident at the same LDS per workgroup.
//   n = r + 256 n1,  r = r2 + 32 r1,  r2 = r4 + 4 r3;     bin k = k1 + 8 k2 + 64 k3 + 512 k4
//   stage A  lane r:            DFT-8 over n1 -> k1, twiddle W2048^(k1 r), LDS [k1][r]
//   stage B  lane (k1, r2):     DFT-8 over r1 -> k2, twiddle W256^(r2 k2),  LDS [k1][k2][r2]
//   stage C  lane (k1, k2, r4): DFT-8 over r3 -> k3, twiddle W32^(r4 k3)
//   stage D  DFT-4 over r4 = the four lanes of a quad: two butterflies through DPP quad_perm, no third LDS exchange; lane q of the quad
//            ends up with k4 = bit-reversed q for all eight k3
// LDS: rows of 288 complex values per k1 (8 x 36: the second layout's k2 pitch 36 puts the eight k2 rows a stage-C lane group reads on
// disjoint banks; every other access of either exchange is 16 / 32 consecutive values) = 18 KB, + 224 stage-B/C twiddles.
constexpr int P8_ROW = 288;
constexpr int P8_K2 = 36;
constexpr int TW8_N = 224;     // W256^m for m <= 31 * 7 = 217 (stage B); stage C reads W32^(r4 k3) = W256^(8 r4 k3), 8 * 21 = 168

template <int M> __device__ __forceinline__ cf cf_lane_xor(cf v)
{
#ifdef HIPEMU
    return cf_make(__shfl_xor(v.x, M), __shfl_xor(v.y, M));
#else
    // (through float temporaries: __builtin_bit_cast applied to the vector-element expression v.y itself reads element 0 with this compiler)
    const float vx = v.x, vy = v.y;
    return cf_make(__builtin_bit_cast(float, lane_xor<M>(__builtin_bit_cast(int, vx))), __builtin_bit_cast(float, lane_xor<M>(__builtin_bit_cast(int, vy))));
#endif
}

__device__ inline void fft8_stage_b_twiddles(cf *twB, const float2 *tw)
{
    for (int m = threadIdx.x; m < TW8_N; m += blockDim.x) twB[m] = cf_of(tw[8 * m]);
}

//  in : x[n1] = sample tid + 256 n1
//  out: x[k3] = bin (tid >> 5) + 8 ((tid >> 2) & 7) + 64 k3 + 512 k4,  k4 = 2 (tid & 1) + ((tid >> 1) & 1)
__device__ inline void fft2048_wg8(cf *x, cf *lds, const float2 *twA, const cf *twB)
{
    const int tid = threadIdx.x;
    dft8(x);
#pragma unroll
    for (int k1 = 0; k1 < 8; k1++) {
        cf v = x[k1];
        if (k1) v = cmul(v, cf_of(twA[(k1 - 1) * 256 + tid]));
        lds[k1 * P8_ROW + tid] = v;
    }
    __syncthreads();
    {
        const int k1 = tid >> 5, r2 = tid & 31;
#pragma unroll
        for (int r1 = 0; r1 < 8; r1++) x[r1] = lds[k1 * P8_ROW + r2 + 32 * r1];
        dft8(x);
        __syncthreads();
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) {
            cf v = x[k2];
            if (k2) v = cmul(v, twB[r2 * k2]);
            lds[k1 * P8_ROW + k2 * P8_K2 + r2] = v;
        }
    }
    __syncthreads();
    const int q = tid & 3;
    {
        const int k1 = tid >> 5, k2 = (tid >> 2) & 7;
#pragma unroll
        for (int r3 = 0; r3 < 8; r3++) x[r3] = lds[k1 * P8_ROW + k2 * P8_K2 + q + 4 * r3];
        dft8(x);
#pragma unroll
        for (int k3 = 1; k3 < 8; k3++) x[k3] = cmul(x[k3], twB[8 * q * k3]);
    }
    // stage D: u = the quad's four values of one k3.  Butterfly 1 (partner = lane ^ 2): lanes 0, 1 <- u_q + u_(q+2), lanes 2, 3 <- u_(q-2) - u_q.
    // Lane 3 turns its difference by -j; butterfly 2 (partner = lane ^ 1): even lanes <- own + partner, odd lanes <- partner - own:
    // lane 0: X[0], lane 1: X[2], lane 2: X[1], lane 3: X[3].
    const float s1 = (q & 2) ? -1.0f : 1.0f, s2 = (q & 1) ? -1.0f : 1.0f;
    const bool turn = q == 3;
#pragma unroll
    for (int k3 = 0; k3 < 8; k3++) {
        const cf a = cadd(cf_lane_xor<2>(x[k3]), emul(x[k3], cf_make(s1, s1)));
        const cf w = turn ? mul_mj(a) : a;
        x[k3] = cadd(cf_lane_xor<1>(w), emul(w, cf_make(s2, s2)));
    }
}

// the 16 consecutive dwords of the capture work-item t < 240 needs: its nine outputs 9 t .. 9 t + 8 (240 x 9 = 2160 = one symbol)
__device__ __forceinline__ void raw_symbol_load8(const uint8_t *raw, long long a0, uint32_t (&W)[16], int tid)
{
    const uint32_t *rw = (const uint32_t *)raw;
    const long long d0 = a0 + 9 * tid - 7;
    const bool live = tid < 240;
    if (a0 >= 7) {
#ifdef HIPEMU
        struct u32x4 { uint32_t x, y, z, w; };
        typedef u32x4 u32x4_dw;
        const uint32_t *gw = rw;
#else
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef u32x4 u32x4_dw __attribute__((aligned(4)));
        const __attribute__((address_space(1))) uint32_t *gw = (const __attribute__((address_space(1))) uint32_t *)rw;
#endif
#pragma unroll
        for (int k = 0; k < 4; k++) {
            u32x4 v = {0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu, 0x7f7f7f7fu};
            if (live) v = *(const u32x4_dw *)(gw + d0 + 4 * k);
            W[4 * k] = v.x; W[4 * k + 1] = v.y; W[4 * k + 2] = v.z; W[4 * k + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) W[k] = live ? hb_raw_dword(rw, d0 + k) : 0x7f7f7f7fu;
    }
}

// as raw_symbol_halfband, nine outputs per work-item (work-items 240 .. 255 park nothing that is read: their slots are the tile's tail).
// In two halves -- outputs 0..3 from E[0..10], outputs 4..8 from E[4..15], the seven shared samples unpacked twice -- with a
// scheduling fence between them: unpacked all at once the sixteen samples (32 VGPRs) beside the raw dwords pushed the kernel to 80 VGPRs.
template <int I0, int NOUT>
__device__ __forceinline__ void raw_halfband8_part(const uint32_t (&W)[16], cf *tile, const hb_v2 *T, int m0)
{
    const hb_v2 off = hb_make(-254.0f, -254.0f);
    hb_v2 E[NOUT + 7];                                          // E[k] = even sample I0 + k
#pragma unroll
    for (int k = 0; k < NOUT + 7; k++) {
        E[k] = hb_make(hb_byte(W[I0 + k], 0), hb_byte(W[I0 + k], 1));
        if (!((I0 + k) & 1)) E[k] = hb_add(E[k], off);
    }
    auto start = [&](int i) -> hb_v2 {
        const float c = HB_BIAS - 127.0f * 64.0f;
        return hb_make(__builtin_fmaf(hb_byte(W[I0 + i + 3], 2), 64.0f, c), __builtin_fmaf(hb_byte(W[I0 + i + 3], 3), 64.0f, c));
    };
    auto pairs = [&](int i, hb_v2 *p) {
#pragma unroll
        for (int j = 0; j < 4; j++) p[j] = hb_add(E[i + j], E[i + 7 - j]);
    };
    auto park = [&](int i, hb_v2 acc) { tile[m0 + I0 + i] = cf_make(acc.x - HB_BIAS, HB_BIAS - acc.y); };
#pragma unroll
    for (int i = 0; i + 1 < NOUT; i += 2) {
        hb_v2 pa[4], pb[4];
        pairs(i, pa); pairs(i + 1, pb);
        hb_v2 a = start(i), b = start(i + 1);
        hb_fma4x2_s(a, b, pa, pb, T);
        park(i, a); park(i + 1, b);
    }
    if (NOUT & 1) {
        hb_v2 pa[4];
        pairs(NOUT - 1, pa);
        hb_v2 a = start(NOUT - 1);
        hb_fma4_s(a, pa, T);
        park(NOUT - 1, a);
    }
}

__device__ inline void raw_symbol_halfband8(const uint32_t (&W)[16], cf *tile, const HbTaps &taps, int tid)
{
    auto uni = [](float f) -> float {
#ifndef HIPEMU
        return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, f)));   // the taps ride in scalar register pairs
#else
        return f;
#endif
    };
    const float t0 = uni(taps.t0), t1 = uni(taps.t1), t2 = uni(taps.t2), t3 = uni(taps.t3);
    const hb_v2 T[4] = {hb_make(t0, t0), hb_make(t1, t1), hb_make(t2, t2), hb_make(t3, t3)};
    hb_round_down();
    raw_halfband8_part<0, 4>(W, tile, T, 9 * tid);
#ifndef HIPEMU
    __builtin_amdgcn_sched_barrier(0);
#endif
    raw_halfband8_part<4, 5>(W, tile, T, 9 * tid);
    hb_round_nearest();
}

template <bool RAW, bool EXACT = false>
__device__ __forceinline__ void mixfft_symbol8(const DevTables &tb, const DevBuffers &db, const StreamState &st, const SymParams &sp, int s, cf *lds, cf *twB)
{
    const int tid = threadIdx.x, sym = blockIdx.x;
    const long long a0 = sp.a00 + (long long)sym * SYM_N;
    uint32_t W[16];
    if (RAW) raw_symbol_load8(st.raw, a0, W, tid);
    fft8_stage_b_twiddles(twB, tb.twiddle);                    // first read two barriers from here
    const double dth = sp.dtheta;
    double a1 = 256.0 * dth;
    a1 -= 2 * M_PI * rint(a1 * (1.0 / (2 * M_PI)));
    double a0p = sp.theta + (double)sym * SYM_N * dth + (double)tid * dth;
    a0p -= 2 * M_PI * rint(a0p * (1.0 / (2 * M_PI)));
    float a1f = (float)a1, a0f = (float)a0p;                   // the double-precision part runs under the capture loads; only two floats cross the half-band
#ifndef HIPEMU
    asm volatile("" : "+v"(a1f), "+v"(a0f));                   // (finished HERE, not sunk behind the half-band with their double-precision inputs)
#endif
    if (RAW) {
        raw_symbol_halfband8(W, lds, hb_taps(tb.hb_q15), tid);
        __syncthreads();
    }
    cf stp, ph;
    {
        const float g1 = nco_ramp(sp.growth, 256), g0 = nco_ramp(sp.growth, tid) * (1.0f / 32767.0f);
        stp = emul(unit_phasor(a1f), cf_make(g1, g1));
        ph = emul(unit_phasor(a0f), cf_make(g0, g0));          // carries the Q15 scale and the oscillator's amplitude at sample tid
    }
    const c16 *win = db.q15 + (size_t)s * db.q15_cap + a0;
    auto sample = [&](int j) -> cf {
        if (RAW) return lds[j];
        const c16 s16 = win[j];
        return cf_make((float)s16.r, -(float)s16.i);
    };
    // NCO phasor of sample tid + 256 q: one accurate evaluation at q = 0 and one of the 256-sample step, then an 8-step recurrence
    cf x[8];
#pragma unroll
    for (int q = 0; q < 8; q++) {
        cf m = EXACT ? mix_exact(nco_tab_phasor(db, s, sym, tid + 256 * q), sample(tid + 256 * q)) : cmul(ph, sample(tid + 256 * q));
        if (q == 0 && tid < CP_N) { const float w = tb.shape[tid]; m = emul(m, cf_make(w, w)); }
        x[q] = m;
        ph = cmul(ph, stp);
    }
    if (tid < CP_N) {                                          // fold the cyclic extension back (acquire.c:246-247)
        const cf m = EXACT ? mix_exact(nco_tab_phasor(db, s, sym, FFT_N + tid), sample(FFT_N + tid)) : cmul(ph, sample(FFT_N + tid));
        const float w = tb.shape[FFT_N + tid];
        x[0] = cadd(x[0], emul(cf_make(w, w), m));
    }
    if (RAW) __syncthreads();

    fft2048_wg8(x, lds, tb.twiddle_a, twB);

    // fftshift + live-bin cut on the UNSHIFTED index k: upper sideband = k in [UB0 - 1024, UB1 - 1024], lower = k in [LB0 + 1024, LB0 + 1024 + 266].
    // Lane q holds k4 = 0, 2, 1, 3 (q = 0..3): of its eight k3 only 4..7 (k4 = 0, 2) or 0..3 (k4 = 1, 3) can be live.
    cf *out = (cf *)(db.bins + ((size_t)s * NSYM + sym) * LIVE_N);
    const int q = tid & 3, k4 = 2 * (q & 1) + (q >> 1);
    const bool hi = q < 2;
    const int kb = (tid >> 5) + 8 * ((tid >> 2) & 7) + 512 * k4 + (hi ? 256 : 0);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const cf v = hi ? x[4 + i] : x[i];
        const int k = kb + 64 * i;
        if (k >= UB0 - 1024 && k <= UB1 - 1024) out[LIVE_HALF + k - (UB0 - 1024)] = v;
        else if (k >= LB0 + 1024 && k < LB0 + 1024 + LIVE_HALF) out[k - (LB0 + 1024)] = v;
    }
}

#ifndef HIPEMU
#define MIXFFT8_OCCUPANCY __attribute__((amdgpu_waves_per_eu(8, 8)))
#else
#define MIXFFT8_OCCUPANCY
#endif
__global__ __launch_bounds__(256) MIXFFT8_OCCUPANCY void k_mixfft8(DevTables tb, DevBuffers db, const int *ids, int local_prepare)
{
    wave_set_priority_high();
    const int s = wave_uniform(stream_of(ids, blockIdx.y));
    const StreamState &st = db.state[s];
    SymParams sp;
    if (local_prepare) {
        __shared__ SymParams sh_sp;
        if (threadIdx.x == 0) {
            const Prepared p = prepare_values(st, false);
            sh_sp.active = p.active; sh_sp.a00 = (st.rd - st.base) + p.samperr; sh_sp.dtheta = p.dtheta; sh_sp.theta = p.theta; sh_sp.growth = p.growth; sh_sp.nco_mode = 0;   // (the fused seam runs FINE blocks only: closed form)
        }
        __syncthreads();
        sp = sh_sp;
    } else {
        sp.active = st.active; sp.a00 = (st.rd - st.base) + st.samperr_cur; sp.dtheta = st.dtheta; sp.theta = st.theta; sp.growth = st.growth; sp.nco_mode = st.nco_mode;
    }
    sp.active = wave_uniform(sp.active); sp.a00 = uniform64(sp.a00); sp.dtheta = uniform64(sp.dtheta); sp.theta = uniform64(sp.theta); sp.growth = uniform64(sp.growth); sp.nco_mode = wave_uniform(sp.nco_mode);
    if (!sp.active) return;
    __shared__ cf lds[8 * P8_ROW];
    static_assert(8 * P8_ROW >= 9 * 256 && 9 * 240 == SYM_N, "nine decimated samples per work-item fit in the FFT tile");
    __shared__ cf twB[TW8_N];
    if (sp.nco_mode) {
        if (st.raw) mixfft_symbol8<true, true>(tb, db, st, sp, s, lds, twB);
        else mixfft_symbol8<false, true>(tb, db, st, sp, s, lds, twB);
        return;
    }
    if (st.raw) mixfft_symbol8<true>(tb, db, st, sp, s, lds, twB);
    else mixfft_symbol8<false>(tb, db, st, sp, s, lds, twB);
}

// The reference's oscillator for one block, sample by sample (acquire.c:237-252): phase *= phase_increment 2160 times per symbol as the float
// complex product it is (four products, two sums, each rounded), phase /= cabsf(phase) at the symbol's end -- 69 120 DEPENDENT steps, so one
// lane per stream walks them (three packed instructions a step: ~0.5 ms, whatever the number of streams) and leaves every sample's phasor in db.nco_tab for
// the symbol kernel.  Launched only on steps that run the acquisition kernels, and it leaves at once for a stream whose block runs on the closed-form
// phasor -- under NCO_EXACT_UNTIL_FINE that is every block after a stream's first lock; the batch default is NCO_CLOSED_FORM (no exact block at all), the drop-in's NCO_EXACT_FIRST_BLOCK.
// cabsf: glibc's hypotf computes sqrt((double)x * x + (double)y * y) and rounds once to float (verified equal on 5e7 random pairs); the
// divisions are IEEE float divisions (hipcc's default).
// One WAVE per stream with one lane at work: as lane = stream the 64 lanes of a wave stored to 64 different cache lines per instruction and the
// vector memory unit, one line per cycle, became the bound (measured: 1.8 ms for 256 streams instead of the chain's ~0.5 ms).
__global__ __launch_bounds__(64) void k_nco_exact(DevBuffers db, const int *ids, int nstreams)
{
    const int idx = blockIdx.x;
    if (idx >= nstreams || threadIdx.x != 0) return;
    const int s = stream_of(ids, idx);
    StreamState &st = db.state[s];
    if (!st.active || !st.nco_mode) return;
    float pr = st.nco_re, pi = st.nco_im;
    const float c = st.inc_re, d = st.inc_im;
    float2 *tab = db.nco_tab + (size_t)s * NSYM * SYM_N;
    for (int sym = 0; sym < NSYM; sym++) {
#ifdef HIPEMU
        for (int j = 0; j < SYM_N; j++) {
            tab[sym * SYM_N + j] = make_float2(pr, pi);
            const float ac = pr * c, bd = pi * d, ad = pr * d, bc = pi * c;
            pr = ac - bd; pi = ad + bc;
        }
#else
        // THREE packed instructions per step -- (ac, ad), (bd, bc), then (ac - bd, ad + bc) with the sign of the low half in neg_lo -- the same
        // six IEEE operations, each rounded once (the compiler's own vectorisation of the scalar form spends four: it forms the sum AND the
        // difference of both halves); the chain is latency-bound at two dependent instructions per step
        cf P = cf_make(pr, pi);
        const cf K = cf_make(c, d);
        cf *out = (cf *)(tab + sym * SYM_N);
#pragma unroll 8
        for (int j = 0; j < SYM_N; j++) {
            out[j] = P;
            cf t1, t2;                                             // (one statement: plain VALU dependencies are interlocked in hardware, and between separate
                                                                   //  statements the compiler pads with s_nop it cannot know to be unnecessary)
            asm("v_pk_mul_f32 %1, %0, %3 op_sel_hi:[0,1]\n\t"
                "v_pk_mul_f32 %2, %0, %3 op_sel:[1,1] op_sel_hi:[1,0]\n\t"
                "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "+v"(P), "=&v"(t1), "=&v"(t2) : "v"(K));
        }
        pr = P.x; pi = P.y;
#endif
        const float m = (float)sqrt((double)pr * (double)pr + (double)pi * (double)pi);
        pr = pr / m; pi = pi / m;
    }
    st.nco_re = pr; st.nco_im = pi;                            // acquire_t.phase after the block
}

void launch_nco_exact(const DevBuffers &db, int nstreams, const int *stream_ids, hipStream_t st)
{
    if (!db.nco_tab) return;
    hipLaunchKernelGGL(k_nco_exact, dim3(nstreams), dim3(64), 0, st, db, stream_ids, nstreams);
}

// Symbols per workgroup (nrsc5hip_debug_tune NRSC5HIP_TUNE_MIXFFT_SYMS).  The persistent forms -- 2 / 4 / 8 symbols per workgroup, the
// next symbol's 24 capture dwords in flight during the current FFT, 128 VGPRs without a spill, 4 -> one round of 8 workgroups per CU
// at 256 streams -- were MEASURED SLOWER than one symbol per workgroup (profiles/r04_mixfft_persistent.txt: 74 vs 63 us per launch,
// pass 36.0 vs 33.0 ms): the stage-A twiddle loads, which the one-symbol kernel issues at its very start beside the capture loads,
// cannot be held across the loop (28 VGPRs) and are exposed once per symbol behind a barrier.  Default: 1.
void launch_mixfft(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, hipStream_t st, int syms_per_wg, int local_prepare)
{
    if (syms_per_wg == 32) { hipLaunchKernelGGL(k_mixfft8, dim3(NSYM, nstreams), dim3(256), 0, st, tb, db, stream_ids, local_prepare); return; }   // the 256-lane form
    if (syms_per_wg == 16) { hipLaunchKernelGGL((k_mixfft<1, 2>), dim3(NSYM / 2, nstreams), dim3(256), 0, st, tb, db, stream_ids, local_prepare); return; }   // knob value 16: NPAR = 2
    switch (syms_per_wg) {
    case 2: hipLaunchKernelGGL((k_mixfft<2, 1>), dim3(NSYM / 2, nstreams), dim3(128), 0, st, tb, db, stream_ids, local_prepare); break;
    case 4: hipLaunchKernelGGL((k_mixfft<4, 1>), dim3(NSYM / 4, nstreams), dim3(128), 0, st, tb, db, stream_ids, local_prepare); break;
    case 8: hipLaunchKernelGGL((k_mixfft<8, 1>), dim3(NSYM / 8, nstreams), dim3(128), 0, st, tb, db, stream_ids, local_prepare); break;
    default: hipLaunchKernelGGL((k_mixfft<1, 1>), dim3(NSYM, nstreams), dim3(128), syms_per_wg >= 100 ? (size_t)(syms_per_wg - 100) << 10 : 0, st, tb, db, stream_ids, local_prepare); break;   // (>= 100: DIAGNOSTIC, that many KiB of unused dynamic LDS per workgroup -- fewer workgroups per CU)
    }
}

// ---- stage-level entry: plain 2048-point FFTs, natural order in and out (parity tests) ----------------
__global__ __launch_bounds__(128) void k_fft2048(DevTables tb, const float2 *in, float2 *out)
{
    __shared__ cf lds[8 * PITCH_A];
    const int tid = threadIdx.x;
    const cf *src = (const cf *)(in + (size_t)blockIdx.x * FFT_N);
    cf *dst = (cf *)(out + (size_t)blockIdx.x * FFT_N);
    cf x[16];
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int n1 = 0; n1 < 8; n1++) x[8 * h + n1] = src[tid + 128 * h + 256 * n1];
    __shared__ cf twB[256];
    fft_stage_b_twiddles<128>(twB, tb.twiddle);
    fft2048_wg<false>(x, lds, tb.twiddle_a, twB);
    const int kbase = (tid >> 4) + 8 * (tid & 15);
#pragma unroll
    for (int i = 0; i < 16; i++) dst[kbase + 128 * ((i >> 2) + 4 * (i & 3))] = x[i];
}

__global__ __launch_bounds__(256) void k_fft2048_8(DevTables tb, const float2 *in, float2 *out)
{
    __shared__ cf lds[8 * P8_ROW];
    __shared__ cf twB[TW8_N];
    const int tid = threadIdx.x;
    const cf *src = (const cf *)(in + (size_t)blockIdx.x * FFT_N);
    cf *dst = (cf *)(out + (size_t)blockIdx.x * FFT_N);
    cf x[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; n1++) x[n1] = src[tid + 256 * n1];
    fft8_stage_b_twiddles(twB, tb.twiddle);
    fft2048_wg8(x, lds, tb.twiddle_a, twB);
    const int q = tid & 3, kb = (tid >> 5) + 8 * ((tid >> 2) & 7) + 512 * (2 * (q & 1) + (q >> 1));
#pragma unroll
    for (int k3 = 0; k3 < 8; k3++) dst[kb + 64 * k3] = x[k3];
}

void launch_fft2048(const DevTables &tb, const float2 *in, float2 *out, int nffts, hipStream_t st, int form)
{
    if (form == 32) hipLaunchKernelGGL(k_fft2048_8, dim3(nffts), dim3(256), 0, st, tb, in, out);
    else hipLaunchKernelGGL(k_fft2048, dim3(nffts), dim3(128), 0, st, tb, in, out);
}

}  // namespace nrsc5
