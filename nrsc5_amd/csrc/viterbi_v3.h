// K=7 rate-1/3 tail-biting soft Viterbi for the P1 frame on gfx950, third generation: 6 VALU instructions per trellis
// step on the serial chain of one wave64 (the second generation in viterbi_wave.h needs 11).
//
// Replaces conv_dec.c:402-453 (schedule + traceback) and the SSE / NEON / generic ACS of conv_sse.h:233-323,
// conv_neon.h, conv_gen.h:32-101; decisions and decoded bits are the reference's, bit for bit (tie rule included).
//
//  * ROTATING LAYOUT (as before): before step t the lane with LOGICAL index L holds the metric of state rotr6^t(L); the
//    predecessors 2b, 2b+1 of a butterfly then sit in logical lanes L and L ^ (1 << t % 6), and the new metric belongs in
//    the same lane.  New: logical index = physical lane ^ (3 * bit 2), so that the six partner relations are the
//    physical lane XORs 1, 2, 7, 8, 16, 32 -- quad_perm, quad_perm, row_half_mirror, row_ror:8 (each folds into ONE
//    VOP2-DPP subtract) and v_permlane16_swap / v_permlane32_swap.
//  * PARITY-CODED TIE RULE.  A lane keeps u = 2 * metric + s0, s0 = bit 0 of the state it holds (1: it is the odd
//    predecessor 2b+1 of its butterfly).  With D = 2 * branch metric the two candidates u_own + D and u_partner - D have
//    opposite parity, can never tie, and their maximum M carries in bit 0 exactly the reference's decision "the survivor
//    came from 2b+1" -- `if (sum0 > sum1)` with ties to 2b+1 (conv_gen.h:47-53) costs no instruction.  Next step:
//    u = (M & ~1) | s0' (v_and_or_b32).  int32 metrics never overflow in 146 240 steps (|2 m| <= 762 per step), so the
//    reference's every-79-steps min-normalisation is dropped: decisions depend on metric differences only.
//  * DECISIONS stay in the lane: hist = alignbit(M, hist, 1) collects bit 0 of 32 steps per lane (v_alignbit_b32, off the
//    critical chain, issued in the DPP / swap hazard shadow of the NEXT step), stored as one dword per lane every 32
//    steps.  The block-parallel traceback turns a lane's 32 bits back into per-step masks with v_add_co (carry-out =
//    ballot of the MSBs), one instruction per step.
//  * SOFT INPUT: one dword per trellis step (s0 | s1 << 8 | s2 << 16, punctured = 0) read with s_load_dwordx16 straight
//    into SGPRs -- no vector loads, no v_readlane; the branch metric is one VOP3P v_dot4_i32_i8 of that SGPR with the
//    lane's +-2 weights, accumulating into u where the partner exchange allows it.
//  * per step, on the chain:  DPP phases   v_dot4 D | v_add X=u+D | v_alignbit (prev) | v_sub_dpp Y=u'-D | v_max | v_and_or
//                             swap phases  v_dot4 P=u+Dp | v_dot4 Q=u+Dq | v_alignbit (prev) | v_permlane*_swap | v_max | v_and_or
//    The instruction order inside the asm blocks provides the wait states the hardware wants (VALU write -> DPP read: 2,
//    VALU write -> permlane swap read: 2); LLVM's hazard recogniser cannot see into inline asm.
#pragma once
#include "nrsc5_dev.h"
#include "wave_ops.h"
#include "viterbi_wave.h"      // rotr6 / rotl6, TB_SEG
#ifndef HIPEMU
#include "viterbi_v3_asm.h"
#endif

namespace nrsc5 {

#ifdef HIPEMU
struct v16i { int v[16]; int &operator[](int i) { return v[i]; } const int &operator[](int i) const { return v[i]; } };
#else
typedef int v16i __attribute__((ext_vector_type(16)));         // 16 SGPRs: the soft words of 16 trellis steps
#endif

struct Vit3Const {
    int w[4];            // phases 0..3: own-edge branch weights (+-2 per soft value), packed as v_dot4 wants them
    int wp[2], wq[2];    // phases 4, 5: weights of the two registers that go through the permlane swap
    int s0[6];           // bit 0 of the state this lane holds in phase r
};

__device__ __forceinline__ unsigned vit3_logical_lane(unsigned phys) { return phys ^ (((phys >> 2) & 1u) * 3u); }

__device__ inline Vit3Const vit3_consts(unsigned phys)
{
    const unsigned L = vit3_logical_lane(phys & 63u);
    Vit3Const k;
#pragma unroll
    for (int r = 0; r < 6; r++) {
        const unsigned S = rotr6(L, r);                        // state held in phase r
        const unsigned reg = S & 0x3eu;                        // edge 2b -> b (gen_state_info, conv_dec.c:137-153)
        const int g0 = (__popc(reg & 0133u) & 1) ? 2 : -2, g1 = (__popc(reg & 0171u) & 1) ? 2 : -2, g2 = (__popc(reg & 0165u) & 1) ? 2 : -2;
        const int pos = (g0 & 0xff) | ((g1 & 0xff) << 8) | ((g2 & 0xff) << 16);
        const int neg = (-g0 & 0xff) | ((-g1 & 0xff) << 8) | ((-g2 & 0xff) << 16);
        k.s0[r] = (int)(S & 1u);
        if (r < 4) k.w[r] = pos;
        else {
            // register P: own candidate in the lower half (s0 = 0), offer to the partner in the upper half; Q the other way
            k.wp[r - 4] = (S & 1u) ? neg : pos;
            k.wq[r - 4] = (S & 1u) ? pos : neg;
        }
    }
    return k;
}

__device__ __forceinline__ int vit3_push(int hist, int ns)     // hist >> 1 with bit 0 of ns entering at bit 31 (v_alignbit_b32)
{
#ifdef HIPEMU
    return (int)(((unsigned)hist >> 1) | ((unsigned)ns << 31));
#else
    return (int)__builtin_amdgcn_alignbit((unsigned)ns, (unsigned)hist, 1);
#endif
}

// Reference form of one trellis step in phase R (CPU emulator build; documents what the asm below does).
// ns: receives this step's maximum; nsp: the previous step's, whose bit 0 is pushed into hist when PUSH.
template <int R, bool PUSH>
__device__ __forceinline__ void vit3_step_c(int &u, int &hist, int &ns, int nsp, int aw, const Vit3Const &k)
{
    if (PUSH) hist = vit3_push(hist, nsp);
    if constexpr (R < 4) {
        const int D = dot4_i8(aw, k.w[R], 0);
        constexpr int M = R == 0 ? 1 : R == 1 ? 2 : R == 2 ? 7 : 8;
        const int X = u + D, Y = __shfl_xor(u, M) - D;
        ns = X > Y ? X : Y;
    } else {
        constexpr int M = R == 4 ? 16 : 32;
        const int P = dot4_i8(aw, k.wp[R - 4], u), Q = dot4_i8(aw, k.wq[R - 4], u);
        const int Pin = __shfl_xor(P, M), Qin = __shfl_xor(Q, M);
        const bool upper = (threadIdx.x & M) != 0;
        const int P2 = upper ? Qin : P, Q2 = upper ? Q : Pin;   // v_permlane*_swap: vdst upper half <-> vsrc lower half
        ns = P2 > Q2 ? P2 : Q2;
    }
    u = (ns & ~1) | k.s0[(R + 1) % 6];
}

// ---- 8 trellis steps as ONE asm statement: viterbi_v3_asm.h, generated by tools/gen_vit3_asm.py, which schedules the
// steps (branch metrics one step ahead, history push as filler) and enforces gfx950's wait-state rules on the result.
// operands: u, h (history word), ns (in: the previous block's last maximum, still to be pushed unless the block opens a
// history word; out: this block's), x, d0..d3 (scratch); a0..a7 soft words (SGPRs); w0..w3, p4, q4, p5, q5 branch
// weights; z0..z5 = s0 per phase
#define V3_OPERANDS                                                                                                      \
    : [u] "+v"(u), [h] "+v"(hist), [ns] "+v"(ns), [x] "=&v"(x), [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3)   \
    : [a0] "s"(a0), [a1] "s"(a1), [a2] "s"(a2), [a3] "s"(a3), [a4] "s"(a4), [a5] "s"(a5), [a6] "s"(a6), [a7] "s"(a7),   \
      [w0] "v"(k.w[0]), [w1] "v"(k.w[1]), [w2] "v"(k.w[2]), [w3] "v"(k.w[3]),                                           \
      [p4] "v"(k.wp[0]), [q4] "v"(k.wq[0]), [p5] "v"(k.wp[1]), [q5] "v"(k.wq[1]),                                       \
      [z0] "v"(k.s0[0]), [z1] "v"(k.s0[1]), [z2] "v"(k.s0[2]), [z3] "v"(k.s0[3]), [z4] "v"(k.s0[4]), [z5] "v"(k.s0[5])

// 8 steps starting in phase PH (0, 2 or 4); W0: the first of them opens a history word (nothing to push).
template <int PH, bool W0>
__device__ __forceinline__ void vit3_block8(int &u, int &hist, int &ns, int a0, int a1, int a2, int a3, int a4, int a5, int a6, int a7, const Vit3Const &k)
{
    static_assert(PH == 0 || PH == 2 || PH == 4, "8-step blocks start in an even phase");
#ifdef HIPEMU
    int na;
    vit3_step_c<(PH + 0) % 6, !W0>(u, hist, na, ns, a0, k);
    vit3_step_c<(PH + 1) % 6, true>(u, hist, ns, na, a1, k);
    vit3_step_c<(PH + 2) % 6, true>(u, hist, na, ns, a2, k);
    vit3_step_c<(PH + 3) % 6, true>(u, hist, ns, na, a3, k);
    vit3_step_c<(PH + 4) % 6, true>(u, hist, na, ns, a4, k);
    vit3_step_c<(PH + 5) % 6, true>(u, hist, ns, na, a5, k);
    vit3_step_c<(PH + 6) % 6, true>(u, hist, na, ns, a6, k);
    vit3_step_c<(PH + 7) % 6, true>(u, hist, ns, na, a7, k);
#else
    int x, d0, d1, d2, d3;
    if constexpr (PH == 0) { if constexpr (W0) asm(VIT3_ASM_PH0_OPEN V3_OPERANDS); else asm(VIT3_ASM_PH0_CONT V3_OPERANDS); }
    else if constexpr (PH == 2) { if constexpr (W0) asm(VIT3_ASM_PH2_OPEN V3_OPERANDS); else asm(VIT3_ASM_PH2_CONT V3_OPERANDS); }
    else { if constexpr (W0) asm(VIT3_ASM_PH4_OPEN V3_OPERANDS); else asm(VIT3_ASM_PH4_CONT V3_OPERANDS); }
#endif
}

// 16 consecutive steps whose soft words sit in `a`; PH = phase of the first one, W0: it opens a history word
template <int PH, bool W0>
__device__ __forceinline__ void vit3_run16(int &u, int &hist, int &nsp, const v16i &a, const Vit3Const &k)
{
    vit3_block8<PH, W0>(u, hist, nsp, a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], k);
    vit3_block8<(PH + 8) % 6, false>(u, hist, nsp, a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15], k);
}

// 16 soft words from a wave-uniform address: scalar loads (the data was written by an earlier kernel)
__device__ __forceinline__ v16i vit3_load16(const int *p)
{
#ifdef HIPEMU
    v16i v;
    for (int i = 0; i < 16; i++) v[i] = p[i];
    return v;
#else
    typedef const __attribute__((address_space(4))) v16i *cptr;
    return *(cptr)(uintptr_t)p;
#endif
}

constexpr int VIT3_WARM = 6;                                   // chunks (64 steps each) of look-ahead of the L2 warm-up load
#ifdef HIPEMU
__device__ inline void vit3_keep(int) {}
#else
__device__ __forceinline__ void vit3_keep(int v) { asm volatile("" :: "v"(v)); }
#endif
#ifdef HIPEMU
#define VIT3_SCHED_BARRIER() do { } while (0)
#define VIT3_WAIT_SCALAR() do { } while (0)
#else
#define VIT3_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#define VIT3_WAIT_SCALAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_sched_barrier(0); } while (0)   /* lgkmcnt(0) */
#endif

// One history word = 32 steps starting in phase PH, soft words in (c0, c1): wait for them (their loads were issued a whole
// word -- ~900 cycles -- ago), issue the next word's two scalar loads into (n0, n1), run the steps, flush the last decision
// and store the word.  Scalar loads return out of order, so the only wait there is is lgkmcnt(0); that is why exactly one
// word's loads are in flight.  jn: index (mod len) of the NEXT word's soft words; words never straddle the tail-biting
// wrap (len % 32 == 0).  Callers alternate two register sets so that no 16-SGPR tuple is ever copied.
template <int PH>
__device__ __forceinline__ void vit3_word(int &u, int &hist, int &nsp, const v16i &c0, const v16i &c1, v16i &n0, v16i &n1,
                                          const int *soft, int len, int &jn, const Vit3Const &k, uint32_t *&dec_word)
{
    VIT3_WAIT_SCALAR();
    n0 = vit3_load16(soft + jn); n1 = vit3_load16(soft + jn + 16);
    jn += 32; if (jn >= len) jn -= len;
    VIT3_SCHED_BARRIER();
    vit3_run16<PH, true>(u, hist, nsp, c0, k);
    vit3_run16<(PH + 16) % 6, false>(u, hist, nsp, c1, k);
    hist = vit3_push(hist, nsp);
    *dec_word = (uint32_t)hist;
    dec_word += 64;
}

// ---- forward pass, in SEGMENTS ------------------------------------------------------------------------------------------
// The trellis of a frame is one serial chain of len + 64 steps (2.0 ms for a P1 frame at 13.75 ns per step, however many
// SIMDs idle beside it).  It is cut at "trip" boundaries (3 chunks = 192 steps: the phase pattern and the register roles
// repeat there) into G segments, one wave each, running CONCURRENTLY:
//   * segment 0 starts from the tail-biting reset (all-zero metrics), as the reference does;
//   * segment g > 0 starts VIT3_WARM_TRIPS trips early from all-zero metrics (a speculation: its history words go to a scratch
//     line), snapshots its metrics where its own steps begin, then runs them for real;
//   * every segment leaves its end metrics.
// EXACTNESS is not left to chance: the state of the recursion is the vector of metric DIFFERENCES (the decisions of all later
// steps are a function of them and of the input alone -- max / add / subtract only, int32 never overflows, the parity bits are
// positional).  So segment g's decisions are the sequential decoder's iff the differences of its snapshot equal the differences
// of segment g-1's end metrics, and g-1's are themselves right.  viterbi3_forward_fix walks that chain: a segment whose check
// fails is run again from the true end metrics of its predecessor (sequentially, by the one fixing wave).  With decodable
// input the survivors merge long before 384 steps and no segment is ever repaired; on pure noise or the all-erasure frame the
// repair brings back exactly the sequential result at, in the worst case, the sequential cost.
constexpr int VIT3_WARM_TRIPS = 2;                             // speculative warm-up of a segment: 384 steps ~ 55 constraint lengths
// (VIT3_GMAX, VIT3_META: nrsc5_dev.h -- the engine sizes the segment metadata with them)

__device__ __host__ inline int vit3_trips(int len) { return (len / 64 + 1) / 3; }
// segments a frame of `len` steps can be cut into: every segment but the first needs room for its warm-up behind it
__device__ __host__ inline int vit3_segments(int len, int want)
{
    const int T = vit3_trips(len);
    int g = want < 1 ? 1 : want > VIT3_GMAX ? VIT3_GMAX : want;
    while (g > 1 && T / g < VIT3_WARM_TRIPS + 1) g--;
    return g;
}

// Trips [trip0, trip1) of one frame by one wave (+ the frame's last 1..2 chunks when `last`), history words into dec.
// warm > 0: `warm` speculative trips first, from all-zero metrics, history to `scratch`, metrics then to snap[64];
// warm == 0: start from this lane's metric u0 (end metric of the predecessor) when have_u0, else from the tail-biting reset.
// uend[64] (may be null) receives the metrics after the last step.  Metric arrays are indexed by LOGICAL lane.  Returns u.
__device__ __forceinline__ int viterbi3_forward_span(const int *soft, int len, uint32_t *dec, uint32_t *scratch, int trip0, int trip1, int warm,
                                                      bool last, bool have_u0, int u0, int *snap, int *uend)
{
    const unsigned phys = threadIdx.x & 63u, L = vit3_logical_lane(phys);
    const Vit3Const k = vit3_consts(phys);
    const int nchunks = len / 64 + 1;
    const int tstart = trip0 - warm;
    int u = (have_u0 && !warm) ? u0 : k.s0[0], hist = 0, nsp = 0;  // reset_decoder: all-zero metrics for tail biting
    int jn = (len - VIT_EXTRA + 192 * tstart) % len;           // step t reads soft[(len - 32 + t) % len] (conv_dec.c:407-412)
    // L2 warm-up: the scalar loads have one word (~900 cycles) of cover, enough for an L2 / Infinity Cache hit but not for
    // HBM; one vector load per 3 chunks touches the 768 bytes that will be needed VIT3_WARM chunks from now
    int jw = (jn + 64 * VIT3_WARM) % len, warmv = 0;
    v16i a0 = vit3_load16(soft + jn), a1 = vit3_load16(soft + jn + 16), b0, b1;
    jn += 32; if (jn >= len) jn -= len;
    uint32_t *w = dec + (size_t)384 * trip0 + L;
    // 3 chunks = 192 steps = 6 history words per trip: the phase pattern (0, 2, 4, 0, 2, 4) and the roles of the two
    // soft-word register sets repeat exactly, so the loop carries no register shuffles
    for (int t = tstart; t < trip1; t++) {
        if (warm) {                                            // wave-uniform
            if (t < trip0) w = scratch + L;
            else if (t == trip0) { snap[L] = u; w = dec + (size_t)384 * trip0 + L; }
        }
        int jl = jw + (int)phys; if (jl >= len) jl -= len;
        const int w0 = soft[jl], w1 = soft[jl + 64 < len ? jl + 64 : jl + 64 - len], w2 = soft[jl + 128 < len ? jl + 128 : jl + 128 - len];
        jw += 192; if (jw >= len) jw -= len;
        vit3_word<0>(u, hist, nsp, a0, a1, b0, b1, soft, len, jn, k, w);
        vit3_word<2>(u, hist, nsp, b0, b1, a0, a1, soft, len, jn, k, w);
        vit3_word<4>(u, hist, nsp, a0, a1, b0, b1, soft, len, jn, k, w);
        vit3_word<0>(u, hist, nsp, b0, b1, a0, a1, soft, len, jn, k, w);
        vit3_word<2>(u, hist, nsp, a0, a1, b0, b1, soft, len, jn, k, w);
        vit3_word<4>(u, hist, nsp, b0, b1, a0, a1, soft, len, jn, k, w);
        warmv ^= w0 ^ w1 ^ w2;                                 // consumed at the end of the trip: their vmcnt wait is free by then
    }
    if (last) {
        const int rest = nchunks % 3;                          // 1 or 2 chunks more (phases 0, then 4)
        if (rest >= 1) {
            vit3_word<0>(u, hist, nsp, a0, a1, b0, b1, soft, len, jn, k, w);
            vit3_word<2>(u, hist, nsp, b0, b1, a0, a1, soft, len, jn, k, w);
        }
        if (rest == 2) {
            vit3_word<4>(u, hist, nsp, a0, a1, b0, b1, soft, len, jn, k, w);
            vit3_word<0>(u, hist, nsp, b0, b1, a0, a1, soft, len, jn, k, w);
        }
    }
    vit3_keep(warmv);
    if (uend) uend[L] = u;
    return u;
}

// end state: first maximum in STATE order (conv_dec.c:310-318); logical lane L holds state rotr6^steps(L).  Returns the logical
// lane of the winning end state (wave-uniform).
__device__ __forceinline__ int vit3_end_lane(int u, int len)
{
    const unsigned L = vit3_logical_lane(threadIdx.x & 63u);
    const int nchunks = len / 64 + 1;
    const int rend = (64 * nchunks) % 6;
    const int pm = u >> 1;
    const int best = wave_max_i32(pm);
    const int smin = wave_min_i32(pm == best ? (int)rotr6(L, rend) : 64);
    return wave_uniform((int)rotl6((unsigned)smin, rend));
}

// segment g of G of a frame: meta = the frame's VIT3_GMAX x VIT3_META ints.  warm: speculative warm-up trips of the segments
// g > 0 (VIT3_WARM_TRIPS; the test hook passes 0: the speculation is then wrong wherever the input carries information, and every
// segment goes through the repair)
__device__ __forceinline__ void viterbi3_forward_segment(const int *soft, int len, uint32_t *dec, int *meta, int g, int G, int warm)
{
    const int T = vit3_trips(len);
    int *m = meta + (size_t)g * VIT3_META;
    if (g && !warm) m[vit3_logical_lane(threadIdx.x & 63u)] = (int)(rotr6(vit3_logical_lane(threadIdx.x & 63u), 0) & 1u);   // the snapshot of a cold start: u = s0
    viterbi3_forward_span(soft, len, dec, (uint32_t *)(m + 128), g * T / G, (g + 1) * T / G, g ? warm : 0, g == G - 1, false, 0, m, m + 64);
}

// One wave per frame, after all G segment waves are done: verify the chain of segment boundaries, repair what the speculation got
// wrong, return the logical lane of the winning end state.  stats (may be null): [0] boundaries checked, [1] segments repaired.
__device__ __forceinline__ int viterbi3_forward_fix(const int *soft, int len, uint32_t *dec, int *meta, int G, int *stats)
{
    const unsigned L = vit3_logical_lane(threadIdx.x & 63u);
    const int T = vit3_trips(len);
    int repaired = 0;
    int ue = meta[64 + L];                                     // true end metrics of the segment before the boundary (segment 0: by construction)
    for (int g = 1; g < G; g++) {
        int *m = meta + (size_t)g * VIT3_META;
        const int sn = m[L];
        const int d = (ue - wave_uniform(ue)) - (sn - wave_uniform(sn));
        if (wave_max_i32(d < 0 ? -d : d) != 0) {              // wave-uniform
            ue = viterbi3_forward_span(soft, len, dec, nullptr, g * T / G, (g + 1) * T / G, 0, g == G - 1, true, ue, nullptr, m + 64);
            repaired++;
        } else {
            ue = m[64 + L];
        }
    }
    if (stats && (threadIdx.x & 63) == 0) { atomicAdd(&stats[0], G - 1); if (repaired) atomicAdd(&stats[1], repaired); }
    return vit3_end_lane(ue, len);
}

// Forward pass of one frame by ONE wave, start to end (len % 64 == 0).  soft: len dwords; dec: 2 * (len / 64 + 1) history
// words per lane, word i of LOGICAL lane L at dec[64 i + L] (bit j = decision of step 32 i + j for the state that lane then
// held).  Returns the logical lane of the winning end state (wave-uniform).
__device__ __forceinline__ int viterbi3_forward(const int *soft, int len, uint32_t *dec)
{
    const int u = viterbi3_forward_span(soft, len, dec, nullptr, 0, vit3_trips(len), 0, true, false, 0, nullptr, nullptr);
    return vit3_end_lane(u, len);
}

// ---- block-parallel traceback ---------------------------------------------------------------------------------------------
// value of `v` in lane (l4 >> 2): ds_bpermute_b32 (the LDS crossbar; no LDS storage involved)
__device__ __forceinline__ unsigned vit3_fetch_lane(unsigned l4, unsigned v)
{
#ifdef HIPEMU
    return (unsigned)__shfl((int)v, (int)(l4 >> 2));
#else
    return (unsigned)__builtin_amdgcn_ds_bpermute((int)l4, (int)v);
#endif
}

// 64 steps of traceback for ALL 64 candidate end lanes of a chunk at once (lane = candidate, logical numbering), two independent
// chunks per call so that the two dependency chains hide each other's crossbar latency.  Per step and chunk: the candidate sits
// in lane l; d = the decision that lane took at this step = bit (S & 31) of ITS history word (ds_bpermute + v_bfe); then bit R of
// l <- d: that is the lane that held the surviving predecessor 2b + d (its state's bit 0 is lane bit R in this phase).
// l4 = l << 2, the byte address form ds_bpermute wants.  acc collects the d of 32 steps.  Four VALU + one LDS-crossbar
// instruction per step and chunk (the second generation: 7.7 VALU + a VALU->SGPR hazard stall).
template <int PHA, int PHB, int S> struct Vit3Map2 {
    static __device__ __forceinline__ void run(unsigned &la, unsigned h0a, unsigned h1a, unsigned &ahia, unsigned &aloa,
                                               unsigned &lb, unsigned h0b, unsigned h1b, unsigned &ahib, unsigned &alob)
    {
        constexpr int RA = (PHA + S) % 6, RB = (PHB + S) % 6;
        const unsigned va = vit3_fetch_lane(la, S >= 32 ? h1a : h0a), vb = vit3_fetch_lane(lb, S >= 32 ? h1b : h0b);
        const unsigned da = (va >> (S & 31)) & 1u, db = (vb >> (S & 31)) & 1u;
        if (S >= 32) { ahia = (ahia << 1) | da; ahib = (ahib << 1) | db; } else { aloa = (aloa << 1) | da; alob = (alob << 1) | db; }
        la = (la & ~(4u << RA)) | (da << (RA + 2));
        lb = (lb & ~(4u << RB)) | (db << (RB + 2));
        Vit3Map2<PHA, PHB, S - 1>::run(la, h0a, h1a, ahia, aloa, lb, h0b, h1b, ahib, alob);
    }
};
template <int PHA, int PHB> struct Vit3Map2<PHA, PHB, -1> {
    static __device__ __forceinline__ void run(unsigned &, unsigned, unsigned, unsigned &, unsigned &, unsigned &, unsigned, unsigned, unsigned &, unsigned &) {}
};

// The decoded bit of a step is the state bit the step shifts out; in this shift-register trellis that is the decision taken SIX
// steps later on the same path (K - 1 = 6 state bits), and for the last six steps of a chunk it is a bit of the candidate's end
// lane itself.  So the 64 output bits of a candidate are its 64 decisions shifted by six, topped up from the lane index:
// out bit S = d[S + 6] (S <= 57), = bit (PH + S) % 6 of the end lane (S >= 58).  PH = phase of the chunk's first step.
template <int PH>
__device__ __forceinline__ void vit3_outputs(unsigned lane, unsigned ahi, unsigned alo, unsigned &ohi, unsigned &olo)
{
    unsigned top = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) top |= ((lane >> ((PH + 58 + k) % 6)) & 1u) << k;
    ohi = (ahi >> 6) | (top << 26);
    olo = (alo >> 6) | (ahi << 26);
}

__device__ __host__ inline size_t vit3_traceback_smem(int len)
{
    const int nchunks = len / 64 + 1, nseg = (nchunks + TB_SEG - 1) / TB_SEG;
    return (size_t)nseg * 64 + (size_t)nchunks;
}

// Block-parallel traceback over the history words of the forward pass (blockDim.x a multiple of 64):
//   1  chunk maps AND the 64 output bits of all 64 candidate end lanes (two chunks per wave at a time); the candidate outputs
//      replace the chunk's history words in place                                           -> gmap[c][64], dec
//      (the engine runs this pass as its own launch, k_p1_tbmap, with as many workgroups per frame as keeps the chip busy: a thin
//      window's few frames then spread over many CUs instead of queueing 2285 chunks on one)
//   2  per segment of TB_SEG chunks: composition of its maps, all 64 candidates             -> segmap[sg][64]
//   3  the true end lane of every segment, last to first (one thread)
//   4  per segment: the chosen end lane of each of its chunks (one thread per segment)      -> chosen[c]
//   5  per chunk: the output words of its chosen candidate
// (Keeping all chunk maps of a frame in LDS -- 146 KB -- was measured: the composition gets faster, but a workgroup that owns a
// CU's LDS keeps the block-step kernels off that CU and the whole pass loses 20 %: profiles/r03_traceback_variants.txt.)
// pass 1 of the traceback for the chunk pairs wave `gw` of `nw` owns (any number of workgroups per frame: the chunks are independent)
__device__ __forceinline__ void viterbi3_traceback_maps(uint32_t *dec, int len, uint8_t *gmap, int gw, int nw)
{
    const int lane = threadIdx.x & 63;
    const int nchunks = len / 64 + 1;
    // a wave takes chunks in adjacent pairs; the phase of a chunk's first step is (64 c) % 6 = 0, 4, 2 for c % 3 = 0, 1, 2
    for (int ca = 2 * gw; ca < nchunks; ca += 2 * nw) {
        const int cb = ca + 1;
        const bool two = cb < nchunks;
        const unsigned h0a = dec[(size_t)(2 * ca) * 64 + lane], h1a = dec[(size_t)(2 * ca + 1) * 64 + lane];
        const unsigned h0b = two ? dec[(size_t)(2 * cb) * 64 + lane] : 0u, h1b = two ? dec[(size_t)(2 * cb + 1) * 64 + lane] : 0u;
        unsigned la = (unsigned)lane << 2, lb = la, ahia = 0, aloa = 0, ahib = 0, alob = 0, ohia, oloa, ohib, olob;
        switch (ca % 3) {                                       // wave-uniform
        case 0: Vit3Map2<0, 4, 63>::run(la, h0a, h1a, ahia, aloa, lb, h0b, h1b, ahib, alob);
                vit3_outputs<0>((unsigned)lane, ahia, aloa, ohia, oloa); vit3_outputs<4>((unsigned)lane, ahib, alob, ohib, olob); break;
        case 1: Vit3Map2<4, 2, 63>::run(la, h0a, h1a, ahia, aloa, lb, h0b, h1b, ahib, alob);
                vit3_outputs<4>((unsigned)lane, ahia, aloa, ohia, oloa); vit3_outputs<2>((unsigned)lane, ahib, alob, ohib, olob); break;
        default: Vit3Map2<2, 0, 63>::run(la, h0a, h1a, ahia, aloa, lb, h0b, h1b, ahib, alob);
                vit3_outputs<2>((unsigned)lane, ahia, aloa, ohia, oloa); vit3_outputs<0>((unsigned)lane, ahib, alob, ohib, olob); break;
        }
        gmap[64 * ca + lane] = (uint8_t)(la >> 2);
        dec[(size_t)(2 * ca) * 64 + lane] = oloa; dec[(size_t)(2 * ca + 1) * 64 + lane] = ohia;
        if (two) {
            gmap[64 * cb + lane] = (uint8_t)(lb >> 2);
            dec[(size_t)(2 * cb) * 64 + lane] = olob; dec[(size_t)(2 * cb + 1) * 64 + lane] = ohib;
        }
    }
}

// passes 1 (unless maps_done: an earlier launch ran viterbi3_traceback_maps over all chunks) .. 5
__device__ inline void viterbi3_traceback_block(uint32_t *dec, int len, int endlane, uint32_t *out, uint8_t *gmap, uint8_t *smem, bool maps_done = false)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int nchunks = len / 64 + 1, nseg = (nchunks + TB_SEG - 1) / TB_SEG;
    uint8_t *segmap = smem, *chosen = smem + (size_t)nseg * 64;
    __shared__ uint8_t segend[64];                             // end lane of each segment (nseg <= 64)
    if (!maps_done) viterbi3_traceback_maps(dec, len, gmap, wave, nwaves);
    __threadfence_block();
    __syncthreads();
    for (int sg = wave; sg < nseg; sg += nwaves) {             // pass 2
        const int c0 = sg * TB_SEG, c1 = min(nchunks, c0 + TB_SEG);
        unsigned e = (unsigned)lane;
        for (int c = c1 - 1; c >= c0; c--) e = gmap[64 * c + e];
        segmap[64 * sg + lane] = (uint8_t)e;
    }
    __syncthreads();
    if (tid == 0) {                                            // pass 3
        unsigned e = (unsigned)endlane;
        for (int sg = nseg - 1; sg >= 0; sg--) { segend[sg] = (uint8_t)e; e = segmap[64 * sg + e]; }
    }
    __syncthreads();
    for (int sg = tid; sg < nseg; sg += blockDim.x) {          // pass 4
        const int c0 = sg * TB_SEG, c1 = min(nchunks, c0 + TB_SEG);
        unsigned e = segend[sg];
        for (int c = c1 - 1; c >= c0; c--) { chosen[c] = (uint8_t)e; e = gmap[64 * c + e]; }
    }
    __syncthreads();
    for (int c = tid; c < nchunks; c += blockDim.x) {          // pass 5
        if (c < nchunks - 1) out[2 * c] = dec[(size_t)(2 * c + 1) * 64 + chosen[c]];       // steps 64c+32 .. 64c+63
        if (c >= 1) out[2 * c - 1] = dec[(size_t)(2 * c) * 64 + chosen[c]];                // steps 64c .. 64c+31
    }
}

// ---- single-path traceback, lane = chunk (round 4) --------------------------------------------------------------------------
// The block-parallel traceback above follows ALL 64 candidate end lanes through every chunk (4 VALU + one crossbar round trip per
// step and chunk, 64 candidate outputs written over the history words): 12 G SIMD-cycles and 13.8 GB of HBM traffic per pass to
// emit 63 MB of decoded bits.  But survivors merge: a few constraint lengths back from ANY state the paths are one path.  So every
// chunk is walked ONCE, speculatively: lane j of a wave owns chunk c0 + j, starts from lane 0 at the END of chunk c + 1, walks that
// chunk as a run-in (64 steps = 9 constraint lengths) and arrives at chunk c's last step on what is almost surely the true
// survivor; then walks its own 64 steps, emits the chunk's two output words and notes the lane it entered its chunk with and the
// lane it left it with.  Exactness is not left to the speculation (as with the forward segments): the walk of chunk c is right iff
// it entered with the lane the walk of chunk c + 1 left with, and chunk c + 1 is itself right; the frame's last chunk starts from
// the true end lane.  viterbi3_traceback_check verifies that chain for every chunk at once and re-walks what fails, round by round
// until nothing does -- worst case the sequential cost, result always the sequential decoder's.
// One wave per 64 chunks; the history words of its 65 chunks (33 KB) are staged in LDS, each chunk's row rotated by the chunk's
// phase so that the bit a step rewrites is the same lane bit (step % 6) for every lane of the wave.
constexpr int TB2_CHUNKS = 64;
constexpr int TB2_LDS_WORDS = (TB2_CHUNKS + 1) * 128;
__device__ __host__ inline int vit3_tb2_waves(int len) { return ((len / 64 + 1) + TB2_CHUNKS - 1) / TB2_CHUNKS; }
// per frame: entry lane [nchunks] then exit lane [nchunks] of every chunk's walk (logical lane numbers)
__device__ __host__ inline size_t vit3_tb2_meta_bytes(int len) { return (size_t)3 * (len / 64 + 1); }   // + the chunk's re-encode disagreements

// 64 steps back through one chunk whose rows lie in LDS in rotated order (row word h of rotated lane q at rows[64 h + q]);
// q = rotated lane at the chunk's last step on entry, at its first step on return; ahi / alo: decisions of steps 32..63 / 0..31
__device__ __forceinline__ void vit3_walk_rotated(const uint32_t *rows, unsigned &q, unsigned &ahi, unsigned &alo)
{
    ahi = 0; alo = 0;
#pragma unroll
    for (int S = 63; S >= 0; S--) {
        const int u = S % 6;                                   // the lane bit this step rewrites, in the rotated numbering
        const unsigned v = rows[(S >= 32 ? 64 : 0) + q];
        const unsigned d = (v >> (S & 31)) & 1u;
        if (S >= 32) ahi |= d << (S & 31); else alo |= d << (S & 31);
        q = (q & ~(1u << u)) | (d << u);
    }
}

// the two output words of a chunk from its 64 decisions and the ROTATED lane it was entered with (see vit3_outputs: the last six
// bits are lane bits (PH + 58 + k) % 6 = rotated bits (58 + k) % 6)
__device__ __forceinline__ void vit3_outputs_rotated(unsigned qend, unsigned ahi, unsigned alo, unsigned &ohi, unsigned &olo)
{
    unsigned top = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) top |= ((qend >> ((58 + k) % 6)) & 1u) << k;
    ohi = (ahi >> 6) | (top << 26);
    olo = (alo >> 6) | (ahi << 26);
}

// Re-encode check of ONE chunk (decode.c:234-265 for its 64 steps): the decoded bits of steps 64c .. 64c+63 (olo, ohi) and of the six
// steps before them (prev6: bit k = step -6 + k, the last outputs of the chunk below -- which are bits of THAT chunk's end lane, i.e. of
// the lane this chunk's walk left with) against the signs of the received soft bits; frame bit i = step - 32.  Frame bits 0..5 need the
// frame's LAST bits (tail biting) and are left to the caller.  -> disagreements at unpunctured positions (<= 160)
__device__ __forceinline__ unsigned vit3_chunk_errors(const int *soft, int len, int c, unsigned prev6, unsigned olo, unsigned ohi)
{
    const unsigned long long wlo = (unsigned long long)prev6 | ((unsigned long long)olo << 6) | ((unsigned long long)ohi << 38);
    const unsigned whi = ohi >> 26;
    unsigned errors = 0;
    for (int S = 0; S < 64; S++) {
        const int i = 64 * c + S - VIT_EXTRA;
        if (i < 6 || i >= len) continue;
        const unsigned r = (unsigned)((S <= 57 ? (wlo >> S) : ((wlo >> S) | ((unsigned long long)whi << (64 - S)))) & 0x7full);   // bit 6 = bit i, bit 0 = bit i - 6
        const int w = soft[i];
        const int c0 = (int8_t)w, c1 = (int8_t)(w >> 8), c2 = (int8_t)(w >> 16);
        const int p0 = __popc(r & 0133u) & 1, p1 = __popc(r & 0171u) & 1, p2 = __popc(r & 0165u) & 1;
        errors += ((c0 > 0) != p0) + ((c1 > 0) != p1) + (((i & 1) == 0) && ((c2 > 0) != p2));   // odd i: the third bit is punctured [1,1,1,1,1,0]
    }
    return errors;
}
// the six outputs in front of chunk c from the LOGICAL lane its walk left with: step 58 + k of chunk c - 1 is bit (ph(c-1) + 58 + k) % 6
__device__ __forceinline__ unsigned vit3_prev6(int c, unsigned exit_lane)
{
    const int php = (4 * ((c + 2) % 3)) % 6;                  // phase of chunk c - 1's first step
    unsigned v = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) v |= ((exit_lane >> ((php + 58 + k) % 6)) & 1u) << k;
    return v;
}

// one wave: chunks [64 w, 64 w + 64) of a frame.  lds: TB2_LDS_WORDS dwords.  endlane: the frame's true end lane (logical).
// soft (may be null): the frame's trellis inputs -- each lane then also files the re-encode disagreements of its chunk (meta[2 n + c])
__device__ __forceinline__ void viterbi3_traceback_walk(const uint32_t *dec, int len, int endlane, uint32_t *out, uint8_t *meta, int w, uint32_t *lds, const int *soft = nullptr)
{
    const int lane = threadIdx.x & 63;
    const int nchunks = len / 64 + 1;
    const int c0 = w * TB2_CHUNKS, nrows = min(TB2_CHUNKS + 1, nchunks - c0);   // chunks staged: own + the run-in chunk of the last lane
    // stage: history word h of logical lane L of chunk c0 + k -> lds[128 k + 64 h + rotr6(L, ph)], ph = phase of the chunk's first step
    for (int i = lane; i < nrows * 128; i += 64) {
        const int k = i >> 7, h = (i >> 6) & 1, L = i & 63;
        const int ph = (4 * ((c0 + k) % 3)) % 6;               // (64 c) % 6
        lds[128 * k + 64 * h + (int)rotr6((unsigned)L, ph)] = dec[(size_t)(2 * (c0 + k) + h) * 64 + L];
    }
    WAVE_LDS_SYNC();
    const int c = c0 + lane;
    if (c >= nchunks) return;
    const int ph = (4 * (c % 3)) % 6;
    unsigned q, ahi, alo;
    if (c == nchunks - 1) {
        q = rotr6((unsigned)endlane, ph);
    } else {
        // run-in through chunk c + 1 from (rotated) lane 0 at its last step; its first step's lane is where chunk c ends.
        // rotated numbering of chunk c = that of chunk c + 1 rotated left by 4 (ph(c + 1) - ph(c) = 4 mod 6)
        unsigned qa = 0, t0, t1;
        vit3_walk_rotated(lds + 128 * (lane + 1), qa, t0, t1);
        q = rotl6(qa, 4);
    }
    const unsigned qend = q;
    vit3_walk_rotated(lds + 128 * lane, q, ahi, alo);
    unsigned ohi, olo;
    vit3_outputs_rotated(qend, ahi, alo, ohi, olo);
    if (c < nchunks - 1) out[2 * c] = ohi;                     // steps 64c+32 .. 64c+63
    if (c >= 1) out[2 * c - 1] = olo;                          // steps 64c .. 64c+31
    meta[c] = (uint8_t)rotl6(qend, ph);                        // entered with (logical lane at the chunk's last step)
    const unsigned lexit = rotl6(q, ph);
    meta[nchunks + c] = (uint8_t)lexit;                        // left with (logical lane at its first step)
    if (soft) meta[2 * nchunks + c] = (uint8_t)vit3_chunk_errors(soft, len, c, vit3_prev6(c, lexit), olo, ohi);
}

// one chunk walked from logical lane `e`, history words straight from global memory (the repair path)
__device__ inline unsigned vit3_rewalk(const uint32_t *dec, int c, unsigned e, unsigned &ohi, unsigned &olo)
{   // (the caller re-counts the chunk's disagreements from l, olo, ohi: vit3_chunk_errors)
    const int ph = (4 * (c % 3)) % 6;
    unsigned l = e, ahi = 0, alo = 0;
    for (int S = 63; S >= 0; S--) {
        const int R = (ph + S) % 6;
        const unsigned v = dec[(size_t)(2 * c + (S >> 5)) * 64 + l];
        const unsigned d = (v >> (S & 31)) & 1u;
        if (S >= 32) ahi |= d << (S & 31); else alo |= d << (S & 31);
        l = (l & ~(1u << R)) | (d << R);
    }
    unsigned top = 0;
    for (int k = 0; k < 6; k++) top |= ((e >> ((ph + 58 + k) % 6)) & 1u) << k;
    ohi = (ahi >> 6) | (top << 26);
    olo = (alo >> 6) | (ahi << 26);
    return l;
}

// whole workgroup, after every wave of viterbi3_traceback_walk of the frame is done (a later launch): verify the chain of chunk
// boundaries and re-walk what the speculation got wrong.  stats (may be null): [0] boundaries checked, [1] chunks re-walked.
__device__ inline void viterbi3_traceback_check(const uint32_t *dec, int len, uint32_t *out, uint8_t *meta, int *stats, const int *soft = nullptr)
{
    const int nchunks = len / 64 + 1;
    __shared__ int tb2_bad;
    int rewalked = 0;
    for (int round = 0; round <= nchunks; round++) {
        if (threadIdx.x == 0) tb2_bad = 0;
        __syncthreads();
        // phase A: who entered with the wrong lane?  (reads only)
        int mine[4]; unsigned want[4]; int nm = 0;
        for (int c = threadIdx.x; c < nchunks - 1; c += blockDim.x) {
            const unsigned w = meta[nchunks + c + 1];
            if (meta[c] != w && nm < 4) { mine[nm] = c; want[nm] = w; nm++; }
        }
        if (nm) tb2_bad = 1;
        __syncthreads();
        if (!tb2_bad) break;                                   // block-uniform
        // phase B: re-walk them from the lane their upper neighbour left with (right if that neighbour is right: the topmost wrong
        // chunk of every run is settled for good in this round)
        for (int k = 0; k < nm; k++) {
            const int c = mine[k];
            unsigned ohi, olo;
            const unsigned l = vit3_rewalk(dec, c, want[k], ohi, olo);
            out[2 * c] = ohi;
            if (c >= 1) out[2 * c - 1] = olo;
            meta[c] = (uint8_t)want[k]; meta[nchunks + c] = (uint8_t)l;
            if (soft) meta[2 * nchunks + c] = (uint8_t)vit3_chunk_errors(soft, len, c, vit3_prev6(c, l), olo, ohi);
            rewalked++;
        }
        __threadfence_block();
        __syncthreads();
    }
    if (stats) { if (threadIdx.x == 0) atomicAdd(&stats[0], nchunks - 1); if (rewalked) atomicAdd(&stats[1], rewalked); }
}

}  // namespace nrsc5
