// K=7 rate-1/3 tail-biting soft Viterbi on ONE wave64: lane = trellis state.
//
// Replaces the reference's conv_dec.c:402-453 (schedule + traceback) and the SSE/NEON/generic
// ACS of conv_sse.h:233-323 / conv_gen.h:32-101 outright:
//   * path metrics live one per lane in a VGPR (int32, so the reference's every-79-steps
//     min-normalisation -- needed only to keep int16 from overflowing -- is unnecessary;
//     decisions depend on metric differences only and are bit-identical),
//   * the butterfly reads its two predecessor metrics 2b / 2b+1 with two cross-lane reads,
//   * the 64 survivor decisions of a step are one __ballot -> one 64-bit word
//     (1.17 MB per P1 frame instead of the reference's 18.7 MB int16 path matrix),
//   * ties pick predecessor 2b+1 exactly as `if (sum0 > sum1)` does (conv_gen.h:47-53).
#pragma once
#include <type_traits>
#include "nrsc5_dev.h"
#include "wave_ops.h"

namespace nrsc5 {

// Per-lane constants of the butterfly that produces new state `lane`.
struct VitLane {
    int sg0, sg1, sg2;   // +-1: expected NRZ outputs on edge 2b -> b, negated for the upper half
    int src_e, src_o;    // lanes holding the predecessor metrics
};

__device__ inline VitLane vit_lane_k7(int lane)
{
    const int b = lane & 31;
    const unsigned reg = ((unsigned)b << 1) & 0x3e;            // gen_state_info, conv_dec.c:137-153
    const int flip = (lane >> 5) ? -1 : 1;                     // states >= 32 use -metric (acs_butterfly)
    VitLane v;
    v.sg0 = flip * ((__popc(reg & 0133u) & 1) ? 1 : -1);       // generators decode.c:39-45
    v.sg1 = flip * ((__popc(reg & 0171u) & 1) ? 1 : -1);
    v.sg2 = flip * ((__popc(reg & 0165u) & 1) ? 1 : -1);
    v.src_e = 2 * b; v.src_o = 2 * b + 1;
    return v;
}

// Forward pass + traceback.  `coded` = 3*len soft values (punctured = 0), any address space the
// caller can read generically; `dec` = len+64 decision words of scratch; `out` = ceil(len/32)
// packed words, bit i of the frame at out[i>>5] bit (i&31).  Must be called by all 64 lanes.
__device__ inline void viterbi_k7_wave(const int8_t *coded, int len, unsigned long long *dec, uint32_t *out)
{
    const int lane = threadIdx.x & 63;
    const VitLane vl = vit_lane_k7(lane);
    const int steps = len + 2 * VIT_EXTRA;
    const int j0 = len - VIT_EXTRA;                            // conv_dec.c:407-408
    const int nchunks = (steps + 63) >> 6;
    int pm = 0;                                                // reset_decoder: all-zero for tail biting

    for (int c = 0; c < nchunks; c++) {
        const int t0 = c << 6;
        int s0 = 0, s1 = 0, s2 = 0;
        if (t0 + lane < steps) {
            const int j = (j0 + t0 + lane) % len;
            s0 = coded[3 * j]; s1 = coded[3 * j + 1]; s2 = coded[3 * j + 2];
        }
        int wlo = 0, whi = 0;
        const int nst = min(64, steps - t0);
#pragma unroll 8
        for (int s = 0; s < nst; s++) {
            const int a0 = wave_readlane(s0, s), a1 = wave_readlane(s1, s), a2 = wave_readlane(s2, s);
            const int m = a0 * vl.sg0 + a1 * vl.sg1 + a2 * vl.sg2;
            const int e = __shfl(pm, vl.src_e), o = __shfl(pm, vl.src_o);
            const int pa = e + m, pc = o - m;
            const bool take_e = pa > pc;
            pm = take_e ? pa : pc;
            const unsigned long long w = __ballot(!take_e);   // bit = 1: survivor came from 2b+1
            wlo = wave_writelane(wlo, (int)(uint32_t)w, s);
            whi = wave_writelane(whi, (int)(uint32_t)(w >> 32), s);
        }
        if (t0 + lane < steps)
            dec[t0 + lane] = ((unsigned long long)(uint32_t)whi << 32) | (uint32_t)wlo;
    }

    // end state: first maximum wins (conv_dec.c:310-318)
    const int best = wave_max_i32(pm);
    unsigned state = (unsigned)(__ffsll((long long)__ballot(pm == best)) - 1);

    for (int c = nchunks - 1; c >= 0; c--) {
        const int t0 = c << 6;
        unsigned long long mine = (t0 + lane < steps) ? dec[t0 + lane] : 0ull;
        const int mlo = (int)(uint32_t)mine, mhi = (int)(uint32_t)(mine >> 32);
        unsigned long long obits = 0;
        const int nst = min(64, steps - t0);
        for (int s = nst - 1; s >= 0; s--) {
            const int t = t0 + s;
            const uint32_t lo = (uint32_t)wave_readlane(mlo, s), hi = (uint32_t)wave_readlane(mhi, s);
            const unsigned bit = (state < 32 ? (lo >> state) : (hi >> (state - 32))) & 1u;
            if (t >= VIT_EXTRA && t < len + VIT_EXTRA)
                obits |= (unsigned long long)((state >> 5) & 1u) << s;     // vals[state], conv_dec.c:275-282
            state = ((state << 1) & 0x3eu) | bit;                           // vstate_lshift
        }
        if (lane == 0) {
            const int wl = 2 * c - 1, wh = 2 * c;              // (t0 - 32) / 32 and the next word
            if (wl >= 0 && wl * 32 < len) out[wl] = (uint32_t)obits;
            if (wh * 32 < len) out[wh] = (uint32_t)(obits >> 32);
        }
    }
}


// =====================================================================================================
// Fast path (len % 64 == 0: P1 146176, P3/P4 4608 / 2304): same trellis, same tie rule, same output bits,
// restructured for the gfx950 register file:
//
//  * ROTATING LAYOUT.  Before step t lane L holds the metric of state rotr6^t(L).  The two predecessors
//    2b, 2b+1 of a butterfly then always sit in lanes L and L ^ (1 << (t % 6)), so the exchange is a
//    DPP / v_permlane swap (lane_xor<>) instead of two ds_bpermute round trips through the LDS crossbar,
//    and the new metric max(own + m, partner - m) belongs in the same lane (state rotr6^(t+1)(L)).
//  * the branch metric is one v_dot4_i32_i8 of the packed soft triple with per-lane, per-phase signs.
//  * the stored decision bit is "the survivor came through MY lane" (own-wins), so the traceback runs in
//    lane space: lane ^= (own ? 0 : 1 << (t % 6)); the decoded bit of step t is bit (t % 6) of the lane.
//    Tie rule of the reference (`if (sum0 > sum1)`, ties -> predecessor 2b+1): own is the even
//    predecessor when bit0(state) = 0 (strict >) and the odd one when bit0 = 1 (>=), i.e. X + s0 > Y.
//  * 64 steps are unrolled at compile time so lane selects are inline constants (v_readlane/v_writelane).
// =====================================================================================================

__device__ __forceinline__ unsigned rotr6(unsigned v, int k) { k %= 6; return k ? (((v >> k) | (v << (6 - k))) & 63u) : (v & 63u); }
__device__ __forceinline__ unsigned rotl6(unsigned v, int k) { k %= 6; return k ? (((v << k) | (v >> (6 - k))) & 63u) : (v & 63u); }

struct VitFastConst { int sgw[6]; int s0[6]; };

__device__ inline VitFastConst vit_fast_consts(int lane)
{
    VitFastConst k;
#pragma unroll
    for (int r = 0; r < 6; r++) {
        const unsigned S = rotr6((unsigned)lane, r);           // state held by this lane in phase r
        const unsigned reg = S & 0x3eu;                        // edge 2b -> b (gen_state_info, conv_dec.c:137-153)
        const int g0 = (__popc(reg & 0133u) & 1) ? 1 : -1, g1 = (__popc(reg & 0171u) & 1) ? 1 : -1, g2 = (__popc(reg & 0165u) & 1) ? 1 : -1;
        k.sgw[r] = (g0 & 0xff) | ((g1 & 0xff) << 8) | ((g2 & 0xff) << 16);
        k.s0[r] = (int)(S & 1u);
    }
    return k;
}

// one trellis step in phase R with branch metric m; parks the previous step's ballot in lane PARK
template <int R, int PARK>
__device__ __forceinline__ unsigned long long vit_fast_step(int &pm, int m, const VitFastConst &k, unsigned long long prev, int &wlo, int &whi)
{
    const int X = pm + m;
    const int Y = lane_xor<(1 << R)>(pm) - m;
    return acs_select_park<PARK>(pm, X, X + k.s0[R], Y, prev, wlo, whi);   // own-wins: X + s0 > Y
}

template <int R> __device__ __forceinline__ int vit_branch_metric(int aw, int s, const VitFastConst &k)
{
    return dot4_i8(wave_readlane(aw, s), k.sgw[R], 0);
}

template <int PH0, int S> struct VitFwd {
    static __device__ __forceinline__ void run(int &pm, int aw, const VitFastConst &k, int m, unsigned long long prev, int &wlo, int &whi)
    {
        // branch metric of the NEXT step first: it does not depend on the path metrics, so it fills the
        // dependency stalls of this step's exchange/compare chain
        const int m_next = (S + 1 < 64) ? vit_branch_metric<(PH0 + S + 1) % 6>(aw, (S + 1) & 63, k) : 0;
        // S == 0 parks a dummy (lane 0 is rewritten by step 1 with the real ballot of step 0)
        const unsigned long long b = vit_fast_step<(PH0 + S) % 6, (S == 0 ? 0 : S - 1)>(pm, m, k, prev, wlo, whi);
        VitFwd<PH0, S + 1>::run(pm, aw, k, m_next, b, wlo, whi);
    }
};
template <int PH0> struct VitFwd<PH0, 64> {
    static __device__ __forceinline__ void run(int &, int, const VitFastConst &, int, unsigned long long prev, int &wlo, int &whi)
    {
        park_ballot<63>(prev, wlo, whi);
    }
};

template <int PH0, int S> struct VitBack {
    static __device__ __forceinline__ void run(unsigned &l, int mlo, int mhi, unsigned &ohi, unsigned &olo)
    {
        constexpr int R = (PH0 + S) % 6;
        const unsigned long long w = ((unsigned long long)(uint32_t)wave_readlane(mhi, S) << 32) | (uint32_t)wave_readlane(mlo, S);
        const unsigned own = (unsigned)(w >> l) & 1u;
        if (S >= 32) ohi = (ohi << 1) | ((l >> R) & 1u); else olo = (olo << 1) | ((l >> R) & 1u);
        l ^= own ? 0u : (1u << R);
        VitBack<PH0, S - 1>::run(l, mlo, mhi, ohi, olo);
    }
};
template <int PH0> struct VitBack<PH0, -1> {
    static __device__ __forceinline__ void run(unsigned &, int, int, unsigned &, unsigned &) {}
};

__device__ __forceinline__ int vit_load_soft(const int8_t *coded, int len, int t)
{
    const int j = (len - VIT_EXTRA + t) % len;                 // conv_dec.c:407-412
    return ((int)(uint8_t)coded[3 * j]) | ((int)(uint8_t)coded[3 * j + 1] << 8) | ((int)(uint8_t)coded[3 * j + 2] << 16);
}

// requires len % 64 == 0; all 64 lanes
__device__ inline void viterbi_k7_wave_fast(const int8_t *coded, int len, unsigned long long *dec, uint32_t *out, int phases = 3)
{
    const int lane = threadIdx.x & 63;
    const VitFastConst k = vit_fast_consts(lane);
    const int nchunks = len / 64 + 1;                          // steps = len + 2 * VIT_EXTRA
    int pm = 0;
    int aw = vit_load_soft(coded, len, lane);
    for (int c = 0; c < ((phases & 1) ? nchunks : 0); c++) {
        const int aw_next = (c + 1 < nchunks) ? vit_load_soft(coded, len, 64 * (c + 1) + lane) : 0;
        int wlo = 0, whi = 0;
        switch (c % 3) {                                       // (64 c) % 6
        case 0: VitFwd<0, 0>::run(pm, aw, k, vit_branch_metric<0>(aw, 0, k), 0ull, wlo, whi); break;
        case 1: VitFwd<4, 0>::run(pm, aw, k, vit_branch_metric<4>(aw, 0, k), 0ull, wlo, whi); break;
        default: VitFwd<2, 0>::run(pm, aw, k, vit_branch_metric<2>(aw, 0, k), 0ull, wlo, whi); break;
        }
        dec[64 * c + lane] = ((unsigned long long)(uint32_t)whi << 32) | (uint32_t)wlo;
        aw = aw_next;
    }

    // end state: first maximum in STATE order (conv_dec.c:310-318); lane L holds state rotr6^steps(L)
    const int rend = (64 * nchunks) % 6;
    const int best = wave_max_i32(pm);
    const int smin = wave_min_i32(pm == best ? (int)rotr6((unsigned)lane, rend) : 64);
    unsigned l = (unsigned)wave_uniform((int)rotl6((unsigned)smin, rend));   // lane of the survivor, kept in an SGPR

    for (int c = ((phases & 2) ? nchunks - 1 : -1); c >= 0; c--) {
        const unsigned long long mine = dec[64 * c + lane];
        const int mlo = (int)(uint32_t)mine, mhi = (int)(uint32_t)(mine >> 32);
        unsigned ohi = 0, olo = 0;
        l = (unsigned)wave_uniform((int)l);                    // keep the survivor chain on the scalar ALU
        switch (c % 3) {
        case 0: VitBack<0, 63>::run(l, mlo, mhi, ohi, olo); break;
        case 1: VitBack<4, 63>::run(l, mlo, mhi, ohi, olo); break;
        default: VitBack<2, 63>::run(l, mlo, mhi, ohi, olo); break;
        }
        // steps 64c+32..64c+63 -> out word 2c; steps 64c..64c+31 -> out word 2c-1 (bit = step & 31)
        if (lane == 0) {
            if (c < nchunks - 1) out[2 * c] = ohi;
            if (c >= 1) out[2 * c - 1] = olo;
        }
    }
}

// The same trellis in COMPACT form for short frames whose len + 64 steps are a multiple of six (the PIDS frame: 80 bits, 144 steps):
// a rolled loop over groups of six steps -- one per phase of the rotating layout, so the lane exchanges stay compile-time DPP /
// permlane forms -- instead of 64-step unrolled instruction streams.  Identical metrics, tie rule, tail-biting schedule and
// decision words, hence identical output.  Why: the PIDS decode sits on the streaming seam's block-step chain.  On the reference-
// shaped ds_bpermute form (`viterbi_k7_wave`: two crossbar round trips per step on the serial chain) it took 22.7 us per block; the
// unrolled fast path with a 16-step last chunk 19 us -- ~19 KB of straight-line code executed once per launch, all of it
// instruction-cache misses (profiles/r04_dropin_timeline.txt).  This form is ~1 KB.
// The history stays in the lane -- five VGPRs (the fast path parks 64-lane ballots instead: a ballot store per step); the traceback fetches the word of the survivor's
// lane with v_readlane and runs on the scalar ALU.  `dec` is not used.
template <int len>
__device__ __forceinline__ void viterbi_k7_wave_compact(const int8_t *coded, unsigned long long *dec, uint32_t *out)
{
    static_assert(len + 2 * VIT_EXTRA <= 150 && (len + 2 * VIT_EXTRA) % 6 == 0, "five map words of five groups, whole groups of six");
    const int lane = threadIdx.x & 63;
    const VitFastConst k = vit_fast_consts(lane);
    constexpr int steps = len + 2 * VIT_EXTRA;
    // soft inputs: lane i of word c holds step 60 c + i (60 = ten groups of six: a group never straddles two words)
    int aw[3];
#pragma unroll
    for (int c = 0; c < 3; c++) aw[c] = (60 * c + lane < steps) ? vit_load_soft(coded, len, 60 * c + lane) : 0;
    int pm = 0;
    // Survivor MAPS instead of decision bits (round 6): over the six steps of a group -- one per phase of the rotating layout -- every lane carries along the lane its survivor
    // path STARTED the group in (own wins: keep it; the partner wins: take the partner's, one more register-file move off the metric chain), so the traceback is one v_readlane
    // per GROUP, 24 hops, where it was one per step (144 SALU <-> VALU round trips, ~5 000 shader cycles of a chain the streaming seam's host waits for).  Word j: groups
    // 5 j .. 5 j + 4, six bits each.  Composing the steps' choices forwards or walking them backwards names the same path: identical output.
    uint32_t maps[5] = {0, 0, 0, 0, 0};
    auto fwd = [&](auto R, int awc, int i, int &orig) __attribute__((always_inline)) {
        constexpr int r = decltype(R)::value;
        const int m = dot4_i8(wave_readlane(awc, i), k.sgw[r], 0);
        const int X = pm + m;
        const int Y = lane_xor<(1 << r)>(pm) - m;
        const int po = lane_xor<(1 << r)>(orig);
        const bool own = X + k.s0[r] > Y;                      // own-wins; ties as the reference's `if (sum0 > sum1)`
        pm = own ? X : Y;
        orig = own ? orig : po;
    };
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int awc = aw[j >> 1], i0 = 30 * (j & 1);
#pragma unroll 1
        for (int g = 0; g < 5; g++) {
            if (30 * j + 6 * g >= steps) break;                // wave-uniform (the last word of a 144-step frame holds four groups)
            const int i = i0 + 6 * g;
            int orig = lane;
            fwd(std::integral_constant<int, 0>{}, awc, i, orig);     fwd(std::integral_constant<int, 1>{}, awc, i + 1, orig);
            fwd(std::integral_constant<int, 2>{}, awc, i + 2, orig); fwd(std::integral_constant<int, 3>{}, awc, i + 3, orig);
            fwd(std::integral_constant<int, 4>{}, awc, i + 4, orig); fwd(std::integral_constant<int, 5>{}, awc, i + 5, orig);
            maps[j] |= (uint32_t)orig << (6 * g);
        }
    }
    // end state: first maximum in STATE order (conv_dec.c:310-318); lane L holds state rotr6^steps(L) = L
    const int best = wave_max_i32_rf(pm);
    const int smin = wave_min_i32_rf(pm == best ? lane : 64);
    unsigned l = (unsigned)wave_uniform(smin);                 // lane of the survivor, kept in an SGPR
    // the decoded bits of a group of six steps are the survivor's lane bits at the group's last step (bit r at step r of the group: each step only rewrites its own bit
    // afterwards); the walk is unrolled, so every group lands in the output words with constant shifts -- no memory, no barrier: the decoder can run on one wave of a larger workgroup
    uint32_t o[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 4; j >= 0; j--) {
#pragma unroll
        for (int g = 4; g >= 0; g--) {
            if (30 * j + 6 * g >= steps) continue;             // compile-time
            const unsigned six = l & 63u;
#pragma unroll
            for (int r = 0; r < 6; r++) {
                const int ob = 30 * j + 6 * g + r - VIT_EXTRA;  // the frame bit step 30 j + 6 g + r decodes
                if (ob >= 0 && ob < len) o[ob >> 5] |= ((six >> r) & 1u) << (ob & 31);
            }
            l = (unsigned)wave_uniform((int)(((uint32_t)wave_readlane((int)maps[j], (int)l) >> (6 * g)) & 63u));      // where that path stood before the group
        }
    }
    (void)dec;
    if (lane == 0) {
#pragma unroll
        for (int w = 0; w < (len + 31) / 32; w++) out[w] = o[w];
    }
}

// The P1 frame's split form -- forward pass by one wave + block-parallel traceback -- lives in viterbi_v3.h.
constexpr int TB_SEG = 40;                                     // chunks per segment of the block-parallel traceback: a P1 frame's 2285 chunks make 58
                                                               // segments (<= 64); the two sequential compositions chase 40 map entries each

// dispatcher used by the kernels
__device__ inline void viterbi_k7_decode(const int8_t *coded, int len, unsigned long long *dec, uint32_t *out, int phases = 3)
{
    if ((len & 63) == 0) viterbi_k7_wave_fast(coded, len, dec, out, phases);
    else if (len == PIDS_LEN) viterbi_k7_wave_compact<PIDS_LEN>(coded, dec, out);
    else viterbi_k7_wave(coded, len, dec, out);
}

}  // namespace nrsc5
