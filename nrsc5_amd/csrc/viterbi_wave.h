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

}  // namespace nrsc5
