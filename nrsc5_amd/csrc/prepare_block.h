// Per-block bookkeeping that precedes the FFTs (top of acquire_process, acquire.c:98-119,153-168): decides
// whether the stream has a complete 33-symbol window, picks timing / CFO for the block from the tracking
// feedback (FINE) or from the coarse acquisition results, opens the block's record.  Runs either as its own
// tiny kernel (k_prepare) or, in the batch pipeline, in the tail of the previous block's k_sync.
#pragma once
#include "kernels.h"
#include "fastmath.h"

namespace nrsc5 {

// PSMI -> compatibility mode (Table 6-4 of 1011s, sync.c:29-35)
__device__ inline int compat_mode_for_psmi(int psmi)
{
    const int m6 = psmi & 63, low = m6 & 15;
    // compatibility_mode[]: 0,1,2,3,1,5,6,5,6,1,2,11,1,5,6,5 then the same 16 entries repeating with [16k] = 6;
    // one nibble per entry
    constexpr unsigned long long TAB16 = 0x5651B21656513210ull;
    int mode = (int)((TAB16 >> (4 * low)) & 15ull);
    if (low == 0 && m6 != 0) mode = 6;
    return mode;
}

// ... -> partitions per sideband that sync_process_fm tracks, equalises and measures (sync.c:343-358)
__device__ inline int partitions_for_psmi(int psmi)
{
    switch (compat_mode_for_psmi(psmi)) {
    case 2: return 11;
    case 3: return 12;
    case 5: case 6: case 11: return 14;
    default: return 10;
    }
}

// ... -> partitions per sideband whose soft bits are ROUTED (primary main + PX1 / PX2, sync.c:537-596): the compatibility modes
// 5 and 6 equalise 14 partitions like MP11 but the reference hands none of their extended partitions to the decoder
__device__ inline int routed_partitions_for_psmi(int psmi)
{
    switch (compat_mode_for_psmi(psmi)) {
    case 2: return 11;
    case 3: return 12;
    case 11: return 14;
    default: return 10;
    }
}

template <typename S> __device__ inline bool window_ready(const S &st) { return st.wr - st.rd >= WIN_N; }

// The words of the stream state the bookkeeping reads, fetched in ONE burst.  Read through the state itself -- each field where the logic
// first needs it, behind the branch that decides whether it is needed -- they were up to eight dependent round trips to L2 by the single
// work-item that runs this, at the tail of every k_sync and at the head of every fast-seam k_mixfft.
struct PrepView {
    long long wr, rd;
    int sync_state, samperr, cfo, coarse_samperr;
    float angle, prev_angle, coarse_re, coarse_im;
    double theta, dtheta, growth;
    float nco_re, nco_im;
    int nco_exact, nblocks;
};
__device__ __forceinline__ PrepView prep_view(const StreamState &st)
{
    PrepView v;
    v.wr = st.wr; v.rd = st.rd; v.sync_state = st.sync_state; v.samperr = st.samperr; v.cfo = st.cfo; v.coarse_samperr = st.coarse_samperr;
    v.angle = st.angle; v.prev_angle = st.prev_angle; v.coarse_re = st.coarse_re; v.coarse_im = st.coarse_im; v.theta = st.theta; v.dtheta = st.dtheta; v.growth = st.growth;
    v.nco_re = st.nco_re; v.nco_im = st.nco_im; v.nco_exact = st.nco_exact; v.nblocks = st.nblocks;
    return v;
}

// What the top of acquire_process decides for the block at st.rd, as values (nothing is written): the symbol kernel of the fast
// streaming seam computes them for itself -- every workgroup from the same unchanged state -- and the sync kernel that follows
// commits them with prepare_block; as a launch of its own (k_prepare) the bookkeeping was ~5 us of a chain the host waits for.
struct Prepared {
    int active, pending;        // process a block this step / a complete window is waiting (for the acquisition kernels)
    int samperr;                // timing pick
    float prev_angle;           // carrier angle after this block's update
    int to_coarse;              // the block moves the stream from NONE to COARSE
    double dtheta, theta;       // NCO step and start phase of the block
    double growth;              // |phase_increment| - 1: the reference's oscillator grows / shrinks by this much per sample until it is renormalised at the symbol's end (acquire.c:250-252)
    int nco_mode;               // 1: this block's phasors are the reference's own float recurrence (k_nco_exact), from the start state below
    float ph_re, ph_im;         // ... acquire_t.phase after the block-start rotation (acquire.c:166), bit for bit
    float inc_c, inc_s;         // phase_increment (acquire.c:168)
};

// acq_ran: the acquisition kernels ran in this step, i.e. coarse_samperr / coarse_re / coarse_im belong to the window at
// st.rd.  A stream that is not FINE only advances on such steps (the host launches them whenever counters[1] > 0 at the
// last burst boundary); anywhere else it waits -- never a block on stale coarse results.
// exact-oscillator mode for this block?  Only while the float state is still the reference's own (nco_exact: every block since the reset ran exact)
template <typename S> __device__ inline bool nco_wants_exact(const S &st, int policy)
{
    if (!st.nco_exact) return false;
    return policy == NCO_EXACT_ALWAYS || (policy == NCO_EXACT_UNTIL_FINE && st.sync_state != SYNC_FINE) || (policy == NCO_EXACT_FIRST_BLOCK && st.nblocks == 0);
}

template <typename S> __device__ inline Prepared prepare_values_of(const S &st, bool acq_ran, int nco_policy = NCO_CLOSED_FORM)
{
    Prepared p;
    p.nco_mode = 0; p.ph_re = 1.0f; p.ph_im = 0.0f; p.inc_c = 1.0f; p.inc_s = 0.0f;
    const bool ready = window_ready(st);
    p.active = (ready && (st.sync_state == SYNC_FINE || acq_ran)) ? 1 : 0;
    p.pending = ready ? 1 : 0;
    p.samperr = 0; p.prev_angle = st.prev_angle; p.to_coarse = 0; p.dtheta = st.dtheta; p.theta = st.theta; p.growth = st.growth;
    if (!p.active) return p;
    float angle;
    if (st.sync_state == SYNC_FINE) {
        p.samperr = SYM_N / 2 + st.samperr;                    // acquire.c:112-113
        const float angle_diff = -st.angle;
        angle = st.prev_angle + angle_diff;
    } else {
        p.samperr = st.coarse_samperr;
        // angle_diff = arg(max_v * e^{-i prev_angle})        (acquire.c:153)
        float sn, cs; ref_sincosf(-st.prev_angle, sn, cs);     // cexpf(I * -prev_angle): glibc's sincosf restated (fastmath.h), bit for bit
        const float pr = st.coarse_re * cs - st.coarse_im * sn;
        const float pi = st.coarse_re * sn + st.coarse_im * cs;
#ifdef NRSC5HIP_ACCURATE_TRIG
        const float angle_diff = (float)atan2((double)pi, (double)pr);
#else
        const float angle_diff = ref_atan2f(pi, pr);
#endif
        const float angle_factor = (st.prev_angle != 0.0f) ? 0.25f : 1.0f;
        angle = st.prev_angle + (angle_diff * angle_factor);
        p.to_coarse = st.sync_state != SYNC_COARSE;
    }
    p.prev_angle = angle;
    angle = (float)((double)angle - 2 * M_PI * st.cfo);        // acquire.c:164
    const float dtheta = angle / FFT_N;
    // The reference rotates by the float pair (cosf, sinf)(dtheta) once per sample (acquire.c:168,250);
    // the angle of that rounded unit vector, not dtheta itself, is its effective NCO step.
    // (round 6) phase_increment = cexpf(angle / fft * I) is glibc's float sincosf, a 0.56-ulp function: a correctly rounded cosine (what rounds 3 - 5 used here:
    // a double series rounded once) is a DIFFERENT float for 1.3 % of arguments, and one ulp of inc_c is 6e-8 of |phase_increment| -- which the reference multiplies
    // up 2160 times per symbol (the amplitude ramp below): a block whose inc_c was the other float carried a ramp off by up to 1.3e-4 across every symbol, the size
    // of difference the CFO search is known to amplify (DESIGN (c) limit 2).  ref_sincosf (fastmath.h) IS glibc's function, bit for bit.
    float inc_c, inc_s;
    ref_sincosf(dtheta, inc_s, inc_c);
    if (fabsf(dtheta) < 0.25f) p.dtheta = small_atan((double)inc_s / (double)inc_c);   // |integer CFO| up to 80 bins: always, in practice
    else p.dtheta = atan2((double)inc_s, (double)inc_c);
    // ... and its LENGTH, 1 + g with |g| up to 6e-8, is the oscillator's amplitude: phase *= phase_increment 2160 times between two
    // renormalisations (acquire.c:250-252) makes the amplitude run as (1 + g)^j across the symbol, up to 1.3e-4 at its last sample -- deterministic,
    // and 100 x the rounding noise of the recurrence.  The symbol kernel gives its closed-form phasor the same ramp (k_mixfft: nco_ramp).
    p.growth = sqrt((double)inc_c * (double)inc_c + (double)inc_s * (double)inc_s) - 1.0;
    // phase *= e^{-i (1080 - samperr) angle / 2048}            (acquire.c:166)
    const float rot = -(float)(SYM_N / 2 - p.samperr) * angle / FFT_N;
    double th = st.theta + (double)rot;
    th -= 2 * M_PI * rint(th / (2 * M_PI));
    p.theta = th;
    p.inc_c = inc_c; p.inc_s = inc_s;
    if (nco_wants_exact(st, nco_policy)) {
        // st->phase *= cexpf(rot * I) as the reference computes it: glibc's cexpf is sincosf of the float argument (ref_sincosf: the same function, restated),
        // the product the plain four-multiplication form in float (gcc -O3, no contraction; oracle/Makefile)
        float rc, rs; ref_sincosf(rot, rs, rc);
        const float a = st.nco_re, b = st.nco_im;
        const float ac = a * rc, bd = b * rs, ad = a * rs, bc = b * rc;
        p.ph_re = ac - bd; p.ph_im = ad + bc;
        p.nco_mode = 1;
    }
    return p;
}
__device__ inline Prepared prepare_values(const StreamState &st, bool acq_ran, int nco_policy = NCO_CLOSED_FORM) { return prepare_values_of(prep_view(st), acq_ran, nco_policy); }

__device__ inline void prepare_block(const DevBuffers &db, StreamState &st, int s, bool acq_ran)
{
    const int was_active = st.active, psmi = st.psmi, nblocks = st.nblocks;
    const PrepView v = prep_view(st);                          // (with the three words above: one burst of loads)
    if (was_active) return;                                    // already prepared (fused into the previous k_sync)
    if (v.sync_state != SYNC_FINE) atomicAdd(&db.counters[1], 1);   // host: keep launching acquisition
    if (v.sync_state != SYNC_FINE || routed_partitions_for_psmi(psmi) > PM_PART) atomicAdd(&db.counters[2], 1);   // ... and the PX kernels
    const Prepared p = prepare_values_of(v, acq_ran, db.nco_tab ? db.nco_policy : NCO_CLOSED_FORM);
    st.active = p.active;
    if (!p.active) {
        if (p.pending) atomicAdd(&db.counters[0], 1);          // work is pending: the host must keep stepping
        return;
    }
    atomicAdd(&db.counters[0], 1);

    BlockRecord &rec = db.records[(size_t)s * db.rec_cap + (nblocks % db.rec_cap)];
    BlockRecord r;
    r.flags = REC_PROCESSED; r.state_before = v.sync_state; r.state_after = 0;
    r.samperr = 0; r.cfo = 0; r.keep = 0; r.bc = 0; r.psmi = 0; r.cfo_wait = 0; r.next_samperr = 0;
    r.prev_angle = 0; r.phase_re = 0; r.phase_im = 0; r.next_angle = 0; r.freq_offset = 0; r.mer_lb = 0; r.mer_ub = 0;
    r.ber = 0; r.p1_slot = -1; r.bc_decoded = -1; r.pids[0] = r.pids[1] = r.pids[2] = 0; r.sis = 0;
    if (v.sync_state == SYNC_FINE) { st.samperr = 0; st.angle = 0; }
    else if (p.to_coarse) { r.flags |= REC_TO_COARSE; st.sync_state = SYNC_COARSE; }
    st.prev_angle = p.prev_angle;
    rec = r;
    st.samperr_cur = p.samperr;
    st.dtheta = p.dtheta;
    st.growth = p.growth;
    st.nco_mode = p.nco_mode;
    if (p.nco_mode) { st.nco_re = p.ph_re; st.nco_im = p.ph_im; st.inc_re = p.inc_c; st.inc_im = p.inc_s; }
    else st.nco_exact = 0;                                     // a block on the closed-form phasor: the float state is no longer the reference's
    st.theta = p.theta;
}

}  // namespace nrsc5
