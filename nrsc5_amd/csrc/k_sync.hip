// K5 (+ PIDS part of K6/K7/K8) -- per-block synchronisation, equalisation and soft demodulation for
// gfx950.  One workgroup per stream and block; replaces sync_process_fm (sync.c:339-610), its helpers
// adjust_ref / decode_ref_fm / find_ref_fm / detect_cfo / adjust_data (sync.c:90-337), sync_adjust
// (sync.c:769-777), decode_push_pm + decode_process_pids (decode.c:378-391,463-472) and the tail of
// acquire_process (acquire.c:259-262).
//
// Parallel axes inside the workgroup: lanes = reference carriers for the Costas loops (sequential in
// the 32 symbols by construction), lanes = (partition, symbol, carrier) cells for equalisation / MER /
// soft bits, lanes = live bins for the brute-force CFO search, one wave for the 144-step PIDS trellis.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "wave_ops.h"
#include "viterbi_wave.h"
#include "prepare_block.h"
#include "l2_header.h"
#include "fastmath.h"
#include "flow_ops.h"
#include "mixfft_body.h"                                       // (defines stream_of; the symbol transform for k_flow at the end of this file)

namespace nrsc5 {

constexpr int NREF_MAX = 30;          // 15 per sideband (14 partitions + 1)
constexpr int CFO_LO = -2 * PW, CFO_HI = 2 * PW;   // candidates -38..37 (sync.c:294)

__device__ inline int ref_bin(int r) { const int i = r >> 1; return (r & 1) ? UB1 - PW * i : LB0 + PW * i; }

// sync word used to resolve the pi ambiguity (sync.c:96-99): +1 / -1 masks over the 32 symbols
constexpr uint32_t PAT_POS = (1u << 1) | (1u << 5) | (1u << 6) | (1u << 8) | (1u << 21);
constexpr uint32_t PAT_NEG = (1u << 0) | (1u << 2) | (1u << 3) | (1u << 4) | (1u << 9) | (1u << 13) | (1u << 14) | (1u << 20) | (1u << 22) | (1u << 31);
// needle of decode_ref_fm / find_ref_fm (sync.c:171-174): fixed positions and their values (rsid bits added per ref)
constexpr uint32_t NEEDLE_MASK = 0x7fu | (0xfu << 8) | (3u << 13) | (7u << 20) | (1u << 31);
constexpr uint32_t NEEDLE_VAL0 = (1u << 1) | (1u << 5) | (1u << 6) | (1u << 8) | (1u << 21);
__device__ inline uint32_t needle_val(unsigned rsid) { return NEEDLE_VAL0 | ((rsid >> 1) << 10) | ((((rsid >> 1) ^ rsid) & 1u) << 11); }

struct LoopGains { float alpha, beta; };
__device__ inline LoopGains loop_gains()                      // sync.c:832-841
{
    const float loop_bw = 0.05f, damping = 0.70710678f;
    const float denom = 1 + (2 * damping * loop_bw) + (loop_bw * loop_bw);
    LoopGains g; g.alpha = (4 * damping * loop_bw) / denom; g.beta = (4 * loop_bw * loop_bw) / denom;
    return g;
}

// ---- the loops as the reference computes them, operation for operation (round 6; DevBuffers::loop_exact) ----------------------------------------
// One step of adjust_ref (sync.c:101-113) on z with loop state (freq, phase): cexpf is glibc's sincosf of the float argument (ref_sincosf), cargf its atan2f
// (ref_atan2f), the complex products gcc's four multiplications and two sums in float, none contracted (-ffp-contract=off here, no FMA in the reference's
// x86-64 baseline build), the wrap of the phase a double comparison / double difference rounded once.  The fast form above it in this file reaches the same
// values to ~5e-7 (v_sin / v_cos, a 4-term arc tangent, e^{2i phase} by the double-angle identities); this one reaches them to the last bit WHEN its inputs
// are the reference's -- and where they are not quite (the transform's own rounding), it at least adds no difference of its own for the CFO search to amplify.
// (The two sincosf evaluations are independent and interleave; in the CFO search that only fits the register budget because the exact visits have a loop of their own --
// k_sync below: sharing one loop with the fast form spilled four VGPRs, a private segment EVERY launch of the sync kernel would pay for.)
__device__ __forceinline__ float2 costas_step_exact(const float2 z, float &freq, float &phase, float cfo_freq, const LoopGains g)
{
    float s2, c2; ref_sincosf(-(2.0f * phase), s2, c2);        // cexpf(-I * 2 * phase)
    float s1, c1; ref_sincosf(-phase, s1, c1);                 // cexpf(-I * phase)
    const float wr = z.x * z.x - z.y * z.y, wi = z.x * z.y + z.y * z.x;          // buffer * buffer
    const float ur = wr * c2 - wi * s2, ui = wr * s2 + wi * c2;                  // ... * cexpf(-2 i phase)
    const float error = ref_atan2f(ui, ur) * 0.5f;
    const float2 zr = make_float2(z.x * c1 - z.y * s1, z.x * s1 + z.y * c1);     // buffer *= cexpf(-i phase)
    freq += g.beta * error;
    if (freq > 0.5f) freq = 0.5f;
    if (freq < -0.5f) freq = -0.5f;
    phase += freq + cfo_freq + (g.alpha * error);
    if ((double)phase > M_PI) phase = (float)((double)phase - 2 * M_PI);
    if ((double)phase < -M_PI) phase = (float)((double)phase + 2 * M_PI);
    return zr;
}

// adjust_ref (sync.c:90-130) in place on col[n * stride], n = 0..31, the loop phase of every symbol filed in ph[n * ph_stride]; returns the sign bits of the
// derotated real parts (after the flip).  RESET: followed by reset_ref (sync.c:132-136) -- what the CFO search does at every visit of a bin: the derotated values
// are rotated back by cexpf(I * phases[n]), which restores them to within rounding, NOT bit for bit, and a bin the search visits again (up to 11 times) starts from
// the values the previous visit left.
template <bool RESET>
__device__ inline uint32_t adjust_ref_exact(float2 *col, int stride, float *ph, int ph_stride, float &freq, float &phase, int cfo, const LoopGains g)
{
    const float cfo_freq = (float)(2 * M_PI * cfo * CP_N / FFT_N);
    float x = 0.0f;
#pragma unroll 1
    for (int n = 0; n < NSYM; n++) {
        ph[n * ph_stride] = phase;
        const float2 zr = costas_step_exact(col[n * stride], freq, phase, cfo_freq, g);
        col[n * stride] = zr;
        const float sgn = ((PAT_POS >> n) & 1u) ? 1.0f : (((PAT_NEG >> n) & 1u) ? -1.0f : 0.0f);
        x += zr.x * sgn;
    }
    const bool flip = x < 0;
    if (flip) phase = (float)((double)phase + M_PI);
    uint32_t pos = 0;
    if (flip || RESET) {
#pragma unroll 1
        for (int n = 0; n < NSYM; n++) {
            float2 zr = col[n * stride];
            float pn = ph[n * ph_stride];
            if (flip) { pn = (float)((double)pn + M_PI); zr = make_float2(zr.x * -1.0f, zr.y * -1.0f); ph[n * ph_stride] = pn; }
            if (zr.x > 0) pos |= 1u << n;
            if (RESET) {
                float sn, cs; ref_sincosf(pn, sn, cs);         // reset_ref: buffer *= cexpf(I * phases[n])
                zr = make_float2(zr.x * cs - zr.y * sn, zr.x * sn + zr.y * cs);
            }
            col[n * stride] = zr;
        }
    } else {
        for (int n = 0; n < NSYM; n++) if (col[n * stride].x > 0) pos |= 1u << n;
    }
    return pos;
}

// One reference carrier through its second-order Costas loop for the 32 symbols of a block
// (sync.c:90-130).  z(n) is fetched through `src` with stride `stride`; optionally the derotated
// values / loop phases are stored.  Returns the sign bits of the derotated real parts (bit n = re > 0).
template <bool STORE>
__device__ inline uint32_t costas_block(const float2 *src, int stride, float &freq, float &phase, int cfo,
                                        LoopGains g, float2 *zout, float *phout)
{
    const float cfo_freq = (float)(2 * M_PI * cfo * CP_N / FFT_N);
    uint32_t pos = 0, neg = 0;
    float x = 0.0f;
    // STORE (the tracking call): the carrier's 32 bins wait in LDS -- the workgroup brought them in together, into the very cells
    // that receive the derotated values (zout == src, stride 1) -- and the loop, unrolled in full, reads one symbol ahead: as 32
    // prefetched registers per lane they made this the kernel's register peak (120 VGPRs x 3 waves per SIMD: the 12-wave workgroup
    // no longer fitted beside the decode waves that share its CU).  The search variant keeps four bins in flight in four named
    // registers and a rolled loop: indexed dynamically an array lands in scratch memory -- a load per symbol on the serial chain,
    // and a private segment that every launch of the kernel pays for.
    constexpr int NZ = STORE ? 1 : 4;
    float2 zin[NZ];
#pragma unroll
    for (int n = 0; n < NZ; n++) zin[n] = src[n * stride];     // independent loads in flight
    float s1, c1; fast_sincos(phase, s1, c1);                  // cexpf(-I phase), see fastmath.h; the next symbol's at the end of each step
    auto step = [&](int n, const float2 z) __attribute__((always_inline)) {
        const float s2 = 2.0f * s1 * c1, c2 = c1 * c1 - s1 * s1;                // e^{2i phase}
        const float2 w = make_float2(z.x * z.x - z.y * z.y, z.x * z.y + z.y * z.x);
        const float ur = w.x * c2 + w.y * s2, ui = w.y * c2 - w.x * s2;         // w * e^{-2i phase}
        const float error = fast_atan2(ui, ur) * 0.5f;
        const float2 zr = make_float2(z.x * c1 + z.y * s1, z.y * c1 - z.x * s1);   // z * e^{-i phase}
        if (STORE) { zout[n] = zr; phout[n] = phase; }
        if (zr.x > 0) pos |= 1u << n;
        if (zr.x < 0) neg |= 1u << n;
        if (!STORE) {
            const float sgn = ((PAT_POS >> n) & 1u) ? 1.0f : (((PAT_NEG >> n) & 1u) ? -1.0f : 0.0f);
            x += zr.x * sgn;
        }
        freq += g.beta * error;
        if (freq > 0.5f) freq = 0.5f;
        if (freq < -0.5f) freq = -0.5f;
        phase += freq + cfo_freq + (g.alpha * error);
        if (STORE && fabsf(phase) < 12.0f) {
            // Steady tracking: the phase is within two turns.  The reference's `if (phase > M_PI) phase -= 2 * M_PI` (double
            // comparison, double difference rounded to float) without leaving float32: (double)phase > M_PI <=> phase > the
            // largest float below pi; phase - 2 pi as an exact difference with float(2 pi) (Sterbenz: pi < phase < 4 pi) plus
            // the rest of the constant, one rounding (equal on 2e6 random phases, tests/test_halfband_float.py).  A phase this
            // small needs no double-precision argument reduction for the next symbol's rotation either.
            constexpr float PI_BELOW = 3.14159250259399414f, TWO_PI_HI = 6.28318548202514648f, TWO_PI_LO = -1.74845553e-7f;
            if (phase > PI_BELOW) phase = (phase - TWO_PI_HI) - TWO_PI_LO;
            if (phase < -PI_BELOW) phase = (phase + TWO_PI_HI) + TWO_PI_LO;
            fast_sincos_reduced(phase, s1, c1);
        } else {
            // after a large timing correction (sync_adjust rotates every loop by up to ~1800 rad) and in the CFO search
            if ((double)phase > M_PI) phase = (float)((double)phase - 2 * M_PI);
            if ((double)phase < -M_PI) phase = (float)((double)phase + 2 * M_PI);
            fast_sincos(phase, s1, c1);
        }
    };
    if (STORE) {
#pragma unroll
        for (int n = 0; n < NSYM; n++) {
            const float2 z = zin[0];
            if (n + 1 < NSYM) zin[0] = src[(n + 1) * stride];  // before this step's store to zout[n]: the next symbol's bin
            step(n, z);
        }
        // the sync-word correlation from the stored values, same terms in the same order (a zero weight adds +-0 to a sum that
        // starts at +0): accumulated inside the unrolled loop the compiler kept the 15 operands alive to the end -- in scratch
#pragma unroll
        for (int n = 0; n < NSYM; n++) {
            if ((PAT_POS >> n) & 1u) x += zout[n].x;
            else if ((PAT_NEG >> n) & 1u) x -= zout[n].x;
        }
    } else {
#pragma unroll 1
        for (int n = 0; n < NSYM; n += 4) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float2 z = zin[j];
                if (n + 4 < NSYM) zin[j] = src[(n + 4 + j) * stride];     // four symbols ahead
                step(n + j, z);
            }
        }
    }
    if (x < 0) {                                               // off by pi: flip (sync.c:119-129)
        if (STORE) for (int n = 0; n < NSYM; n++) { phout[n] = (float)((double)phout[n] + M_PI); zout[n] = make_float2(-zout[n].x, -zout[n].y); }
        phase = (float)((double)phase + M_PI);
        pos = neg;                                             // a zero real part stays "not positive" after the flip
    }
    return pos;
}

// find_ref_fm (sync.c:188-207): smallest cyclic shift at which the needle matches, trying the sign
// pattern and then its complement.
__device__ inline int needle_search(uint32_t d, unsigned rsid)
{
    const uint32_t val = needle_val(rsid);
    for (int n = 0; n < NSYM; n++) { const uint32_t rot = (d >> n) | (n ? (d << (32 - n)) : 0u); if ((rot & NEEDLE_MASK) == val) return n; }
    d = ~d;
    for (int n = 0; n < NSYM; n++) { const uint32_t rot = (d >> n) | (n ? (d << (32 - n)) : 0u); if ((rot & NEEDLE_MASK) == val) return n; }
    return -1;
}

__device__ inline float half_turn_diff(float a, float b)    // phase_diff, sync.c:284-290
{
    float d = a - b;
    while ((double)d > M_PI / 2) d = (float)((double)d - M_PI);
    while ((double)d < -M_PI / 2) d = (float)((double)d + M_PI);
    return d;
}

__device__ inline float2 cdiv(float2 a, float2 b)
{
    const float inv = 1.0f / (b.x * b.x + b.y * b.y);
    return make_float2((a.x * b.x + a.y * b.y) * inv, (a.y * b.x - a.x * b.y) * inv);
}

__device__ inline int soft_bit(float x, float mult)          // demod, sync.c:69-73
{
    const float c = fmaxf(fminf(x, 1.0f), -1.0f);
    return (int)lroundf(c * mult);
}

// cell index -> (side, partition from the band edge, symbol, carrier 1..18); PPB > 0 folds the divisions
template <int PPB> __device__ __forceinline__ void cell_coords(int c, int ppb_rt, int &side, int &part, int &n, int &k)
{
    const int ppb = PPB > 0 ? PPB : ppb_rt;
    k = 1 + c % 18; n = (c / 18) % NSYM; part = (c / (18 * NSYM)) % ppb; side = c / (18 * NSYM * ppb);
}

// adjust_data (sync.c:263-282) for one cell: C = (19+19j) / (k m19 e^{j phi19} + (19-k) m0 e^{j phi0})
template <int PPB>
__device__ __forceinline__ float2 equalise_cell(int c, int ppb_rt, const float2 *bins, const float2 (*refcs)[NSYM], const float *smag, int &side)
{
    int k, n, part;
    cell_coords<PPB>(c, ppb_rt, side, part, n, k);
    // side 0: refs i=part (low) and part+1 (high); side 1: low = upper-sideband ref part+1, high = ref part
    const int r_lo = side ? 2 * (part + 1) + 1 : 2 * part, r_hi = side ? 2 * part + 1 : 2 * (part + 1);
    const int b = ref_bin(r_lo) + k;
    const float2 z = bins[n * LIVE_N + bin_to_live(b)];
    const float2 lp = refcs[r_lo][n], up = refcs[r_hi][n];
    const float a = k * smag[r_hi], bq = (PW - k) * smag[r_lo];
    const float2 den = make_float2(a * up.x + bq * lp.x, a * up.y + bq * lp.y);
    const float2 C = cdiv(make_float2((float)PW, (float)PW), den);
    return make_float2(z.x * C.x - z.y * C.y, z.x * C.y + z.y * C.x);
}

__device__ __forceinline__ float cell_error(float2 v)           // |ideal - v|^2 against the nearest QPSK point (sync.c:465-483)
{
    const float ix = v.x >= 0 ? 1.0f : -1.0f, iy = v.y >= 0 ? 1.0f : -1.0f;
    const float dx = ix - v.x, dy = iy - v.y;
    return dx * dx + dy * dy;
}

// partitions 0..9 = lower sideband from the edge; 10..19 = upper sideband in ascending frequency (sync.c:514-536):
// the upper-sideband cell of partition `part` (from the edge) is matrix partition 19 - part
__device__ __forceinline__ void store_soft(int8_t *pm_blk, float2 v, int side, int part, int n, int k, float mult_lb, float mult_ub)
{
    const int part20 = side ? 19 - part : part;
    const float mult = side ? mult_ub : mult_lb;
    char2 o; o.x = (signed char)soft_bit(v.x, mult); o.y = (signed char)soft_bit(v.y, mult);
    *(char2 *)(pm_blk + n * 720 + part20 * 36 + (k - 1) * 2) = o;
}

// extended partitions (sync.c:537-596): cell (side, part >= 10, n, k) -> PX1 (1 partition per sideband in MP2, 2 in
// MP3 / MP11) or PX2 (2 more in MP11, where BOTH sidebands use the lower sideband's gain -- the reference's quirk)
__device__ __forceinline__ void store_px(int8_t *pair, float2 v, int side, int part, int n, int k, int ppb, int odd, float mult_lb, float mult_ub)
{
    const int nx1 = ppb == 11 ? 1 : 2;
    const int ch = (part - PM_PART) >= nx1 ? 1 : 0;
    const int count = ch ? 2 : nx1, q = part - PM_PART - (ch ? 2 : 0);
    const int per_sym = 72 * count, len = NSYM * per_sym;
    const int idx = side ? 36 * count + (count - 1 - q) * 36 + (k - 1) * 2 : q * 36 + (k - 1) * 2;
    const float mult = (side && !ch) ? mult_ub : mult_lb;
    char2 o; o.x = (signed char)soft_bit(v.x, mult); o.y = (signed char)soft_bit(v.y, mult);
    *(char2 *)(pair + (size_t)ch * 2 * PX_MAX + odd * len + n * per_sym + idx) = o;
}

// Work-items per stream (template parameter; launch_sync picks).  768 = 12 waves, three per SIMD: the equaliser cells of MP1 divide
// evenly (11520 = 768 x 15) and the three resident waves hide each other's LDS / memory latency -- with 256 (one wave per SIMD, 45
// cells each) the equalising and soft-bit phases ran at the latency of one dependent chain: 24 k + 11 k shader cycles per block
// for ~6 k of issue work; a lone stream's block went from 38 to 25 us.  Inside a full batch the 12-wave workgroup has to find three
// free wave slots on every SIMD of a CU beside the decode waves (a 16-wave traceback workgroup + the forward pass' waves): at 120
// VGPRs it waited for them to drain (the workgroup itself needed 30 us, the launch 75); held to <= 80 VGPRs (six waves per SIMD's
// worth: three of its waves and four traceback waves of 64 VGPRs share a SIMD's 512 registers) the wide form is the faster one
// there too, by a little (profiles/r04_sync_lanes.txt).  The narrow form stays selectable (NRSC5HIP_TUNE_SYNC_LANES).
#ifndef HIPEMU
#define SYNC_OCCUPANCY(NT) __attribute__((amdgpu_waves_per_eu((NT) > 512 ? 6 : 1, (NT) > 512 ? 6 : 8)))
#else
#define SYNC_OCCUPANCY(NT)
#endif
// The workgroup's LDS as one struct (round 6): the kernel k_sync places it in a static array of its own; the dataflow kernel (k_flow.hip), whose workgroups are
// symbol transforms OR block steps, places it in the region both roles share.
enum { PRE_STATE, PRE_BC, PRE_PXS, PRE_STARTED_PM, PRE_P1_COUNT, PRE_FINE_EPOCH, PRE_PM_SLOT, PRE_MER_CNT, PRE_ERR_LB, PRE_ERR_UB, PRE_N };
constexpr int SYNC_OFF_REFPH = NREF_MAX * NSYM * (int)sizeof(float2), SYNC_OFF_REFCS = SYNC_OFF_REFPH + NREF_MAX * NSYM * (int)sizeof(float);
constexpr int SYNC_OFF_CFO = SYNC_OFF_REFCS + NREF_MAX * NSYM * (int)sizeof(float2), SYNC_REF_BYTES = SYNC_OFF_CFO + (CFO_HI - CFO_LO) * 22;
template <int SYNC_NT> struct SyncLds {
    alignas(16) uint8_t lds_raw[SYNC_REF_BYTES > PM_BLOCK ? SYNC_REF_BYTES : PM_BLOCK];
    alignas(16) int8_t sh_pids_coded[3 * PIDS_LEN];
    double red[2][SYNC_NT / 64];
    long long sh_tstamp;
    float smag[NREF_MAX];
    int ref_ok[NREF_MAX], ref_bc[NREF_MAX], ref_psmi[NREF_MAX];
    int sh_i[8];
    float sh_f[8];
    float sh_diff[2 * 14];
    uint32_t sh_pids_out[4];
    int sh_seen[16 + 80];                                     // (the CFO search: [0..3] vote masks, [8..8 + 76) the candidates' best offsets)
    float ref_freq[NREF_MAX];
    int sh_pre[PRE_N];
    uint16_t sh_gather[PIDS_CODED];
};

// FLOW (k_flow.hip): the block step of stream s as one work item of the dataflow grid -- what the launch boundary in front of k_sync guarantees (the
// symbol transforms' bins are visible) and what the one behind it guarantees (this step's state is visible to the next step's work items) are the caller's
// business there; the body is the same.
template <int SYNC_NT>
__device__ __forceinline__ void sync_body(uint8_t *lds_base, const DevTables &tb, const DevBuffers &db, const int s, int parity, int slot, int fuse_prepare, int window, int pids_inline, int do_prepare, int ext_refs)
{
    SyncLds<SYNC_NT> &L = *reinterpret_cast<SyncLds<SYNC_NT> *>(lds_base);
    StreamState &st = db.state[s];
    if (do_prepare) {                                          // block-uniform (fast streaming seam): this block's bookkeeping is committed here -- the
        if (threadIdx.x == 0) prepare_block(db, st, s, false); // symbol kernel computed the same values for itself (prepare_values)
        __threadfence_block();
        __syncthreads();
    }
    // One burst of loads before the first wait: the state words the kernel starts from, the active reference carriers' bins (for the LARGEST carrier
    // set: which of them are active depends on psmi, their addresses do not) and each Costas lane's loop state.  Read where they are used they were four
    // DEPENDENT trips to memory -- active, then samperr / psmi / nblocks, then the bins, then the loop state behind a barrier -- at ~1 us apiece for data
    // the previous kernel wrote on other XCDs (profiles/r04_mixfft_phases.txt has the same finding for the symbol kernel).
    const int tid = threadIdx.x;
    constexpr int NREFBIN = (NREF_MAX * NSYM + SYNC_NT - 1) / SYNC_NT;
    const int e_active = st.active, e_nblocks = st.nblocks, e_samperr = st.samperr_cur, e_psmi = st.psmi;
    // (round 5) ... and the words the later phases used to fetch one dependent trip at a time (profiles/r04_sync_lanes.txt, "what is still exposed"): the
    // tracking state and block count every work-item tests behind the barrier that opens the equalising phase, the frame hand-off words and the MER
    // accumulators work-item 0 reads between its stores, the PIDS gather index of this block count.  A block that LOCKS rewrites some of them (work-item 0,
    // COARSE section): it publishes the new values in LDS (sh_i[4..6]) and the copies below are replaced there.
    // One word per work-item (work-items 0 .. PRE_N - 1), parked in LDS at the first barrier: held in registers across the kernel the ten values and the
    // gather index cost 11 spilled VGPRs and 204 spilled SGPRs of the 80-register budget (the 12-wave workgroup must fit beside the decode waves).
    int e_word = 0;
    {
        const int *w = &st.sync_state;
        w = tid == PRE_BC ? &st.bc : tid == PRE_PXS ? &st.px_started : tid == PRE_STARTED_PM ? &st.started_pm : tid == PRE_P1_COUNT ? &st.p1_count : w;
        w = tid == PRE_FINE_EPOCH ? &st.fine_epoch : tid == PRE_PM_SLOT ? &st.pm_slot : tid == PRE_MER_CNT ? &st.mer_cnt : w;
        w = tid == PRE_ERR_LB ? (const int *)&st.error_lb : tid == PRE_ERR_UB ? (const int *)&st.error_ub : w;
        if (tid < PRE_N) e_word = *w;
    }
    const int e_bc_v = st.bc;                                  // (every work-item: the gather index below needs it before anything is in LDS)
    // ext_refs = 0: the host's last look at the counters found every stream FINE on 10 partitions per sideband (MP1: the engine's px_needed flag), so only
    // the 22 carriers of that set are fetched here; a stream that turns out to need more (it re-locked on another service mode since that look) fetches
    // the rest below, one dependent trip later.  With the largest set fetched for every stream the pass read 2.3 GB it never used (whole path 3.11 -> 3.25 x).
    float2 e_bin[NREFBIN];
    constexpr int NCOMMON = 2 * (PM_PART + 1) * NSYM;          // 704 bins: the reference carriers of 10 partitions per sideband
    {
        const float2 *bins0 = db.bins + (size_t)s * NSYM * LIVE_N;
#pragma unroll
        for (int i = 0; i < NREFBIN; i++) {
            int k = min(tid + i * SYNC_NT, NREF_MAX * NSYM - 1);                  // (clamped, not predicated: no branch between the loads)
            if (!ext_refs) k = min(k, NCOMMON - 1);                              // (block-uniform condition; the clamped lanes re-read a line that is fetched anyway)
            e_bin[i] = bins0[(k % NSYM) * LIVE_N + bin_to_live(ref_bin(k / NSYM))];
        }
    }
    const int e_l = bin_to_live(ref_bin(min(tid, NREF_MAX - 1)));
    const float e_freq = st.costas_freq[e_l], e_phase = st.costas_phase[e_l];
    if (!e_active) {                                           // block-uniform
        // no block this step; with the fused pipeline the stream may have become ready since (new samples).  Fused steps
        // run without the acquisition kernels, so only FINE streams can be prepared here (prepare_block.h).
        if (fuse_prepare && threadIdx.x == 0) prepare_block(db, st, s, false);
        return;
    }
    // second burst (the first has arrived: e_active was needed): where this block count's PIDS cells sit in the soft-bit rows (decode.c:324-342) -- a table
    // load that feeds an address, in flight from here to the first barrier instead of in front of the gather
    const int e_gather = tb.pids_gather[(e_bc_v & 15) * PIDS_CODED + min(tid, PIDS_CODED - 1)];
    constexpr int SYNC_NW = SYNC_NT / 64;
    // phase instrumentation (nrsc5hip_debug_sync_phases): the running time stamp lives in LDS -- as a variable it was a register pair
    // alive across the whole kernel, spilled and reloaded around every barrier
    long long &sh_tstamp = L.sh_tstamp;
    if (db.sync_phase_cycles && s == 0 && tid == 0) sh_tstamp = (long long)clock64();
#define SYNC_MARK(i) do { if (db.sync_phase_cycles && s == 0 && tid == 0) { const long long now = (long long)clock64(); db.sync_phase_cycles[i] += now - sh_tstamp; sh_tstamp = now; } } while (0)

    // One LDS region, two lives: the reference-carrier scratch of the tracking and equalising phases, then -- once the last
    // equalised cell sits in a register (barrier after the MER sums) -- the block's soft-bit rows on their way to the matrix.
    constexpr int OFF_REFPH = SYNC_OFF_REFPH, OFF_REFCS = SYNC_OFF_REFCS, OFF_CFO = SYNC_OFF_CFO;
    uint8_t *lds_raw = L.lds_raw;
    float2 (*refz)[NSYM] = (float2 (*)[NSYM])lds_raw;                            // derotated reference carriers
    float (*refph)[NSYM] = (float (*)[NSYM])(lds_raw + OFF_REFPH);               // loop phase per symbol (phases[][] of the reference)
    float2 (*refcs)[NSYM] = (float2 (*)[NSYM])(lds_raw + OFF_REFCS);             // e^{+i refph}
    int8_t (*cfo_offs)[22] = (int8_t (*)[22])(lds_raw + OFF_CFO);
    int8_t *pm_tile = (int8_t *)lds_raw;                                         // MP1: this block's soft-bit rows (second life)
    static_assert(PM_BLOCK % 16 == 0 && PM_FRAME % 16 == 0, "soft-bit rows leave in 16-byte pieces");
    auto &smag = L.smag; auto &ref_ok = L.ref_ok; auto &ref_bc = L.ref_bc; auto &ref_psmi = L.ref_psmi; auto &sh_i = L.sh_i; auto &sh_f = L.sh_f; auto &red = L.red;
    auto &sh_diff = L.sh_diff; auto &sh_pids_coded = L.sh_pids_coded; auto &sh_pids_out = L.sh_pids_out; auto &sh_seen = L.sh_seen; auto &ref_freq = L.ref_freq;
    auto &sh_pre = L.sh_pre; auto &sh_gather = L.sh_gather;
    static_assert(SYNC_NW == SYNC_NT / 64, "one partial sum per wave");
    static_assert(PM_BLOCK <= 65536, "a gather index fits 16 bits");
    if (tid == 0) { sh_i[2] = 0; sh_i[3] = 0; }                // [2] set when this block completes a P1 frame (replay checkpoint below), [3] when a PIDS frame was decoded here

    float2 *bins = db.bins + (size_t)s * NSYM * LIVE_N;       // [sym][live]
    const int nblocks0 = wave_uniform(e_nblocks);
    BlockRecord &rec = db.records[(size_t)s * db.rec_cap + (nblocks0 % db.rec_cap)];
    const LoopGains g = loop_gains();
    const int samperr = wave_uniform(e_samperr);
    const int ppb = partitions_for_psmi(wave_uniform(e_psmi));
    const int nref = 2 * (ppb + 1);

    // ---- sync_adjust (sync.c:769-777): timing pick moved by adj samples -> rotate every loop phase
    {
        const int adj = SYM_N / 2 - samperr;                   // block-uniform; 0 (nothing to rotate: x - 0.0 == x) on most blocks of a
        if (adj != 0)                                          // stream without a sample-clock error
        for (int l = tid; l < LIVE_N; l += SYNC_NT) {
            const int b = live_to_bin(l);
            st.costas_phase[l] = (float)((double)st.costas_phase[l] - (adj * (b - FFT_N / 2)) * 2 * M_PI / FFT_N);
        }
    }
    // the active reference carriers' bins -> refz[r][n] (the Costas loops below derotate them in place)
#pragma unroll
    for (int i = 0; i < NREFBIN; i++) {
        const int k = tid + i * SYNC_NT;
        if (k < nref * NSYM) {
            if (!ext_refs && k >= NCOMMON) e_bin[i] = bins[(k % NSYM) * LIVE_N + bin_to_live(ref_bin(k / NSYM))];   // the host's hint was stale for this stream
            refz[k / NSYM][k % NSYM] = e_bin[i];
        }
    }
    if (tid < PRE_N) sh_pre[tid] = e_word;
    if (tid < PIDS_CODED) sh_gather[tid] = (uint16_t)e_gather;
    __syncthreads();
    SYNC_MARK(0);

    // ---- Costas loops of the active reference carriers (sync.c:360-364)
    // loop_exact (DevBuffers): 0 = the fast forms everywhere, 1 = the reference's own operations (ref_sincosf / ref_atan2f, in-place derotate and reset_ref) in every
    // block that starts un-synchronised -- the tracking pass over garbage and the CFO search, where a last-bit difference can be amplified into a different loop
    // state (DESIGN (c) limit 2) --, 2 = in every block
    const bool exact_blk = db.loop_exact == 2 || (db.loop_exact == 1 && sh_pre[PRE_STATE] != SYNC_FINE);
    if (tid < nref) {
        const int l = bin_to_live(ref_bin(tid));
        float f = e_freq, p = e_phase;
        if (SYM_N / 2 - samperr != 0) p = st.costas_phase[l];  // block-uniform: sync_adjust above has just rotated the phases
        if (exact_blk) adjust_ref_exact<false>(refz[tid], 1, refph[tid], 1, f, p, 0, g);   // block-uniform: the reference's own operations (loop_exact)
        else costas_block<true>(refz[tid], 1, f, p, 0, g, refz[tid], refph[tid]);         // in place: refz holds the carrier's raw bins
        int l2 = l;
#ifndef HIPEMU
        asm volatile("" : "+v"(l2));                           // the address is computed again instead of surviving the loops in a (spilled) register pair
#endif
        st.costas_freq[l2] = f; st.costas_phase[l2] = p;
        ref_freq[tid] = f;
    }
    __syncthreads();
    SYNC_MARK(1);

    // ---- COARSE: try to lock (sync.c:366-423)
    if (sh_pre[PRE_STATE] == SYNC_COARSE) {
        if (tid < nref) {
            // decode_ref_fm (sync.c:169-186)
            uint32_t d = 0;
            for (int n = 0; n < NSYM; n++) if (refz[tid][n].x > 0) d |= 1u << n;
            const unsigned rsid = (30 - (tid >> 1)) & 3;
            const int ok = ((d & NEEDLE_MASK) == needle_val(rsid));
            const uint32_t dd = d ^ (d << 1);                  // DBPSK: data[n] = bit[n] ^ bit[n-1]
            ref_ok[tid] = ok;
            ref_bc[tid] = (int)((((dd >> 16) & 1) << 3) | (((dd >> 17) & 1) << 2) | (((dd >> 18) & 1) << 1) | ((dd >> 19) & 1));
            ref_psmi[tid] = (int)((((dd >> 25) & 1) << 5) | (((dd >> 26) & 1) << 4) | (((dd >> 27) & 1) << 3) | (((dd >> 28) & 1) << 2) | (((dd >> 29) & 1) << 1) | ((dd >> 30) & 1));
        }
        __syncthreads();
        if (tid == 0) {
            int good = 0;
            int *seen_bc = sh_seen, *seen_psmi = sh_seen + 16;  // LDS: indexed by decoded values (as private arrays they were scratch memory)
            for (int k = 0; k < 16; k++) seen_bc[k] = 0;
            for (int k = 0; k < 64; k++) seen_psmi[k] = 0;
            for (int r = 0; r < nref; r++) if (ref_ok[r]) { good++; seen_bc[ref_bc[r]]++; seen_psmi[ref_psmi[r]]++; }
            int action = 0;                                    // 0: none, 1: locked, 2: run CFO search
            if (good >= 4) {
                int maj_bc = -1, maj_psmi = -1;
                for (int v = 0; v < 16; v++) if (seen_bc[v] > good / 2) maj_bc = v;
                for (int v = 0; v < 16; v++) if (seen_psmi[v] > good / 2) maj_psmi = v;     // 0..15 only (sync.c:396)
                if (maj_bc >= 0 && maj_psmi >= 0) {
                    st.bc = maj_bc; st.psmi = maj_psmi;
                    // input_set_sync_state(FINE): EVENT_SYNC payload (input.c:179-185)
                    rec.freq_offset = (float)(((double)st.prev_angle - 2 * M_PI * st.cfo) * 744187.5 / (2 * M_PI * FFT_N));
                    rec.flags |= REC_TO_FINE;
                    st.sync_state = SYNC_FINE; st.fine_epoch++;
                    st.started_pm = 0;                         // decode_reset (decode.c:563-572)
                    st.px_pos = 0; st.px_ready = 0; st.px_started = 0;   // interleaver_iv_reset
                    sh_i[4] = maj_bc; sh_i[5] = maj_psmi;              // the copies of the head burst are stale from here on
                    action = 1;
                }
            } else if (st.cfo_wait == 0) {
                action = 2;
            } else {
                st.cfo_wait--;
            }
            sh_i[0] = action;
        }
        __syncthreads();
        if (sh_i[0] == 2) {
            // ---- detect_cfo (sync.c:292-337): every candidate offset x every reference position.
            // Lane = live bin; a bin is visited by at most 11 (cfo, i) pairs, in ascending cfo order,
            // and each visit advances that bin's loop state exactly as adjust_ref does.
            for (int k = tid; k < (CFO_HI - CFO_LO) * 22; k += SYNC_NT) (&cfo_offs[0][0])[k] = -1;
            __syncthreads();
            if constexpr (SYNC_NT >= LIVE_N) {
                // ONE pass (round 5).  A work-item owns one live bin and files the loop state after each of its (at most 11) visits in a per-stream slab of
                // global memory (db.cfo_snap: stores nobody waits for); the workgroup-wide vote then says where the search stopped and each work-item
                // reads back the ONE snapshot of its last visit at or below that candidate and commits it.  (Round 3 kept the snapshots in a private array:
                // scratch memory, paid for by every launch of the kernel; round 4 ran every visit twice instead; as named registers of an unrolled visit
                // loop they cost 17 spilled VGPRs.)  The search is the straggler of the block-step chain: the launch of a step in which ONE stream searches
                // lasted 230 - 535 us against 37 us for a step without (profiles/r05_trace_*.txt), two thirds of it the second pass and a 76-candidate vote
                // on a single work-item.
                const int l = tid < LIVE_N ? tid : LIVE_N - 1;
                const bool mine = tid < LIVE_N;
                const int b = live_to_bin(l);
                const bool lower = l < LIVE_HALF;
                float f = st.costas_freq[l], p = st.costas_phase[l];
                int rslot = -1;
                if (lower) { if ((b - LB0) % PW == 0 && (b - LB0) / PW <= ppb) rslot = 2 * ((b - LB0) / PW); }
                else { if ((UB1 - b) % PW == 0 && (UB1 - b) / PW <= ppb) rslot = 2 * ((UB1 - b) / PW) + 1; }
                const float2 *src = rslot >= 0 ? (const float2 *)&refz[rslot][0] : (const float2 *)(bins + l);
                const int stride = rslot >= 0 ? 1 : LIVE_N;
                float2 *snap = db.cfo_snap + ((size_t)s * LIVE_N + l) * (PM_PART + 1);
                // (two loops, one per arithmetic: in one loop the exact form's double-precision constants were held in registers across the fast form's visits too,
                //  and the fast form's prefetch addresses went to scratch memory -- a private segment every launch of this kernel would pay for)
                if (mine && !exact_blk)
                for (int q = 0; q <= PM_PART; q++) {
                    const int i = lower ? (PM_PART - q) : q;   // ascending cfo
                    const int cfo = lower ? (b - LB0 - PW * i) : (b - UB1 + PW * i);
                    if (cfo < CFO_LO || cfo >= CFO_HI) continue;
                    const uint32_t d = costas_block<false>(src, stride, f, p, cfo, g, nullptr, nullptr);
                    cfo_offs[cfo - CFO_LO][2 * i + (lower ? 0 : 1)] = (int8_t)needle_search(d, (30 - i) & 3);
                    snap[q] = make_float2(f, p);
                }
                if (mine && exact_blk)
                for (int q = 0; q <= PM_PART; q++) {
                    const int i = lower ? (PM_PART - q) : q;
                    const int cfo = lower ? (b - LB0 - PW * i) : (b - UB1 + PW * i);
                    if (cfo < CFO_LO || cfo >= CFO_HI) continue;
                    // adjust_ref + reset_ref in place on the bin's own column (this work-item is its only visitor), phases in the stream's scratch slab
                    const uint32_t d = adjust_ref_exact<true>(const_cast<float2 *>(src), stride, db.cfo_phase + (size_t)s * NSYM * LIVE_N + l, LIVE_N, f, p, cfo, g);
                    cfo_offs[cfo - CFO_LO][2 * i + (lower ? 0 : 1)] = (int8_t)needle_search(d, (30 - i) & 3);
                    snap[q] = make_float2(f, p);
                }
                __syncthreads();
                // the vote (sync.c:316-335), one candidate per work-item: the most frequent needle offset among the candidate's 22 reference positions, the
                // smallest such offset on ties (the reference scans the offsets upwards with `>`), at least three of them
                constexpr int NCAND = CFO_HI - CFO_LO;
                static_assert(NCAND <= 128, "two waves vote");
                int my_best = -1;
                if (tid < NCAND) {
                    int best = -1, best_count = 0;
                    for (int r = 0; r < 22; r++) {
                        const int v = cfo_offs[tid][r];
                        if (v < 0) continue;
                        int cnt = 0;
                        for (int r2 = 0; r2 < 22; r2++) cnt += cfo_offs[tid][r2] == v;
                        if (cnt > best_count || (cnt == best_count && v < best)) { best = v; best_count = cnt; }
                    }
                    if (best >= 0 && best_count >= 3) my_best = best;
                }
                if (tid < 128) {
                    const unsigned long long m = __ballot(my_best >= 0);
                    if ((tid & 63) == 0) { sh_seen[2 * (tid >> 6)] = (int)(uint32_t)m; sh_seen[2 * (tid >> 6) + 1] = (int)(uint32_t)(m >> 32); }
                    if (tid < NCAND) sh_seen[8 + tid] = my_best;
                }
                __syncthreads();
                if (tid == 0) {
                    int found = 0x7fffffff;
                    for (int w = 0; w < 4 && found == 0x7fffffff; w++) {
                        const uint32_t m = (uint32_t)sh_seen[w];
                        if (m) found = 32 * w + __ffs((int)m) - 1;
                    }
                    if (found != 0x7fffffff) {
                        const int best = sh_seen[8 + found];
                        st.keep_extra = ((NSYM - best) % NSYM) * SYM_N;          // acquire_keep_extra
                        st.cfo += found + CFO_LO;                                 // acquire_cfo_adjust
                        st.cfo_wait = 8;
                        found += CFO_LO;
                    }
                    sh_i[1] = found;
                }
                __syncthreads();
                {
                    const int last_cfo = sh_i[1];
                    int q_last = -1;                           // the visits ascend in cfo with q
                    for (int q = 0; q <= PM_PART; q++) {
                        const int i = lower ? (PM_PART - q) : q;
                        const int cfo = lower ? (b - LB0 - PW * i) : (b - UB1 + PW * i);
                        if (cfo >= CFO_LO && cfo < CFO_HI && cfo <= last_cfo) q_last = q;
                    }
                    if (mine && q_last >= 0) { const float2 v = snap[q_last]; st.costas_freq[l] = v.x; st.costas_phase[l] = v.y; }
                }
            } else {
                // Two passes over the same visits instead of a snapshot per visit (snapshots indexed by a running count lived in scratch
                // memory, and the private segment was paid for by EVERY launch of this kernel): pass 0 files the needle offsets and the
                // first workgroup-wide match decides where the search stops; pass 1 repeats the visits up to that candidate from the
                // saved loop state -- the search runs a handful of times per acquisition.
                for (int pass = 0; pass < 2; pass++) {
                    const int last_cfo = pass ? sh_i[1] : 0x7fffffff;  // pass 1: visits with cfo <= last_cfo happened
                    for (int l = tid; l < LIVE_N; l += SYNC_NT) {
                        const int b = live_to_bin(l);
                        const bool lower = l < LIVE_HALF;
                        float f = st.costas_freq[l], p = st.costas_phase[l];   // untouched by pass 0
                        // is this bin one of the already-derotated active references?
                        int rslot = -1;
                        if (lower) { if ((b - LB0) % PW == 0 && (b - LB0) / PW <= ppb) rslot = 2 * ((b - LB0) / PW); }
                        else { if ((UB1 - b) % PW == 0 && (UB1 - b) / PW <= ppb) rslot = 2 * ((UB1 - b) / PW) + 1; }
                        const float2 *src = rslot >= 0 ? (const float2 *)&refz[rslot][0] : (const float2 *)(bins + l);
                        const int stride = rslot >= 0 ? 1 : LIVE_N;
                        bool visited = false;
                        for (int q = 0; q <= PM_PART; q++) {
                            const int i = lower ? (PM_PART - q) : q;   // ascending cfo
                            const int cfo = lower ? (b - LB0 - PW * i) : (b - UB1 + PW * i);
                            if (cfo < CFO_LO || cfo >= CFO_HI || cfo > last_cfo) continue;
                            const uint32_t d = costas_block<false>(src, stride, f, p, cfo, g, nullptr, nullptr);
                            visited = true;
                            if (pass == 0) cfo_offs[cfo - CFO_LO][2 * i + (lower ? 0 : 1)] = (int8_t)needle_search(d, (30 - i) & 3);
                        }
                        if (pass == 1 && visited) { st.costas_freq[l] = f; st.costas_phase[l] = p; }
                    }
                    if (pass == 1) break;
                    __syncthreads();
                    if (tid == 0) {
                        int found = 0x7fffffff;
                        for (int c = 0; c < CFO_HI - CFO_LO && found == 0x7fffffff; c++) {
                            int *count = sh_seen;
                            for (int k = 0; k < NSYM; k++) count[k] = 0;
                            for (int r = 0; r < 22; r++) if (cfo_offs[c][r] >= 0) count[cfo_offs[c][r]]++;
                            int best = -1, best_count = 0;
                            for (int k = 0; k < NSYM; k++) if (count[k] > best_count) { best = k; best_count = count[k]; }
                            if (best >= 0 && best_count >= 3) {
                                st.keep_extra = ((NSYM - best) % NSYM) * SYM_N;      // acquire_keep_extra
                                st.cfo += c + CFO_LO;                                 // acquire_cfo_adjust
                                st.cfo_wait = 8;
                                found = c + CFO_LO;
                            }
                        }
                        sh_i[1] = found;
                    }
                    __syncthreads();
                }
        }
            }
        __syncthreads();
    }

    SYNC_MARK(2);
    // ---- FINE: equalise, measure, demodulate (sync.c:425-609)
    // (state / block count / service mode from the head burst -- unless this very block locked: then work-item 0 has just written them)
    const bool locked_now = sh_pre[PRE_STATE] == SYNC_COARSE && sh_i[0] == 1;
    const int state_now = locked_now ? (int)SYNC_FINE : sh_pre[PRE_STATE];
    if (state_now == SYNC_FINE) {
        const int bc = locked_now ? sh_i[4] : sh_pre[PRE_BC];
        const int psmi_now = locked_now ? sh_i[5] : wave_uniform(e_psmi);
        const int pxs_now = locked_now ? 0 : sh_pre[PRE_PXS];
        // The block that achieves lock keeps equalising with the partition count of the PREVIOUS service mode (computed at
        // the top of sync_process_fm, sync.c:343-358) but already routes PX soft bits by the new one (sync.c:537-596).
        const int ppb_px = routed_partitions_for_psmi(psmi_now);
        const bool px_on = ppb_px > PM_PART && (pxs_now || (bc & 1) == 0);   // decode_push_px1/2 (decode.c:393-437)
        for (int k = tid; k < nref * NSYM; k += SYNC_NT) {
            const int r = k / NSYM, n = k % NSYM;
            float sn, cs; fast_sincos(refph[r][n], sn, cs);
            refcs[r][n] = make_float2(cs, sn);
        }
        if (tid < nref) {                                      // calc_smag (sync.c:254-261)
            float sum = 0.0f;
            for (int n = 0; n < NSYM; n++) sum += fabsf(refz[tid][n].x);
            smag[tid] = sum / NSYM;
        }
        if (tid >= 64 && tid < 64 + ppb) {                     // the phase differences of the timing estimate below, one partition per lane
            const int i = tid - 64;
            sh_diff[2 * i] = half_turn_diff(refph[2 * i][0], refph[2 * (i + 1)][0]);
            sh_diff[2 * i + 1] = half_turn_diff(refph[2 * (i + 1) + 1][0], refph[2 * i + 1][0]);
        }
        __syncthreads();

        if (tid == 0) {
            // timing error from the phase slope across each partition, residual CFO from the loop
            // frequencies (sync.c:426-463); same summation order as the reference
            float se = 0.0f, angle = 0.0f, sum_xy = 0.0f, sum_x2 = 0.0f;
            for (int i = 0; i < 2 * ppb; i++) se += sh_diff[i];
            se = (float)(se / (ppb * 2) * FFT_N / PW / (2 * M_PI));
            for (int i = 0; i <= ppb; i++) {
                float x = (float)(LB0 + PW * i - FFT_N / 2), y = ref_freq[2 * i];
                angle += y; sum_xy += x * y; sum_x2 += x * x;
                x = (float)(UB1 - PW * i - FFT_N / 2); y = ref_freq[2 * i + 1];
                angle += y; sum_xy += x * y; sum_x2 += x * x;
            }
            se = (float)(se - (sum_xy / sum_x2) * FFT_N / (2 * M_PI) * NSYM);
            st.samperr = (int)roundf(se);
            angle /= (ppb + 1) * 2;
            st.angle = angle;
            sh_f[0] = angle;
        }
        __syncthreads();
        if (tid < nref) st.costas_freq[bin_to_live(ref_bin(tid))] = ref_freq[tid] - sh_f[0];
        SYNC_MARK(3);

        // cell (side, part, n, k): data carrier k = 1..18 of partition `part` (counted from the band edge);
        // lane tid owns cells c = tid + SYNC_NT i.  MP1 (10 partitions, 15 or 45 cells per lane) keeps the equalised values
        // in registers between the MER pass and the soft-bit pass; the wider service modes recompute them.
        const int ncell = 2 * ppb * NSYM * 18;
        constexpr int MP1C = 2 * PM_PART * NSYM * 18 / SYNC_NT;                 // 15 (768 work-items) or 45 (256)
        static_assert(MP1C * SYNC_NT == 2 * PM_PART * NSYM * 18, "the MP1 cells divide evenly over the work-items");
        float2 cellv[MP1C];
        double e_lb = 0.0, e_ub = 0.0;
        if (ppb == PM_PART) {
            // operands from the cell table (DevTables::eq_cell), all 15 bins of the lane requested before the first is used.
            // (Requested before the Costas loops and held across them they cost 30 VGPRs that the 80-register budget does not
            // have: the compiler spilled every one of them.)
            uint32_t cw[MP1C];
#pragma unroll
            for (int i = 0; i < MP1C; i++) cw[i] = tb.eq_cell[tid + SYNC_NT * i];
#pragma unroll
            for (int i = 0; i < MP1C; i++) cellv[i] = bins[((cw[i] >> 10) & 31u) * LIVE_N + (cw[i] & 1023u)];
#pragma unroll
            for (int i = 0; i < MP1C; i++) {
                const uint32_t w = cw[i];
                const int n = (w >> 10) & 31u, r_lo = (w >> 15) & 31u, r_hi = (w >> 20) & 31u, k = (w >> 25) & 31u;
                const float2 z = cellv[i];
                const float2 lp = refcs[r_lo][n], up = refcs[r_hi][n];
                const float a = k * smag[r_hi], bq = (PW - k) * smag[r_lo];
                const float2 den = make_float2(a * up.x + bq * lp.x, a * up.y + bq * lp.y);
                const float2 C = cdiv(make_float2((float)PW, (float)PW), den);
                const float2 v = make_float2(z.x * C.x - z.y * C.y, z.x * C.y + z.y * C.x);
                cellv[i] = v;
                const float e = cell_error(v);
                if (w >> 30) e_ub += e; else e_lb += e;
            }
        } else {
            for (int c = tid; c < ncell; c += SYNC_NT) {
                int side;
                const float e = cell_error(equalise_cell<0>(c, ppb, bins, refcs, smag, side));
                if (side) e_ub += e; else e_lb += e;
            }
        }
        e_lb = wave_sum_f64_rf(e_lb); e_ub = wave_sum_f64_rf(e_ub);            // (register-file moves: 24 LDS crossbar round trips less on every wave's path to the MER barrier)
        if ((tid & 63) == 0) { red[0][tid >> 6] = e_lb; red[1][tid >> 6] = e_ub; }
        __syncthreads();
        if (tid == 0) {
            double sl = 0.0, su = 0.0;
            for (int w = 0; w < SYNC_NW; w++) { sl += red[0][w]; su += red[1][w]; }
            const float error_lb = (float)sl, error_ub = (float)su;
            // (accumulators and counter from the head burst: nothing else in this kernel writes them)
            float acc_lb = __builtin_bit_cast(float, sh_pre[PRE_ERR_LB]) + error_lb, acc_ub = __builtin_bit_cast(float, sh_pre[PRE_ERR_UB]) + error_ub;
            int cnt = sh_pre[PRE_MER_CNT] + 1;
            if (cnt == 16) {                                   // EVENT_MER every 16 blocks (sync.c:490-501)
                const float signal = (float)(2 * NSYM * (ppb * 18) * cnt);
                rec.mer_lb = 10 * log10f(signal / acc_lb);
                rec.mer_ub = 10 * log10f(signal / acc_ub);
                rec.flags |= REC_MER;
                cnt = 0; acc_lb = 0; acc_ub = 0;
            }
            st.error_lb = acc_lb; st.error_ub = acc_ub; st.mer_cnt = cnt;
            const float mer_lb = 2.0f * NSYM * (float)(ppb * 18) / error_lb;
            const float mer_ub = 2.0f * NSYM * (float)(ppb * 18) / error_ub;
            sh_f[1] = fmaxf(fminf(mer_lb * 10, 127.0f), 1.0f);
            sh_f[2] = fmaxf(fminf(mer_ub * 10, 127.0f), 1.0f);
        }
        __syncthreads();
        const float mult_lb = sh_f[1], mult_ub = sh_f[2];
        SYNC_MARK(4);

        // primary-main soft bits -> row `bc` of the stream's 16 x 32 x 720 interleaver matrix (decode.c:380).
        // Partitions 0..9 = lower sideband from the edge; 10..19 = upper sideband in ascending frequency
        // (sync.c:514-536): the upper-sideband cell of partition `part` (from the edge) is matrix partition 19 - part.
        const int pm_slot = sh_pre[PRE_PM_SLOT];
        int8_t *pm_blk = db.pm + ((size_t)s * NPM + pm_slot) * PM_FRAME + (size_t)bc * PM_BLOCK;
        if (ppb == PM_PART) {
            // the 11520 two-byte cells of the block's 32 x 720 soft-bit rows are assembled in LDS and leave in 16-byte rows:
            // six coalesced stores per lane instead of 45 scattered two-byte ones
#pragma unroll
            for (int i = 0; i < MP1C; i++) {
                const int c = tid + SYNC_NT * i;
                const float mult = c >= ncell / 2 ? mult_ub : mult_lb;              // cells of the upper sideband come second
                char2 o; o.x = (signed char)soft_bit(cellv[i].x, mult); o.y = (signed char)soft_bit(cellv[i].y, mult);
                *(char2 *)(pm_tile + tb.eq_out[c]) = o;
            }
            __syncthreads();
            for (int q = tid; q < PM_BLOCK / 16; q += SYNC_NT) ((uint4 *)pm_blk)[q] = ((const uint4 *)pm_tile)[q];
        } else {
            for (int c = tid; c < ncell; c += SYNC_NT) {
                int k, n, part, side;
                cell_coords<0>(c, ppb, side, part, n, k);
                if (part < PM_PART) store_soft(pm_blk, equalise_cell<0>(c, ppb, bins, refcs, smag, side), side, part, n, k, mult_lb, mult_ub);
                else if (px_on && part < ppb_px) store_px(db.px_pair + (size_t)s * 4 * PX_MAX, equalise_cell<0>(c, ppb, bins, refcs, smag, side), side, part, n, k, ppb_px, bc & 1, mult_lb, mult_ub);
            }
        }
        if (px_on && ppb_px > ppb) {
            // lock block only: extended partitions that were not equalised yet -- the reference demodulates the raw bins
            const int p0 = ppb > PM_PART ? ppb : PM_PART, np = ppb_px - p0;
            for (int c = tid; c < 2 * np * NSYM * 18; c += SYNC_NT) {
                const int k = 1 + c % 18, n = (c / 18) % NSYM, part = p0 + (c / (18 * NSYM)) % np, side = c / (18 * NSYM * np);
                const int b = (side ? UB1 - PW * (part + 1) : LB0 + PW * part) + k;
                store_px(db.px_pair + (size_t)s * 4 * PX_MAX, bins[n * LIVE_N + bin_to_live(b)], side, part, n, k, ppb_px, bc & 1, mult_lb, mult_ub);
            }
        }
        __threadfence_block();
        __syncthreads();
        SYNC_MARK(5);

        // ---- PIDS: gather + depuncture now (decode.c:324-342); the 80-bit Viterbi + descramble run in
        // k_pids_decode, off this kernel's critical path (results only feed the record, not the loops)
        int8_t *stage = db.pids_stage + (((size_t)s * NWIN + parity) * 16 + slot) * (3 * PIDS_LEN);
        const int8_t *pm_src = ppb == PM_PART ? pm_tile : pm_blk;      // MP1: the rows are still in LDS
        if (pids_inline) {
            // streaming seam (block-uniform): the 80-bit frame is decoded right here, by the second wave, while the first lane does the
            // block's bookkeeping and the record -- as its own launch (or in the report kernel) it was 10-20 us on a chain the host
            // waits for
            for (int n = tid; n < PIDS_CODED; n += SYNC_NT) sh_pids_coded[n + n / 5] = pm_src[!locked_now ? (int)sh_gather[n] : (int)tb.pids_gather[bc * PIDS_CODED + n]];
            for (int n = tid; n < PIDS_CODED / 5; n += SYNC_NT) sh_pids_coded[6 * n + 5] = 0;
            __syncthreads();
            if ((tid >> 6) == 1) {
                uint32_t *out = sh_pids_out;
                viterbi_k7_wave_compact<PIDS_LEN>(sh_pids_coded, nullptr, out);
                WAVE_LDS_FENCE();
                {
                    const uint32_t p[3] = { out[0] ^ tb.scr_pids[0], out[1] ^ tb.scr_pids[1], (out[2] ^ tb.scr_pids[2]) & 0xffffu };   // descramble (decode.c:470)
                    const bool crc_ok = pids_crc_ok_wave(p);   // the whole wave (one lane's bit loop was a fifth of this decode)
                    WAVE_LDS_FENCE();                          // every lane has read out[] before lane 0 rewrites it
                    if (tid == 64) { out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; out[3] = crc_ok ? 1u : 0u; }
                }
            }
        } else {
            for (int n = tid; n < PIDS_CODED; n += SYNC_NT) stage[n + n / 5] = pm_src[!locked_now ? (int)sh_gather[n] : (int)tb.pids_gather[bc * PIDS_CODED + n]];
            for (int n = tid; n < PIDS_CODED / 5; n += SYNC_NT) stage[6 * n + 5] = 0;
        }
        SYNC_MARK(6);
        if (tid == 0) {
            // (hand-off words from the head burst: read here they were five loads, each behind the previous store)
            int started_pm = locked_now ? 0 : sh_pre[PRE_STARTED_PM];    // decode_reset at the lock (above)
            const int fine_epoch = sh_pre[PRE_FINE_EPOCH] + (locked_now ? 1 : 0);
            const int e_p1_count = sh_pre[PRE_P1_COUNT];
            if (!pids_inline) db.pids_rec[((size_t)s * NWIN + parity) * 16 + slot] = nblocks0 % db.rec_cap;
            else sh_i[3] = 1;                                  // the tail files the frame
            rec.flags |= REC_PIDS;
            rec.bc_decoded = bc;
            if (bc == 0) { started_pm = 1; st.started_pm = 1; }   // decode.c:383-390
            if (started_pm && bc == 15) {
                const int slot = e_p1_count % db.p1_slots;
                st.p1_count = e_p1_count + 1;
                st.p1_pending[parity] = 1; st.p1_slot[parity] = slot; st.p1_record[parity] = nblocks0 % db.rec_cap; st.p1_epoch[parity] = fine_epoch;
                st.p1_pmslot[parity] = pm_slot;
                st.p1_verdict[parity] = 0; st.p1_recabs[parity] = nblocks0; st.p1_window[parity] = window;
                sh_i[2] = 1;
                rec.p1_slot = slot; rec.flags |= REC_P1;
            }
            if (ppb_px > PM_PART) {
                if ((bc & 1) == 0) st.px_started = 1;
                if ((pxs_now || (bc & 1) == 0) && (bc & 1)) {
                    // a block pair is complete: k_px_deint runs interleaver IV next; frames appear once it has wrapped
                    st.px_go = NSYM * 72 * (ppb_px == 11 ? 1 : 2);
                    st.px_nch = ppb_px == 14 ? 2 : 1;
                    st.px_record = nblocks0 % db.rec_cap;
                    if (st.px_ready || st.px_pos == 32 * st.px_go) {
                        st.px_slot = st.px_count % db.px_slots; st.px_count++;
                        rec.sis = (uint32_t)st.px_slot;
                        rec.flags |= REC_P3 | (ppb_px == 14 ? (uint32_t)REC_P4 : 0u);
                    } else st.px_slot = -1;
                }
            }
            st.bc = (bc + 1) % 16;
            st.last_pm_slot = pm_slot;
            if (bc == 15) st.pm_slot = (pm_slot + 1) % NPM;      // the next frame fills a fresh matrix
        }
    }
    __syncthreads();
    SYNC_MARK(14);                                             // (the inline PIDS decode on wave 1 beside tid 0's bookkeeping, waited for)

    // ---- end of acquire_process (acquire.c:259-262) + record.  Two lanes of different waves share the work: the NCO phase with
    // its double-precision sine / cosine (a diagnostic of the record) on one, the FIFO / counters / record on the other, each with
    // its state loads issued together (a load behind every store of the other kind cost an L2 round trip apiece)
    if (tid == 64) {
        if (st.nco_mode) {
            // exact-oscillator block: k_nco_exact left acquire_t.phase as the reference has it after this block; a later closed-form
            // block continues from its angle
            const float pr = st.nco_re, pi = st.nco_im;
            st.theta = atan2((double)pi, (double)pr);
            rec.phase_re = pr; rec.phase_im = pi;
        } else {
            double th = st.theta + (double)NSYM * SYM_N * st.dtheta;
            th -= 2 * M_PI * rint(th / (2 * M_PI));
            st.theta = th;
            rec.phase_re = (float)cos(th); rec.phase_im = (float)sin(th);
        }
    }
    if (tid == 0) {
        const int keep_extra = st.keep_extra, state = st.sync_state, cfo = st.cfo, bc_now = st.bc, psmi = st.psmi, cfo_wait = st.cfo_wait, next_samperr = st.samperr, nblocks = st.nblocks;
        const long long rd = st.rd;
        const float prev_angle = st.prev_angle, next_angle = st.angle;
        const int keep = SYM_N + (SYM_N / 2 - samperr) + keep_extra;
        st.keep_extra = 0;
        st.rd = rd + (WIN_N - keep);
        rec.state_after = state; rec.samperr = samperr; rec.cfo = cfo; rec.keep = keep;
        rec.bc = bc_now; rec.psmi = psmi; rec.cfo_wait = cfo_wait; rec.next_samperr = next_samperr;
        rec.prev_angle = prev_angle;
        rec.next_angle = next_angle;
        if (sh_i[3]) {                                         // the PIDS frame wave 1 decoded (streaming seam)
            rec.pids[0] = sh_pids_out[0]; rec.pids[1] = sh_pids_out[1]; rec.pids[2] = sh_pids_out[2];
            if (sh_pids_out[3]) rec.flags |= REC_PIDS_CRC;
        }
        st.nblocks = nblocks + 1;
        st.active = 0;
    }
    if (fuse_prepare && !db.ckpt) { __threadfence_block(); __syncthreads(); }     // block-uniform: the next block's bookkeeping reads the NCO phase
    if (db.ckpt) {                                             // block-uniform: window pipeline with the on-device L2 feedback
        // Replay checkpoint.  The reference judges a P1 frame's first L2 header inside this block (frame.c:535-540) and starts
        // the next one from SYNC_STATE_NONE when it fails; here the verdict comes from the deferred decode, windows later.
        // The state as of now -- after this block, before the next block's bookkeeping -- is what k_rollback rewinds to.
        __threadfence_block();
        __syncthreads();
        if (sh_i[2]) {
            const uint32_t *src = (const uint32_t *)&st;
            uint32_t *dst = (uint32_t *)(db.ckpt + (size_t)s * NWIN + parity);
            for (int k = tid; k < (int)(sizeof(StreamState) / 4); k += SYNC_NT) dst[k] = src[k];
        }
        __syncthreads();
    }
    if (tid == 0 && fuse_prepare) prepare_block(db, st, s, false);   // top of the NEXT block's acquire_process (FINE streams only)
    SYNC_MARK(7);
}

// Streaming seam, ONE stream, a block that needs no kernel behind the sync kernel (no P1 frame to decode, no extended sidebands): the report -- k_stream_tail's job -- by the
// sync kernel's own workgroup, one launch and ~8 us less on a chain the host waits for.  What the other waves of the workgroup stored is read past the vector L1
// (fence + barrier, then agent-scope loads): the record, the read position, the step counters.
template <int NT> __device__ __forceinline__ void stream_report_wg(const DevBuffers &db, int s, int first_rec, StreamReport *out, unsigned seq)
{
    __threadfence();
    __syncthreads();
    const int t = threadIdx.x;
    const StreamState &st = db.state[s];
    const int nblocks = (int)flow_load_u32((const unsigned *)&st.nblocks);
    const int n = min(max(nblocks - first_rec, 0), 4);
    constexpr int RW = sizeof(BlockRecord) / 4;
    for (int q = t; q < n * RW; q += NT) {
        const int k = q / RW, w = q % RW;
        ((uint32_t *)&out->rec[k])[w] = flow_load_u32((const unsigned *)&db.records[(size_t)s * db.rec_cap + ((first_rec + k) % db.rec_cap)] + w);
    }
    if (t < 4) { out->counters[t] = (int)flow_load_u32((const unsigned *)&db.counters[t]); flow_store_u32((unsigned *)&db.counters[t], 0u); }   // (through the caches: nothing of this kernel may still be on its way when the host, having seen the report, launches the next step)
    if (t == 0) { out->rd = (long long)flow_load_u64((const unsigned long long *)&st.rd); out->nblocks = nblocks; out->nrec = n; }
    __threadfence_system();
    __syncthreads();
    if (t == 0) { *(volatile unsigned *)&out->seq = seq; __threadfence_system(); }
}

template <int SYNC_NT>
__global__ __launch_bounds__(SYNC_NT) SYNC_OCCUPANCY(SYNC_NT) void k_sync(DevTables tb, DevBuffers db, const int *ids, int parity, int slot, int fuse_prepare, int window, int pids_inline, int do_prepare, int ext_refs)
{
    wave_set_priority_high();                                  // block-step chain = critical path; decode waves run at priority 0
    const int s = wave_uniform(stream_of(ids, blockIdx.x));    // in a scalar register: every address derived from it stays off the VGPR budget
    __shared__ __attribute__((aligned(16))) uint8_t lds[sizeof(SyncLds<SYNC_NT>)];
    sync_body<SYNC_NT>(lds, tb, db, s, parity, slot, fuse_prepare, window, pids_inline, do_prepare, ext_refs);
}

// The fast seam's form for ONE stream: the same body, then the report.  A kernel of its own, so that the batch kernel above keeps the code (and the resource footprint: 79 VGPRs, no
// scratch -- tests/test_codegen_guards.py) it had before the report existed.
struct SyncReportTail { StreamReport *out; unsigned seq; int first_rec; };
// the kernel's arguments as the kernarg segment lays them out (each at its natural alignment, like the members of a struct): where `tail` sits
struct SyncReportKernargs { DevTables tb; DevBuffers db; const int *ids; int parity, slot, fuse_prepare, window, pids_inline, do_prepare, ext_refs; SyncReportTail tail; };
__global__ __launch_bounds__(768) SYNC_OCCUPANCY(768) void k_sync_report(DevTables tb, DevBuffers db, const int *ids, int parity, int slot, int fuse_prepare, int window, int pids_inline, int do_prepare, int ext_refs,
                                                                       SyncReportTail tail)
{
    wave_set_priority_high();
    const int s = wave_uniform(stream_of(ids, blockIdx.x));
    __shared__ __attribute__((aligned(16))) uint8_t lds[sizeof(SyncLds<768>)];
    sync_body<768>(lds, tb, db, s, parity, slot, fuse_prepare, window, pids_inline, do_prepare, ext_refs);
#ifndef HIPEMU
    // The report's three arguments are fetched from the kernarg segment HERE, behind a fence the compiler cannot look through: as ordinary parameters they are loaded at the kernel's entry
    // with all the others and stay in scalar registers through the whole body, which has none to spare (104 SGPRs, ~250 spilled to VGPR lanes)
    const char *ka = (const char *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka) : : "memory");
    const SyncReportTail t = *(const SyncReportTail *)(ka + offsetof(SyncReportKernargs, tail));
    const DevBuffers &dbr = *(const DevBuffers *)(ka + offsetof(SyncReportKernargs, db));      // (likewise the three buffer pointers the report reads)
    (void)tail;
#else
    const SyncReportTail t = tail;
    const DevBuffers &dbr = db;
#endif
    stream_report_wg<768>(dbr, s, t.first_rec, t.out, t.seq);
}

void launch_sync(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, int parity, int slot, int fuse_prepare, int window, hipStream_t st, int lanes, int pids_inline, int do_prepare, int ext_refs,
                 StreamReport *report, unsigned report_seq, int report_first)
{
    // lanes: 0 = by the size of the stream set (see SYNC_OCCUPANCY above), else 256 / 768 (nrsc5hip_debug_tune NRSC5HIP_TUNE_SYNC_LANES)
    const int nt = lanes ? lanes : 768;                        // measured at 256 streams: 32.8 ms per pass with 768, 33.8 with 256 (profiles/r04_sync_lanes.txt)
    if (report && nstreams == 1 && nt == 768) { hipLaunchKernelGGL(k_sync_report, dim3(1), dim3(768), 0, st, tb, db, stream_ids, parity, slot, fuse_prepare, window, pids_inline, do_prepare, ext_refs, SyncReportTail{ report, report_seq, report_first }); return; }
    if (nt == 768) hipLaunchKernelGGL(k_sync<768>, dim3(nstreams), dim3(768), 0, st, tb, db, stream_ids, parity, slot, fuse_prepare, window, pids_inline, do_prepare, ext_refs);
    else hipLaunchKernelGGL(k_sync<256>, dim3(nstreams), dim3(256), 0, st, tb, db, stream_ids, parity, slot, fuse_prepare, window, pids_inline, do_prepare, ext_refs);
}

// ---- deferred PIDS decode: one wave per (slot, stream) with a staged frame -----------------------------------
// called by the 64 lanes of one wave; LDS scratch from the caller
__device__ __forceinline__ void pids_decode_wave(const DevTables &tb, const DevBuffers &db, int s, int parity, int slot, int8_t *coded, unsigned long long *dec, uint32_t *out)
{
    int *recp = db.pids_rec + ((size_t)s * NWIN + parity) * 16 + slot;
    const int r = *recp;
    if (r < 0) return;                                         // wave-uniform
    const int8_t *stage = db.pids_stage + (((size_t)s * NWIN + parity) * 16 + slot) * (3 * PIDS_LEN);
    const int lane = threadIdx.x & 63;
    static_assert((3 * PIDS_LEN) % 4 == 0 && 3 * PIDS_LEN / 4 <= 64, "the staged frame is one dword per lane");
    if (lane < 3 * PIDS_LEN / 4) ((uint32_t *)coded)[lane] = ((const uint32_t *)stage)[lane];
    WAVE_LDS_SYNC();
    viterbi_k7_wave_compact<PIDS_LEN>(coded, dec, out);        // the rotating-layout trellis in its compact form, inlined with the frame length a constant
    WAVE_LDS_SYNC();
    // (the CRC runs over a local copy: handed the record itself it re-read the words from global memory for each of its 80 bits -- ~17 us of the 22.7 us this decode used to
    //  take; round 6: by the whole wave, pids_crc_ok_wave)
    const uint32_t p[3] = { out[0] ^ tb.scr_pids[0], out[1] ^ tb.scr_pids[1], (out[2] ^ tb.scr_pids[2]) & 0xffffu };   // descramble (decode.c:470)
    const bool crc_ok = pids_crc_ok_wave(p);
    if (lane == 0) {
        BlockRecord &rec = db.records[(size_t)s * db.rec_cap + r];
        rec.pids[0] = p[0]; rec.pids[1] = p[1]; rec.pids[2] = p[2];
        if (crc_ok) atomicOr(&rec.flags, (uint32_t)REC_PIDS_CRC);
        *recp = -1;
    }
}

__global__ __launch_bounds__(64) void k_pids_decode(DevTables tb, DevBuffers db, const int *ids, int parity)
{
    const int s = stream_of(ids, blockIdx.y), slot = blockIdx.x;
    __shared__ __attribute__((aligned(16))) int8_t coded[3 * PIDS_LEN];
    __shared__ unsigned long long dec[PIDS_LEN + 64];
    __shared__ uint32_t out[4];
    pids_decode_wave(tb, db, s, parity, slot, coded, dec, out);
}

// Streaming seam: the tail of ONE stream's block step in one launch -- the block's PIDS frame (in-order mode files it in slot 0 of
// window slot 0), then what the host needs into pinned host memory: the step's counters, the FIFO read position and the records
// [first_rec, nblocks) (a block's record is final when its step ends) and, last of all, the sequence number the host is waiting
// for.  Leaves the step counters at zero for the next step.  (As two launches, k_pids_decode + the report: one more ~4 us dispatch
// on a chain the host waits for.)
__global__ __launch_bounds__(64) void k_stream_tail(DevTables tb, DevBuffers db, int s, int first_rec, StreamReport *out, unsigned seq, int do_pids)
{
    __shared__ __attribute__((aligned(16))) int8_t coded[3 * PIDS_LEN];
    __shared__ unsigned long long dec[PIDS_LEN + 64];
    __shared__ uint32_t bits[4];
    const int t = threadIdx.x;
    if (do_pids) pids_decode_wave(tb, db, s, 0, 0, coded, dec, bits);       // one wave: the whole workgroup
    __threadfence();
    __syncthreads();
    const StreamState &st = db.state[s];
    const int n = min(max(st.nblocks - first_rec, 0), 4);
    constexpr int RW = sizeof(BlockRecord) / 4;
    for (int q = t; q < n * RW; q += 64) {
        const int k = q / RW, w = q % RW;
        ((uint32_t *)&out->rec[k])[w] = ((const uint32_t *)&db.records[(size_t)s * db.rec_cap + ((first_rec + k) % db.rec_cap)])[w];
    }
    if (t < 4) { out->counters[t] = db.counters[t]; db.counters[t] = 0; }
    if (t == 0) { out->rd = st.rd; out->nblocks = st.nblocks; out->nrec = n; }
    __threadfence_system();
    __syncthreads();
    if (t == 0) { *(volatile unsigned *)&out->seq = seq; __threadfence_system(); }
}

void launch_stream_tail(const DevTables &tb, const DevBuffers &db, int s, int first_rec, StreamReport *out, unsigned seq, int do_pids, hipStream_t st)
{
    hipLaunchKernelGGL(k_stream_tail, dim3(1), dim3(64), 0, st, tb, db, s, first_rec, out, seq, do_pids);
}

void launch_pids_decode(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, int parity, int nslots, hipStream_t st)
{
    hipLaunchKernelGGL(k_pids_decode, dim3(nslots, nstreams), dim3(64), 0, st, tb, db, stream_ids, parity);
}

// ---- extended sidebands: interleaver IV (decode.c:344-376) for a completed block pair ----------------------------
// The interleaver is convolutional: the bit read at position i of a pair was written delay[i] positions earlier
// (1..N, N = 32 blocks), either earlier in this pair (take it from the pair buffer) or in the memory.  All reads of a
// pair happen before its writes, as in the reference's read-then-write loop, by splitting the pass at a barrier.
__global__ __launch_bounds__(1024) void k_px_deint(DevTables tb, DevBuffers db, const int *ids, int parity, int slot)
{
    const int s = stream_of(ids, blockIdx.y), ch = blockIdx.x;
    StreamState &st = db.state[s];
    const int len = st.px_go;                                  // block-uniform
    if (len == 0 || ch >= st.px_nch) return;
    const int N = 32 * len, tid = threadIdx.x;
    int I = st.px_pos, ready = st.px_ready;
    if (I == N) { I = 0; ready = 1; }
    const uint32_t *delay = len == PX_MAX ? tb.px_delay_wide : tb.px_delay_narrow;
    int8_t *mem = db.px_mem + ((size_t)s * 2 + ch) * PX_MEM;
    const int8_t *pair = db.px_pair + ((size_t)s * 2 + ch) * 2 * PX_MAX;
    int8_t *stage = db.px_stage + ((((size_t)s * NWIN + parity) * 8 + (slot >> 1)) * 2 + ch) * PX_DEPUNCT;
    int8_t vals[2 * PX_MAX / 1024];
#pragma unroll
    for (int r = 0; r < 2 * PX_MAX / 1024; r++) {
        const int i = tid + 1024 * r;
        vals[r] = 0;
        if (i < 2 * len) {
            const int d = (int)delay[i];
            vals[r] = d <= i ? pair[i - d] : mem[(I + i - d + N) % N];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2 * PX_MAX / 1024; r++) {
        const int i = tid + 1024 * r;
        if (i < 2 * len) {
            mem[I + i] = pair[i];
            const int o = (i >> 2) * 6 + (i & 3) + ((i & 3) >= 1) + ((i & 3) >= 3);   // kept positions 0, 2, 3, 5 of [1,0,1,1,0,1]
            stage[o] = vals[r];
            if ((i & 3) == 0) { stage[o + 1] = 0; stage[o + 4] = 0; }
        }
    }
    if (tid == 0) {
        PxJob &job = db.px_job[(((size_t)s * NWIN + parity) * 8 + (slot >> 1)) * 2 + ch];
        job.rec = ready ? st.px_record : -1; job.slot = st.px_slot; job.len = len; job.pad = 0;
    }
}

// both channels read px_pos / px_ready above; advance them once both are done
__global__ void k_px_commit(DevBuffers db, const int *ids, int nstreams)
{
    const int sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= nstreams) return;
    StreamState &st = db.state[stream_of(ids, sidx)];
    if (st.px_go == 0) return;
    const int N = 32 * st.px_go;
    if (st.px_pos == N) { st.px_pos = 0; st.px_ready = 1; }
    st.px_pos += 2 * st.px_go;
    st.px_go = 0;
}

void launch_px_deint(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, int parity, int slot, hipStream_t st)
{
    hipLaunchKernelGGL(k_px_deint, dim3(2, nstreams), dim3(1024), 0, st, tb, db, stream_ids, parity, slot);
    hipLaunchKernelGGL(k_px_commit, dim3((nstreams + 63) / 64), dim3(64), 0, st, db, stream_ids, nstreams);
}

// ---- staged P3 / P4 frames: K=7 Viterbi (nrsc5_conv_decode_p3_p4), descramble (decode.c:407-409,430-432) --------
__global__ __launch_bounds__(64) void k_px_decode(DevTables tb, DevBuffers db, const int *ids, int parity, int lane_id)
{
    const int s = stream_of(ids, blockIdx.y), j = blockIdx.x;   // j = pair slot * 2 + channel
    PxJob &job = db.px_job[((size_t)s * NWIN + parity) * 16 + j];
    if (job.rec < 0) return;                                   // wave-uniform
    const int len = job.len, ch = j & 1;
    const int8_t *coded = db.px_stage + (((size_t)s * NWIN + parity) * 16 + j) * PX_DEPUNCT;
    unsigned long long *dec = db.px_dec + (((size_t)lane_id * db.nstreams_alloc + s) * 16 + j) * (PX_MAX + 64);
    uint32_t *out = db.px_ring + (((size_t)s * db.px_slots + job.slot) * 2 + ch) * PX_WORDS;
    viterbi_k7_decode(coded, len, dec, out);
    __threadfence_block();
    __syncthreads();
    for (int w = threadIdx.x; w < len / 32; w += 64) out[w] ^= tb.scr_p1[w];
    if (threadIdx.x == 0) { job.pad = db.l2_px_ring ? 1 : 0; job.rec = -1; }     // pad: k_l2_index_px_window owes this frame its index
}

void launch_px_decode(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, int parity, int lane_id, hipStream_t st)
{
    hipLaunchKernelGGL(k_px_decode, dim3(16, nstreams), dim3(64), 0, st, tb, db, stream_ids, parity, lane_id);
    if (db.l2_px_ring) launch_l2_index_px_window(db, nstreams, stream_ids, parity, st);
}


// =====================================================================================================================================================
// k_flow (round 6): K consecutive block steps of a set of FINE zero-copy streams as ONE launch whose workgroups are work items of two kinds -- a PAIR of
// symbol transforms of one stream (the body of k_mixfft<1, 2>) or one stream's block step (the body of k_sync<256>) -- ordered so that every item depends only
// on items in front of it:
//     step e, stream t:   16 symbol pairs (e, t)  ->  block step (e, t)  ->  symbol pairs (e + 1, t)  -> ...
// As two launches per step (k_mixfft, then k_sync) the chain costs the SUM of the two kernels' latencies, 46 + 38 us, although the first is a throughput kernel
// and the second a latency chain that issues on ~12 % of its cycles: a launch boundary makes stream 0's block step wait for stream 255's last symbol, and the
// next step's symbols of stream 0 for stream 255's block step.  Two queues do not interleave at workgroup granularity (profiles/r05_two_engines.txt); one grid
// does: the block step of stream t is dispatched FLOW_LAG streams behind its symbols and runs beside the symbols of the streams that follow; the next step's
// symbols of stream t come a whole round later, when its block step has long finished.
//   * Work lists, one per XCD (streams idx % 8 == x: the hand-offs of a stream stay inside one L2 where the dispatcher's observed placement -- workgroup b on XCD
//     b % 8 -- holds; nothing depends on it).  A workgroup reads the XCD it runs on and draws a ticket from that list; if the list is exhausted, from the next one:
//     every item is drawn exactly once, and an item's dependencies have SMALLER tickets of the same list -- they were drawn by workgroups that are running or
//     done -- so no wait can be circular, whatever the dispatch order.
//   * Hand-offs (flow_ops.h; cdna_hip_programming.md Guideline 16): bins leave write-through and are counted per stream (flow.sym); the block step polls the
//     counter, takes ONE agent-scope acquire and reads with plain loads.  At its end the block step releases (its state is read by the stream's next block step,
//     possibly on another CU) and publishes the next block's symbol parameters as 8-byte {step tag, value} granules: the data is the flag.
//   * Every poll is bounded; a poll that gives up files an error word the host turns into NRSC5HIP_EDEVICE, the item completes without its work (the counters
//     still move: the grid drains instead of hanging).
// What stays with the two-kernel form: steps that run the acquisition kernels (a stream that is not FINE), the extended service modes' PX kernels, FIFO input,
// the exact oscillator, small stream sets (the per-stream chain is no shorter here; what is gained is the overlap between streams).
constexpr int FLOW_LAG = 8;                // a stream's block step is dispatched this many streams (x 16 symbol items) behind its symbols
constexpr int FLOW_GRANULES = 9;           // a00 (2), dtheta (2), theta (2), growth (2), active
constexpr unsigned FLOW_SPIN_LIMIT = 1u << 21;   // polls of ~0.25 us: ~0.5 s

struct FlowItem { int e, t, pair; };       // pair < 0: the block step of stream t in step e
// ticket q of a list of nx streams -> item (see the order above)
__device__ __forceinline__ FlowItem flow_item(int q, int nx)
{
    const int per = 17 * nx, e = q / per, r = q - e * per;
    const int L = nx < FLOW_LAG ? nx : FLOW_LAG;
    FlowItem it; it.e = e;
    if (r < 16 * L) { it.t = r >> 4; it.pair = r & 15; return it; }
    const int rb = r - 16 * L, nb = 17 * (nx - L);
    if (rb < nb) { const int k = rb / 17, j = rb - 17 * k; if (j < 16) { it.t = L + k; it.pair = j; } else { it.t = k; it.pair = -1; } return it; }
    it.t = nx - L + (rb - nb); it.pair = -1;
    return it;
}

union FlowLds { MixLds<2> mix; SyncLds<256> sync; };

// flow words (zeroed by the host before every launch): head[8] | err[8] | sym[n] | granules[n][FLOW_GRANULES] (8-byte aligned)
__global__ __launch_bounds__(256) void k_flow(DevTables tb, DevBuffers db, const int *ids, int n, int K, unsigned *flow, unsigned *err_host, int parity, int slot0, int window)
{
    wave_set_priority_high();
    __shared__ __attribute__((aligned(16))) uint8_t lds[sizeof(FlowLds)];
    __shared__ int sh_role[4];
    const int tid = threadIdx.x;
    unsigned *head = flow, *err = flow + 8, *symc = flow + 16;
    unsigned long long *gran = (unsigned long long *)(flow + 16 + ((n + 1) & ~1));
    if (tid == 0) {
        int x = flow_xcc_id(), q = -1, nx = 0;
        for (int k = 0; k < 8; k++, x = (x + 1) & 7) {
            nx = x < n ? (n - x + 7) >> 3 : 0;
            if (!nx) continue;
            q = (int)flow_add_u32(&head[x], 1u);
            if (q < K * 17 * nx) break;
            q = -1;
        }
        sh_role[0] = q; sh_role[1] = x; sh_role[2] = nx;
    }
    __syncthreads();
    const int q = sh_role[0];
    if (q < 0) return;                                         // more workgroups on this XCD than items anywhere: nothing left
    const FlowItem it = flow_item(q, sh_role[2]);
    const int idx = sh_role[1] + 8 * it.t;                     // position in the stream set
    const int s = wave_uniform(stream_of(ids, idx));
    if (it.pair >= 0) {
        // ---- two symbol transforms.  Their parameters: step 0 from the stream state (the launch boundary published it), later steps from the granules
        SymParams &sp = reinterpret_cast<FlowLds *>(lds)->mix.sh_sp;
        if (it.e == 0) {
            if (tid == 0) {
                const StreamState &st = db.state[s];
                sp.active = st.active; sp.a00 = (st.rd - st.base) + st.samperr_cur; sp.dtheta = st.dtheta; sp.theta = st.theta; sp.growth = st.growth; sp.nco_mode = 0;
            }
        } else if (tid < 64) {
            const unsigned long long *g = gran + (size_t)idx * FLOW_GRANULES + (tid < FLOW_GRANULES ? tid : 0);
            unsigned long long v = 0; bool ok = false, dead = false;
            for (unsigned spins = 0; ; spins++) {
                v = flow_load_u64(g);
                ok = (unsigned)(v >> 32) == (unsigned)it.e;
                if (__all(ok)) break;
                if (spins > FLOW_SPIN_LIMIT) { dead = true; break; }
                flow_sleep();
            }
            const unsigned lo = (unsigned)v;
            // lanes 0..8 hold one granule each: a00 = [0] | [1] << 32, dtheta = [2,3], theta = [4,5], growth = [6,7], active = [8]
            unsigned w[FLOW_GRANULES];                         // (every lane of the wave takes part in the reads: the CPU twin's readlane is a collective)
#pragma unroll
            for (int l = 0; l < FLOW_GRANULES; l++) w[l] = (unsigned)wave_readlane((int)lo, l);
            if (tid == 0) {
                const unsigned long long a = w[0] | (unsigned long long)w[1] << 32, d = w[2] | (unsigned long long)w[3] << 32, th = w[4] | (unsigned long long)w[5] << 32, gr = w[6] | (unsigned long long)w[7] << 32;
                sp.a00 = (long long)a; __builtin_memcpy(&sp.dtheta, &d, 8); __builtin_memcpy(&sp.theta, &th, 8); __builtin_memcpy(&sp.growth, &gr, 8);
                sp.active = dead ? 0 : (int)w[8]; sp.nco_mode = 0;
                if (dead) { flow_store_u32(&err[0], 1u + (unsigned)idx); err_host[0] = 1u + (unsigned)idx; }
            }
        }
        __syncthreads();
        mixfft_wg<1, 2, true>(lds, tb, db, s, it.pair, 0, &sp);
        flow_drain_stores();                                   // every wave: its write-through stores have left
        __syncthreads();
        if (tid == 0) flow_add_u32(&symc[idx], 1u);
        return;
    }
    // ---- the stream's block step: wait for its 32 symbols of this step, one acquire for the workgroup, then k_sync's body
    if (tid == 0) {
        const unsigned want = 16u * (unsigned)(it.e + 1);
        bool dead = false;
        for (unsigned spins = 0; flow_load_u32(&symc[idx]) < want; spins++) {
            if (spins > FLOW_SPIN_LIMIT) { dead = true; break; }
            flow_sleep();
        }
        if (dead) { flow_store_u32(&err[1], 1u + (unsigned)idx); err_host[1] = 1u + (unsigned)idx; }
        flow_acquire();
    }
    __syncthreads();
    sync_body<256>(lds, tb, db, s, parity, (slot0 + it.e) & 15, 1, window, 0, 0, 0);
    if (it.e + 1 >= K) return;                                 // the launch's last step: the launch boundary publishes
    flow_drain_stores();
    __syncthreads();
    if (tid == 0) {
        flow_release();                                        // this step's state, for the stream's next block step (another workgroup, maybe another XCD)
        const StreamState &st = db.state[s];
        const long long a00 = (st.rd - st.base) + st.samperr_cur;
        unsigned long long d, th, gr;
        { const double v = st.dtheta; __builtin_memcpy(&d, &v, 8); } { const double v = st.theta; __builtin_memcpy(&th, &v, 8); } { const double v = st.growth; __builtin_memcpy(&gr, &v, 8); }
        const unsigned w[FLOW_GRANULES] = { (unsigned)a00, (unsigned)((unsigned long long)a00 >> 32), (unsigned)d, (unsigned)(d >> 32), (unsigned)th, (unsigned)(th >> 32),
                                            (unsigned)gr, (unsigned)(gr >> 32), (unsigned)st.active };
        unsigned long long *g = gran + (size_t)idx * FLOW_GRANULES;
        const unsigned long long tag = (unsigned long long)(unsigned)(it.e + 1) << 32;
#pragma unroll
        for (int k = 0; k < FLOW_GRANULES; k++) flow_store_u64(&g[k], tag | w[k]);
    }
}

size_t flow_words(int n) { return 16 + (size_t)((n + 1) & ~1) + 2 * (size_t)n * FLOW_GRANULES; }

// K block steps of the n listed streams (all FINE, zero-copy, MP1 routing, closed-form oscillator: the caller checks); `flow` = flow_words(n) zeroed dwords;
// err_host: two words of device-visible host memory, written only when a poll gives up ([0]: a symbol item, [1]: a block step; 1 + position of the stream)
void launch_flow(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, int K, unsigned *flow, unsigned *err_host, int parity, int slot0, int window, hipStream_t st)
{
    const unsigned items = (unsigned)K * 17u * (unsigned)nstreams;
    hipLaunchKernelGGL(k_flow, dim3(items), dim3(256), 0, st, tb, db, stream_ids, nstreams, K, flow, err_host, parity, slot0, window);
}

}  // namespace nrsc5
