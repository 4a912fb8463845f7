// AM (hybrid MA1 / all-digital MA3) path for gfx950.  Replaces, for NRSC5_MODE_AM:
//   decimate_samples' 5-stage 32:1 cascade (input.c:70-91)                         -> k_am_decimate_cu8
//   acquire_process incl. the AM carrier regression (acquire.c:98-263)             -> k_am_block (fused)
//   sync_push / sync_process_am, find_ref_am, find_block_am (sync.c:209-252,612-767)-> k_am_block
//   decode_process_pids_am (decode.c:474-505)                                      -> k_am_block tail
//   decode_process_p1_p3_am, nrsc5_conv_decode_e1 / _e2_e3 (decode.c:507-554)      -> k_am_viterbi
//   interleaver_ma1 incl. the 3-frame diversity delay (decode.c:74-231)            -> k_am_interleave
//
// The AM stream is 32 x slower than FM (46.5 kS/s), so one workgroup owns one stream for a whole block:
// the 32 x 256-point FFTs, the carrier line fit and sync_process_am all run out of one 64 KB LDS tile and
// only hard symbols (3.2 KB per block) go back to HBM.  The K=9 trellis has 256 states = one per work-item.
#include <atomic>
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "fastmath.h"
#include "wave_ops.h"
#include "l2_header.h"

namespace nrsc5 {

__device__ inline int stream_of(const int *ids, int idx) { return ids ? ids[idx] : idx; }

// =====================================================================================================
// K1-AM: cu8 -> (Q15 >> 4) -> five cascaded 15-tap half-bands
// =====================================================================================================
// Output m of the cascade depends on raw samples [32 m - 434, 32 m] and exists once raw sample 32 m + 31 has
// arrived.  A tile of AMD_T outputs recomputes the dependency cone from raw samples (history kept in the stream
// state), stage by stage in LDS, so tiles are independent of each other and of how the caller chunks its pushes.
constexpr int AMD_T = 32;
constexpr int AMD_N0 = 32 * (AMD_T - 1) + 435;   // raw samples per tile
constexpr int AMD_N1 = 16 * (AMD_T - 1) + 211;
constexpr int AMD_N2 = 8 * (AMD_T - 1) + 99;
constexpr int AMD_N3 = 4 * (AMD_T - 1) + 43;
constexpr int AMD_N4 = 2 * (AMD_T - 1) + 15;

__device__ inline int am_hb_dot(const int16_t *a, int stride, int t0, int t1, int t2, int t3)
{
    int acc = 0;                                               // firdecim_q15.c:137-151: shift before add, int16 accumulator
    acc = (int16_t)(acc + (((a[0] + a[14 * stride]) * t0) >> 15));
    acc = (int16_t)(acc + (((a[2 * stride] + a[12 * stride]) * t1) >> 15));
    acc = (int16_t)(acc + (((a[4 * stride] + a[10 * stride]) * t2) >> 15));
    acc = (int16_t)(acc + (((a[6 * stride] + a[8 * stride]) * t3) >> 15));
    return (int16_t)(acc + a[7 * stride]);
}

// raw sample r (absolute index since reset) of the virtual stream [history | chunk]
__device__ inline void am_raw_fetch(const AmStream &am, const uint8_t *iq, long long r, long long nraw_new, int &re, int &im)
{
    re = 0; im = 0;
    if (r < 0) return;
    unsigned a, b;
    if (r < am.raw_count) {
        const long long h = r - (am.raw_count - AM_RAW_HIST);
        if (h < 0) return;
        a = am.raw_hist[2 * h]; b = am.raw_hist[2 * h + 1];
    } else {
        const long long k = r - am.raw_count;
        if (k >= nraw_new) return;
        a = iq[2 * k]; b = iq[2 * k + 1];
    }
    re = (((int)a - 127) * 64) >> 4;                           // U8_Q15 then x >>= 4 (input.c:67-70)
    im = (((int)b - 127) * 64) >> 4;
}

__global__ __launch_bounds__(256) void k_am_decimate_cu8(DevTables tb, DevBuffers db, const int *ids,
                                                         const uint8_t *iq_base, long long iq_stride, const unsigned *nbytes)
{
    const int sidx = blockIdx.y;
    const int s = stream_of(ids, sidx);
    const StreamState &st = db.state[s];
    const AmStream &am = db.am[s];
    const long long nraw = nbytes[sidx] / 2;
    const long long m_first = am.raw_count / 32, m_end = (am.raw_count + nraw) / 32;
    const long long m0 = m_first + (long long)blockIdx.x * AMD_T;
    if (m0 >= m_end) return;
    const uint8_t *iq = iq_base + (size_t)sidx * iq_stride;
    __shared__ int16_t bufA[2 * AMD_N0], bufB[2 * AMD_N1];     // interleaved re, im
    const int tid = threadIdx.x;
    const int t0 = tb.hb_q15[0], t1 = tb.hb_q15[1], t2 = tb.hb_q15[2], t3 = tb.hb_q15[3];

    const long long lo0 = 32 * m0 - 434;
    for (int k = tid; k < AMD_N0; k += 256) {
        int re, im;
        am_raw_fetch(am, iq, lo0 + k, nraw, re, im);
        bufA[2 * k] = (int16_t)re; bufA[2 * k + 1] = (int16_t)im;
    }
    // The first outputs after a reset: what lies in front of sample 0 of stage l's input is the window content the reset left there
    // (AmStream::seed; zeros for a fresh session), not something computed from earlier raw samples.  Stage l's local sample k is its
    // absolute sample base_l + k, base = lo0, 16 m0 - 210, 8 m0 - 98, 4 m0 - 42, 2 m0 - 14.
    const bool edge = m0 < 14;                                 // block-uniform: only then does the cone reach below sample 0
    if (edge) {
        __syncthreads();
        if (tid < 14 && -14 + tid - lo0 >= 0 && -14 + tid - lo0 < AMD_N0) { const int k = (int)(-14 + tid - lo0); bufA[2 * k] = am.seed[0][tid].r; bufA[2 * k + 1] = am.seed[0][tid].i; }
    }
#define AM_SEED_STAGE(buf, n, base, l) do { if (edge) { __syncthreads(); const long long kk = -14 + tid - (base); \
        if (tid < 14 && kk >= 0 && kk < (n)) { (buf)[2 * kk] = am.seed[l][tid].r; (buf)[2 * kk + 1] = am.seed[l][tid].i; } } } while (0)
    __syncthreads();
    // local index jl of a stage's output reads the previous stage's local samples 2 jl .. 2 jl + 14
    for (int k = tid; k < 2 * AMD_N1; k += 256) bufB[k] = (int16_t)am_hb_dot(bufA + 4 * (k >> 1) + (k & 1), 2, t0, t1, t2, t3);
    AM_SEED_STAGE(bufB, AMD_N1, 16 * m0 - 210, 1);
    __syncthreads();
    for (int k = tid; k < 2 * AMD_N2; k += 256) bufA[k] = (int16_t)am_hb_dot(bufB + 4 * (k >> 1) + (k & 1), 2, t0, t1, t2, t3);
    AM_SEED_STAGE(bufA, AMD_N2, 8 * m0 - 98, 2);
    __syncthreads();
    for (int k = tid; k < 2 * AMD_N3; k += 256) bufB[k] = (int16_t)am_hb_dot(bufA + 4 * (k >> 1) + (k & 1), 2, t0, t1, t2, t3);
    AM_SEED_STAGE(bufB, AMD_N3, 4 * m0 - 42, 3);
    __syncthreads();
    for (int k = tid; k < 2 * AMD_N4; k += 256) bufA[k] = (int16_t)am_hb_dot(bufB + 4 * (k >> 1) + (k & 1), 2, t0, t1, t2, t3);
    AM_SEED_STAGE(bufA, AMD_N4, 2 * m0 - 14, 4);
#undef AM_SEED_STAGE
    __syncthreads();
    if (tid < AMD_T && m0 + tid < m_end) {
        c16 y;
        y.r = (int16_t)am_hb_dot(bufA + 4 * tid, 2, t0, t1, t2, t3);
        y.i = (int16_t)am_hb_dot(bufA + 4 * tid + 1, 2, t0, t1, t2, t3);
        db.q15[(size_t)s * db.q15_cap + (st.wr - st.base) + (m0 - m_first) + tid] = y;
    }
}

__global__ __launch_bounds__(256) void k_am_decimate_commit(DevTables tb, DevBuffers db, const int *ids, const uint8_t *iq_base, long long iq_stride, const unsigned *nbytes)
{
    const int sidx = blockIdx.x;
    const int s = stream_of(ids, sidx);
    StreamState &st = db.state[s];
    AmStream &am = db.am[s];
    const long long nraw = nbytes[sidx] / 2;
    if (nraw == 0) return;
    const uint8_t *iq = iq_base + (size_t)sidx * iq_stride;
    __shared__ uint8_t nh[2 * AM_RAW_HIST];
    const long long first = am.raw_count + nraw - AM_RAW_HIST;
    for (int k = threadIdx.x; k < AM_RAW_HIST; k += 256) {
        const long long r = first + k;
        uint8_t a = 127, b = 127;                              // never read: indices before the stream start
        if (r >= am.raw_count) { a = iq[2 * (r - am.raw_count)]; b = iq[2 * (r - am.raw_count) + 1]; }
        else if (r >= 0 && r >= am.raw_count - AM_RAW_HIST) { const long long h = r - (am.raw_count - AM_RAW_HIST); a = am.raw_hist[2 * h]; b = am.raw_hist[2 * h + 1]; }
        nh[2 * k] = a; nh[2 * k + 1] = b;
    }
    // What the last compaction of each stage's window inside this chunk leaves at its front (StaleWindows, nrsc5_dev.h).  decim[0] takes
    // every raw sample (>> 4): an FM session after the next reset starts from these too.
    const long long p0 = stale_start(st.stale.hb_pushed, nraw, 14);
    if (p0 != STALE_NONE && threadIdx.x < 14) {
        int re, im;
        am_raw_fetch(am, iq, am.raw_count + p0 + threadIdx.x, nraw, re, im);
        st.stale.hb[threadIdx.x].r = (int16_t)re; st.stale.hb[threadIdx.x].i = (int16_t)im;
    }
    // decim[l], l = 1..4, has taken 2 floor(raw / 2^(l+1)) samples y_l (y_0 = raw >> 4, y_l[j] = half-band over y_(l-1)[2j-14 .. 2j]): the 14 in
    // front of its last compaction are recomputed from raw samples -- 14 -> 41 -> 95 -> 203 -> 419, at most 465 raw samples back from the
    // chunk's first (AM_RAW_HIST covers it)
    {
        __shared__ int16_t cA[2 * 419], cB[2 * 203];
        const int t0 = tb.hb_q15[0], t1 = tb.hb_q15[1], t2 = tb.hb_q15[2], t3 = tb.hb_q15[3];
        for (int l = 1; l <= 4; l++) {
            const long long a_l = 2 * (am.raw_count >> (l + 1)), b_l = 2 * ((am.raw_count + nraw) >> (l + 1));
            const long long p = stale_start(a_l, b_l - a_l, 14);
            if (p == STALE_NONE) continue;                     // block-uniform
            long long base[5]; int n[5];
            base[l] = a_l + p; n[l] = 14;
            for (int q = l; q >= 1; q--) { base[q - 1] = 2 * base[q] - 14; n[q - 1] = 2 * n[q] + 13; }
            // level q is held in cA when l - q is even, in cB when odd: the 419 samples of l = 4 and the 203 of l = 3 land in the buffer that fits them
            int16_t *cur = (l & 1) ? cB : cA;
            for (int k = threadIdx.x; k < n[0]; k += 256) {
                int re, im;
                am_raw_fetch(am, iq, base[0] + k, nraw, re, im);
                cur[2 * k] = (int16_t)re; cur[2 * k + 1] = (int16_t)im;
            }
            __syncthreads();
            for (int q = 1; q <= l; q++) {
                int16_t *nxt = ((l - q) & 1) ? cB : cA;
                for (int k = threadIdx.x; k < 2 * n[q]; k += 256) nxt[k] = (int16_t)am_hb_dot(cur + 4 * (k >> 1) + (k & 1), 2, t0, t1, t2, t3);
                __syncthreads();
                cur = nxt;
            }
            if (threadIdx.x < 14) { st.stale.am_stage[l - 1][threadIdx.x].r = cur[2 * threadIdx.x]; st.stale.am_stage[l - 1][threadIdx.x].i = cur[2 * threadIdx.x + 1]; }
            __syncthreads();
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 2 * AM_RAW_HIST; k += 256) am.raw_hist[k] = nh[k];
    if (threadIdx.x == 0) {
        st.wr += (am.raw_count + nraw) / 32 - am.raw_count / 32;
        am.raw_count += nraw;
        st.stale.hb_pushed += nraw;
    }
}

void launch_am_decimate_cu8(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids,
                            const uint8_t *iq_base, long long iq_stride, const unsigned *nbytes, unsigned max_nbytes, hipStream_t st)
{
    const unsigned max_out = max_nbytes / 64 + 1;
    hipLaunchKernelGGL(k_am_decimate_cu8, dim3((max_out + AMD_T - 1) / AMD_T, nstreams), dim3(256), 0, st, tb, db, stream_ids, iq_base, iq_stride, nbytes);
    hipLaunchKernelGGL(k_am_decimate_commit, dim3(nstreams), dim3(256), 0, st, tb, db, stream_ids, iq_base, iq_stride, nbytes);
}

// =====================================================================================================
// K=9 rate-1/3 tail-biting Viterbi, 256 states = 256 work-items (conv_dec.c:402-453, conv_gen.h:32-123)
// =====================================================================================================
// Work-item n owns new state n: predecessors 2b, 2b+1 with b = n & 127, branch metric +m for n < 128 and -m above
// (acs_butterfly).  int32 metrics, no normalisation (inputs are +-1/0: |metric| <= 3 per step); ties pick
// predecessor 2b+1 as `if (sum0 > sum1)` does.  Decisions: one ballot per wave and step (4 x u64 per step).
struct K9Smem {
    int metric[2][256];
    int8_t soft[3 * 256];
    unsigned long long chunk[4 * 256];
    int red_val[4], red_idx[4];
    int state;
};

__device__ inline void viterbi_k9_block(const int8_t *coded, int len, unsigned g0, unsigned g1, unsigned g2,
                                        unsigned long long *dec, uint32_t *out, K9Smem &sm)
{
    const int n = threadIdx.x, b = n & 127;
    const unsigned reg = ((unsigned)b << 1) & 0xfeu;           // gen_state_info, conv_dec.c:137-153
    const int flip = n >= 128 ? -1 : 1;
    const int sg0 = flip * ((__popc(reg & g0) & 1) ? 1 : -1);
    const int sg1 = flip * ((__popc(reg & g1) & 1) ? 1 : -1);
    const int sg2 = flip * ((__popc(reg & g2) & 1) ? 1 : -1);
    const int steps = len + 2 * VIT_EXTRA, j0 = len - VIT_EXTRA;
    int cur = 0;
    sm.metric[0][n] = 0;                                       // reset_decoder: all-zero for tail biting
    for (int t0 = 0; t0 < steps; t0 += 256) {
        __syncthreads();
        if (t0 + n < steps) {
            const int j = (j0 + t0 + n) % len;
            sm.soft[3 * n] = coded[3 * j]; sm.soft[3 * n + 1] = coded[3 * j + 1]; sm.soft[3 * n + 2] = coded[3 * j + 2];
        }
        __syncthreads();
        const int nst = min(256, steps - t0);
        for (int s = 0; s < nst; s++) {
            const int m = sm.soft[3 * s] * sg0 + sm.soft[3 * s + 1] * sg1 + sm.soft[3 * s + 2] * sg2;
            const int e = sm.metric[cur][2 * b], o = sm.metric[cur][2 * b + 1];
            const int pa = e + m, pc = o - m;
            const bool take_e = pa > pc;
            sm.metric[cur ^ 1][n] = take_e ? pa : pc;
            const unsigned long long w = __ballot(!take_e);    // bit = 1: survivor came from 2b+1
            if ((n & 63) == 0) dec[(size_t)(t0 + s) * 4 + (n >> 6)] = w;
            cur ^= 1;
            __syncthreads();
        }
    }
    // end state: first maximum in state order (conv_dec.c:310-318)
    {
        int v = sm.metric[cur][n], idx = n;
        for (int m = 32; m >= 1; m >>= 1) {
            const int ov = __shfl_xor(v, m), oi = __shfl_xor(idx, m);
            if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
        }
        if ((n & 63) == 0) { sm.red_val[n >> 6] = v; sm.red_idx[n >> 6] = idx; }
        __threadfence_block();
        __syncthreads();
        if (n == 0) {
            for (int w = 1; w < 4; w++) if (sm.red_val[w] > v) { v = sm.red_val[w]; idx = sm.red_idx[w]; }
            sm.state = idx;
        }
    }
    // traceback: decisions staged through LDS 256 steps at a time, walked by one work-item
    const int nchunks = (steps + 255) / 256;
    uint32_t accw = 0; int accidx = -1;
    for (int c = nchunks - 1; c >= 0; c--) {
        const int t0 = c * 256, nst = min(256, steps - t0);
        __syncthreads();
        for (int k = n; k < 4 * nst; k += 256) sm.chunk[k] = dec[(size_t)t0 * 4 + k];
        __syncthreads();
        if (n == 0) {
            unsigned state = (unsigned)sm.state;
            for (int s = nst - 1; s >= 0; s--) {
                const int t = t0 + s;
                const unsigned bit = (unsigned)(sm.chunk[4 * s + (state >> 6)] >> (state & 63)) & 1u;
                if (t >= VIT_EXTRA && t < len + VIT_EXTRA) {
                    const int i = t - VIT_EXTRA;
                    if ((i >> 5) != accidx) { if (accidx >= 0) out[accidx] = accw; accidx = i >> 5; accw = 0; }
                    accw |= ((state >> 7) & 1u) << (i & 31);   // vals[state]: the newest input bit
                }
                state = ((state << 1) & 0xfeu) | bit;           // vstate_lshift
            }
            sm.state = (int)state;
        }
    }
    if (n == 0 && accidx >= 0) out[accidx] = accw;
    __threadfence_block();
    __syncthreads();
}

// ---- the same trellis on ONE wave64, two steps per LDS round trip ------------------------------------------------------
// Lane L reads old states 4L..4L+3 (one 16-byte LDS read).  Step t: butterflies 2L and 2L+1 give the four intermediate
// states (i0 << 7) | (2L + xh); step t+1: butterflies L and L + 64 combine them into the four new states
// (i1 << 7) | (i0 << 6) | L, written at stride 64 (conflict-free) into the other half of a ping-pong buffer, where they are
// again states 4L'..4L'+3 of some lane L'.  Same eight compares per lane as two single steps, same tie rule, same int32
// metrics -> bit-identical decisions; half the LDS latency and loop overhead per trellis step.
// No workgroup barrier: a single-wave workgroup orders its own LDS traffic (WAVE_LDS_SYNC, wave_ops.h).
// Decisions per step pair: one byte per lane -- bit i0 * 2 + xh for step t (1 = survivor from old state 4L + 2 xh + 1),
// bit 4 + i1 * 2 + i0 for step t+1 (1 = survivor from xh = 1) -- so a step pair of the traceback needs ONE byte, the one
// of lane n & 63: prev = ((n & 63) << 2) | xh << 1 | xl, read at a wave-uniform address from a chunk staged in LDS.
struct K9WSmem { int metric[2][256]; };   // 2 KB per frame in flight (the traceback stages decisions in the same 2 KB): 32 decode
                                         // workgroups per CU still leave room for k_am_block's 70 KB tile

__device__ inline int k9_sign_word(unsigned b, unsigned g0, unsigned g1, unsigned g2)
{
    const unsigned reg = (b << 1) & 0xfeu;
    const int s0 = (__popc(reg & g0) & 1) ? 1 : -1, s1 = (__popc(reg & g1) & 1) ? 1 : -1, s2 = (__popc(reg & g2) & 1) ? 1 : -1;
    return (s0 & 0xff) | ((s1 & 0xff) << 8) | ((s2 & 0xff) << 16);
}

__device__ inline int k9_soft_word(const int8_t *coded, int j)
{
    return (coded[3 * j] & 0xff) | ((coded[3 * j + 1] & 0xff) << 8) | ((coded[3 * j + 2] & 0xff) << 16);
}

struct K9Signs { int a, b, c, d; };
__device__ inline K9Signs k9_signs(int lane, unsigned g0, unsigned g1, unsigned g2)
{
    K9Signs sg;
    sg.a = k9_sign_word(2u * lane, g0, g1, g2); sg.b = k9_sign_word(2u * lane + 1u, g0, g1, g2);             // step t: b = 2L + xh
    sg.c = k9_sign_word((unsigned)lane, g0, g1, g2); sg.d = k9_sign_word((unsigned)lane + 64u, g0, g1, g2);   // step t+1: b = (i0 << 6) | L
    return sg;
}

// A frame is steps = len + 2 * VIT_EXTRA trellis steps (even for every frame length of the AM path) = npairs step pairs, walked
// forward in chunks of 64 step pairs and backward in chunks of 32.
__host__ __device__ inline int k9_pairs(int len) { return (len + 2 * VIT_EXTRA) >> 1; }
__host__ __device__ inline int k9_chunks(int len) { return (k9_pairs(len) + 63) >> 6; }

// Forward pass over the chunks [c0, c1): the metrics are in sm.metric[0] on entry (every chunk but the frame's last is an even
// number of step pairs, so a chunk boundary always finds them there); returns the half that holds them after the last pair.
// Decisions are written for chunks >= cstore only (a segment wave's warm-up chunks belong to its predecessor), and the metrics
// the wave enters chunk `cstore` with go to `snap` (k9_forward_fix checks them against the predecessor's end metrics).
__device__ inline int k9_forward_chunks(const int8_t *coded, int len, const K9Signs &sg, unsigned long long *dec, K9WSmem &sm,
                                        int c0, int c1, int cstore, int *snap)
{
    const int lane = threadIdx.x & 63;
    const int j0 = len - VIT_EXTRA, npairs = k9_pairs(len);
    int cur = 0;
    for (int c = c0; c < c1; c++) {
        const int p0 = c << 6, np = min(64, npairs - p0);
        if (snap && c == cstore) *(int4 *)&snap[4 * lane] = *(const int4 *)&sm.metric[cur][4 * lane];
        const bool store = c >= cstore;                        // wave-uniform
        int aw0 = 0, aw1 = 0;                                   // this lane's step pair of the chunk
        if (lane < np) {
            const int t = 2 * (p0 + lane);
            aw0 = k9_soft_word(coded, (j0 + t) % len);
            aw1 = k9_soft_word(coded, (j0 + t + 1) % len);
        }
        for (int s = 0; s < np; s++) {
            const int a0 = wave_readlane(aw0, s), a1 = wave_readlane(aw1, s);
            const int mA = dot4_i8(a0, sg.a, 0), mB = dot4_i8(a0, sg.b, 0), nC = dot4_i8(a1, sg.c, 0), nD = dot4_i8(a1, sg.d, 0);
            const int4 old = *(const int4 *)&sm.metric[cur][4 * lane];
            // step t
            const int e00 = old.x + mA, o00 = old.y - mA, e10 = old.x - mA, o10 = old.y + mA;   // xh = 0: i0 = 0, i0 = 1
            const int e01 = old.z + mB, o01 = old.w - mB, e11 = old.z - mB, o11 = old.w + mB;   // xh = 1
            const bool t00 = e00 > o00, t01 = e01 > o01, t10 = e10 > o10, t11 = e11 > o11;       // t[i0][xh]: survivor from the even predecessor
            const int u00 = t00 ? e00 : o00, u01 = t01 ? e01 : o01, u10 = t10 ? e10 : o10, u11 = t11 ? e11 : o11;
            // step t+1: new state (i1, i0, L) from u[i0][0] (even) and u[i0][1] (odd)
            const int f00 = u00 + nC, p00 = u01 - nC, f10 = u00 - nC, p10 = u01 + nC;           // i0 = 0: i1 = 0, i1 = 1
            const int f01 = u10 + nD, p01 = u11 - nD, f11 = u10 - nD, p11 = u11 + nD;           // i0 = 1
            const bool r00 = f00 > p00, r01 = f01 > p01, r10 = f10 > p10, r11 = f11 > p11;       // r[i1][i0]
            int *nxt = sm.metric[cur ^ 1];
            nxt[lane] = r00 ? f00 : p00;                        // state (0, 0, L)
            nxt[64 + lane] = r01 ? f01 : p01;                   // state (0, 1, L)
            nxt[128 + lane] = r10 ? f10 : p10;                  // state (1, 0, L)
            nxt[192 + lane] = r11 ? f11 : p11;                  // state (1, 1, L)
            // this lane's eight decisions of the step pair in one byte: bit i0*2+xh for step t, bit 4+i1*2+i0 for step t+1
            const unsigned dbyte = (t00 ? 0u : 1u) | (t01 ? 0u : 2u) | (t10 ? 0u : 4u) | (t11 ? 0u : 8u)
                                 | (r00 ? 0u : 16u) | (r01 ? 0u : 32u) | (r10 ? 0u : 64u) | (r11 ? 0u : 128u);
            if (store) ((uint8_t *)dec)[(size_t)(p0 + s) * 64 + lane] = (uint8_t)dbyte;     // one 64-byte row per step pair, fire and forget
            cur ^= 1;
            WAVE_LDS_SYNC();
        }
    }
    return cur;
}

// end state: first maximum in state order (conv_dec.c:310-318); m = this lane's metrics of states 4L .. 4L+3
__device__ inline unsigned k9_end_state(int4 m)
{
    const int lane = threadIdx.x & 63;
    int v = m.x, idx = 4 * lane;
    if (m.y > v) { v = m.y; idx = 4 * lane + 1; }
    if (m.z > v) { v = m.z; idx = 4 * lane + 2; }
    if (m.w > v) { v = m.w; idx = 4 * lane + 3; }
    for (int k = 32; k >= 1; k >>= 1) {
        const int ov = __shfl_xor(v, k), oi = __shfl_xor(idx, k);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    return (unsigned)wave_uniform(idx);
}

// Traceback over the 32-pair chunks c_hi - 1 .. c_lo (downwards) from `state`, two steps per iteration; a chunk's decisions are
// staged in LDS (the metrics are dead: 2 KB = 32 step pairs) and looked up at a wave-uniform address.  Output words are written for
// chunks < c_out only (the chunks above are a segment wave's run-in); `arrive` = the state on entering chunk c_out - 1.
// Chunk c holds steps 64 c .. 64 c + 63 = frame bits 64 c - 32 .. 64 c + 31: words 2c - 1 (low half) and 2c (high half), and no
// other chunk writes those words.
__device__ inline unsigned k9_traceback_chunks(const unsigned long long *dec, int len, K9WSmem &sm, unsigned state, int c_hi, int c_lo, int c_out,
                                               uint32_t *out, unsigned &arrive)
{
    const int lane = threadIdx.x & 63;
    const int steps = len + 2 * VIT_EXTRA, npairs = k9_pairs(len);
    unsigned long long *stage = (unsigned long long *)&sm.metric[0][0];
    const uint8_t *db8 = (const uint8_t *)stage;
    for (int c = c_hi - 1; c >= c_lo; c--) {
        if (c == c_out - 1) arrive = state;
        const int p0 = c << 5, np = min(32, npairs - p0);
        for (int k = lane; k < 8 * np; k += 64) stage[k] = dec[(size_t)p0 * 8 + k];
        WAVE_LDS_SYNC();
        unsigned long long obits = 0;                           // output bits of steps 2 p0 .. 2 p0 + 63
        for (int s = np - 1; s >= 0; s--) {
            const unsigned i1 = state >> 7, i0 = (state >> 6) & 1u, L = state & 63u;
            const unsigned q = db8[64 * s + L];                  // lane L's decision byte of this step pair: one broadcast read
            const unsigned xh = (q >> (4 + 2 * i1 + i0)) & 1u;
            const unsigned xl = (q >> (2 * i0 + xh)) & 1u;
            obits = (obits << 2) | (unsigned long long)(i0 | (i1 << 1));    // the pair walked last (s = 0) ends in bits 0..1
            state = (unsigned)wave_uniform((int)((L << 2) | (xh << 1) | xl));
        }
        WAVE_LDS_SYNC();
        if (lane == 0 && c < c_out) {
            const int wl = 2 * c - 1, wh = 2 * c;
            if (wl >= 0 && wl * 32 < len) out[wl] = (uint32_t)obits;
            if (wh * 32 < len && 2 * p0 + 32 < steps) out[wh] = (uint32_t)(obits >> 32);
        }
    }
    return state;
}

__device__ inline void viterbi_k9_wave(const int8_t *coded, int len, unsigned g0, unsigned g1, unsigned g2,
                                       unsigned long long *dec, uint32_t *out, K9WSmem &sm, int phases = 3)
{
    const int lane = threadIdx.x & 63;
    const K9Signs sg = k9_signs(lane, g0, g1, g2);
    const int npairs = k9_pairs(len), nchunks = k9_chunks(len);
    for (int k = 0; k < 4; k++) sm.metric[0][4 * lane + k] = 0;   // reset_decoder: all-zero for tail biting
    WAVE_LDS_SYNC();
    const int cur = (phases & 1) ? k9_forward_chunks(coded, len, sg, dec, sm, 0, nchunks, 0, nullptr) : 0;
    unsigned state = k9_end_state(*(const int4 *)&sm.metric[cur][4 * lane]);
    __threadfence_block();
    __syncthreads();
    unsigned arrive = 0;
    if (phases & 2) k9_traceback_chunks(dec, len, sm, state, (npairs + 31) >> 5, 0, (npairs + 31) >> 5, out, arrive);
    __threadfence_block();
    __syncthreads();
}

// ---- the same decode in segment waves ---------------------------------------------------------------------------------
// FORWARD.  The chunks of a frame are cut into up to K9_GMAX segments, one wave each, all running at once.  Segment g > 0 cannot
// know the metrics its first step starts from, so it starts `warm` chunks early from all-zero metrics -- survivor paths merge
// within a few constraint lengths, after which metric DIFFERENCES no longer depend on where the wave started -- and notes the
// metrics it reaches its first own step with (snap).  Decisions depend on metric differences only (int32 sums, no saturation,
// no normalisation): k9_forward_fix walks the boundaries in order and accepts segment g iff snap[g] - snap[g][0] equals the TRUE
// end metrics of segment g - 1 minus their element 0; otherwise it re-runs segment g from those.  Exact for any segment count and
// any warm-up, including 0 (the test hook that forces every repair).
// TRACEBACK.  Segment g < last starts K9_TB_RUNIN chunks above its own chunks from state 0 -- survivors merge going backwards
// too -- and notes the state it enters its own chunks with (arrive) and leaves them with (leave); the last segment starts from
// the true end state.  k9_traceback_fix walks down from the last segment: segment g is accepted iff arrive[g] is the state the
// segment above truly left with, else it is walked again from that state.  Output words are per chunk, so a repair rewrites
// exactly the words of its segment.
// K9_GMAX, K9_WARM (chunks of 64 step pairs) and K9_TB_RUNIN (chunks of 32 step pairs): nrsc5_dev.h

__host__ __device__ inline int k9_seg_chunks(int len, int G) { return (k9_chunks(len) + G - 1) / G; }
__host__ __device__ inline int k9_seg_count(int len, int G) { const int per = k9_seg_chunks(len, G); return (k9_chunks(len) + per - 1) / per; }

__device__ inline void k9_forward_segment(const int8_t *coded, int len, unsigned g0, unsigned g1, unsigned g2, unsigned long long *dec,
                                          K9Meta &meta, K9WSmem &sm, int g, int G, int warm)
{
    const int lane = threadIdx.x & 63;
    const int nch = k9_chunks(len), per = k9_seg_chunks(len, G);
    const int c0 = g * per, c1 = min(nch, c0 + per);
    if (c0 >= nch) return;                                     // wave-uniform
    const K9Signs sg = k9_signs(lane, g0, g1, g2);
    for (int k = 0; k < 4; k++) sm.metric[0][4 * lane + k] = 0;
    WAVE_LDS_SYNC();
    const int cur = k9_forward_chunks(coded, len, sg, dec, sm, g ? max(0, c0 - warm) : 0, c1, c0, g ? meta.snap[g] : nullptr);
    *(int4 *)&meta.uend[g][4 * lane] = *(const int4 *)&sm.metric[cur][4 * lane];
}

// one wave per frame, after every segment wave has finished: returns the end state of the frame
__device__ inline unsigned k9_forward_fix(const int8_t *coded, int len, unsigned g0, unsigned g1, unsigned g2, unsigned long long *dec,
                                          K9Meta &meta, K9WSmem &sm, int G, unsigned *stats)
{
    const int lane = threadIdx.x & 63;
    const int nch = k9_chunks(len), per = k9_seg_chunks(len, G), nseg = k9_seg_count(len, G);
    const K9Signs sg = k9_signs(lane, g0, g1, g2);
    int4 tend = *(const int4 *)&meta.uend[0][4 * lane];        // true end metrics of the segment below, up to a constant
    unsigned repairs = 0;
    for (int g = 1; g < nseg; g++) {
        const int4 a = *(const int4 *)&meta.snap[g][4 * lane];
        const int a0 = wave_readlane(a.x, 0), b0 = wave_readlane(tend.x, 0);
        const bool same = a.x - a0 == tend.x - b0 && a.y - a0 == tend.y - b0 && a.z - a0 == tend.z - b0 && a.w - a0 == tend.w - b0;
        if (__all(same)) { tend = *(const int4 *)&meta.uend[g][4 * lane]; continue; }
        *(int4 *)&sm.metric[0][4 * lane] = tend;
        WAVE_LDS_SYNC();
        const int c0 = g * per, c1 = min(nch, c0 + per);
        const int cur = k9_forward_chunks(coded, len, sg, dec, sm, c0, c1, c0, nullptr);
        tend = *(const int4 *)&sm.metric[cur][4 * lane];
        WAVE_LDS_SYNC();
        repairs++;
    }
    if (stats && lane == 0) { atomicAdd(&stats[0], (unsigned)(nseg - 1)); if (repairs) atomicAdd(&stats[1], repairs); }
    return k9_end_state(tend);
}

__device__ inline void k9_traceback_segment(const unsigned long long *dec, int len, K9Meta &meta, K9WSmem &sm, uint32_t *out, int g, int G, int runin)
{
    const int nch = k9_chunks(len), per = k9_seg_chunks(len, G), nseg = k9_seg_count(len, G);
    if (g >= nseg) return;                                     // wave-uniform
    const int ntb = (k9_pairs(len) + 31) >> 5;
    const int lo = 2 * g * per, hi = min(ntb, 2 * min(nch, (g + 1) * per));
    const bool last = g == nseg - 1;
    unsigned arrive = last ? meta.end_state : 0u;
    const unsigned leave = k9_traceback_chunks(dec, len, sm, arrive, last ? hi : min(ntb, hi + runin), lo, hi, out, arrive);
    if ((threadIdx.x & 63) == 0) { meta.arrive[g] = arrive; meta.leave[g] = leave; }
}

// one wave per frame, after every traceback segment wave has finished
__device__ inline void k9_traceback_fix(const unsigned long long *dec, int len, K9Meta &meta, K9WSmem &sm, uint32_t *out, int G, unsigned *stats)
{
    const int nch = k9_chunks(len), per = k9_seg_chunks(len, G), nseg = k9_seg_count(len, G);
    const int ntb = (k9_pairs(len) + 31) >> 5;
    unsigned truth = (unsigned)wave_uniform((int)meta.leave[nseg - 1]), repairs = 0;
    for (int g = nseg - 2; g >= 0; g--) {
        if ((unsigned)wave_uniform((int)meta.arrive[g]) == truth) { truth = (unsigned)wave_uniform((int)meta.leave[g]); continue; }
        const int lo = 2 * g * per, hi = min(ntb, 2 * min(nch, (g + 1) * per));
        unsigned arrive = 0;
        truth = k9_traceback_chunks(dec, len, sm, truth, hi, lo, hi, out, arrive);
        repairs++;
    }
    if (stats && (threadIdx.x & 63) == 0) { atomicAdd(&stats[2], (unsigned)(nseg - 1)); if (repairs) atomicAdd(&stats[3], repairs); }
    __threadfence_block();
    __syncthreads();
}

// re-encode the decoded (still scrambled) bits and count sign disagreements at unpunctured positions
// (bit_errors, decode.c:234-261); returns the block-wide total in every work-item
__device__ inline int am_bit_errors(const int8_t *coded, const uint32_t *bits, int len, unsigned g0, unsigned g1, unsigned g2,
                                    unsigned pmask, int plen, int *red /* [4] */)
{
    int errors = 0;
    for (int i = threadIdx.x; i < len; i += blockDim.x) {
        unsigned r = 0;                                        // r bit 8-k = bits[i-k]
#pragma unroll
        for (int k = 0; k < 9; k++) {
            int q = i - k; if (q < 0) q += len;
            r |= ((bits[q >> 5] >> (q & 31)) & 1u) << (8 - k);
        }
        const int j = 3 * i;
        if (((pmask >> (j % plen)) & 1u) && ((coded[j] > 0) != (int)(__popc(r & g0) & 1))) errors++;
        if (((pmask >> ((j + 1) % plen)) & 1u) && ((coded[j + 1] > 0) != (int)(__popc(r & g1) & 1))) errors++;
        if (((pmask >> ((j + 2) % plen)) & 1u) && ((coded[j + 2] > 0) != (int)(__popc(r & g2) & 1))) errors++;
    }
    errors = wave_sum_i32(errors);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = errors;
    __syncthreads();
    int total = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) total += red[w];
    return total;
}

constexpr unsigned GEN_E1_0 = 0561, GEN_E1_1 = 0657, GEN_E1_2 = 0711;     // decode.c:47-53
constexpr unsigned GEN_E2_0 = 0561, GEN_E2_1 = 0753, GEN_E2_2 = 0711;     // decode.c:55-61
constexpr unsigned PUNCT_E1 = 0x7f6d, PUNCT_E2 = 0x0d;                    // bit k = pattern[k]: {1,0,1,1,0,1,1,0,1,1,1,1,1,1,1}, {1,0,1,1,0,0}

// =====================================================================================================
// the block step
// =====================================================================================================
struct AmBlockSmem {
    float2 X[NSYM * AM_FFT];        // 64 KB: coarse-acquisition scratch, then the 32 symbol spectra (natural order)
    float2 tw[AM_FFT / 2];
    float shape[AM_SYM];
    float2 mult[4][AM_PW];
    float marg[2][AM_PW];
    float2 carrier[NSYM];
    float magsum[2 * 53 + 1];
    float red_mag[16]; int red_idx[16]; float2 red_v[16];
    uint8_t pids_sym[2 * NSYM];
    int8_t pids_coded[3 * PIDS_LEN];
    uint32_t pids_out[3];
    // block-uniform scalars produced by work-item 0
    int active, fine, samperr, ma3, refmask;
    int deliver;                // replay: P1 PDU this block delivered (0..7), -1: none
    double theta, dtheta;
    float2 step270, step256;        // e^{i 270 dtheta}, e^{i 256 dtheta}
    float dphi[NSYM];           // carrier phase advance per symbol (line fit), one work-item each
    double targ;                // argument handed from work-item 0 to the work-items that evaluate its cosine / sine
    int bc_now, psmi_now, rdbi_now;   // the stream's block count / service mode / RDBI after the reference decode (block-uniform copies)
};
// the PIDS trellis runs after the spectra are consumed: its scratch aliases the head of X
static_assert(sizeof(K9Smem) + 4 * (PIDS_LEN + 64) * sizeof(unsigned long long) <= sizeof(float2) * NSYM * AM_FFT, "PIDS scratch must fit in X");

__device__ inline float2 cmulf(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ inline float2 cdivf(float2 a, float2 b)
{
    const float d = b.x * b.x + b.y * b.y;
    return make_float2((a.x * b.x + a.y * b.y) / d, (a.y * b.x - a.x * b.y) / d);
}
__device__ inline unsigned bitrev8(unsigned v) { return __brev(v) >> 24; }

// sync.c:37-88
__device__ inline unsigned slice4(float f) { return f < -1 ? 0u : f < 0 ? 2u : f < 1 ? 3u : 1u; }
__device__ inline unsigned slice8(float f) { return f < -3 ? 0u : f < -2 ? 4u : f < -1 ? 6u : f < 0 ? 2u : f < 1 ? 3u : f < 2 ? 7u : f < 3 ? 5u : 1u; }
__device__ inline unsigned sym_qpsk(float2 c) { return (c.x < 0 ? 0u : 1u) | (c.y < 0 ? 0u : 2u); }
__device__ inline unsigned sym_qam16(float2 c) { return slice4(c.x) | (slice4(c.y) << 2); }
__device__ inline unsigned sym_qam64(float2 c) { return slice8(c.x) | (slice8(c.y) << 3); }

__device__ inline float half_turn_diff(float a, float b)   // phase_diff, sync.c:284-290
{
    float d = a - b;
    while (d > (float)(M_PI / 2)) d = (float)((double)d - M_PI);
    while (d < (float)(-M_PI / 2)) d = (float)((double)d + M_PI);
    return d;
}

// Mix one block down with the NCO (phase theta + dtheta * sample), fold the cyclic prefix (rotated by 121 samples:
// carrier phases are referenced to the symbol centre, acquire.c:239-247) and leave the 32 inputs in NATURAL order for the
// 16 x 16 transform below.  Work-item j owns input slot j of every symbol.
template <int NT> __device__ inline void am_fold(AmBlockSmem &sm, const c16 *win, int samperr, double theta, float2 step270, float2 step256)
{
    // work-item (j, g): sample j of the eight symbols of group g; the phasor is evaluated in closed form at the head of every
    // group and advanced by recurrence inside it, whatever the block size (256 work-items take the four groups in turn, 512
    // two each), so that both launch shapes produce the same bits
    const int j = threadIdx.x & 255;
    const unsigned slot = (unsigned)(j + (AM_FFT - AM_CP) / 2) & 255u;
    constexpr int NG = (NSYM / 8) / (NT >> 8);
    // every sample this work-item will mix, requested before the first is used: one at a time behind the phasor recurrence the 16 loads of a
    // work-item were 16 dependent trips to L2 / HBM (the first fold of a block: 33 k of its 139 k shader cycles; profiles/r05_am_phases.txt)
    c16 a[NG][8], c[NG][8];
#pragma unroll
    for (int q = 0; q < NG; q++) {
        const int g = ((int)threadIdx.x >> 8) + q * (NT >> 8);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            a[q][k] = win[(8 * g + k) * AM_SYM + j + samperr];
            c[q][k] = win[(8 * g + k) * AM_SYM + (j < AM_CP ? j + AM_FFT : j) + samperr];      // (work-items >= AM_CP: the same line again, unused)
        }
    }
#pragma unroll
    for (int q = 0; q < NG; q++) {
        const int g = ((int)threadIdx.x >> 8) + q * (NT >> 8);
        float2 p;
        {
            double th = theta + sm.dtheta * (double)(j + g * 8 * AM_SYM);
            th -= 2 * M_PI * rint(th / (2 * M_PI));
            float sn, cs; sincosf((float)th, &sn, &cs);
            p = make_float2(cs, sn);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int i = 8 * g + k;
            float2 v = cmulf(p, make_float2((float)a[q][k].r / 32767.0f, (float)a[q][k].i / 32767.0f));    // cq15_to_cf, defines.h:106
            if (j < AM_CP) {
                const float2 w = cmulf(cmulf(p, step256), make_float2((float)c[q][k].r / 32767.0f, (float)c[q][k].i / 32767.0f));
                const float sa = sm.shape[j], sb = sm.shape[j + AM_FFT];
                v = make_float2(sa * v.x + sb * w.x, sa * v.y + sb * w.y);
            }
            sm.X[i * AM_FFT + slot] = v;
            p = cmulf(p, step270);
        }
    }
}

// forward 4-point DFT in place, natural order
__device__ __forceinline__ void am_dft4(float2 &a0, float2 &a1, float2 &a2, float2 &a3)
{
    const float2 t0 = make_float2(a0.x + a2.x, a0.y + a2.y), t1 = make_float2(a0.x - a2.x, a0.y - a2.y);
    const float2 t2 = make_float2(a1.x + a3.x, a1.y + a3.y), d = make_float2(a1.x - a3.x, a1.y - a3.y);
    const float2 t3 = make_float2(d.y, -d.x);                  // (a1 - a3) * (-j)
    a0 = make_float2(t0.x + t2.x, t0.y + t2.y); a1 = make_float2(t1.x + t3.x, t1.y + t3.y);
    a2 = make_float2(t0.x - t2.x, t0.y - t2.y); a3 = make_float2(t1.x - t3.x, t1.y - t3.y);
}
// forward 16-point DFT in place: input v[n], output X[4 k1 + k2] in v[4 k2 + k1]
__device__ __forceinline__ void am_dft16(float2 *v)
{
#pragma unroll
    for (int n1 = 0; n1 < 4; n1++) am_dft4(v[n1], v[n1 + 4], v[n1 + 8], v[n1 + 12]);     // v[n1 + 4 k2] = Y[n1][k2]
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, c2 = 0.70710678118654752440f;
    // W16^(n1 k2) = e^{-2 pi i n1 k2 / 16}
    v[5] = cmulf(v[5], make_float2(c1, -s1));     // 1
    v[6] = cmulf(v[6], make_float2(c2, -c2));     // 2
    v[7] = cmulf(v[7], make_float2(s1, -c1));     // 3
    v[9] = cmulf(v[9], make_float2(c2, -c2));     // 2
    v[10] = make_float2(v[10].y, -v[10].x);       // 4: -j
    v[11] = cmulf(v[11], make_float2(-c2, -c2));  // 6
    v[13] = cmulf(v[13], make_float2(s1, -c1));   // 3
    v[14] = cmulf(v[14], make_float2(-c2, -c2));  // 6
    v[15] = cmulf(v[15], make_float2(-c1, s1));   // 9
#pragma unroll
    for (int k2 = 0; k2 < 4; k2++) am_dft4(v[4 * k2], v[4 * k2 + 1], v[4 * k2 + 2], v[4 * k2 + 3]);
}

// 32 x 256-point forward FFT in LDS, natural order in and out: 256 = 16 x 16, sixteen points per work-item in registers, ONE exchange through the tile
// (n = 16 a + b, k = c + 16 d: work-item (symbol, b) transforms over a and applies W256^(b c); work-item (symbol, c) transforms over b).  The
// exchange is in place: Y[c][b] sits at c * 16 + (b ^ c), which keeps both the writes (b runs across the work-items) and the reads (c runs across
// them: a plain row-major tile would put all sixteen on one bank) conflict-free without a second buffer.  Round 4's form -- eight radix-2 passes, each
// a read-modify-write of the whole 64 KB tile behind a barrier -- was 21 k of the block's 139 k shader cycles.
template <int NT> __device__ inline void am_fft_all(AmBlockSmem &sm)
{
    static_assert((NSYM * 16) % NT == 0, "whole rounds");
    for (int id0 = 0; id0 < NSYM * 16; id0 += NT) {
        const int id = id0 + (int)threadIdx.x, n = id >> 4, l = id & 15;
        float2 *x = sm.X + n * AM_FFT;
        float2 v[16];
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 16; a++) v[a] = x[16 * a + l];                        // b = l
        am_dft16(v);
        __syncthreads();
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++)
#pragma unroll
            for (int k1 = 0; k1 < 4; k1++) {
                const int c = 4 * k1 + k2, m = l * c;                             // W256^(b c), b c <= 225
                float2 w = sm.tw[m & 127];
                if (m & 128) w = make_float2(-w.x, -w.y);
                x[c * 16 + (l ^ c)] = c ? cmulf(v[4 * k2 + k1], w) : v[4 * k2 + k1];
            }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 16; b++) v[b] = x[l * 16 + (b ^ l)];                  // c = l
        am_dft16(v);
        __syncthreads();
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++)
#pragma unroll
            for (int k1 = 0; k1 < 4; k1++) x[l + 16 * (4 * k1 + k2)] = v[4 * k2 + k1];
    }
    __syncthreads();
}

// spectrum bin `off` relative to the carrier (fftshift folded into the index), symbol n
__device__ inline float2 &am_bin(AmBlockSmem &sm, int off, int n) { return sm.X[n * AM_FFT + (off & 255)]; }

// e^{i 270 dtheta}, e^{i 256 dtheta} from sm.dtheta: four double-precision cosines / sines, one per wave (work-items 0, 64, 128, 192), instead of four in
// a row on work-item 0.  Same functions of the same arguments: the same bits.  Callers put a barrier before (sm.dtheta) and after.
__device__ __forceinline__ void am_nco_steps(AmBlockSmem &sm)
{
    const int tid = threadIdx.x;
    if ((tid & 63) == 0 && tid < 256) {
        const int w = tid >> 6;
        const double x = sm.dtheta * (double)(w < 2 ? AM_SYM : AM_FFT);
        float v;
        if (fabs(x) <= 0.25) { double c, sn; small_cos_sin(x, c, sn); v = (w & 1) ? (float)sn : (float)c; }   // integer CFO 0 (every block once tuned): the series of fastmath.h, a few ulps of a double
        else v = (w & 1) ? (float)sin(x) : (float)cos(x);
        float2 &dst = w < 2 ? sm.step270 : sm.step256;
        if (w & 1) dst.y = v; else dst.x = v;
    }
}

// 256 work-items per stream (the in-order K=9 PIDS trellis below owns one state per work-item).  Every wide phase strides by the
// block size, but 1024 work-items were measured SLOWER in the window pipeline (am-cs16 88.6 -> 108.4 ms): a 16-wave workgroup with
// 64 KB of LDS waits for a whole CU's worth of slots while the decode streams keep the chip full of long one-wave trellis passes.
// (NT, the block size, is a template constant: read as blockDim.x it is two dependent global loads -- implicit-argument pointer, dispatch packet -- in
// front of the first loop that strides by it; profiles/r04_mixfft_phases.txt)
template <int NT>
__global__ __launch_bounds__(NT) void k_am_block(DevTables tb, DevBuffers db, const int *ids, int pipeline, int parity, int slot)
{
    wave_set_priority_high();                                  // block-step chain = critical path; the decode waves run at priority 0
    const int s = stream_of(ids, blockIdx.x);
    StreamState &st = db.state[s];
    AmStream &am = db.am[s];
    HIP_DYNAMIC_SHARED(uint8_t, smem_raw)
    AmBlockSmem &sm = *(AmBlockSmem *)smem_raw;
    const int tid = threadIdx.x;
    static_assert(NT >= 256 && NT % 256 == 0, "four waves share the NCO's cosines / sines; the fold strides by groups of 256");
    const bool ready = st.wr - st.rd >= AM_WIN;                 // block-uniform
    if (tid == 0) {
        am.dec_bc = -1; st.active = ready ? 1 : 0;
        sm.deliver = -1;
    }
    __syncthreads();
    if (!ready) return;
    const c16 *win = db.q15 + (size_t)s * db.q15_cap + (st.rd - st.base);
    const int state_before = st.sync_state;
    // phase instrumentation (nrsc5hip_debug_tune NRSC5HIP_TUNE_SYNC_PHASES; tools/gpu_am_phases.py): shader cycles of stream 0's workgroup between the marks
    __shared__ long long am_tstamp;
    if (db.sync_phase_cycles && s == 0 && tid == 0) am_tstamp = (long long)clock64();
#define AM_MARK(i) do { if (db.sync_phase_cycles && s == 0 && tid == 0) { const long long now = (long long)clock64(); db.sync_phase_cycles[i] += now - am_tstamp; am_tstamp = now; } } while (0)

    for (int k = tid; k < AM_FFT / 2; k += NT) sm.tw[k] = tb.am_twiddle[k];
    for (int k = tid; k < AM_SYM; k += NT) sm.shape[k] = tb.am_shape[k];

    // ---- coarse acquisition while not FINE (acquire.c:120-158 with the AM filter / geometry) ------------------
    if (state_before != SYNC_FINE) {
        c16 *filt = (c16 *)sm.X;                               // [AM_WIN]
        float2 *sums = (float2 *)(filt + AM_WIN + 2);          // [AM_SYM], 8-byte aligned (AM_WIN even)
        for (int t = tid; t < AM_WIN; t += NT) {
            int sr = 0, si = 0;
#pragma unroll
            for (int i = 1; i < 16; i++) {
                const int ka = t - 31 + i, kb = t - 31 + (32 - i);
                const c16 xa = ka >= 0 ? win[ka] : st.fir_hist[31 + ka];
                const c16 xb = kb >= 0 ? win[kb] : st.fir_hist[31 + kb];
                const int q = tb.am_acq_q15[i];
                sr = (int16_t)(sr + (((xa.r + xb.r) * q) >> 15));
                si = (int16_t)(si + (((xa.i + xb.i) * q) >> 15));
            }
            const int kc = t - 15;
            const c16 xc = kc >= 0 ? win[kc] : st.fir_hist[31 + kc];
            const int q = tb.am_acq_q15[16];
            sr = (int16_t)(sr + ((xc.r * q) >> 15));
            si = (int16_t)(si + ((xc.i * q) >> 15));
            c16 y; y.r = (int16_t)sr; y.i = (int16_t)si;
            filt[t] = y;
        }
        __syncthreads();
        for (int i = tid; i < AM_SYM; i += NT) {
            float sr = 0.0f, si = 0.0f;
            for (int j = 0; j < NSYM; j++) {
                const c16 qa = filt[i + j * AM_SYM], qb = filt[i + j * AM_SYM + AM_FFT];
                const float ax = (float)qa.r / 32767.0f, ay = (float)qa.i / 32767.0f;
                const float bx = (float)qb.r / 32767.0f, by = -((float)qb.i / 32767.0f);      // conjf
                sr += ax * bx - ay * by; si += ax * by + ay * bx;
            }
            sums[i] = make_float2(sr, si);
        }
        __syncthreads();
        float best_mag = -1.0f; int best_i = 0x7fffffff; float2 best_v = make_float2(0.0f, 0.0f);
        for (int i = tid; i < AM_SYM; i += NT) {
            float vr = 0.0f, vi = 0.0f;
            int k = i;
            for (int j = 0; j < AM_CP; j++) {
                const float2 z = sums[k];
                vr += (z.x * sm.shape[j]) * sm.shape[j + AM_FFT];
                vi += (z.y * sm.shape[j]) * sm.shape[j + AM_FFT];
                if (++k == AM_SYM) k = 0;
            }
            const float mag = vr * vr + vi * vi;
            if (mag > best_mag) { best_mag = mag; best_i = i; best_v = make_float2(vr, vi); }
        }
        for (int m = 32; m >= 1; m >>= 1) {                    // first maximum in index order wins
            const float om = __shfl_xor(best_mag, m); const int oi = __shfl_xor(best_i, m);
            const float ox = __shfl_xor(best_v.x, m), oy = __shfl_xor(best_v.y, m);
            if (om > best_mag || (om == best_mag && oi < best_i)) { best_mag = om; best_i = oi; best_v = make_float2(ox, oy); }
        }
        if ((tid & 63) == 0) { sm.red_mag[tid >> 6] = best_mag; sm.red_idx[tid >> 6] = best_i; sm.red_v[tid >> 6] = best_v; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < NT / 64; w++)
                if (sm.red_mag[w] > best_mag || (sm.red_mag[w] == best_mag && sm.red_idx[w] < best_i)) { best_mag = sm.red_mag[w]; best_i = sm.red_idx[w]; best_v = sm.red_v[w]; }
            st.coarse_samperr = (best_i + AM_SYM - 15) % AM_SYM;       // FILTER_DELAY, acquire.c:149
            st.coarse_re = best_v.x; st.coarse_im = best_v.y;
        }
        if (tid < 31) {
            st.fir_hist[tid] = win[AM_WIN - 31 + tid];
            const long long p = stale_start(st.stale.fir_pushed[MODE_AM], AM_WIN, 31);     // filter_am's compaction inside this block (>= 0: AM_WIN > 2 * 2017)
            if (p != STALE_NONE) st.stale.fir[MODE_AM][tid] = win[p + tid];
        }
        __syncthreads();
        if (tid == 0) st.stale.fir_pushed[MODE_AM] += AM_WIN;
    }

    AM_MARK(0);                                                // tables + coarse acquisition (while not FINE)
    // ---- per-block bookkeeping (top of acquire_process, acquire.c:110-119,153-168) -------------------------------
    BlockRecord &rec = db.records[(size_t)s * db.rec_cap + (st.nblocks % db.rec_cap)];
    if (tid == 0) {
        atomicAdd(&db.counters[0], 1);
        BlockRecord r;
        r.flags = REC_PROCESSED; r.state_before = state_before; r.state_after = 0;
        r.samperr = 0; r.cfo = 0; r.keep = 0; r.bc = 0; r.psmi = 0; r.cfo_wait = 0; r.next_samperr = 0;
        r.prev_angle = 0; r.phase_re = 0; r.phase_im = 0; r.next_angle = 0; r.freq_offset = 0; r.mer_lb = 0; r.mer_ub = 0;
        r.ber = 0; r.p1_slot = -1; r.bc_decoded = -1; r.pids[0] = r.pids[1] = r.pids[2] = 0; r.sis = 0;
        // the state words this section reads, in one burst (each behind the branch that needs it they were a chain of L2 round trips)
        const int st_samperr = st.samperr, st_cfo = st.cfo, st_psmi = st.psmi, st_coarse_samperr = st.coarse_samperr;
        int sync_state = st.sync_state;
        const float st_prev_angle = st.prev_angle, st_coarse_re = st.coarse_re, st_coarse_im = st.coarse_im;
        const double st_theta = st.theta;
        int samperr; float angle;
        if (state_before == SYNC_FINE) {
            samperr = AM_SYM / 2 + st_samperr; st.samperr = 0;
            // sync_t.angle is only written by the FM path: zero, except in the first synchronised block after a reset that followed an FM session (nrsc5hip_stream_reset
            // keeps it as sync_reset does) -- acquire.c:115-118 consumes it whatever the mode
            angle = st_prev_angle + -st.angle;
            st.angle = 0.0f; st.prev_angle = angle;
        } else {
            samperr = st_coarse_samperr;
            float sn, cs; ref_sincosf(-st_prev_angle, sn, cs);   // cexpf(I * -prev_angle), acquire.c:153: glibc's sincosf restated (fastmath.h)
            const float pr = st_coarse_re * cs - st_coarse_im * sn;
            const float pi = st_coarse_re * sn + st_coarse_im * cs;
            const float angle_diff = ref_atan2f(pi, pr);
            const float angle_factor = (st_prev_angle != 0.0f) ? 0.25f : 1.0f;
            angle = st_prev_angle + (angle_diff * angle_factor);
            st.prev_angle = angle;
            if (sync_state != SYNC_COARSE) { r.flags |= REC_TO_COARSE; sync_state = SYNC_COARSE; st.sync_state = SYNC_COARSE; }
        }
        rec = r;
        angle = (float)((double)angle - 2 * M_PI * st_cfo);    // acquire.c:164
        const float dth = angle / AM_FFT;
        double th = st_theta + (double)(-(float)(AM_SYM / 2 - samperr) * angle / AM_FFT);  // acquire.c:166
        th -= 2 * M_PI * rint(th / (2 * M_PI));
        sm.theta = th; sm.targ = (double)dth;
        sm.samperr = samperr; sm.fine = sync_state == SYNC_FINE; sm.ma3 = st_psmi == AM_MA3;
    }
    __syncthreads();
    // phase_increment = cexpf(angle / fft * I) (acquire.c:168; kept for the slope correction below): glibc's sincosf of the float argument, restated bit for bit
    // (ref_sincosf, fastmath.h) -- round 6; until round 5 a double-precision cosine and sine on two waves, rounded once (another float for 1.3 % of arguments) and
    // two barriers and ~8 k cycles more.  The effective step of the closed-form oscillator is the angle of that rounded pair.
    if (tid == 0) {
        float sn, cs; ref_sincosf((float)sm.targ, sn, cs);       // (sm.targ holds the float dth exactly)
        sm.red_v[0] = make_float2(cs, sn);
        const double t = (double)sn / (double)cs;
        sm.dtheta = (cs > 0.0f && fabs(t) <= 0.26) ? small_atan(t) : atan2((double)sn, (double)cs);
    }
    __syncthreads();
    am_nco_steps(sm);
    __syncthreads();
    const int samperr = sm.samperr;
    const bool fine_at_top = sm.fine != 0;
    AM_MARK(1);                                                // bookkeeping + NCO set-up (one work-item, double-precision trigonometry)

    // ---- pass 1: phase of the analog carrier per symbol, line fit over the block (acquire.c:170-235) -------------
    am_fold<NT>(sm, win, samperr, sm.theta, sm.step270, sm.step256);
    AM_MARK(2);                                                // pass-1 fold
    if (fine_at_top) {
        // only the carrier bin is needed: sum of the folded inputs (bin 0 before fftshift)
        __syncthreads();
        if (tid < 256) {
            const int n = tid >> 3, part = tid & 7;            // 8 work-items per symbol, 32 inputs each
            float sr = 0.0f, si = 0.0f;
            for (int k = 0; k < 32; k++) { const float2 v = sm.X[n * AM_FFT + part * 32 + k]; sr += v.x; si += v.y; }
            for (int m = 4; m >= 1; m >>= 1) { sr += __shfl_xor(sr, m); si += __shfl_xor(si, m); }
            if (part == 0) sm.carrier[n] = make_float2(sr, si);
        }
    } else {
        am_fft_all<NT>(sm);
        if (tid < NSYM) sm.carrier[tid] = am_bin(sm, 0, tid);
        if (tid <= 2 * 53) {                                   // |bins| summed over the block, carrier +-53
            float acc = 0.0f;
            for (int n = 0; n < NSYM; n++) { const float2 v = am_bin(sm, tid - 53, n); acc += sqrtf(v.x * v.x + v.y * v.y); }
            sm.magsum[tid] = acc;
        }
    }
    __syncthreads();
    // The line fit (acquire.c:199-231).  The phase advance of every symbol -- a complex division and an arc tangent, 32 of them in a row on one
    // work-item as the reference's loop is written -- is a function of two neighbouring carriers only: one work-item each; the running sums stay
    // one sequential chain in the reference's order (same additions, same bits).
    if (tid < NSYM) {
        const float2 c = sm.carrier[tid];
        float d;
        if (tid == 0) d = ref_atan2f(c.y, c.x);
        else { const float2 q = cdivf(c, sm.carrier[tid - 1]); d = ref_atan2f(q.y, q.x); }
        sm.dphi[tid] = d;
    }
    __syncthreads();
    if (tid == 0) {
        float y = 0, sum_y = 0, sum_xy = 0, sum_x2 = 0;
        for (int i = 0; i < NSYM; i++) {
            const float x = AM_SYM * (i - (float)(NSYM - 1) / 2);
            if (i == 0) y = sm.dphi[0];
            else y += sm.dphi[i];
            sum_y += y; sum_xy += x * y; sum_x2 += x * x;
        }
        if (!fine_at_top) {
            float max_mag = -1.0f; int max_index = -1;
            for (int j = 0; j <= 2 * 53; j++) if (sm.magsum[j] > max_mag) { max_mag = sm.magsum[j]; max_index = j; }
            st.cfo += max_index - 53;                          // acquire_cfo_adjust: effective from the next block
        }
        const float slope = sum_xy / sum_x2;
        const float a = -sum_y / NSYM + slope * NSYM * AM_SYM / 2;
        const float corr = (float)((double)a - 0.06);          // acquire.c:233-234
        double th = sm.theta + (double)corr;
        th -= 2 * M_PI * rint(th / (2 * M_PI));
        sm.theta = th; sm.targ = (double)slope;
    }
    __syncthreads();
    // phase_increment *= cexpf(-slope I) (acquire.c:232): glibc's sincosf of -slope (ref_sincosf), then the float complex product of the two rounded unit vectors
    if (tid == 0) {
        float rs, rc; ref_sincosf(-(float)sm.targ, rs, rc);     // (sm.targ holds the float slope exactly)
        const float2 inc = sm.red_v[0];
        const float2 inc2 = make_float2(inc.x * rc - inc.y * rs, inc.x * rs + inc.y * rc);
        const double t = (double)inc2.y / (double)inc2.x;
        sm.dtheta = (inc2.x > 0.0f && fabs(t) <= 0.26) ? small_atan(t) : atan2((double)inc2.y, (double)inc2.x);
    }
    __syncthreads();
    am_nco_steps(sm);
    __syncthreads();
    AM_MARK(3);                                                // carrier (or FFT while not FINE) + line fit + NCO correction

    // ---- pass 2: the block's spectra (acquire.c:237-257) -> sync_process_am on the LDS tile --------------------
    am_fold<NT>(sm, win, samperr, sm.theta, sm.step270, sm.step256);
    AM_MARK(4);                                                // pass-2 fold
    am_fft_all<NT>(sm);
    AM_MARK(5);                                                // 32 x FFT-256

    // lower sideband: z = -conj(z); complementary sidebands of the hybrid waveform add coherently (sync.c:616-633)
    for (int id = tid; id < NSYM * AM_IDX_MAX; id += NT) {
        const int n = id / AM_IDX_MAX, i = 1 + id % AM_IDX_MAX;
        float2 &lo = am_bin(sm, -i, n);
        lo = make_float2(-lo.x, lo.y);
        if (!sm.ma3 && i <= 53) { float2 &up = am_bin(sm, i, n); up = make_float2(up.x + lo.x, up.y + lo.y); }
    }
    __syncthreads();
    if (tid < 64) {
        const unsigned long long m = __ballot(tid < NSYM && am_bin(sm, 1, tid & 31).y > 0);
        if (tid == 0) sm.refmask = (int)(uint32_t)m;           // bit n = reference-carrier BPSK bit of symbol n
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned d = (unsigned)sm.refmask;
        // the state words of this section in one burst, worked on as values, stored where they change (read through the state at every test they were
        // a chain of dependent L2 round trips: every store in between forces the next read back to memory)
        int sync_state = st.sync_state, cfo_wait = st.cfo_wait, psmi = st.psmi, bc_now = st.bc, rdbi = am.rdbi;
        // fixed part of the reference sequence (find_ref_am / find_block_am needles, sync.c:211-213,242-244)
        const unsigned care23 = 0x60427fu, val23 = 0x600226u;  // positions 0-6, 9, 14, 21, 22; ones at 1, 2, 5, 9, 21, 22
        if (sync_state == SYNC_COARSE && cfo_wait == 0) {
            int off = -1;
            for (int r = 0; r < NSYM && off < 0; r++) {
                const unsigned rot = r ? ((d >> r) | (d << (32 - r))) : d;      // rot bit i = d[(r + i) % 32]
                if ((rot & care23) == val23) off = r;
            }
            if (off > 0) { st.keep_extra = ((NSYM - off) % NSYM) * AM_SYM; st.cfo_wait = 8; }
        } else {
            st.cfo_wait = cfo_wait - 1;
        }
        if (sync_state == SYNC_COARSE) {
            int bc = -1;
            auto bit = [&](int k) { return (d >> k) & 1u; };
            if ((d & care23) == val23
                && !(bit(7) ^ bit(8))
                && !(bit(10) ^ bit(11) ^ bit(12) ^ bit(13))
                && !(bit(15) ^ bit(16) ^ bit(17) ^ bit(18) ^ bit(19) ^ bit(20))
                && !(__popc(d & 0xff800000u) & 1)) {
                bc = (int)((bit(17) << 2) | (bit(18) << 1) | bit(19));
                if (bc == 0) {
                    psmi = (int)((bit(26) << 4) | (bit(27) << 3) | (bit(28) << 2) | (bit(29) << 1) | bit(30));
                    st.psmi = psmi;
                    rdbi = (int)bit(15);
                    am.pli = bit(7); am.hppi = bit(11); am.aabi = bit(12); am.rdbi = rdbi;
                }
            }
            unsigned history = bc == -1 ? 0u : ((am.offset_history << 4) | (unsigned)bc);
            if ((history & 0xffffu) == 0x5670u) {
                bc_now = 0; st.bc = 0;
                sync_state = SYNC_FINE; st.sync_state = SYNC_FINE; st.fine_epoch++;    // input_set_sync_state: EVENT_SYNC payload (input.c:179-185)
                rec.flags |= REC_TO_FINE;
                rec.freq_offset = (float)(((double)st.prev_angle - 2 * M_PI * st.cfo) * 46511.71875 / (2 * M_PI * AM_FFT));
                rec.sis = (uint32_t)((am.pli & 1) | ((am.hppi & 1) << 1) | ((am.aabi & 1) << 2) | ((rdbi & 1) << 3) | 16);
                am.am_errors = 0; am.am_diversity_wait = 4;    // decode_reset (decode.c:563-572)
                history = 0;
            }
            am.offset_history = history;
        }
        sm.fine = sync_state == SYNC_FINE;
        sm.ma3 = psmi == AM_MA3;
        sm.bc_now = bc_now; sm.psmi_now = psmi; sm.rdbi_now = rdbi;
    }
    __syncthreads();
    AM_MARK(6);                                                // sideband combine + reference-sequence decode (one work-item)

    if (sm.fine) {
        const bool ma3 = sm.ma3 != 0;
        const int bc = sm.bc_now;
        // PIDS carriers: normalise by the two training symbols (8 and 24), slice QAM16 (sync.c:661-678)
        if (tid < 2 * NSYM) {
            const int n = tid >> 1, which = tid & 1;
            const int off = which == 0 ? (ma3 ? -27 : 27) : (ma3 ? 27 : 53);
            const float2 t8 = am_bin(sm, off, 8), t24 = am_bin(sm, off, 24);
            const float2 mult = cdivf(make_float2(3.0f, -1.0f), make_float2(t8.x + t24.x, t8.y + t24.y));
            sm.pids_sym[tid] = (uint8_t)sym_qam16(cmulf(am_bin(sm, off, n), mult));
        }
        // per-carrier equaliser taps from the two training cells of each carrier (sync.c:689-714)
        if (tid < 4 * AM_PW) {
            const int part = tid / AM_PW, col = tid % AM_PW;
            const int t1 = (5 + 11 * col) % 32, t2 = (21 + 11 * col) % 32;
            const int pri = ma3 ? 2 : 57, ter = ma3 ? 28 : 2;
            int off; float2 ideal;
            if (part == 0) { off = -(pri + col); ideal = make_float2(5.0f, -5.0f); }
            else if (part == 1) { off = pri + col; ideal = make_float2(5.0f, -5.0f); }
            else if (part == 2) { off = 28 + col; ideal = ma3 ? make_float2(5.0f, -5.0f) : make_float2(3.0f, -1.0f); }
            else { off = ma3 ? -(ter + col) : ter + col; ideal = ma3 ? make_float2(5.0f, -5.0f) : make_float2(-1.0f, 1.0f); }
            const float2 a = am_bin(sm, off, t1), b2 = am_bin(sm, off, t2);
            const float2 m = cdivf(ideal, make_float2(a.x + b2.x, a.y + b2.y));
            sm.mult[part][col] = m;
            if (part < 2) sm.marg[part][col] = ref_atan2f(m.y, m.x);
        }
        __syncthreads();
        AM_MARK(7);                                            // PIDS carriers + equaliser taps
        if (tid == 0) {
            float se = 0;
            for (int col = 1; col < AM_PW; col++) {
                se += half_turn_diff(sm.marg[0][col], sm.marg[0][col - 1]);
                se += half_turn_diff(sm.marg[1][col], sm.marg[1][col - 1]);
            }
            se = (float)((double)(se / (2 * (AM_PW - 1)) * AM_FFT) / (2 * M_PI));
            st.samperr = (int)roundf(se);
        }
        // equalise and slice the four partitions: hard symbols of this block go to the frame matrices (decode.c:439-449)
        uint8_t *symbase = db.am_sym + (size_t)s * 4 * AM_SYMS;
        for (int id = tid; id < 4 * NSYM * AM_PW; id += NT) {
            const int part = id / (NSYM * AM_PW), r = id % (NSYM * AM_PW), n = r / AM_PW, col = r % AM_PW;
            const int pri = ma3 ? 2 : 57, ter = ma3 ? 28 : 2;
            int off;
            if (part == 0) off = -(pri + col); else if (part == 1) off = pri + col; else if (part == 2) off = 28 + col;
            else off = ma3 ? -(ter + col) : ter + col;
            const float2 v = cmulf(am_bin(sm, off, n), sm.mult[part][col]);
            unsigned code;
            if (part < 2 || ma3) code = sym_qam64(v);
            else if (part == 2) code = sym_qam16(v);
            else code = sym_qpsk(v);
            symbase[(size_t)part * AM_SYMS + bc * (NSYM * AM_PW) + r] = (uint8_t)code;
        }
        __syncthreads();
        AM_MARK(8);                                            // timing estimate + equalise / slice
        // PIDS: bit gather (decode.c:476-501), K=9 E2 decode, descramble
        if (tid < 120) {
            const int n = tid, p = n % 4;
            int k = (n + (n / 60) + 11) % 30, row = (11 * (k + (k / 15)) + 3) % 32;
            const int il = (sm.pids_sym[row * 2] >> p) & 1;
            k = (n + (n / 60)) % 30; row = (11 * (k + (k / 15)) + 3) % 32;
            const int iu = (sm.pids_sym[row * 2 + 1] >> p) & 1;
            const int i = n / 12, j = n % 12;
            const int il_pos[12] = { 0, 1, 12, 13, 6, 5, 18, 17, 11, 7, 23, 19 };      // decode.c:63-64
            const int iu_pos[12] = { 2, 4, 14, 16, 3, 8, 15, 20, 9, 10, 21, 22 };
            const bool pids1_disabled = (sm.psmi_now == 1) && sm.rdbi_now;
            sm.pids_coded[i * 24 + il_pos[j]] = pids1_disabled ? 0 : (il ? 1 : -1);
            sm.pids_coded[i * 24 + iu_pos[j]] = iu ? 1 : -1;
        }
        __syncthreads();
        if (pipeline) {
            // window pipeline: the 144-step PIDS trellis leaves the step chain -- stage its input for k_am_decode
            int8_t *stage = db.am_pids_stage + (((size_t)s * NWIN + parity) * 8 + slot) * (3 * PIDS_LEN);
            if (tid < 3 * PIDS_LEN) stage[tid] = sm.pids_coded[tid];
            if (tid == 0) db.am_pids_rec[((size_t)s * NWIN + parity) * 8 + slot] = st.nblocks % db.rec_cap;
        } else {
            K9Smem &k9 = *(K9Smem *)sm.X;
            unsigned long long *pids_dec = (unsigned long long *)((uint8_t *)sm.X + sizeof(K9Smem));
            viterbi_k9_block(sm.pids_coded, PIDS_LEN, GEN_E2_0, GEN_E2_1, GEN_E2_2, pids_dec, sm.pids_out, k9);
        }
        if (tid == 0) {
            if (!pipeline) {
                // (CRC over a local copy: on the record itself every one of its 80 bits was a round trip to global memory)
                const uint32_t p[3] = { sm.pids_out[0] ^ tb.scr_pids[0], sm.pids_out[1] ^ tb.scr_pids[1], (sm.pids_out[2] ^ tb.scr_pids[2]) & 0xffffu };
                rec.pids[0] = p[0]; rec.pids[1] = p[1]; rec.pids[2] = p[2];
                rec.flags |= pids_crc_ok(p) ? (uint32_t)REC_PIDS_CRC : 0u;
            }
            rec.flags |= REC_PIDS;
            rec.bc_decoded = bc;
            // hand this block to the P1 / P3 decoders (decode_process_p1_p3_am runs next, in k_am_viterbi)
            am.dec_bc = bc; am.dec_record = st.nblocks % db.rec_cap; am.dec_rdbi = am.rdbi; am.dec_psmi = st.psmi;
            if (bc == 0) {
                am.am_errors = 0;
                if (am.am_diversity_wait == 0) { am.frame_slot = am.next_slot; am.vit_parity = am.next_job; }
            }
            if (pipeline && am.am_diversity_wait == 0) {
                // the frame's decodes run (or ran) on a decode stream from the trellis inputs of the previous L1 frame:
                // this block only announces what decode_process_p1_p3_am delivers here (decode.c:507-554)
                rec.flags |= REC_P1 | ((bc == 7 && !am.rdbi) ? (uint32_t)REC_P3 : 0u);
                rec.p1_slot = am.frame_slot;
                if (db.am_ckpt) {
                    // frame_process judges this PDU's first header inside this very block (frame.c:535-540).  If the deferred
                    // decode has already filed a failure, apply it now; else run on and let k_rollback_am rewind to the
                    // checkpoint k_am_interleave takes at the end of this step (k_replay.hip)
                    AmJob &dj = db.am_job[(size_t)s * NWIN + am.vit_parity];
                    dj.deliver_abs[bc] = st.nblocks;
                    sm.deliver = bc;
                    if (atomicAdd(&dj.verdict[bc], 0) == 2) { dj.verdict[bc] = 3; st.sync_state = SYNC_NONE; rec.flags |= REC_LOST_SYNC; }
                }
            }
            st.bc = (bc + 1) % 8;
        }
    }

    AM_MARK(9);                                                // PIDS gather / stage + frame hand-off (one work-item)
    // ---- tail of acquire_process (acquire.c:259-262) + record -------------------------------------------------------
    // Two work-items of different waves share it, as in k_sync: the NCO phase with its double-precision cosine / sine (a diagnostic of the record) on
    // one, the FIFO position and the record on the other with its state loads issued together (a load behind every store of the other kind cost an
    // L2 round trip apiece).
    if (tid == 64) {
        double th = sm.theta + sm.dtheta * (double)(NSYM * AM_SYM);
        th -= 2 * M_PI * rint(th / (2 * M_PI));
        st.theta = th;
        rec.phase_re = (float)cos(th); rec.phase_im = (float)sin(th);
    }
    if (tid == 0) {
        const int keep_extra = st.keep_extra, state = st.sync_state, cfo = st.cfo, bc_now = st.bc, psmi = st.psmi, cfo_wait = st.cfo_wait, next_samperr = st.samperr, nblocks = st.nblocks;
        const long long rd = st.rd;
        const float prev_angle = st.prev_angle, next_angle = st.angle;   // (sync_t.angle: zero in this mode unless a reset carried it over from an FM session)
        const int keep = AM_SYM + (AM_SYM / 2 - samperr) + keep_extra;
        st.keep_extra = 0;
        st.rd = rd + (AM_WIN - keep);
        rec.state_after = state; rec.samperr = samperr; rec.cfo = cfo; rec.keep = keep; rec.bc = bc_now;
        rec.psmi = psmi; rec.cfo_wait = cfo_wait; rec.next_samperr = next_samperr;
        rec.prev_angle = prev_angle; rec.next_angle = next_angle;
        st.nblocks = nblocks + 1;
    }
    AM_MARK(10);                                               // tail
    if (db.am_ckpt) {                                          // block-uniform: window pipeline with the on-device L2 feedback
        // Replay checkpoint of a block that delivered a P1 PDU (k_replay.hip): the state as of now.  For block 7 the
        // de-interleaver's bookkeeping (k_am_interleave, next) still belongs to the block: k_rollback_am adds it on restore.
        __threadfence_block();
        __syncthreads();
        const int j = sm.deliver;
        if (j >= 0) {
            AmCkpt &ck = db.am_ckpt[((size_t)s * NWIN + am.vit_parity) * 8 + j];
            const uint32_t *a = (const uint32_t *)&st, *b = (const uint32_t *)&am;
            uint32_t *da = (uint32_t *)&ck.st, *dbp = (uint32_t *)&ck.am;
            for (int k = tid; k < (int)(sizeof(StreamState) / 4); k += NT) da[k] = a[k];
            for (int k = tid; k < (int)(sizeof(AmStream) / 4); k += NT) dbp[k] = b[k];
        }
    }
}

// ---- this block's P1 frame, and after block 7 the P3 frame (decode_process_p1_p3_am, decode.c:507-554) ----------
__global__ __launch_bounds__(64) void k_am_viterbi(DevTables tb, DevBuffers db, const int *ids, int l2_feedback)
{
    const int s = stream_of(ids, blockIdx.y);
    const StreamState &st = db.state[s];
    AmStream &am = db.am[s];
    if (!st.active || am.dec_bc < 0 || am.am_diversity_wait != 0) return;      // block-uniform
    const int role = blockIdx.x, bc = am.dec_bc;                                 // 0: P1, 1: P3 (in-order mode)
    if (role == 1 && (bc != 7 || am.dec_rdbi)) return;
    __shared__ K9WSmem k9;
    __shared__ int red[4];
    const bool ma3 = am.dec_psmi == AM_MA3;
    uint32_t *slot = db.p1_ring + ((size_t)s * db.p1_slots + am.frame_slot) * P1_WORDS;
    unsigned long long *dec = db.am_dec + (size_t)s * (size_t)(8 * AM_DEC_P1 + AM_DEC_P3);
    BlockRecord &rec = db.records[(size_t)s * db.rec_cap + am.dec_record];
    if (role == 0) {
        const int8_t *in = db.am_vit + (size_t)s * db.am_nvit * 2 * AM_VIT + (size_t)bc * AM_P1_LEN * 3;
        uint32_t *out = slot + bc * AM_P1_WORDS;
        viterbi_k9_wave(in, AM_P1_LEN, GEN_E1_0, GEN_E1_1, GEN_E1_2, dec, out, k9);
        const int err = am_bit_errors(in, out, AM_P1_LEN, GEN_E1_0, GEN_E1_1, GEN_E1_2, PUNCT_E1, 15, red);
        for (int w = threadIdx.x; w < AM_P1_WORDS; w += 64)      // descramble; the last word holds 6 frame bits
            out[w] = (out[w] ^ tb.scr_p1[w]) & (w == AM_P1_WORDS - 1 ? (1u << (AM_P1_LEN & 31)) - 1u : 0xffffffffu);
        __threadfence_block();
        __syncthreads();
        static_assert(sizeof(L2Smem) <= sizeof(K9WSmem), "L2 scratch aliases the trellis scratch");
        L2Smem &l2 = *(L2Smem *)&k9;                           // the trellis scratch is dead by now
        const bool hdr_ok = l2_feedback ? l2_first_header_ok_am_block(out, l2) : true;     // frame.c:535-540 for the 466-byte AM PDU
        if (threadIdx.x == 0) {
            atomicAdd(&am.am_errors, (unsigned)err);
            atomicOr(&rec.flags, (uint32_t)REC_P1);
            rec.p1_slot = am.frame_slot;
            StreamState &stw = db.state[s];
            if (!hdr_ok && stw.sync_state == SYNC_FINE) { stw.sync_state = SYNC_NONE; rec.state_after = SYNC_NONE; atomicOr(&rec.flags, (uint32_t)REC_LOST_SYNC); }
        }
    } else {
        const int8_t *in = db.am_vit + (size_t)s * db.am_nvit * 2 * AM_VIT + AM_VIT;
        uint32_t *out = slot + AM_P3_WORD0;
        const int len = ma3 ? AM_P3_LEN_MA3 : AM_P3_LEN_MA1;
        int err;
        if (!ma3) {
            viterbi_k9_wave(in, len, GEN_E2_0, GEN_E2_1, GEN_E2_2, dec + (size_t)8 * AM_DEC_P1, out, k9);
            err = am_bit_errors(in, out, len, GEN_E2_0, GEN_E2_1, GEN_E2_2, PUNCT_E2, 6, red);
        } else {
            viterbi_k9_wave(in, len, GEN_E1_0, GEN_E1_1, GEN_E1_2, dec + (size_t)8 * AM_DEC_P1, out, k9);
            err = am_bit_errors(in, out, len, GEN_E1_0, GEN_E1_1, GEN_E1_2, PUNCT_E1, 15, red);
        }
        const int words = (len + 31) / 32;
        const uint32_t tailmask = (len & 31) ? (1u << (len & 31)) - 1u : 0xffffffffu;
        for (int w = threadIdx.x; w < words; w += 64) out[w] = (out[w] ^ tb.scr_p1[w]) & (w == words - 1 ? tailmask : 0xffffffffu);
        if (threadIdx.x == 0) {
            atomicAdd(&am.am_errors, (unsigned)err);
            atomicOr(&rec.flags, (uint32_t)REC_P3);
        }
    }
}

// ---- after block 7: BER of the frame just decoded, then interleaver_ma1 for the frame just received -------------
// Output-centric and table-driven: every depunctured trellis input looks up which bit of which hard-symbol matrix it is
// (DevTables::am_deint_*, built from decode.c:66-231 by engine.hip), the main (m*) bits pass through the 3-frame
// diversity delay lines -- a 3-slot ring with one cell per trellis input, whose oldest slot is read and refilled in place by the
// same work-item (consecutive work-items, consecutive bytes).
// Four inputs per work-item and trip: the table entries AND the delay cells (their address does not depend on the entry -- a cell
// that is not a delayed input's is simply not used) are requested together, then the LDS look-ups, then the stores; one input per
// trip left the kernel waiting for three dependent round trips per input.
__device__ inline void am_deint_span(const uint32_t *tab, int n, const uint8_t *sym, uint8_t *q, int8_t *v, int i0, int step)
{
    for (int i = i0; i < n; i += 4 * step) {
        uint32_t e[4]; uint8_t old[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int k = i + u * step;
            e[u] = k < n ? tab[k] : AMT_PUNCT;
            old[u] = k < n ? q[k] : (uint8_t)0;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int k = i + u * step;
            if (k >= n) break;
            int8_t out = 0;
            if (!(e[u] & AMT_PUNCT)) {
                const uint8_t *m = sym + (size_t)((e[u] >> 16) & 3u) * AM_SYMS;
                int bit = (m[e[u] & 0x1fffu] >> ((e[u] >> 13) & 7u)) & 1;
                if (e[u] & AMT_DELAYED) { q[k] = (uint8_t)bit; bit = old[u]; }
                out = bit ? 1 : -1;
            }
            v[k] = out;
        }
    }
}

// The 162 000 (MA1) / 180 000 (MA3) trellis inputs of an L1 frame are independent table look-ups (every cell of the diversity delay
// lines is visited by exactly one of them), so the frame is cut into AM_IL_PARTS slices, one workgroup each: with diverse streams an
// eighth of the batch finishes an L1 frame in any one step, and one workgroup per stream left 7 of 8 CUs idle for the 90 us such a
// step then took.  The bookkeeping that ends the frame -- delay-line head, ring slot and decode job of the next frame -- is done by
// the slice that finishes last (AmStream::il_done counts them).
constexpr int AM_IL_PARTS = 32;
constexpr int AM_IL_THREADS = 256;          // 4-wave workgroups find room between the decode waves; 16-wave ones wait for it

__device__ inline void am_deinterleave_slice(const DevTables &tb, const DevBuffers &db, int s, int parity, int part)
{
    AmStream &am = db.am[s];
    const bool ma3 = am.dec_psmi == AM_MA3;
    const int tid = threadIdx.x;
    const int vslot = parity < 0 ? 0 : parity;                 // window pipeline: one set of trellis inputs per window in flight
    if (part == 0 && tid == 0 && parity < 0 && am.am_diversity_wait == 0) {
        unsigned total = 8 * (AM_P1_LEN * 12 / 5);
        if (!am.dec_rdbi) total += ma3 ? AM_P3_LEN_MA3 * 12 / 5 : AM_P3_LEN_MA1 * 3 / 2;
        db.records[(size_t)s * db.rec_cap + am.dec_record].ber = (float)am.am_errors / (float)total;
    }
    // the frame's four hard-symbol matrices (25.6 KB) are gathered from byte by byte in interleaver order: stage them in LDS
    // first (the scattered byte loads were what the kernel waited for)
    __shared__ __attribute__((aligned(16))) uint8_t sym[4 * AM_SYMS];   // [pl, pu, s, t][8 blocks][32][25]
    static_assert((4 * AM_SYMS) % 16 == 0, "symbol matrices are copied in 16-byte pieces");
    {
        const uint4 *src = (const uint4 *)(db.am_sym + (size_t)s * 4 * AM_SYMS);
        for (int k = tid; k < 4 * AM_SYMS / 16; k += AM_IL_THREADS) ((uint4 *)sym)[k] = src[k];
    }
    __syncthreads();
    uint8_t *q1 = db.am_q + ((size_t)s * 3 + am.q_head) * 2 * AM_VIT, *q3 = q1 + AM_VIT;   // the oldest of the three frames in the delay lines
    int8_t *v1 = db.am_vit + ((size_t)s * db.am_nvit + vslot) * 2 * AM_VIT, *v3 = v1 + AM_VIT;
    const int i0 = part * AM_IL_THREADS + tid, step = AM_IL_THREADS * AM_IL_PARTS;
    am_deint_span(tb.am_deint_p1, AM_VIT, sym, q1, v1, i0, step);
    if (!ma3) am_deint_span(tb.am_deint_p3_ma1, 3 * AM_P3_LEN_MA1, sym, q3, v3, i0, step);
    else am_deint_span(tb.am_deint_p3_ma3, AM_VIT, sym, q3, v3, i0, step);
}

// after EVERY slice of the frame: advance the delay lines, reserve the ring slot and the decode job of the next L1 frame
__device__ inline void am_deinterleave_commit(const DevBuffers &db, int s, int parity, int window)
{
    AmStream &am = db.am[s];
    am.q_head = (am.q_head + 1) % 3;
    if (am.am_diversity_wait > 0) am.am_diversity_wait--;
    if (am.am_diversity_wait == 0) {
        // the next L1 frame delivers what these trellis inputs decode to: reserve its ring slot now
        StreamState &stw = db.state[s];
        am.next_slot = stw.p1_count % db.p1_slots; stw.p1_count++;
        if (parity >= 0) {
            AmJob &job = db.am_job[(size_t)s * NWIN + parity];
            job.slot = am.next_slot; job.psmi = am.dec_psmi; job.rdbi = am.dec_rdbi; job.errors = 0; job.done = 0; job.pad = 0; job.epoch = stw.fine_epoch;
            for (int j = 0; j < 8; j++) { job.verdict[j] = 0; job.deliver_abs[j] = -1; }
            job.window = window;
            am.next_job = parity;
            job.valid = 1;
        }
    }
}

__global__ __launch_bounds__(AM_IL_THREADS) void k_am_interleave(DevTables tb, DevBuffers db, const int *ids, int parity, int window)
{
    wave_set_priority_high();
    const int s = stream_of(ids, blockIdx.y);
    const StreamState &st = db.state[s];
    AmStream &am = db.am[s];
    if (!st.active || am.dec_bc != 7) return;                  // block-uniform
    am_deinterleave_slice(tb, db, s, parity, (int)blockIdx.x);
    // the slice that finishes last commits the frame: every other slice has read q_head / dec_* by then (their loads completed
    // before the barrier below; no fence -- an agent-scope fence per work-item writes back and invalidates L2 two million times a
    // launch, measured: the pass 98 -> 143 ms)
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(&am.il_done, 1u) == (unsigned)(AM_IL_PARTS - 1)) {
        am.il_done = 0;
        am_deinterleave_commit(db, s, parity, window);
    }
}

// ---- window pipeline: all nine frames of an L1 frame decode concurrently on a decode stream ---------------------------
// Four launches (the P3 frame is 6.4 / 8 times a P1 frame: as ONE wave it kept the launch -- and a decode stream -- alive for
// 4-5 ms after the P1 waves had gone; in K9_GMAX segment waves every wave of the launch is about one P1 frame long):
//   k_am_decode_fwd     forward pass: 8 P1 waves, G P3 segment waves, 8 PIDS frames (whole: 144 steps)
//   k_am_decode_fix     P3: segment boundaries checked / re-run, end state
//   k_am_decode_tb      traceback: P1 frames whole + BER, descramble, first-header verdict; P3 in G segment waves
//   k_am_decode_finish  P3: traceback boundaries checked / re-walked, BER, descramble; frame accounting
struct AmDecodeFrame { const int8_t *in; uint32_t *out; unsigned long long *dec; int len; unsigned g0, g1, g2; };

__device__ inline AmDecodeFrame am_decode_frame(const DevBuffers &db, const AmJob &job, int s, int parity, int lane_id, int role)
{
    AmDecodeFrame f;
    const int8_t *vit = db.am_vit + ((size_t)s * db.am_nvit + parity) * 2 * AM_VIT;
    uint32_t *slot = db.p1_ring + ((size_t)s * db.p1_slots + job.slot) * P1_WORDS;
    unsigned long long *dec = db.am_dec + ((size_t)lane_id * db.nstreams_alloc + s) * (size_t)(8 * AM_DEC_P1 + AM_DEC_P3);
    if (role < 8) {
        f.in = vit + (size_t)role * AM_P1_LEN * 3; f.out = slot + role * AM_P1_WORDS; f.dec = dec + (size_t)role * AM_DEC_P1;
        f.len = AM_P1_LEN; f.g0 = GEN_E1_0; f.g1 = GEN_E1_1; f.g2 = GEN_E1_2;
    } else {
        const bool ma3 = job.psmi == AM_MA3;
        f.in = vit + AM_VIT; f.out = slot + AM_P3_WORD0; f.dec = dec + (size_t)8 * AM_DEC_P1;
        f.len = ma3 ? AM_P3_LEN_MA3 : AM_P3_LEN_MA1;
        f.g0 = ma3 ? GEN_E1_0 : GEN_E2_0; f.g1 = ma3 ? GEN_E1_1 : GEN_E2_1; f.g2 = ma3 ? GEN_E1_2 : GEN_E2_2;
    }
    return f;
}

// the frame is decoded: count it, and the last of the L1 frame's decodes closes the job (nrsc5_report_ber's value, decode.c:545)
__device__ inline void am_decode_account(const DevBuffers &db, AmJob &job, int s, int err)
{
    if (threadIdx.x != 0) return;
    atomicAdd(&job.errors, (unsigned)err);
    __threadfence();
    const int expected = job.rdbi ? 8 : 9;
    if (atomicAdd(&job.done, 1) == expected - 1) {
        const bool ma3 = job.psmi == AM_MA3;
        unsigned total = 8 * (AM_P1_LEN * 12 / 5);
        if (!job.rdbi) total += ma3 ? AM_P3_LEN_MA3 * 12 / 5 : AM_P3_LEN_MA1 * 3 / 2;
        db.am_ber[(size_t)s * db.p1_slots + job.slot] = (float)atomicAdd(&job.errors, 0u) / (float)total;
        job.pad = db.l2_am_ring ? (job.rdbi ? 0xff : 0x1ff) : 0;      // frames k_l2_index_am_window owes their index
        job.valid = 0;
    }
}

__global__ __launch_bounds__(64) void k_am_decode_fwd(DevTables tb, DevBuffers db, const int *ids, int parity, int lane_id, int G, int warm)
{
    const int s = stream_of(ids, blockIdx.y), role = blockIdx.x;           // 0..7: P1 frame of that block, 8..8+G-1: P3 segment, then 8 PIDS frames
    __shared__ K9WSmem k9;
    if (role >= 8 + G) {
        // decode_process_pids_am's trellis (decode.c:502-504) for the block processed in step `pb` of this window
        const int pb = role - 8 - G;
        int *recp = db.am_pids_rec + ((size_t)s * NWIN + parity) * 8 + pb;
        const int r = *recp;
        if (r < 0) return;                                                 // wave-uniform
        __shared__ uint32_t pout[4];
        const int8_t *stage = db.am_pids_stage + (((size_t)s * NWIN + parity) * 8 + pb) * (3 * PIDS_LEN);
        unsigned long long *pdec = db.am_dec + ((size_t)lane_id * db.nstreams_alloc + s) * (size_t)(8 * AM_DEC_P1 + AM_DEC_P3)
                                 + (size_t)8 * AM_DEC_P1 + AM_DEC_P3 - (size_t)(9 - pb) * 4 * (PIDS_LEN + 64);   // tail of the P3 scratch: its frame is shorter than AM_P3_LEN_MA3 + 64 only by the slack reserved here
        viterbi_k9_wave(stage, PIDS_LEN, GEN_E2_0, GEN_E2_1, GEN_E2_2, pdec, pout, k9);
        if (threadIdx.x == 0) {
            BlockRecord &rec = db.records[(size_t)s * db.rec_cap + r];
            const uint32_t p[3] = { pout[0] ^ tb.scr_pids[0], pout[1] ^ tb.scr_pids[1], (pout[2] ^ tb.scr_pids[2]) & 0xffffu };
            rec.pids[0] = p[0]; rec.pids[1] = p[1]; rec.pids[2] = p[2];
            if (pids_crc_ok(p)) atomicOr(&rec.flags, (uint32_t)REC_PIDS_CRC);
            *recp = -1;
        }
        return;
    }
    const AmJob &job = db.am_job[(size_t)s * NWIN + parity];
    if (!job.valid) return;                                                // wave-uniform
    if (role >= 8 && job.rdbi) return;
    K9Meta &meta = db.am_k9meta[(size_t)lane_id * db.nstreams_alloc + s];
    const AmDecodeFrame f = am_decode_frame(db, job, s, parity, lane_id, role);
    if (role < 8) {
        const int lane = threadIdx.x;
        const K9Signs sg = k9_signs(lane, f.g0, f.g1, f.g2);
        for (int k = 0; k < 4; k++) k9.metric[0][4 * lane + k] = 0;
        WAVE_LDS_SYNC();
        const int cur = k9_forward_chunks(f.in, f.len, sg, f.dec, k9, 0, k9_chunks(f.len), 0, nullptr);
        const unsigned end = k9_end_state(*(const int4 *)&k9.metric[cur][4 * lane]);
        if (lane == 0) meta.p1_end[role] = end;
    } else {
        k9_forward_segment(f.in, f.len, f.g0, f.g1, f.g2, f.dec, meta, k9, role - 8, G, warm);
    }
}

__global__ __launch_bounds__(64) void k_am_decode_fix(DevBuffers db, const int *ids, int parity, int lane_id, int G)
{
    const int s = stream_of(ids, blockIdx.x);
    const AmJob &job = db.am_job[(size_t)s * NWIN + parity];
    if (!job.valid || job.rdbi) return;                                    // wave-uniform
    __shared__ K9WSmem k9;
    K9Meta &meta = db.am_k9meta[(size_t)lane_id * db.nstreams_alloc + s];
    const AmDecodeFrame f = am_decode_frame(db, job, s, parity, lane_id, 8);
    const unsigned end = k9_forward_fix(f.in, f.len, f.g0, f.g1, f.g2, f.dec, meta, k9, G, db.am_k9stats);
    if (threadIdx.x == 0) meta.end_state = end;
}

__global__ __launch_bounds__(64) void k_am_decode_tb(DevTables tb, DevBuffers db, const int *ids, int parity, int lane_id, int G, int runin, int l2_feedback)
{
    const int s = stream_of(ids, blockIdx.y), role = blockIdx.x;           // 0..7: P1 frame of that block, 8..8+G-1: P3 segment
    AmJob &job = db.am_job[(size_t)s * NWIN + parity];
    if (!job.valid) return;                                                // wave-uniform
    if (role >= 8 && job.rdbi) return;
    __shared__ K9WSmem k9;
    __shared__ int red[4];
    K9Meta &meta = db.am_k9meta[(size_t)lane_id * db.nstreams_alloc + s];
    const AmDecodeFrame f = am_decode_frame(db, job, s, parity, lane_id, role);
    if (role >= 8) { k9_traceback_segment(f.dec, f.len, meta, k9, f.out, role - 8, G, runin); return; }
    unsigned arrive = 0;
    const int ntb = (k9_pairs(f.len) + 31) >> 5;
    k9_traceback_chunks(f.dec, f.len, k9, (unsigned)wave_uniform((int)meta.p1_end[role]), ntb, 0, ntb, f.out, arrive);
    __threadfence_block();
    __syncthreads();
    uint32_t *out = f.out;
    const int err = am_bit_errors(f.in, out, AM_P1_LEN, GEN_E1_0, GEN_E1_1, GEN_E1_2, PUNCT_E1, 15, red);
    for (int w = threadIdx.x; w < AM_P1_WORDS; w += 64)
        out[w] = (out[w] ^ tb.scr_p1[w]) & (w == AM_P1_WORDS - 1 ? (1u << (AM_P1_LEN & 31)) - 1u : 0xffffffffu);
    __threadfence_block();
    __syncthreads();
    if (l2_feedback) {                                         // frame.c:535-540: file the verdict for the block that delivers this PDU
        L2Smem &l2 = *(L2Smem *)&k9;                           // the trellis scratch is dead by now
        const bool ok = l2_first_header_ok_am_block(out, l2);
        if (threadIdx.x == 0) { __threadfence(); atomicExch(&job.verdict[role], ok ? 1 : 2); }
    }
    am_decode_account(db, job, s, err);
}

__global__ __launch_bounds__(64) void k_am_decode_finish(DevTables tb, DevBuffers db, const int *ids, int parity, int lane_id, int G)
{
    const int s = stream_of(ids, blockIdx.x);
    AmJob &job = db.am_job[(size_t)s * NWIN + parity];
    if (!job.valid || job.rdbi) return;                                    // wave-uniform
    __shared__ K9WSmem k9;
    __shared__ int red[4];
    K9Meta &meta = db.am_k9meta[(size_t)lane_id * db.nstreams_alloc + s];
    const AmDecodeFrame f = am_decode_frame(db, job, s, parity, lane_id, 8);
    k9_traceback_fix(f.dec, f.len, meta, k9, f.out, G, db.am_k9stats);
    const bool ma3 = job.psmi == AM_MA3;
    const int err = ma3 ? am_bit_errors(f.in, f.out, f.len, GEN_E1_0, GEN_E1_1, GEN_E1_2, PUNCT_E1, 15, red)
                        : am_bit_errors(f.in, f.out, f.len, GEN_E2_0, GEN_E2_1, GEN_E2_2, PUNCT_E2, 6, red);
    const int words = (f.len + 31) / 32;
    const uint32_t tailmask = (f.len & 31) ? (1u << (f.len & 31)) - 1u : 0xffffffffu;
    for (int w = threadIdx.x; w < words; w += 64) f.out[w] = (f.out[w] ^ tb.scr_p1[w]) & (w == words - 1 ? tailmask : 0xffffffffu);
    am_decode_account(db, job, s, err);
}

void launch_am_decode(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, int parity, int lane_id, int l2_feedback, hipStream_t st,
                      int segments, int warm, int runin)
{
    const int G = segments < 1 ? 1 : segments > K9_GMAX ? K9_GMAX : segments;
    hipLaunchKernelGGL(k_am_decode_fwd, dim3(8 + G + 8, nstreams), dim3(64), 0, st, tb, db, stream_ids, parity, lane_id, G, warm);
    hipLaunchKernelGGL(k_am_decode_fix, dim3(nstreams), dim3(64), 0, st, db, stream_ids, parity, lane_id, G);
    hipLaunchKernelGGL(k_am_decode_tb, dim3(8 + G, nstreams), dim3(64), 0, st, tb, db, stream_ids, parity, lane_id, G, runin, l2_feedback);
    hipLaunchKernelGGL(k_am_decode_finish, dim3(nstreams), dim3(64), 0, st, tb, db, stream_ids, parity, lane_id, G);
    if (db.l2_am_ring) launch_l2_index_am_window(db, nstreams, stream_ids, parity, st);
}

void launch_am_step(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, hipStream_t st, int l2_feedback, int pipeline_parity, int slot, int window)
{
    // (per device: the attribute belongs to the function as loaded on the current device -- one process may drive several, include/nrsc5hip.h)
    static std::atomic<bool> attr_set[64];
    int dev = 0; (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {      // (a device index beyond the table: set it on every launch rather than never)
        (void)hipFuncSetAttribute((const void *)k_am_block<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(AmBlockSmem));
        (void)hipFuncSetAttribute((const void *)k_am_block<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(AmBlockSmem));
        if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
    }
    if (pipeline_parity >= 0) hipLaunchKernelGGL(k_am_block<512>, dim3(nstreams), dim3(512), sizeof(AmBlockSmem), st, tb, db, stream_ids, 1, pipeline_parity, slot);
    else hipLaunchKernelGGL(k_am_block<256>, dim3(nstreams), dim3(256), sizeof(AmBlockSmem), st, tb, db, stream_ids, 0, pipeline_parity, slot);
    if (pipeline_parity < 0) {
        hipLaunchKernelGGL(k_am_viterbi, dim3(2, nstreams), dim3(64), 0, st, tb, db, stream_ids, l2_feedback);
        if (db.l2_am_ring) launch_l2_index_am_step(db, nstreams, stream_ids, st);
    }
    hipLaunchKernelGGL(k_am_interleave, dim3(AM_IL_PARTS, nstreams), dim3(AM_IL_THREADS), 0, st, tb, db, stream_ids, pipeline_parity, window);
}

// ---- stage-level entry: decode `nframes` independent K=9 frames (parity tests) ------------------------------------
__global__ __launch_bounds__(256) void k_viterbi_k9_frames(const int8_t *coded, int len, unsigned g0, unsigned g1, unsigned g2,
                                                           unsigned long long *dec, uint32_t *out)
{
    __shared__ K9Smem k9;
    const int f = blockIdx.x;
    viterbi_k9_block(coded + (size_t)f * 3 * len, len, g0, g1, g2, dec + (size_t)f * 4 * (len + 64), out + (size_t)f * ((len + 31) / 32), k9);
}
__global__ __launch_bounds__(64) void k_viterbi_k9_frames_wave(const int8_t *coded, int len, unsigned g0, unsigned g1, unsigned g2,
                                                               unsigned long long *dec, uint32_t *out, int phases)
{
    __shared__ K9WSmem k9;
    const int f = blockIdx.x;
    viterbi_k9_wave(coded + (size_t)f * 3 * len, len, g0, g1, g2, dec + (size_t)f * 4 * (len + 64), out + (size_t)f * ((len + 31) / 32), k9, phases);
}
// the segment-wave form, one launch per stage (what k_am_decode_* do for the P3 frame)
__global__ __launch_bounds__(64) void k_k9seg_fwd(const int8_t *coded, int len, unsigned g0, unsigned g1, unsigned g2, unsigned long long *dec, K9Meta *meta, int G, int warm)
{
    __shared__ K9WSmem k9;
    const int f = blockIdx.y;
    k9_forward_segment(coded + (size_t)f * 3 * len, len, g0, g1, g2, dec + (size_t)f * 4 * (len + 64), meta[f], k9, (int)blockIdx.x, G, warm);
}
__global__ __launch_bounds__(64) void k_k9seg_fix(const int8_t *coded, int len, unsigned g0, unsigned g1, unsigned g2, unsigned long long *dec, K9Meta *meta, int G, unsigned *stats)
{
    __shared__ K9WSmem k9;
    const int f = blockIdx.x;
    const unsigned end = k9_forward_fix(coded + (size_t)f * 3 * len, len, g0, g1, g2, dec + (size_t)f * 4 * (len + 64), meta[f], k9, G, stats);
    if (threadIdx.x == 0) meta[f].end_state = end;
}
__global__ __launch_bounds__(64) void k_k9seg_tb(const unsigned long long *dec, int len, K9Meta *meta, uint32_t *out, int G, int runin)
{
    __shared__ K9WSmem k9;
    const int f = blockIdx.y;
    k9_traceback_segment(dec + (size_t)f * 4 * (len + 64), len, meta[f], k9, out + (size_t)f * ((len + 31) / 32), (int)blockIdx.x, G, runin);
}
__global__ __launch_bounds__(64) void k_k9seg_finish(const unsigned long long *dec, int len, K9Meta *meta, uint32_t *out, int G, unsigned *stats)
{
    __shared__ K9WSmem k9;
    const int f = blockIdx.x;
    k9_traceback_fix(dec + (size_t)f * 4 * (len + 64), len, meta[f], k9, out + (size_t)f * ((len + 31) / 32), G, stats);
}

void launch_viterbi_k9_frames(const int8_t *coded, int len, int nframes, unsigned g0, unsigned g1, unsigned g2,
                              unsigned long long *dec, uint32_t *out, hipStream_t st, int phases, K9Meta *meta, int segments, int warm, int runin, unsigned *stats)
{
    // frames longer than a PIDS frame take the production wave form (in segment waves when `meta` is given); 80-bit frames the
    // 256-work-item form
    if (len > 80 && meta) {
        const int G = segments < 1 ? 1 : segments > K9_GMAX ? K9_GMAX : segments;
        if (phases & 1) {
            hipLaunchKernelGGL(k_k9seg_fwd, dim3(G, nframes), dim3(64), 0, st, coded, len, g0, g1, g2, dec, meta, G, warm);
            hipLaunchKernelGGL(k_k9seg_fix, dim3(nframes), dim3(64), 0, st, coded, len, g0, g1, g2, dec, meta, G, stats);
        }
        if (phases & 2) {
            hipLaunchKernelGGL(k_k9seg_tb, dim3(G, nframes), dim3(64), 0, st, dec, len, meta, out, G, runin);
            hipLaunchKernelGGL(k_k9seg_finish, dim3(nframes), dim3(64), 0, st, dec, len, meta, out, G, stats);
        }
    }
    else if (len > 80) hipLaunchKernelGGL(k_viterbi_k9_frames_wave, dim3(nframes), dim3(64), 0, st, coded, len, g0, g1, g2, dec, out, phases);
    else hipLaunchKernelGGL(k_viterbi_k9_frames, dim3(nframes), dim3(256), 0, st, coded, len, g0, g1, g2, dec, out);
}

}  // namespace nrsc5
