// L1 FEC stage for gfx950: P1 de-interleave (K6), tail-biting Viterbi (K7), re-encode BER and
// descrambler (K8).  Replaces decode.c:296-322,451-461 and conv_dec.c for the P1 logical channel.
#include <hip/hip_runtime.h>
#include "kernels.h"
#include "viterbi_wave.h"
#include "viterbi_v3.h"
#include "l2_header.h"

namespace nrsc5 {

__device__ inline int stream_of(const int *ids, int idx) { return ids ? ids[idx] : idx; }

// ---- K6: P1 de-interleave + depuncture (interleaver I, decode.c:296-322) ---------------------------------------
// Coded bit i = 320 k + j of the frame sits in matrix row (11 k) % 32 of block (j/20 + 7 part) % 16, partition
// part = PM_V[j % 20], column (11 k + k/288) % 36.  All k that share a matrix row r (k = 3r + 32 m) read the same
// 16 x 720-byte lines, so one workgroup per (stream, r) stages those 11.5 KB in LDS with coalesced loads and emits
// its ~36 runs of 128 trellis steps (320 soft bits + 64 erasures [1,1,1,1,1,0]) as one dword per step -- the form the
// forward pass reads with scalar loads (viterbi_v3.h): HBM sees 369 KB in + 585 KB out per frame, instead of one
// cache line per gathered byte.
__global__ __launch_bounds__(256) void k_p1_deint(DevTables tb, DevBuffers db, const int *ids, int parity, int lane_id)
{
    const int s = stream_of(ids, blockIdx.y);
    StreamState &st = db.state[s];
    if (!st.p1_pending[parity]) return;                        // block-uniform
    __shared__ uint32_t tile[16 * 180];                        // [block][720 bytes]
    __shared__ uint16_t lut[384];
    const int r = blockIdx.x, tid = threadIdx.x;
    const int8_t *pm = db.pm + ((size_t)s * NPM + st.p1_pmslot[parity]) * PM_FRAME;
    uint32_t *out = (uint32_t *)(db.coded + ((size_t)lane_id * db.nstreams_alloc + s) * P1_LEN);
    for (int w = tid; w < 16 * 180; w += 256) {
        const int b = w / 180, x = w % 180;
        tile[w] = ((const uint32_t *)(pm + ((size_t)b * 32 + r) * 720))[x];
    }
    for (int q = tid; q < 384; q += 256) lut[q] = tb.deint_lut[q];
    __syncthreads();
    const uint8_t *bytes = (const uint8_t *)tile;
    const int k0 = (3 * r) & 31;                               // 11 k = r (mod 32)  <=>  k = 3 r (mod 32)
    const int nk = (P1_CODED / 320 - k0 + 31) / 32;            // k = k0 + 32 m < 1142
    for (int idx = tid; idx < nk * 128; idx += 256) {
        const int m = idx >> 7, i = idx & 127;                 // step i of the 128-step run of k
        const int k = k0 + 32 * m;
        const int col = (11 * k + k / 288) % 36;
        uint32_t v = 0;
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const unsigned off = lut[3 * i + t];
            if (off != 0xffffu) v |= (uint32_t)bytes[off + col] << (8 * t);
        }
        out[128 * k + i] = v;
    }
}

// ---- K8 helpers ---------------------------------------------------------------------------------
// Re-encode the decoded (still scrambled) bits and count sign disagreements with the received soft
// bits at unpunctured positions (decode.c:234-265).  Block-stride over the frame; returns this
// thread's partial count.
// one frame bit's contribution
__device__ inline int bit_errors_k7_at(const int *soft, const uint32_t *bits, int len, int i)
{
    unsigned r = 0;                                            // r bit 6 = bits[i], bit 6-k = bits[i-k]
#pragma unroll
    for (int k = 0; k < 7; k++) {
        int q = i - k; if (q < 0) q += len;                    // tail biting
        r |= ((bits[q >> 5] >> (q & 31)) & 1u) << (6 - k);
    }
    const int w = soft[i];
    const int c0 = (int8_t)w, c1 = (int8_t)(w >> 8), c2 = (int8_t)(w >> 16);
    const int p0 = __popc(r & 0133u) & 1, p1 = __popc(r & 0171u) & 1, p2 = __popc(r & 0165u) & 1;
    return ((c0 > 0) != p0) + ((c1 > 0) != p1) + (((i & 1) == 0) && ((c2 > 0) != p2));
}

__device__ inline int bit_errors_k7_partial(const int *soft, const uint32_t *bits, int len)
{
    int errors = 0;
    for (int i = threadIdx.x; i < len; i += blockDim.x) {
        unsigned r = 0;                                        // r bit 6 = bits[i], bit 6-k = bits[i-k]
#pragma unroll
        for (int k = 0; k < 7; k++) {
            int q = i - k; if (q < 0) q += len;                // tail biting
            r |= ((bits[q >> 5] >> (q & 31)) & 1u) << (6 - k);
        }
        const int w = soft[i];
        const int c0 = (int8_t)w, c1 = (int8_t)(w >> 8), c2 = (int8_t)(w >> 16);
        const int p0 = __popc(r & 0133u) & 1, p1 = __popc(r & 0171u) & 1, p2 = __popc(r & 0165u) & 1;
        if ((c0 > 0) != p0) errors++;
        if ((c1 > 0) != p1) errors++;
        if ((i & 1) == 0 && ((c2 > 0) != p2)) errors++;        // odd i: the third bit is punctured [1,1,1,1,1,0]
    }
    return errors;
}

// ---- K7: P1 frame = forward pass by one wave, then traceback/BER/descramble by a 16-wave block -------
// Four frames per workgroup, one per wave: the waves of a workgroup land on the four SIMDs of one CU, so no two trellis
// passes of a launch share a SIMD (with one-wave workgroups the dispatcher sometimes stacks them, and the launch lasts as long
// as its slowest wave).
constexpr int FWD_WAVES = 4;
// G segment waves per frame run concurrently (viterbi_v3.h: speculative start, verified and repaired by k_p1_fix): the launch
// lasts 1 / G of the serial chain when the chip has the SIMDs to spare -- thin windows, the single stream, the in-order seam.
__global__ __launch_bounds__(64 * FWD_WAVES) void k_p1_forward(DevTables tb, DevBuffers db, const int *ids, int parity, int lane_id, int prio, int nstreams, int G, int warm)
{
    wave_set_priority(prio);
    const int widx = wave_uniform((int)(blockIdx.x * FWD_WAVES + (threadIdx.x >> 6)));
    if (widx >= nstreams * G) return;                          // wave-uniform
    const int sidx = widx / G, g = widx % G;
    const int s = wave_uniform(stream_of(ids, sidx));
    StreamState &st = db.state[s];
    if (!st.p1_pending[parity]) return;                        // wave-uniform
    const size_t slot = (size_t)lane_id * db.nstreams_alloc + s;
    const int *soft = db.coded + slot * P1_LEN;
    uint32_t *dec = db.dec + slot * (size_t)(2 * (P1_LEN + 64));
    viterbi3_forward_segment(soft, P1_LEN, dec, db.fwd_meta + slot * (size_t)(VIT3_GMAX * VIT3_META), g, G, warm);
}

// one wave per frame: check the segment boundaries of k_p1_forward, re-run what the speculation got wrong, pick the end state
__global__ __launch_bounds__(64 * FWD_WAVES) void k_p1_fix(DevTables tb, DevBuffers db, const int *ids, int parity, int lane_id, int prio, int nstreams, int G)
{
    wave_set_priority(prio);
    const int sidx = wave_uniform((int)(blockIdx.x * FWD_WAVES + (threadIdx.x >> 6)));
    if (sidx >= nstreams) return;                              // wave-uniform
    const int s = wave_uniform(stream_of(ids, sidx));
    StreamState &st = db.state[s];
    if (!st.p1_pending[parity]) return;                        // wave-uniform
    const size_t slot = (size_t)lane_id * db.nstreams_alloc + s;
    const int endlane = viterbi3_forward_fix(db.coded + slot * P1_LEN, P1_LEN, db.dec + slot * (size_t)(2 * (P1_LEN + 64)),
                                             db.fwd_meta + slot * (size_t)(VIT3_GMAX * VIT3_META), G, db.fwd_stats);
    if ((threadIdx.x & 63) == 0) st.p1_endlane[parity] = endlane;
}

constexpr int TB_THREADS = 1024;
constexpr int TBM_WAVES = 4;

// pass 1 of the traceback (chunk maps + candidate outputs) as its own launch: `parts` workgroups of 4 waves per frame
__global__ __launch_bounds__(64 * TBM_WAVES) void k_p1_tbmap(DevTables tb, DevBuffers db, const int *ids, int parity, int lane_id, int prio, int parts)
{
    wave_set_priority(prio);
    const int s = stream_of(ids, blockIdx.y);
    StreamState &st = db.state[s];
    if (!st.p1_pending[parity]) return;                        // block-uniform
    const size_t slot = (size_t)lane_id * db.nstreams_alloc + s;
    viterbi3_traceback_maps(db.dec + slot * (size_t)(2 * (P1_LEN + 64)), P1_LEN, db.tbmap + slot * ((size_t)(P1_LEN / 64 + 1) * 64),
                            (int)(blockIdx.x * TBM_WAVES + (threadIdx.x >> 6)), parts * TBM_WAVES);
}

// Round 4: the single-path traceback (viterbi_v3.h: lane = chunk, run-in through the chunk above, verified by k_p1_traceback).
// 36 one-wave workgroups per frame, 33 KB of LDS each (four per CU).
// Round 5: a PERSISTENT grid.  As 36 x 256 = 9216 workgroups the launch kept thousands of workgroups PENDING in the dispatcher for its ~450 us (four fit a CU: 33 KB
// of LDS each), and the block-step kernels of the chain queue waited behind them: the k_sync launch that coincided with a window's traceback lasted 370 - 550 us
// instead of 37 (one per decode window: ~6 of the pass's 30 ms; profiles/r05_trace_sync.txt).  `ntasks` (part, stream) pairs are walked by gridDim.x resident
// workgroups, parts of one frame by neighbouring workgroups.
__global__ __launch_bounds__(64) void k_p1_tbwalk(DevTables tb, DevBuffers db, const int *ids, int parity, int lane_id, int prio, int nparts, int ntasks)
{
    wave_set_priority(prio);
    __shared__ uint32_t lds[TB2_LDS_WORDS];
    for (int t = (int)blockIdx.x; t < ntasks; t += (int)gridDim.x) {
        const int s = wave_uniform(stream_of(ids, t / nparts));
        StreamState &st = db.state[s];
        if (!st.p1_pending[parity]) continue;                  // wave-uniform (one wave per workgroup)
        const size_t slot = (size_t)lane_id * db.nstreams_alloc + s;
        uint32_t *out = db.p1_ring + ((size_t)s * db.p1_slots + st.p1_slot[parity]) * P1_WORDS;
        viterbi3_traceback_walk(db.dec + slot * (size_t)(2 * (P1_LEN + 64)), P1_LEN, st.p1_endlane[parity], out,
                                db.tbmap + slot * ((size_t)(P1_LEN / 64 + 1) * 64), t % nparts, lds, db.coded + slot * P1_LEN);
        WAVE_LDS_FENCE();                                      // the next task refills the tile
    }
}

// maps_done: 0 = the block-parallel traceback runs all its passes here; 1 = k_p1_tbmap ran pass 1; 2 = k_p1_tbwalk has written the
// frame speculatively: verify / repair only
__global__ __launch_bounds__(TB_THREADS) void k_p1_traceback(DevTables tb, DevBuffers db, const int *ids, int parity, int lane_id, int l2_mode, int prio, int maps_done)
{
    wave_set_priority(prio);
    const int s = stream_of(ids, blockIdx.x);
    StreamState &st = db.state[s];
    if (!st.p1_pending[parity]) return;                        // block-uniform
    HIP_DYNAMIC_SHARED(uint8_t, smem)
    __shared__ int err_total;
    const int tid = threadIdx.x;
    const int *soft = db.coded + ((size_t)lane_id * db.nstreams_alloc + s) * P1_LEN;
    uint32_t *dec = db.dec + ((size_t)lane_id * db.nstreams_alloc + s) * (size_t)(2 * (P1_LEN + 64));
    uint32_t *out = db.p1_ring + ((size_t)s * db.p1_slots + st.p1_slot[parity]) * P1_WORDS;
    if (tid == 0) err_total = 0;
    uint8_t *gmap = db.tbmap + ((size_t)lane_id * db.nstreams_alloc + s) * ((size_t)(P1_LEN / 64 + 1) * 64);
    if (maps_done == 2) viterbi3_traceback_check(dec, P1_LEN, out, gmap, db.tb_stats, soft);
    else viterbi3_traceback_block(dec, P1_LEN, st.p1_endlane[parity], out, gmap, smem, maps_done != 0);
    __threadfence_block();
    __syncthreads();
    int errors;
    if (maps_done == 2) {
        // every chunk's walk filed its own re-encode disagreements (36 workgroups per frame did the counting: as one workgroup's loop
        // over the frame it was most of this kernel); what is left are the frame's first six bits, whose window wraps to its last ones
        constexpr int NCH = P1_LEN / 64 + 1;
        errors = 0;
        for (int c = tid; c < NCH; c += blockDim.x) errors += gmap[2 * NCH + c];
        if (tid < 6) errors += bit_errors_k7_at(soft, out, P1_LEN, tid);
        errors = wave_sum_i32(errors);
    } else {
        errors = wave_sum_i32(bit_errors_k7_partial(soft, out, P1_LEN));
    }
    if ((tid & 63) == 0) atomicAdd(&err_total, errors);
    __syncthreads();
    uint32_t *mirror = db.p1_mirror ? db.p1_mirror + ((size_t)s * db.p1_slots + st.p1_slot[parity]) * P1_WORDS : nullptr;
    for (int w = tid; w < P1_WORDS; w += blockDim.x) {
        const uint32_t v = out[w] ^ tb.scr_p1[w];              // descramble
        out[w] = v;
        if (mirror) mirror[w] = v;                             // the host's copy, posted over PCIe while the pass is still running
    }
    __threadfence_block();
    __syncthreads();
    // frame_process -> input_set_sync_state(NONE) when the first L2 header does not decode (frame.c:535-540).
    // l2_mode 1: in-order decode on the main stream -> the very next block starts from NONE, as in the reference;
    // l2_mode 2: deferred decode -> file the verdict; k_rollback (k_replay.hip) rewinds the stream to the end of the
    //            frame's block and restarts it from NONE there, so the outcome is the reference's all the same
    __shared__ L2Smem l2;
    const bool ok = l2_mode ? l2_first_header_ok_fm_block(out, l2) : true;     // block-uniform; the whole workgroup takes part
    if (tid == 0) {
        BlockRecord &rec = db.records[(size_t)s * db.rec_cap + st.p1_record[parity]];
        rec.ber = (float)err_total / P1_CODED;                 // decode.c:458
        if (l2_mode == 1) { if (!ok && st.sync_state == SYNC_FINE) { st.sync_state = SYNC_NONE; rec.state_after = SYNC_NONE; rec.flags |= REC_LOST_SYNC; } }
        else if (l2_mode) { __threadfence(); st.p1_verdict[parity] = ok ? 1 : 2; }
        if (db.l2_ring) st.p1_l2slot[parity] = st.p1_slot[parity] + 1;
        st.p1_pending[parity] = 0;
    }
}

static size_t traceback_smem(int len) { return vit3_traceback_smem(len); }

// The three stages of a window's P1 decode, launched back to back on the window's decode stream (separate entry points so that
// the engine can time the trellis pass -- the dominant kernel of the whole path -- on its own).
// Wave priorities (s_setprio): the forward pass is long-running background work next to the step chain (priority 3); raising it
// to 1 or 2 was measured again with the 6-instruction trellis: no gain (profiles/r02_naux.txt).  The traceback is the short
// kernel at the end of each decode chain: one step above the trellis waves, but below the chain -- at the chain's own level
// (3) its 4096 waves per launch sit on the serial Costas loops of k_sync (profiles/r02_ab_prio_tb.txt).
void launch_p1_deint(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, int parity, int lane_id, hipStream_t st)
{
    hipLaunchKernelGGL(k_p1_deint, dim3(32, nstreams), dim3(256), 0, st, tb, db, stream_ids, parity, lane_id);
}
void launch_p1_forward(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, int parity, int lane_id, hipStream_t st, int segments, int warm)
{
    constexpr int prio_fwd = 0;
    const int G = vit3_segments(P1_LEN, segments);
    hipLaunchKernelGGL(k_p1_forward, dim3((nstreams * G + FWD_WAVES - 1) / FWD_WAVES), dim3(64 * FWD_WAVES), 0, st, tb, db, stream_ids, parity, lane_id, prio_fwd, nstreams, G, warm);
    hipLaunchKernelGGL(k_p1_fix, dim3((nstreams + FWD_WAVES - 1) / FWD_WAVES), dim3(64 * FWD_WAVES), 0, st, tb, db, stream_ids, parity, lane_id, prio_fwd, nstreams, G);
}
void launch_p1_traceback(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, int parity, int lane_id, hipStream_t st, int l2_mode, int parts, int walk)
{
    constexpr int prio_tb = 1;
    // Thin windows / small stream sets (parts = 16): pass 1 of the traceback as its own launch over `parts` workgroups per frame,
    // so that a handful of frames spread over the chip (one frame: 0.48 -> 0.11 ms).  Full windows keep it inside the 16-wave
    // traceback workgroup: 1024 small workgroups at once crowd the block-step kernels off the SIMDs (measured: k_sync 14 -> 25 ms
    // per pass, the pass 36 -> 50 ms; profiles/r03_traceback_variants.txt).
    if (walk) {
        // walk: 1 = one workgroup per task (round 4), N > 1 = a persistent grid of N workgroups
        const int nparts = vit3_tb2_waves(P1_LEN), ntasks = nparts * nstreams;
        const int grid = walk > 1 ? (walk < ntasks ? walk : ntasks) : ntasks;
        hipLaunchKernelGGL(k_p1_tbwalk, dim3(grid), dim3(64), 0, st, tb, db, stream_ids, parity, lane_id, prio_tb, nparts, ntasks);
        hipLaunchKernelGGL(k_p1_traceback, dim3(nstreams), dim3(TB_THREADS), traceback_smem(P1_LEN), st, tb, db, stream_ids, parity, lane_id, l2_mode, prio_tb, 2);
    } else {
        const int split = parts >= 16 ? 16 : 0;
        if (split) hipLaunchKernelGGL(k_p1_tbmap, dim3(split, nstreams), dim3(64 * TBM_WAVES), 0, st, tb, db, stream_ids, parity, lane_id, prio_tb, split);
        hipLaunchKernelGGL(k_p1_traceback, dim3(nstreams), dim3(TB_THREADS), traceback_smem(P1_LEN), st, tb, db, stream_ids, parity, lane_id, l2_mode, prio_tb, split ? 1 : 0);
    }
    if (db.l2_ring) launch_l2_index_window(db, nstreams, stream_ids, parity, st);
}

// ---- stage-level entry: the first-header check (l2_header.h) on packed, descrambled frames, one workgroup per frame -------------
__global__ void k_stage_first_header(const uint32_t *words, int nwords, int am, int *ok)
{
    __shared__ L2Smem l2;
    const uint32_t *w = words + (size_t)blockIdx.x * nwords;
    const bool v = am ? l2_first_header_ok_am_block(w, l2) : l2_first_header_ok_fm_block(w, l2);
    if (threadIdx.x == 0) ok[blockIdx.x] = v ? 1 : 0;
}
void launch_stage_first_header(const uint32_t *words, int nwords, int nframes, int am, int threads, int *ok, hipStream_t st)
{
    hipLaunchKernelGGL(k_stage_first_header, dim3(nframes), dim3(threads), 0, st, words, nwords, am, ok);
}

// ---- stage-level entry: decode `nframes` independent frames of equal length (parity tests) ----------
// phases (micro-benchmark): bit0 forward, bit1 traceback; bit2 selects the single-wave sequential traceback.
__global__ __launch_bounds__(64) void k_viterbi_frames(const int8_t *coded, int len, unsigned long long *dec, uint32_t *out, int phases)
{
    const int f = blockIdx.x;
    viterbi_k7_decode(coded + (size_t)f * 3 * len, len, dec + (size_t)f * (len + 64), out + (size_t)f * ((len + 31) / 32), phases);
}
// soft values as the reference hands them to nrsc5_conv_decode (3 int8 per step) -> one dword per step
__global__ void k_pack_soft3(const int8_t *coded, int *soft, size_t nsteps)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsteps) return;
    soft[i] = (int)(uint8_t)coded[3 * i] | ((int)(uint8_t)coded[3 * i + 1] << 8) | ((int)(uint8_t)coded[3 * i + 2] << 16);
}
__global__ __launch_bounds__(64) void k_viterbi_frames_fwd(const int *soft, int len, uint32_t *dec, int *meta, int G, int warm)
{
    const int f = blockIdx.x / G, g = blockIdx.x % G;
    viterbi3_forward_segment(soft + (size_t)f * len, len, dec + (size_t)f * 2 * (len + 64), meta + (size_t)f * VIT3_GMAX * VIT3_META, g, G, warm);
}
__global__ __launch_bounds__(64) void k_viterbi_frames_fix(const int *soft, int len, uint32_t *dec, int *meta, int G, int *endlane, int *stats)
{
    const int f = blockIdx.x;
    const int e = viterbi3_forward_fix(soft + (size_t)f * len, len, dec + (size_t)f * 2 * (len + 64), meta + (size_t)f * VIT3_GMAX * VIT3_META, G, stats);
    if ((threadIdx.x & 63) == 0) endlane[f] = e;
}
__global__ __launch_bounds__(64 * TBM_WAVES) void k_viterbi_frames_tbmap(uint32_t *dec, int len, uint8_t *gmap, int parts)
{
    const int f = blockIdx.y;
    viterbi3_traceback_maps(dec + (size_t)f * 2 * (len + 64), len, gmap + (size_t)f * (len / 64 + 1) * 64, (int)(blockIdx.x * TBM_WAVES + (threadIdx.x >> 6)), parts * TBM_WAVES);
}
__global__ __launch_bounds__(TB_THREADS) void k_viterbi_frames_tb(uint32_t *dec, int len, const int *endlane, uint32_t *out, uint8_t *gmap, int maps_done)
{
    HIP_DYNAMIC_SHARED(uint8_t, smem)
    const int f = blockIdx.x;
    viterbi3_traceback_block(dec + (size_t)f * 2 * (len + 64), len, endlane[f], out + (size_t)f * ((len + 31) / 32), gmap + (size_t)f * (len / 64 + 1) * 64, smem, maps_done != 0);
}

__global__ __launch_bounds__(64) void k_viterbi_frames_tbwalk(const uint32_t *dec, int len, const int *endlane, uint32_t *out, uint8_t *gmap)
{
    __shared__ uint32_t lds[TB2_LDS_WORDS];
    const int f = blockIdx.y;
    viterbi3_traceback_walk(dec + (size_t)f * 2 * (len + 64), len, endlane[f], out + (size_t)f * ((len + 31) / 32), gmap + (size_t)f * (len / 64 + 1) * 64, (int)blockIdx.x, lds);
}
__global__ __launch_bounds__(TB_THREADS) void k_viterbi_frames_tbcheck(const uint32_t *dec, int len, uint32_t *out, uint8_t *gmap, int *stats)
{
    const int f = blockIdx.x;
    viterbi3_traceback_check(dec + (size_t)f * 2 * (len + 64), len, out + (size_t)f * ((len + 31) / 32), gmap + (size_t)f * (len / 64 + 1) * 64, stats);
}

void vit_scratch_free(VitScratch &sc)
{
    if (sc.endlane) (void)hipFree(sc.endlane);
    if (sc.gmap) (void)hipFree(sc.gmap);
    if (sc.soft) (void)hipFree(sc.soft);
    if (sc.meta) (void)hipFree(sc.meta);
    sc = VitScratch();
}

template <typename T, typename N> static bool vit_grow(T *&p, N &cap, N need, size_t elem)
{
    if (cap >= need) return true;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    if (hipMalloc((void **)&p, elem * (size_t)need) != hipSuccess) { p = nullptr; return false; }
    cap = need;
    return true;
}

// the block-parallel traceback keeps one end lane per segment of TB_SEG chunks in a fixed 64-entry LDS array (viterbi_v3.h)
static_assert((P1_LEN / 64 + 1 + TB_SEG - 1) / TB_SEG <= 64, "P1 frame exceeds the traceback's segment table");
constexpr int VIT3_MAX_LEN = (64 * TB_SEG - 1) * 64;

int launch_viterbi_frames(VitScratch &sc, const int8_t *coded, int len, int nframes, unsigned long long *dec, uint32_t *out, hipStream_t st, int phases, int segments, int *stats, int warm)
{
    if ((len & 63) == 0 && !(phases & 4)) {                    // production split: forward segment waves + fix + parallel traceback (viterbi_v3.h)
        if (len > VIT3_MAX_LEN) return -1;
        int *&endlane = sc.endlane; uint8_t *&gmap = sc.gmap; int *&soft = sc.soft; int *&meta = sc.meta;
        const int cap_before = sc.cap;
        if (!vit_grow(endlane, sc.cap, nframes, sizeof(int))) return -1;
        if (sc.cap != cap_before) (void)hipMemsetAsync(endlane, 0, sizeof(int) * nframes, st);
        if (!vit_grow(meta, sc.mcap, nframes, sizeof(int) * (size_t)VIT3_GMAX * VIT3_META)) return -1;
        if (!vit_grow(gmap, sc.gcap, (size_t)nframes * (len / 64 + 1) * 64, 1)) return -1;
        if (!vit_grow(soft, sc.scap, (size_t)nframes * len, sizeof(int))) return -1;
        const size_t nsteps = (size_t)nframes * len;
        if (phases & 1) {
            if (!(phases & 8))                                 // bit 3 (micro-benchmark): the soft words of this input are packed already
                hipLaunchKernelGGL(k_pack_soft3, dim3((unsigned)((nsteps + 255) / 256)), dim3(256), 0, st, coded, soft, nsteps);
            const int G = vit3_segments(len, segments);
            hipLaunchKernelGGL(k_viterbi_frames_fwd, dim3(nframes * G), dim3(64), 0, st, (const int *)soft, len, (uint32_t *)dec, meta, G, warm);
            hipLaunchKernelGGL(k_viterbi_frames_fix, dim3(nframes), dim3(64), 0, st, (const int *)soft, len, (uint32_t *)dec, meta, G, endlane, stats);
        }
        if ((phases & 2) && !(phases & 16)) {               // the single-path traceback (production default); bit 4 selects the block-parallel one
            hipLaunchKernelGGL(k_viterbi_frames_tbwalk, dim3(vit3_tb2_waves(len), nframes), dim3(64), 0, st, (const uint32_t *)dec, len, (const int *)endlane, out, gmap);
            hipLaunchKernelGGL(k_viterbi_frames_tbcheck, dim3(nframes), dim3(TB_THREADS), 0, st, (const uint32_t *)dec, len, out, gmap, stats ? stats + 2 : nullptr);
        } else if (phases & 2) {
            const int split = segments >= 16 ? 16 : 0;      // as launch_p1_traceback
            if (split) hipLaunchKernelGGL(k_viterbi_frames_tbmap, dim3(split, nframes), dim3(64 * TBM_WAVES), 0, st, (uint32_t *)dec, len, gmap, split);
            hipLaunchKernelGGL(k_viterbi_frames_tb, dim3(nframes), dim3(TB_THREADS), traceback_smem(len), st, (uint32_t *)dec, len, (const int *)endlane, out, gmap, split ? 1 : 0);
        }
        return 0;
    }
    hipLaunchKernelGGL(k_viterbi_frames, dim3(nframes), dim3(64), 0, st, coded, len, dec, out, phases & 3);
    return 0;
}

// ---- device self-test of the register-file lane exchanges against the generic shuffle ----------------
__global__ __launch_bounds__(64) void k_selftest(int *fail)
{
    const int lane = threadIdx.x & 63;
    int bad = 0;
    for (int rep = 0; rep < 4; rep++) {
        const int v = (lane * 2654435 + rep * 977) ^ (lane << 20);
        bad += lane_xor<1>(v) != __shfl_xor(v, 1);
        bad += lane_xor<2>(v) != __shfl_xor(v, 2);
        bad += lane_xor<4>(v) != __shfl_xor(v, 4);
        bad += lane_xor<8>(v) != __shfl_xor(v, 8);
        bad += lane_xor<16>(v) != __shfl_xor(v, 16);
        bad += lane_xor<32>(v) != __shfl_xor(v, 32);
        int w = wave_writelane_c<37>(v, 12345);
        bad += w != (lane == 37 ? 12345 : v);
        const int a = (int)0x00817f05, b = (int)0x00ff0103;   // bytes (5,127,-127,0) . (3,1,-1,0) = 15+127+127
        bad += dot4_i8(a, b, 7) != 7 + 15 + 127 + 127;
    }
    atomicAdd(fail, bad);
}

void launch_selftest(int *fail, hipStream_t st) { hipLaunchKernelGGL(k_selftest, dim3(4), dim3(64), 0, st, fail); }

}  // namespace nrsc5
