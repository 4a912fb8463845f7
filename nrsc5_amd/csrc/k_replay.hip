// Replay for the window pipeline (p1_async) with the on-device L2 -> L1 feedback (l2_feedback).
//
// The reference judges the first L2 header of every P1 frame inside the block that completes the frame
// (frame_process -> input_set_sync_state(SYNC_STATE_NONE), frame.c:535-540, input.c:172-188) and starts the very next
// block from SYNC_STATE_NONE when the RS(255,247) check fails (acquire.c:110-119).  In the window pipeline the frame is
// decoded windows later, so the stream runs on SPECULATIVELY.  When a verdict "failed" arrives, k_rollback rewinds the
// stream to the state k_sync saved at the end of the frame's block (DevBuffers::ckpt), drops it to SYNC_NONE there and
// lets it run again from that point: LOST_SYNC, re-acquisition and every later frame land on the reference's blocks.
//
// What makes this safe without stopping the decode streams: nothing a speculated block produced is ever reused.
// Record indices, P1 / P3 / P4 ring slots are NOT rewound -- the records of the speculated blocks are marked
// REC_DISCARDED instead (the host skips them) -- so deferred decodes of speculated blocks that are still in flight
// write where nobody looks.  A verdict is honoured only if the record that announced its frame is still valid.
#include <hip/hip_runtime.h>
#include "kernels.h"

namespace nrsc5 {

__device__ inline int stream_of(const int *ids, int idx) { return ids ? ids[idx] : idx; }

__global__ __launch_bounds__(256) void k_rollback(DevBuffers db, const int *ids, int cur_window, int min_age)
{
    const int s = stream_of(ids, blockIdx.x);
    StreamState &st = db.state[s];
    BlockRecord *ring = db.records + (size_t)s * db.rec_cap;
    __shared__ int sh_p;
    const int tid = threadIdx.x;
    if (tid == 0) {
        // earliest failed frame whose block is still part of the stream's history
        int best = -1, best_rec = 0x7fffffff;
        for (int p = 0; p < NWIN; p++) {
            const int v = st.p1_verdict[p];
            if (v == 0 || cur_window - st.p1_window[p] < min_age) continue;   // min_age > 0: test hook, verdicts take effect late
            st.p1_verdict[p] = 0;                                              // consumed
            if (v != 2) continue;
            const int a = st.p1_recabs[p];
            if (ring[a % db.rec_cap].flags & REC_DISCARDED) continue;          // frame of a block that was rewound over already
            if (a < best_rec) { best_rec = a; best = p; }
        }
        sh_p = best;
    }
    __syncthreads();
    const int p = sh_p;
    if (p < 0) return;                                         // block-uniform
    const StreamState &ck = db.ckpt[(size_t)s * NWIN + p];
    const int a = st.p1_recabs[p], n = st.nblocks;
    for (int r = a + 1 + tid; r < n; r += 256) atomicOr(&ring[r % db.rec_cap].flags, (uint32_t)REC_DISCARDED);
    // tracking state as of the end of block a; the input side (wr, base, hb_hist), the record / ring-slot counters and the
    // per-window job descriptors keep their current values
    for (int l = tid; l < LIVE_N; l += 256) { st.costas_freq[l] = ck.costas_freq[l]; st.costas_phase[l] = ck.costas_phase[l]; }
    if (tid < 31) { st.fir_hist[tid] = ck.fir_hist[tid]; st.stale.fir[MODE_FM][tid] = ck.stale.fir[MODE_FM][tid]; }    // the acquisition filter's window as of block a
    if (tid == 0) {
        st.rd = ck.rd;
        st.prev_angle = ck.prev_angle; st.theta = ck.theta; st.keep_extra = ck.keep_extra; st.cfo = ck.cfo;
        st.psmi = ck.psmi; st.cfo_wait = ck.cfo_wait; st.bc = ck.bc; st.samperr = ck.samperr; st.angle = ck.angle;
        st.mer_cnt = ck.mer_cnt; st.error_lb = ck.error_lb; st.error_ub = ck.error_ub;
        st.started_pm = ck.started_pm; st.pm_slot = ck.pm_slot; st.last_pm_slot = ck.last_pm_slot;
        st.px_pos = ck.px_pos; st.px_ready = ck.px_ready; st.px_started = ck.px_started; st.px_go = 0;
        st.fine_epoch = ck.fine_epoch;
        st.stale.fir_pushed[MODE_FM] = ck.stale.fir_pushed[MODE_FM];
        st.sync_state = SYNC_NONE;                             // input_set_sync_state(NONE) at the end of block a
        st.active = 0;                                         // a block the fused bookkeeping already opened is void
        BlockRecord &rec = ring[a % db.rec_cap];
        if (rec.state_after == SYNC_FINE) { rec.state_after = SYNC_NONE; rec.flags |= REC_LOST_SYNC; }
        st.ndiscard += n - a - 1;
        atomicAdd(&db.counters[1], 1);                         // host: the next burst needs the acquisition kernels
        atomicAdd(&db.counters[3], 1);
        atomicAdd(&db.counters[0], 1);                         // ... and there is work again
    }
}

// ---- AM ------------------------------------------------------------------------------------------------------------------
// Every block of an AM L1 frame delivers one P1 PDU (decode_process_p1_p3_am, decode.c:507-554), each judged by frame_process
// on its own, so the deferred decode of an L1 frame files eight verdicts (AmJob::verdict) and k_am_block keeps a checkpoint
// per delivering block (DevBuffers::am_ckpt).  A verdict that is already there when its block runs is applied by k_am_block
// itself; the others are taken here: the earliest failed PDU whose delivering block is still part of the stream's history wins.
// As for FM nothing a speculated block produced is reused: frame slots and records are not rewound, the symbol matrices and the
// diversity delay lines it touched are rewritten in full before the re-locked stream delivers again (am_diversity_wait = 4).
__global__ __launch_bounds__(256) void k_rollback_am(DevBuffers db, const int *ids, int cur_window, int min_age)
{
    const int s = stream_of(ids, blockIdx.x);
    StreamState &st = db.state[s];
    AmStream &am = db.am[s];
    BlockRecord *ring = db.records + (size_t)s * db.rec_cap;
    __shared__ int sh_p, sh_j, sh_best;
    const int tid = threadIdx.x;
    if (tid == 0) { sh_best = 0x7fffffff; sh_p = -1; sh_j = 0; }
    __syncthreads();
    int mine = 0x7fffffff;
    if (tid < NWIN * 8) {                                      // one work-item per (job, PDU)
        const int p = tid >> 3, j = tid & 7;
        AmJob &job = db.am_job[(size_t)s * NWIN + p];
        if (cur_window - job.window >= min_age                 // min_age > 0: test hook, verdicts take effect late
            && job.verdict[j] == 2 && job.deliver_abs[j] >= 0) {                   // failed, not applied yet, and delivered
            job.verdict[j] = 3;                                                    // consumed
            const int a = job.deliver_abs[j];
            if (!(ring[a % db.rec_cap].flags & REC_DISCARDED)) { mine = a; atomicMin(&sh_best, a); }   // else: its block was rewound over already
        }
    }
    __syncthreads();
    if (tid < NWIN * 8 && mine != 0x7fffffff && mine == sh_best) { sh_p = tid >> 3; sh_j = tid & 7; }      // record indices are unique per stream
    __syncthreads();
    const int p = sh_p, j = sh_j;
    if (p < 0) return;                                         // block-uniform
    const AmCkpt &ck = db.am_ckpt[((size_t)s * NWIN + p) * 8 + j];
    const int a = db.am_job[(size_t)s * NWIN + p].deliver_abs[j], n = st.nblocks;
    for (int r = a + 1 + tid; r < n; r += 256) atomicOr(&ring[r % db.rec_cap].flags, (uint32_t)REC_DISCARDED);
    if (tid < 31) { st.fir_hist[tid] = ck.st.fir_hist[tid]; st.stale.fir[MODE_AM][tid] = ck.st.stale.fir[MODE_AM][tid]; }
    if (tid == 0) {
        // tracking state as of the end of block a; the input side (wr, base, the 32:1 decimator's raw history) and the record /
        // frame-slot counters keep their current values
        st.rd = ck.st.rd;
        st.prev_angle = ck.st.prev_angle; st.theta = ck.st.theta; st.keep_extra = ck.st.keep_extra; st.cfo = ck.st.cfo;
        st.psmi = ck.st.psmi; st.cfo_wait = ck.st.cfo_wait; st.bc = ck.st.bc; st.samperr = ck.st.samperr; st.angle = ck.st.angle;
        st.fine_epoch = ck.st.fine_epoch;
        st.stale.fir_pushed[MODE_AM] = ck.st.stale.fir_pushed[MODE_AM];
        am.pli = ck.am.pli; am.hppi = ck.am.hppi; am.aabi = ck.am.aabi; am.rdbi = ck.am.rdbi; am.offset_history = ck.am.offset_history;
        am.am_errors = ck.am.am_errors; am.am_diversity_wait = ck.am.am_diversity_wait;
        am.q_head = j == 7 ? (ck.am.q_head + 1) % 3 : ck.am.q_head;       // block 7's checkpoint predates its de-interleaver pass (k_am_interleave)
        am.frame_slot = ck.am.frame_slot; am.vit_parity = ck.am.vit_parity; am.next_job = ck.am.next_job;
        am.dec_bc = -1;
        st.sync_state = SYNC_NONE;                             // input_set_sync_state(NONE) at the end of block a
        BlockRecord &rec = ring[a % db.rec_cap];
        if (rec.state_after == SYNC_FINE) { rec.state_after = SYNC_NONE; rec.flags |= REC_LOST_SYNC; }
        st.ndiscard += n - a - 1;
        atomicAdd(&db.counters[3], 1);
        atomicAdd(&db.counters[0], 1);                         // there is work again
    }
}

void launch_rollback_am(const DevBuffers &db, int nstreams, const int *stream_ids, int cur_window, int min_age, hipStream_t st)
{
    hipLaunchKernelGGL(k_rollback_am, dim3(nstreams), dim3(256), 0, st, db, stream_ids, cur_window, min_age);
}

void launch_rollback(const DevBuffers &db, int nstreams, const int *stream_ids, int cur_window, int min_age, hipStream_t st)
{
    hipLaunchKernelGGL(k_rollback, dim3(nstreams), dim3(256), 0, st, db, stream_ids, cur_window, min_age);
}

}  // namespace nrsc5
