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
    if (tid < 31) st.fir_hist[tid] = ck.fir_hist[tid];
    if (tid == 0) {
        st.rd = ck.rd;
        st.prev_angle = ck.prev_angle; st.theta = ck.theta; st.keep_extra = ck.keep_extra; st.cfo = ck.cfo;
        st.psmi = ck.psmi; st.cfo_wait = ck.cfo_wait; st.bc = ck.bc; st.samperr = ck.samperr; st.angle = ck.angle;
        st.mer_cnt = ck.mer_cnt; st.error_lb = ck.error_lb; st.error_ub = ck.error_ub;
        st.started_pm = ck.started_pm; st.pm_slot = ck.pm_slot; st.last_pm_slot = ck.last_pm_slot;
        st.px_pos = ck.px_pos; st.px_ready = ck.px_ready; st.px_started = ck.px_started; st.px_go = 0;
        st.fine_epoch = ck.fine_epoch;
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

void launch_rollback(const DevBuffers &db, int nstreams, const int *stream_ids, int cur_window, int min_age, hipStream_t st)
{
    hipLaunchKernelGGL(k_rollback, dim3(nstreams), dim3(256), 0, st, db, stream_ids, cur_window, min_age);
}

}  // namespace nrsc5
