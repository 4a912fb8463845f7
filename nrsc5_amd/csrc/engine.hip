// Host side of libnrsc5hip: engine object, device-resident per-stream state, the block-step
// scheduler and the C ABI of include/nrsc5hip.h.  Mirrors the reference's src/input.c seam
// (input_push_cu8/cs16, input_reset, input_set_sync_state) -- see include/nrsc5hip.h for the map.
#include <array>
#include <atomic>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <deque>
#include <new>
#include <vector>
#include "nrsc5hip.h"
#include "kernels.h"

using namespace nrsc5;

// ONE block-step chain per engine.  Cutting the stream set into half-sets on two chain queues was built and measured in round 3
// (profiles/r03_chain_lanes.txt: 49 vs 38 ms per pass -- the chip is occupancy-bound inside k_mixfft, a second queue only splits
// the same slots) and in round 1 (stream groups on separate HIP streams: nothing gained); the scaffolding for it is gone.
static_assert(sizeof(nrsc5hip_record) == sizeof(BlockRecord), "record ABI mismatch");
static_assert(sizeof(BlockRecord) % 8 == 0, "record alignment");

static thread_local char g_err[512] = "";

// wall-clock totals of the fast streaming seam of the CALLING THREAD's sessions (nrsc5hip_debug_seam_totals): where a drop-in
// session's time goes.  Thread-local: sessions driven from different threads never share a counter.
static thread_local double g_seam[14];  // [0] s copying pushes into pinned staging, [1] s enqueueing H2D + decimator, [2] s enqueueing block steps,
                           // [3] s waiting for the device (the one sync per block), [4] pushes, [5] submissions, [6] block steps, [7] s in drain / frame fetches,
                           // [8] block steps whose wait was deferred, [9] read positions mispredicted, [10] steps without the P1 decode launches,
                           // [11] P1 decodes launched after the fact (the prediction said no frame could complete),
                           // [12] block steps submitted ahead of the previous block's delivery
struct SeamClock {
    int slot; std::chrono::steady_clock::time_point t0;
    explicit SeamClock(int s) : slot(s), t0(std::chrono::steady_clock::now()) {}
    ~SeamClock() { g_seam[slot] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
extern "C" void nrsc5hip_debug_seam_totals(double out[8], int reset)
{
    for (int k = 0; k < 8; k++) { if (out) out[k] = g_seam[k]; if (reset) g_seam[k] = 0; }
}
extern "C" void nrsc5hip_debug_seam_counts(double out[6], int reset)
{
    for (int k = 0; k < 6; k++) { if (out) out[k] = g_seam[8 + k]; if (reset) g_seam[8 + k] = 0; }
}
extern "C" const char *nrsc5hip_last_error(void) { return g_err; }
#ifndef NRSC5HIP_SOURCE_SHA
#define NRSC5HIP_SOURCE_SHA "unknown"
#endif
// the fingerprint behind a marker, so that a build can be identified from the FILE (nrsc5_amd.engine.check_fresh reads the bytes:
// a library that is already mapped into the process keeps answering for the old build after the file has been replaced)
static const char g_source_sha_marker[] = "NRSC5HIP_SOURCE_SHA=" NRSC5HIP_SOURCE_SHA;
extern "C" const char *nrsc5hip_source_sha(void) { return g_source_sha_marker + 20; }

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            snprintf(g_err, sizeof(g_err), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return NRSC5HIP_EHIP;                                                                      \
        }                                                                                              \
    } while (0)
#define FAIL(code, ...) do { snprintf(g_err, sizeof(g_err), __VA_ARGS__); return (code); } while (0)

// Every entry point runs on ITS ENGINE's device, whatever the calling thread's current device is (one process may own one engine
// per GPU, each driven by its own thread or all by one): the guard switches on entry and restores on exit.
struct DeviceGuard {
    int prev = -1, want = -1;
    explicit DeviceGuard(int dev) : want(dev) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != want) (void)hipSetDevice(want); }
    ~DeviceGuard() { if (prev >= 0 && prev != want) (void)hipSetDevice(prev); }
};
#define ON_ENGINE_DEVICE_FAST(e) DeviceGuard _device_guard((e) ? (e)->cfg.device : 0); if (!(e)) FAIL(NRSC5HIP_EINVAL, "null engine")
// ... and, for every entry but the fast streaming seam's own, with no block step in flight (deferred wait, see nrsc5hip_engine)
#define ON_ENGINE_DEVICE(e) ON_ENGINE_DEVICE_FAST(e); do { int _rc = settle(e); if (_rc) return _rc; } while (0)
struct nrsc5hip_engine;
static int settle(nrsc5hip_engine *e);

struct nrsc5hip_engine {
    nrsc5hip_config cfg;
    DevTables tb;
    DevBuffers db;
    // The block-step chain: its HIP stream, the decode streams of the window pipeline and their bookkeeping.
    struct Lane {
        hipStream_t main, aux[NAUX];
        hipEvent_t ev_window[NWIN], ev_decoded[NWIN];
        bool decoded_pending[NWIN];
        int lane_parity[NAUX];         // window slot of the last decode each decode stream was given (-1: none yet)
        bool thin;                     // the last burst advanced fewer than a quarter of the set's streams (the replaying stragglers' tail)
        bool acq_needed;               // some stream of the CURRENT stream set may be un-synchronised: launch the acquisition kernels
        bool px_needed;                // some stream is not FINE yet or runs a service mode with extended sidebands
        unsigned long long set_sig;    // identity of the stream set the two flags above were measured on (0 = none)
        int dec_waited;                // chunks of the current chunked append this lane has already waited for
        bool prepared_by_sync;         // the previous step's k_sync already ran the next block's bookkeeping
        long long step_count;          // block steps issued so far (decode-window bookkeeping in async mode)
        long long am_step_count;       // same for the AM window pipeline (8 steps per window)
        bool am_decoded_pending[NWIN];
        int *counters_dev, *counters_host;
        DevBuffers db;                 // engine buffers with this lane's counters
    } lane;
    int naux;                          // decode streams in use (<= NAUX)
    int naux_am;                       // ... by the AM window pipeline (2 measured best once the P3 frame decodes in segment waves)
    int verdict_lag;                   // test hook (nrsc5hip_debug_tune): replay takes verdicts this many windows late
    int am_segments, am_warm, am_runin;   // K=9 decode of the AM P3 frame: segment waves per frame (8), their forward warm-up / traceback run-in (test hooks: 0)
    int fwd_warm;                      // test hook: speculative warm-up trips of a forward segment (2; 0 makes every speculation fail -> repair path)
    int mixfft_syms;                   // symbols per k_mixfft workgroup (1, 2, 4, 8)
    int sync_lanes;                    // work-items per stream of k_sync: 0 = by the size of the stream set, 256, 768
    int fold_report;                   // 1 (default): fast seam, a step with nothing behind k_sync: k_sync posts the report (NRSC5HIP_TUNE_FOLD_REPORT = 0: k_stream_tail as a launch of its own)
    int fuse_seam_prepare;             // 1 (default): fast seam, FINE stream: no k_prepare launch (NRSC5HIP_TUNE_SEAM_PREPARE = 0: separate launch)
    int tb_walk;                       // > 0: single-path traceback (k_p1_tbwalk + check): 1 (default) = a workgroup per (frame, part), N > 1 = a persistent grid of N workgroups (opt-in); 0: the block-parallel one of round 3
    int fwd_segments;                  // waves per frame of the P1 forward pass; 0 = pick from the size of the stream set (fwd_segments_for)
    int flow_min;                      // dataflow bursts (k_flow, k_sync.hip): stream sets of at least this many streams (0 = never) run the steps of a burst in which every
                                       // stream is FINE as ONE launch
    unsigned *flow_dev; size_t flow_cap;   // its hand-off words (zeroed before every launch) and how many there are
    unsigned *flow_err;                    // two words of pinned host memory the kernel writes when a poll gives up
    long long flow_bursts, flow_steps;     // bursts / block steps issued that way since the engine was created
    hipStream_t main;                  // = lane.main
    std::vector<void *> allocs;
    // host mirrors
    std::vector<long long> wr_host, base_host;
    std::vector<int> drained;          // records already handed out per stream
    std::vector<int> mode_host;        // MODE_FM / MODE_AM per stream
    std::vector<long long> raw_host;   // AM cu8: raw input samples consumed (32:1 decimator phase)
    std::vector<char> attached;        // zero-copy batch: the stream reads the caller's capture (one append per reset)
    // Fast streaming seam (p1_async = 0): the host mirrors the stream's FIFO read position, so a push that cannot complete a
    // block costs one host memcpy into pinned memory, one async H2D and the K1 launch -- no synchronisation at all -- and a
    // push that does complete one ends with ONE sync, after a report kernel has posted the counters, the new read position and
    // the block's record straight into pinned host memory.
    // NSTAGE pinned staging buffers used round robin (a buffer is refilled NSTAGE submissions after it was handed to the device: with
    // two, and three submissions per block, the host waited ~30 us per block for the decimator of the submission before last)
    static constexpr int NSTAGE = 8;
    uint8_t *stage_pin[NSTAGE], *stage_pin_dev[NSTAGE], *stage_dev2[NSTAGE]; hipEvent_t stage_ev[NSTAGE]; bool stage_busy[NSTAGE]; int stage_slot;
    unsigned *decim_ticket;            // k_decimate_fm_cu8_stream: workgroups of the running launch that have finished
    // Ingest stream (round 4): the direct decimator runs on its own HIP stream, beside the block step on `main` (which keeps ONE CU
    // busy): chunks are submitted as they fill (early_flush bytes), so that when the push that completes a block arrives only the
    // remainder is left to decimate and nothing of it sits on the step chain.  Order between the two streams: a step waits for the
    // ingest work submitted before it (ev_ingest); a FIFO compaction on the ingest stream waits for the steps submitted before it
    // (ev_main: it needs the final read position); anything else that touches the stream synchronises both (settle).
    hipStream_t ingest; hipEvent_t ev_ingest, ev_main, ev_appended;
    bool ingest_dirty;                 // work on the ingest stream that `main` has not been ordered behind yet
    bool main_stepped;                 // block steps on `main` that the ingest stream has not been ordered behind yet
    bool main_appended;                // FIFO appends on `main` (a block's last chunk) that the ingest stream has not been ordered behind yet
    size_t early_flush;                // staged bytes at which a chunk is submitted before its block is complete (0: never)
    // samples accepted by a push but not submitted yet: they wait in stage_pin[stage_slot] until the mirror says a block completes
    // (or the buffer is full, or anything else looks at the stream) -- one H2D + one decimator launch per BLOCK, not per push
    int staged_stream; size_t staged_bytes; bool staged_cu8; long long staged_q15;
    StreamReport *report_host[2], *report_dev[2];   // pinned, device-mapped reports: the step with sequence number q posts into [q & 1]
    std::vector<long long> rd_host;            // FIFO read position (absolute decimated samples) as of the last report
    std::vector<int> fetched;                  // records of the stream copied to `pending` so far (absolute index)
    std::vector<char> mirror_ok;               // rd_host / pending are exact: only the streaming seam touched the stream since its reset
    std::vector<std::deque<BlockRecord>> pending;   // records reported but not yet drained
    // Deferred wait (round 4).  A block that starts in FINE consumes a number of samples the host can compute in advance
    // (keep = 2160 - the timing feedback of the previous block, acquire.c:112,259; both are in that block's record), so the mirror
    // is advanced at SUBMISSION and the wait for the step's report moves to the next call that needs its results: the device works
    // on block n while the host copies the pushes of block n + 1 into staging.  At most one step per engine is in flight.
    int inflight_stream;               // stream whose block step is submitted but not harvested (-1: none)
    unsigned report_seq;               // sequence number the most recently launched report kernel posts when it is done
    long long inflight_rd_pred;        // the read position predicted for the step in flight (-1: no prediction, the mirror waits)
    bool inflight_decoded;             // the step in flight carried the P1 de-interleave / trellis / traceback launches
    unsigned inflight_seq;             // its report's sequence number
    // A second step, submitted AHEAD of the delivery of the one in flight (nrsc5hip_stream_step_ahead): allowed when the block in
    // flight starts FINE and cannot complete a P1 frame -- nothing its delivery tells the host can change what the next block does
    // (frame.c's only way back into L1 is the first header of a P1 frame, frame.c:535-540) -- so the device runs block n + 1 while
    // the host still hands block n to L2.  Its read-position prediction needs block n's record: it is made when that is harvested.
    struct Ahead { bool valid; int stream; unsigned seq; bool decoded; } ahead;
    bool inflight_progress;            // the last harvested step processed (or left pending) a block
    bool direct_decimate;              // 1 (default): FM cu8 pushes are decimated straight from the pinned staging buffer; 0 (NRSC5HIP_TUNE_DIRECT_DECIMATE): H2D copy first
    bool defer_wait;                   // 1 (default): predictable steps stay in flight; 0 (NRSC5HIP_TUNE_DEFER_WAIT): every step is waited for at once
    bool counters_clean;               // the step counters are zero: the last kernel that touched them was a report kernel
    std::vector<char> pred_ok;         // the stream's last harvested record left it FINE and nothing else touched it since
    std::vector<int> pred_samperr, pred_bc;    // ... that record's next_samperr and block count
    std::vector<char> manual_step;     // nrsc5hip_stream_set_manual_step: pushes stage and submit samples, the caller steps
    // Host-resident capture (round 6; fast seam, FM cu8, NRSC5HIP_TUNE_HOST_CAPTURE): the pushes of ONE stream of the engine are kept as they arrive in a pinned,
    // device-mapped buffer and the stream reads them in place -- StreamState::raw points into it, and the symbol kernel / the acquisition run the half-band on what they
    // read (halfband_raw.h): the zero-copy batch's kernels, fed across PCIe.  A push is one host memcpy: no decimator launch, no ingest stream, nothing in front of the
    // block step.  The stream's own byte numbering: HC_PREFIX bytes of decimator history (what its reset left in hb_hist), then every byte pushed since that reset;
    // decimated sample a = dword a of that numbering, so the stream's counters start at HC_OFF.  The buffer is linear: when it is full the live tail moves to its
    // front and `raw` moves with it (hc_rebase).  Anything the capture cannot express (a cs16 push, the batch entry points) first turns it back into the FIFO (hc_detach).
    static constexpr long long HC_OFF = 8, HC_PREFIX = 4 * HC_OFF, HC_KEEP = 16384;
    uint8_t *hc_pin, *hc_dev; size_t hc_cap;
    int hc_stream;                     // the stream bound to the buffer, -1: none
    long long hc_abs0, hc_wr;          // byte index (stream numbering) of hc_pin[0] / of the next byte to be written
    bool host_capture;                 // knob (default on where the buffer exists)
    long long hc_rebases, hc_attaches, hc_detaches;
    long long reports_folded;          // block steps whose report the sync kernel posted itself (fold_report)
    std::vector<std::array<c16, 14>> hb_hist_host;   // the decimator history each stream's last reset left on the device (zeros for a fresh session)
    // staging
    uint8_t *stage_dev; size_t stage_bytes;
    size_t stage_ring_bytes;           // size of each of the NSTAGE staging buffers of the fast seam (a block of either mode fits)
    int *ids_dev; unsigned *nbytes_dev;
    int *all_ids_dev;                  // identity list 0..S-1
    // chunked K1 running ahead of the block steps on its own stream (fresh batches in the async pipeline)
    hipStream_t dec_stream;
    std::vector<hipEvent_t> dec_events;    // dec_events[c] fires when output samples [0, (c+1)*dec_chunk) of every stream are committed
    long long dec_chunk;                   // output samples per chunk, 0 = no chunked append outstanding
    unsigned *chunk_nbytes_dev; int chunk_cap;
    // engine-owned pinned result buffers for nrsc5hip_batch_fetch_view (allocated on first use)
    BlockRecord *rec_host; uint32_t *frames_host; int *nblocks_host;
    // optional per-kernel-class timing with HIP events on the launching stream
    bool prof_on;
    int prof_only;                     // -1: every class is timed; else only this one (events cost ~5 us of the chain's time per kernel)
    struct ProfSpan { int cls; hipEvent_t a, b; };
    std::vector<ProfSpan> prof_spans;
    std::vector<hipEvent_t> prof_pool;
    double prof_ms[NRSC5HIP_PROF_CLASSES];
    long long prof_launches[NRSC5HIP_PROF_CLASSES];
    VitScratch vit_scratch;            // scratch of the nrsc5hip_stage_viterbi_* entry points (per engine: nothing process-global)
};

static hipEvent_t prof_event(nrsc5hip_engine *e)
{
    if (!e->prof_pool.empty()) { hipEvent_t ev = e->prof_pool.back(); e->prof_pool.pop_back(); return ev; }
    hipEvent_t ev = nullptr; (void)hipEventCreate(&ev); return ev;
}
struct ProfScope {
    nrsc5hip_engine *e; int cls; hipStream_t st; hipEvent_t a;
    ProfScope(nrsc5hip_engine *e_, int cls_, hipStream_t st_) : e(e_), cls(cls_), st(st_), a(nullptr)
    { if (e->prof_on && (e->prof_only < 0 || e->prof_only == cls)) { a = prof_event(e); (void)hipEventRecord(a, st); } }
    ~ProfScope()
    { if (a) { hipEvent_t b = prof_event(e); (void)hipEventRecord(b, st); e->prof_spans.push_back({cls, a, b}); } }
};
static void prof_collect(nrsc5hip_engine *e)
{
    // caller has synchronised both streams
    for (auto &sp : e->prof_spans) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) { e->prof_ms[sp.cls] += ms; e->prof_launches[sp.cls]++; }
        e->prof_pool.push_back(sp.a); e->prof_pool.push_back(sp.b);
    }
    e->prof_spans.clear();
}

template <typename T> static int dev_alloc(nrsc5hip_engine *e, T **p, size_t count)
{
    void *q = nullptr;
    hipError_t err = hipMalloc(&q, count * sizeof(T) ? count * sizeof(T) : 1);
    if (err != hipSuccess) { snprintf(g_err, sizeof(g_err), "hipMalloc(%zu bytes) failed: %s", count * sizeof(T), hipGetErrorString(err)); return NRSC5HIP_ENOMEM; }
    e->allocs.push_back(q);
    *p = (T *)q;
    return 0;
}
template <typename T> static int dev_upload(nrsc5hip_engine *e, const T **p, const std::vector<T> &v)
{
    T *d; int rc = dev_alloc(e, &d, v.size()); if (rc) return rc;
    HIPCHK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *p = d;
    return 0;
}

// ---- read-only tables --------------------------------------------------------------------------
static int build_tables(nrsc5hip_engine *e)
{
    static const int8_t PM_V[20] = { 10, 2, 18, 6, 14, 8, 16, 0, 12, 4, 11, 3, 19, 7, 15, 9, 17, 1, 13, 5 };   // decode.c:34-37
    // interleaver II for PIDS (decode.c:324-342 with b=200, I0=365440): index inside block bc
    std::vector<uint16_t> pids(16 * PIDS_CODED);
    for (unsigned bc = 0; bc < 16; bc++)
        for (unsigned n = 0; n < (unsigned)PIDS_CODED; n++) {
            const unsigned i = bc * PIDS_CODED + n, part = PM_V[i % 20];
            const unsigned k = ((i / 20) % (PIDS_CODED / 20)) + P1_CODED / (20 * 16);
            const unsigned row = (11 * k) % 32, col = (11 * k + k / (32 * 9)) % 36;
            pids[i] = (uint16_t)(row * 720 + part * 36 + col);
        }
    // descrambler stream (decode.c:279-294), packed LSB-first
    std::vector<uint32_t> scr(P1_WORDS, 0), scr_pids(3, 0);
    {
        unsigned val = 0x3ff;
        for (int i = 0; i < P1_LEN; i++) {
            const unsigned bit = ((val >> 9) ^ val) & 1;
            val |= bit << 11; val >>= 1;
            scr[i >> 5] |= bit << (i & 31);
            if (i < PIDS_LEN) scr_pids[i >> 5] |= bit << (i & 31);
        }
    }
    std::vector<float2> tw(FFT_N);
    for (int k = 0; k < FFT_N; k++) {
        const double a = -2.0 * M_PI * k / FFT_N;
        tw[k].x = (float)cos(a); tw[k].y = (float)sin(a);
    }
    std::vector<float> shape(SYM_N);                           // acquire.c:322-331
    for (int i = 0; i < SYM_N; i++) {
        if (i < CP_N) shape[i] = sinf(M_PI / 2 * i / CP_N);
        else if (i < FFT_N) shape[i] = 1;
        else shape[i] = cosf(M_PI / 2 * (i - FFT_N) / CP_N);
    }
    // Q15 taps: (int16)(tap * 32767.0f) as firdecim_q15_create does (firdecim_q15.c:37-42)
    static const float hb_taps[4] = { 0.6062333583831787f, -0.13481467962265015f, 0.032919470220804214f, -0.00410953676328063f };   // input.c:35-40
    static const float acq_taps[32] = {                        // acquire.c:28-61
        -0.000685643230099231f, 0.005636964458972216f, 0.009015781804919243f, -0.015486305579543114f,
        -0.035108357667922974f, 0.017446253448724747f, 0.08155813068151474f, 0.007995186373591423f,
        -0.13311293721199036f, -0.0727422907948494f, 0.15914097428321838f, 0.16498781740665436f,
        -0.1324498951435089f, -0.2484012246131897f, 0.051773931831121445f, 0.2821577787399292f,
        0.051773931831121445f, -0.2484012246131897f, -0.1324498951435089f, 0.16498781740665436f,
        0.15914097428321838f, -0.0727422907948494f, -0.13311293721199036f, 0.007995186373591423f,
        0.08155813068151474f, 0.017446253448724747f, -0.035108357667922974f, -0.015486305579543114f,
        0.009015781804919243f, 0.005636964458972216f, -0.000685643230099231f, 0.0f };
    std::vector<int16_t> hbq(4), acq(17, 0);
    for (int i = 0; i < 4; i++) hbq[i] = (int16_t)(hb_taps[3 - i] * 32767.0f);
    for (int i = 1; i <= 16; i++) acq[i] = (int16_t)(acq_taps[31 - i] * 32767.0f);

    int rc;
    if ((rc = dev_upload(e, &e->tb.pids_gather, pids))) return rc;
    {
        // MP1 equaliser: every data cell's operands, so that k_sync neither divides nor takes remainders per cell
        const int ncell = 2 * PM_PART * NSYM * 18;
        std::vector<uint32_t> cell(ncell);
        std::vector<uint16_t> outp(ncell);
        for (int c = 0; c < ncell; c++) {
            const int k = 1 + c % 18, n = (c / 18) % NSYM, part = (c / (18 * NSYM)) % PM_PART, side = c / (18 * NSYM * PM_PART);
            const int r_lo = side ? 2 * (part + 1) + 1 : 2 * part, r_hi = side ? 2 * part + 1 : 2 * (part + 1);
            const int ref_lo_bin = (r_lo & 1) ? UB1 - PW * (r_lo >> 1) : LB0 + PW * (r_lo >> 1);
            const int live = bin_to_live(ref_lo_bin + k);
            cell[c] = (uint32_t)live | (uint32_t)n << 10 | (uint32_t)r_lo << 15 | (uint32_t)r_hi << 20 | (uint32_t)k << 25 | (uint32_t)side << 30;
            outp[c] = (uint16_t)(n * 720 + (side ? 19 - part : part) * 36 + (k - 1) * 2);
        }
        if ((rc = dev_upload(e, &e->tb.eq_cell, cell))) return rc;
        if ((rc = dev_upload(e, &e->tb.eq_out, outp))) return rc;
    }
    {   // byte q of the 384-byte run of one k: q%6==5 is the erasure, else j = q - q/6, part = PM_V[j%20], block = (j/20 + 7 part) % 16
        std::vector<uint16_t> lut(384);
        for (int q = 0; q < 384; q++) {
            if (q % 6 == 5) { lut[q] = 0xffff; continue; }
            const int j = q - q / 6, part = PM_V[j % 20], block = (j / 20 + 7 * part) % 16;
            lut[q] = (uint16_t)(block * 720 + part * 36);
        }
        if ((rc = dev_upload(e, &e->tb.deint_lut, lut))) return rc;
    }
    if ((rc = dev_upload(e, &e->tb.scr_p1, scr))) return rc;
    if ((rc = dev_upload(e, &e->tb.scr_pids, scr_pids))) return rc;
    if ((rc = dev_upload(e, &e->tb.twiddle, tw))) return rc;
    {   // k_mixfft's first exchange: work-item r multiplies its k1-th output by W2048^(k1 r) -- from the plain table a gather with stride
        // k1 (up to 28 cache lines per wave and load), from this copy 256 consecutive entries per k1
        std::vector<float2> twa(7 * 256);
        for (int k1 = 1; k1 < 8; k1++) for (int r = 0; r < 256; r++) twa[(k1 - 1) * 256 + r] = tw[(k1 * r) & 2047];
        if ((rc = dev_upload(e, &e->tb.twiddle_a, twa))) return rc;
    }
    if ((rc = dev_upload(e, &e->tb.shape, shape))) return rc;
    if ((rc = dev_upload(e, &e->tb.hb_q15, hbq))) return rc;
    if ((rc = dev_upload(e, &e->tb.acq_q15, acq))) return rc;
    for (int wide = 0; wide < 2; wide++) {
        // interleaver IV (decode.c:344-376) is convolutional: position i of a block pair reads what was written
        // delay[i] positions earlier (1..N).  Same arithmetic as the reference's loop, for the first pair of a cycle.
        const unsigned L = wide ? 4608 : 2304, J = wide ? 4 : 2, C = 36, M = wide ? 2 : 4, N = 32 * L;
        const unsigned bk_bits = 32 * C, bk_adj = 32 * C - 1;
        std::vector<uint32_t> delay(2 * L);
        unsigned taken[4] = { 0, 0, 0, 0 };
        for (unsigned g = 0; g < 2 * L; g++) {
            const unsigned part = ((g + 2 * (M / 4)) / M) % J;
            const unsigned pti = taken[part]++;
            const unsigned block = (pti + (part * 7) - (bk_adj * (pti / bk_bits))) % 32;
            const unsigned row = ((11 * pti) % bk_bits) / C, col = (pti * 11) % C;
            const unsigned rp = (block * 32 + row) * (J * C) + part * C + col;
            const unsigned d = (g + N - rp) % N;
            delay[g] = d ? d : N;
        }
        if ((rc = dev_upload(e, wide ? &e->tb.px_delay_wide : &e->tb.px_delay_narrow, delay))) return rc;
    }
    {   // interleaver_ma1 (decode.c:66-231) folded into one table per code word: for every depunctured trellis input, which
        // bit of which hard-symbol matrix it is (bit_map), whether it passes the 3-frame diversity delay line, or a
        // punctured zero.  Same arithmetic as the reference's loops, evaluated once.
        static const int src12[12] = { 3, 0, 0, 3, 3, 0, 1, 1, 2, 2, 2, 1 };      // position in a 12-bit group -> bl / ml / bu / mu (decode.c:26-30)
        static const int j12[12] = { 2, 1, 0, 1, 0, 2, 1, 2, 1, 2, 0, 0 };
        static const int rank15[15] = { 0, -1, 1, 2, -1, 3, 4, -1, 5, 6, 7, 8, 9, 10, 11 };   // E1 puncture {1,0,1,1,0,1,1,0,1,1,1,1,1,1,1}
        auto cell_of = [](int b, int k) { const int col = (9 * k) % 25, row = (11 * col + 16 * (k / 25) + 11 * (k / 50)) % 32; return 25 * (b * 32 + row) + col; };
        // (the delay-line cell of a delayed input is keyed by the input's own index -- DevBuffers::am_q -- so the reference's queue /
        // position arguments only document which line it is)
        auto entry = [](int cell, int bit, int matrix, int delayed, int /*queue*/, int /*n*/) {
            return (uint32_t)((unsigned)cell | ((unsigned)bit << 13) | ((unsigned)matrix << 16) | (delayed ? AMT_DELAYED : 0u)); };
        std::vector<uint32_t> t1(AM_VIT), t3b(AM_VIT), t3a(3 * AM_P3_LEN_MA1);
        for (int i = 0; i < AM_VIT; i++) {
            const int rk = rank15[i % 15];
            if (rk < 0) { t1[i] = AMT_PUNCT; t3b[i] = t1[i]; continue; }
            const int o = (i / 15) * 12 + rk, g = o / 12, pos = o % 12, n = g * 3 + j12[pos];
            switch (src12[pos]) {       // matrices: 0 pl, 1 pu, 2 s, 3 t
            case 0: t1[i] = entry(cell_of(n / 2250, (n + n / 750 + 1) % 750), n % 3, 0, 0, 0, 0);                 // bl
                    t3b[i] = entry(cell_of((3 * n + 3) % 8, (n + n / 3000 + 3) % 750), n % 3, 3, 0, 0, 0); break;   // ebl
            case 1: t1[i] = entry(cell_of((3 * n + 3) % 8, (n + n / 3000 + 3) % 750), 3 + n % 3, 0, 1, 0, n);      // ml
                    t3b[i] = entry(cell_of((3 * n + 3) % 8, (n + n / 3000 + 3) % 750), 3 + n % 3, 3, 1, 2, n); break;   // eml
            case 2: t1[i] = entry(cell_of(n / 2250, (n + n / 750) % 750), n % 3, 1, 0, 0, 0);                     // bu
                    t3b[i] = entry(cell_of((3 * n) % 8, (n + n / 3000 + 2) % 750), n % 3, 2, 0, 0, 0); break;     // ebu
            default: t1[i] = entry(cell_of((3 * n) % 8, (n + n / 3000 + 2) % 750), 3 + n % 3, 1, 1, 1, n);         // mu
                    t3b[i] = entry(cell_of((3 * n) % 8, (n + n / 3000 + 2) % 750), 3 + n % 3, 2, 1, 3, n); break;  // emu
            }
        }
        for (int i = 0; i < 3 * AM_P3_LEN_MA1; i++) {             // E2 puncture {1,0,1,1,0,0}; 6-bit groups: el {0,1}, eu {2,3,5,4}
            const int r6 = i % 6;
            if (!(r6 == 0 || r6 == 2 || r6 == 3)) { t3a[i] = AMT_PUNCT; continue; }
            const int o = (i / 6) * 3 + (r6 == 0 ? 0 : r6 - 1), g = o / 6, pos = o % 6;
            if (pos < 2) { const int n = g * 2 + pos; t3a[i] = entry(cell_of((3 * n + n / 3000) % 8, (n + n / 6000) % 750), n % 2, 3, 0, 0, 0); }
            else {
                const int j = pos == 2 ? 0 : pos == 3 ? 1 : pos == 5 ? 2 : 3, n = g * 4 + j;
                t3a[i] = entry(cell_of((3 * n + n / 3000 + 2 * (n / 12000)) % 8, (n + n / 6000) % 750), n % 4, 2, 0, 0, 0);
            }
        }
        if ((rc = dev_upload(e, &e->tb.am_deint_p1, t1))) return rc;
        if ((rc = dev_upload(e, &e->tb.am_deint_p3_ma3, t3b))) return rc;
        if ((rc = dev_upload(e, &e->tb.am_deint_p3_ma1, t3a))) return rc;
    }
    {   // AM tables: acquisition FIR (acquire.c:63-96), pulse shape (acquire.c:333-342), 256-point twiddles
        static const float am_taps[32] = {
            -0.00038464731187559664f, -0.00021618751634377986f, 0.0026779419276863337f, -0.00029802651260979474f,
            -0.0012626448879018426f, -0.0013182522961869836f, -0.012252614833414555f, 0.015980124473571777f,
            0.037112727761268616f, -0.05451361835002899f, -0.05804193392395973f, 0.11320608854293823f,
            0.055298302322626114f, -0.16878043115139008f, -0.022917453199625015f, 0.19178225100040436f,
            -0.022917453199625015f, -0.16878043115139008f, 0.055298302322626114f, 0.11320608854293823f,
            -0.05804193392395973f, -0.05451361835002899f, 0.037112727761268616f, 0.015980124473571777f,
            -0.012252614833414555f, -0.0013182522961869836f, -0.0012626448879018426f, -0.00029802651260979474f,
            0.0026779419276863337f, -0.00021618751634377986f, -0.00038464731187559664f, 0.0f };
        std::vector<int16_t> amq(17, 0);
        for (int i = 1; i <= 16; i++) amq[i] = (int16_t)(am_taps[31 - i] * 32767.0f);
        std::vector<float> ashape(AM_SYM);
        for (int i = 0; i < AM_SYM; i++) {
            if (i < AM_CP) ashape[i] = sinf(M_PI / 2 * i / AM_CP);
            else if (i < AM_FFT) ashape[i] = 1;
            else ashape[i] = cosf(M_PI / 2 * (i - AM_FFT) / AM_CP);
        }
        std::vector<float2> atw(AM_FFT);
        for (int k = 0; k < AM_FFT; k++) { const double a = -2.0 * M_PI * k / AM_FFT; atw[k].x = (float)cos(a); atw[k].y = (float)sin(a); }
        if ((rc = dev_upload(e, &e->tb.am_acq_q15, amq))) return rc;
        if ((rc = dev_upload(e, &e->tb.am_shape, ashape))) return rc;
        if ((rc = dev_upload(e, &e->tb.am_twiddle, atw))) return rc;
    }
    return 0;
}

static void init_state(StreamState &st, int mode = MODE_FM)
{
    memset(&st, 0, sizeof(st));
    st.psmi = 1;                                               // sync_reset (sync.c:821)
    st.sync_state = SYNC_NONE;
    st.mode = mode;
    st.nco_re = 1.0f; st.nco_im = 0.0f; st.nco_exact = 1;     // acquire_reset: phase = 1 (acquire.c:296) -- from here on the float state can be kept bit for bit
}

static void init_am_state(AmStream &am)
{
    memset(&am, 0, sizeof(am));
    am.pli = am.hppi = am.aabi = am.rdbi = -1;                 // sync_reset (sync.c:822-825)
    am.am_diversity_wait = 4;                                  // decode_reset (decode.c:568)
    am.dec_bc = -1;
}

extern "C" int nrsc5hip_engine_create(const nrsc5hip_config *cfg, nrsc5hip_engine **out)
{
    if (!cfg || !out) FAIL(NRSC5HIP_EINVAL, "null argument");
    *out = nullptr;
    if (cfg->max_streams < 1 || cfg->q15_capacity < 2 * WIN_N || cfg->record_capacity < 64 || cfg->p1_slots < 2)
        FAIL(NRSC5HIP_EINVAL, "bad config (max_streams>=1, q15_capacity>=%d, record_capacity>=64, p1_slots>=2)", 2 * WIN_N);
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev) FAIL(NRSC5HIP_EINVAL, "device %d out of range (%d devices)", cfg->device, ndev);
    DeviceGuard guard(cfg->device);                            // the caller's current device is restored on return
    nrsc5hip_engine *e = new (std::nothrow) nrsc5hip_engine();
    if (!e) FAIL(NRSC5HIP_ENOMEM, "out of host memory");
    e->cfg = *cfg;
    const size_t S = cfg->max_streams;
    if (cfg->p1_async) {
        // the window pipeline drives 1 chain + 3 decode streams (+ the caller's): with the HIP runtime's default of 4 hardware
        // queues they share queues and the decode / chain overlap is lost silently
        const char *q = getenv("GPU_MAX_HW_QUEUES");
        static std::atomic<bool> warned{false};                // engines of one process share nothing else; this is a once-per-process notice
        if ((!q || atoi(q) < 8) && !warned.exchange(true)) {
            fprintf(stderr, "libnrsc5hip: warning: p1_async engine with GPU_MAX_HW_QUEUES=%s (< 8): decode streams will share hardware queues with "
                            "the block-step chain; export GPU_MAX_HW_QUEUES=8 before the HIP runtime initialises\n", q ? q : "unset (default 4)");
        }
    }
    int rc = 0;
    do {
        {
            // decode streams in use; nrsc5hip_debug_tune changes them.  FM: ONE since round 5 (three in rounds 2 - 4, profiles/r02_naux.txt): with the segmented forward pass
            // and the single-path traceback a window's decode (~1.6 ms) fits the 16 block steps of the next window (~1.7 ms) on one queue, and with two or three the
            // forward pass of one window overlaps the traceback of another -- the k_sync launch that meets both lasts 370 - 790 us instead of 37 (one per window; tools/gpu_trace_sync.sh,
            // profiles/r05_trace_sync_decode_streams.txt): 30.1 -> 29.2 ms per pass.  AM: three, with four segment waves per P3 frame (round 5, on the
            // rewritten block step: 70.4 ms with two streams x eight segments, 67.5 with three x eight, 65.7 with three x four, 91.3 with one: the K=9 decodes are ~80 ms of kernel time per pass)
            e->naux = 1; e->naux_am = 3;
            e->verdict_lag = 0; e->fwd_segments = 0; e->fwd_warm = 2; e->mixfft_syms = 1; e->sync_lanes = 0; e->flow_min = 0; e->tb_walk = 1; e->fuse_seam_prepare = 1; e->fold_report = 1;      // measured: profiles/r04_mixfft_persistent.txt
            e->am_segments = 4; e->am_warm = K9_WARM; e->am_runin = K9_TB_RUNIN;
        }
        {
            nrsc5hip_engine::Lane &ln = e->lane;
            // (queue priorities -- chain stream high, decode streams low -- were measured: nothing for the batch, +15 % per block for
            // a lone stream, profiles/r02_ab_prio_demod.txt)
            if (hipStreamCreate(&ln.main) != hipSuccess) rc = NRSC5HIP_EHIP;
            for (int k = 0; k < NAUX && !rc; k++) if (hipStreamCreate(&ln.aux[k]) != hipSuccess) rc = NRSC5HIP_EHIP;
            for (int k = 0; k < NWIN && !rc; k++) {
                if (hipEventCreate(&ln.ev_window[k]) != hipSuccess || hipEventCreate(&ln.ev_decoded[k]) != hipSuccess) rc = NRSC5HIP_EHIP;
                ln.decoded_pending[k] = false;
            }
            ln.acq_needed = true; ln.px_needed = true; ln.set_sig = 0; ln.step_count = 0; ln.am_step_count = 0;
            for (int k = 0; k < NAUX; k++) ln.lane_parity[k] = -1;
            ln.thin = false;
            for (int k = 0; k < NWIN; k++) ln.am_decoded_pending[k] = false;
            if (!rc && hipHostMalloc((void **)&ln.counters_host, 4 * sizeof(int), hipHostMallocDefault) != hipSuccess) rc = NRSC5HIP_ENOMEM;
        }
        if (rc) { snprintf(g_err, sizeof(g_err), "stream/event creation failed"); break; }
        e->main = e->lane.main;
        // K1 of the copying batch path runs ahead of the block steps on this stream (confining it to a slice of the CUs
        // with hipExtStreamCreateWithCUMask was measured: no gain, profiles/r02_k1cus.txt -- it is the HBM traffic itself that
        // slows the latency-bound step kernels; the zero-copy batch path has no K1 at all)
        if (hipStreamCreate(&e->dec_stream) != hipSuccess) { rc = NRSC5HIP_EHIP; snprintf(g_err, sizeof(g_err), "hipStreamCreate failed"); break; }
        e->dec_chunk = 0; e->chunk_nbytes_dev = nullptr; e->chunk_cap = 0;
        if ((rc = build_tables(e))) break;
        DevBuffers &db = e->db;
        db.q15_cap = cfg->q15_capacity; db.p1_slots = cfg->p1_slots; db.rec_cap = cfg->record_capacity;
        if ((rc = dev_alloc(e, &db.state, S))) break;
        db.ckpt = nullptr;
        if (cfg->p1_async && cfg->l2_feedback) {
            if (cfg->record_capacity < 2 * NWIN * 16) { rc = NRSC5HIP_EINVAL; snprintf(g_err, sizeof(g_err), "p1_async with l2_feedback needs record_capacity >= %d (speculated blocks keep their records)", 2 * NWIN * 16); break; }
            if ((rc = dev_alloc(e, &db.ckpt, S * NWIN))) break;
        }
        if ((rc = dev_alloc(e, &db.q15, S * (size_t)db.q15_cap))) break;
        db.acq_win = nullptr;
        if ((cfg->batch_zero_copy || !cfg->p1_async) && (rc = dev_alloc(e, &db.acq_win, S * WIN_N))) break;   // (fast-seam engines: the host-resident capture reads in place too)
        if ((rc = dev_alloc(e, &db.acq_filt, S * WIN_N))) break;
        if ((rc = dev_alloc(e, &db.acq_list, S + 1))) break;
        if ((rc = dev_alloc(e, &db.acq_sums, S * SYM_N))) break;
        if ((rc = dev_alloc(e, &db.bins, S * NSYM * LIVE_N))) break;
        // the reference's oscillator sample by sample for blocks in exact mode (553 KB per stream; k_nco_exact -> k_mixfft)
        if ((rc = dev_alloc(e, &db.nco_tab, S * NSYM * SYM_N))) break;
        if ((rc = dev_alloc(e, &db.cfo_snap, S * LIVE_N * (PM_PART + 1)))) break;
        if ((rc = dev_alloc(e, &db.cfo_phase, S * NSYM * LIVE_N))) break;     // 68 KB per stream: phases[][] of the exact CFO search's visit in progress
        e->flow_cap = flow_words((int)S); e->flow_bursts = e->flow_steps = 0;
        if ((rc = dev_alloc(e, &e->flow_dev, e->flow_cap))) break;
        if (hipHostMalloc((void **)&e->flow_err, 2 * sizeof(unsigned), hipHostMallocDefault) != hipSuccess) { rc = NRSC5HIP_ENOMEM; break; }
        e->flow_err[0] = e->flow_err[1] = 0;
        db.loop_exact = 1;                                     // the reference's own loop arithmetic in blocks that start un-synchronised (k_sync.hip; NRSC5HIP_TUNE_LOOP_EXACT)
        // Default (round 6): a freshly reset stream's FIRST block -- the block its CFO search runs on -- advances the oscillator by the reference's own float recurrence
        // (k_nco_exact), every later block by the closed-form phasor with the recurrence's amplitude ramp.  Measured on the MI355X with the loop arithmetic of k_sync on the
        // reference's own operations (loop_exact below): 0 of 722 locks through the CFO search deviate in any field (768 / 768 streams strict), against 5 failing + 8 counted
        // streams with the closed form in that block (profiles/r06_nco_policy_decision.txt); cost 1.8 ms of a 30.4 ms pass.  (Round 5, with the fast loop arithmetic, had
        // seen no effect of the policy on the device and defaulted to NCO_CLOSED_FORM: both halves are needed.)  nrsc5hip_debug_tune(NRSC5HIP_TUNE_NCO_EXACT) changes it.
        db.nco_policy = NCO_EXACT_FIRST_BLOCK;
        if ((rc = dev_alloc(e, &db.pm, S * NPM * PM_FRAME))) break;
        db.nstreams_alloc = (int)S;
        if ((rc = dev_alloc(e, &db.coded, (size_t)(cfg->p1_async ? NAUX : 1) * S * P1_LEN))) break;
        if ((rc = dev_alloc(e, &db.dec, (size_t)(cfg->p1_async ? NAUX : 1) * S * (size_t)(2 * (P1_LEN + 64))))) break;
        if ((rc = dev_alloc(e, &db.tbmap, (size_t)(cfg->p1_async ? NAUX : 1) * S * (size_t)(P1_LEN / 64 + 1) * 64))) break;
        if ((rc = dev_alloc(e, &db.fwd_meta, (size_t)(cfg->p1_async ? NAUX : 1) * S * (size_t)(VIT3_GMAX * VIT3_META)))) break;
        if ((rc = dev_alloc(e, &db.fwd_stats, 4))) break;
        if (hipMemset(db.fwd_stats, 0, 4 * sizeof(int)) != hipSuccess) { rc = NRSC5HIP_EHIP; break; }
        db.tb_stats = db.fwd_stats + 2;
        if ((rc = dev_alloc(e, &db.am_k9stats, 4))) break;
        if (hipMemset(db.am_k9stats, 0, 4 * sizeof(unsigned)) != hipSuccess) { rc = NRSC5HIP_EHIP; break; }
        if ((rc = dev_alloc(e, &db.pids_stage, S * NWIN * 16 * 3 * PIDS_LEN))) break;
        if ((rc = dev_alloc(e, &db.pids_rec, S * NWIN * 16))) break;
        if (hipMemset(db.pids_rec, 0xff, S * NWIN * 16 * sizeof(int)) != hipSuccess) { rc = NRSC5HIP_EHIP; break; }
        if ((rc = dev_alloc(e, &db.p1_ring, S * db.p1_slots * P1_WORDS))) break;
        db.p1_mirror = nullptr;
        if ((rc = dev_alloc(e, &db.records, S * db.rec_cap))) break;
        if ((rc = dev_alloc(e, &db.counters, 4))) break;
        {
            const size_t nax = cfg->p1_async ? NAUX : 1;
            db.px_slots = 8 * cfg->p1_slots;
            if ((rc = dev_alloc(e, &db.px_mem, S * 2 * PX_MEM))) break;
            if ((rc = dev_alloc(e, &db.px_pair, S * 4 * PX_MAX))) break;
            if ((rc = dev_alloc(e, &db.px_stage, S * NWIN * 16 * (size_t)PX_DEPUNCT))) break;
            if ((rc = dev_alloc(e, &db.px_job, S * NWIN * 16))) break;
            if ((rc = dev_alloc(e, &db.px_dec, nax * S * 16 * (size_t)(PX_MAX + 64)))) break;
            if ((rc = dev_alloc(e, &db.px_ring, S * (size_t)db.px_slots * 2 * PX_WORDS))) break;
            if (hipMemset(db.px_mem, 0, S * 2 * PX_MEM) != hipSuccess || hipMemset(db.px_pair, 0, S * 4 * PX_MAX) != hipSuccess ||
                hipMemset(db.px_job, 0xff, S * NWIN * 16 * sizeof(PxJob)) != hipSuccess) { rc = NRSC5HIP_EHIP; snprintf(g_err, sizeof(g_err), "PX state init failed"); break; }
        }
        db.l2_ring = nullptr; db.l2_px_ring = nullptr; db.l2_am_ring = nullptr;
        if (cfg->l2_index) {
            if ((rc = dev_alloc(e, &db.l2_ring, S * (size_t)cfg->p1_slots))) break;
            if (hipMemset(db.l2_ring, 0, S * (size_t)cfg->p1_slots * sizeof(nrsc5hip_l2_frame)) != hipSuccess) { rc = NRSC5HIP_EHIP; break; }
            // ... and of every P3 / P4 frame slot, and (AM engines) of the nine frames of every AM L1 frame slot
            const size_t npx = S * (size_t)db.px_slots * 2, nam = cfg->am_enable ? S * (size_t)cfg->p1_slots * 9 : 0;
            if ((rc = dev_alloc(e, &db.l2_px_ring, npx))) break;
            if (hipMemset(db.l2_px_ring, 0, npx * sizeof(nrsc5hip_l2_frame)) != hipSuccess) { rc = NRSC5HIP_EHIP; break; }
            if (nam) {
                if ((rc = dev_alloc(e, &db.l2_am_ring, nam))) break;
                if (hipMemset(db.l2_am_ring, 0, nam * sizeof(nrsc5hip_l2_frame)) != hipSuccess) { rc = NRSC5HIP_EHIP; break; }
            }
        }
        db.am = nullptr; db.am_sym = nullptr; db.am_q = nullptr; db.am_vit = nullptr; db.am_dec = nullptr; db.am_k9meta = nullptr; db.am_job = nullptr; db.am_ckpt = nullptr; db.am_ber = nullptr; db.am_pids_stage = nullptr; db.am_pids_rec = nullptr; db.am_nvit = 1;
        if (cfg->am_enable) {
            if ((rc = dev_alloc(e, &db.am, S))) break;
            if ((rc = dev_alloc(e, &db.am_sym, S * 4 * AM_SYMS))) break;
            if ((rc = dev_alloc(e, &db.am_q, S * 3 * 2 * (size_t)AM_VIT))) break;
            db.am_nvit = cfg->p1_async ? NWIN : 1;
            const size_t ndec = cfg->p1_async ? NAUX : 1;
            if ((rc = dev_alloc(e, &db.am_vit, S * db.am_nvit * 2 * AM_VIT))) break;
            if ((rc = dev_alloc(e, &db.am_dec, ndec * S * (size_t)(8 * AM_DEC_P1 + AM_DEC_P3)))) break;
            if (cfg->p1_async && (rc = dev_alloc(e, &db.am_k9meta, ndec * S))) break;
            if ((rc = dev_alloc(e, &db.am_job, S * NWIN))) break;
            if ((rc = dev_alloc(e, &db.am_ber, S * (size_t)cfg->p1_slots))) break;
            if (cfg->p1_async && cfg->l2_feedback && (rc = dev_alloc(e, &db.am_ckpt, S * NWIN * 8))) break;      // replay checkpoints, one per delivered P1 PDU
            if ((rc = dev_alloc(e, &db.am_pids_stage, S * NWIN * 8 * (size_t)(3 * PIDS_LEN)))) break;
            if ((rc = dev_alloc(e, &db.am_pids_rec, S * NWIN * 8))) break;
            if (hipMemset(db.am_pids_rec, 0xff, S * NWIN * 8 * sizeof(int)) != hipSuccess) { rc = NRSC5HIP_EHIP; break; }
            if (hipMemset(db.am_job, 0, S * NWIN * sizeof(AmJob)) != hipSuccess || hipMemset(db.am_ber, 0, S * (size_t)cfg->p1_slots * sizeof(float)) != hipSuccess) { rc = NRSC5HIP_EHIP; break; }
            std::vector<AmStream> ainit(S);
            for (size_t k = 0; k < S; k++) init_am_state(ainit[k]);
            if (hipMemcpy(db.am, ainit.data(), S * sizeof(AmStream), hipMemcpyHostToDevice) != hipSuccess ||
                hipMemset(db.am_q, 0, S * 3 * 2 * (size_t)AM_VIT) != hipSuccess || hipMemset(db.am_vit, 0, S * db.am_nvit * 2 * AM_VIT) != hipSuccess ||
                hipMemset(db.am_sym, 0, S * 4 * AM_SYMS) != hipSuccess) { rc = NRSC5HIP_EHIP; snprintf(g_err, sizeof(g_err), "AM state init failed"); break; }
        }
        db.sync_phase_cycles = nullptr;        // nrsc5hip_debug_tune(NRSC5HIP_TUNE_SYNC_PHASES) turns the instrumentation on
        e->stage_bytes = 4u << 20; e->stage_ring_bytes = 1u << 20;
        if ((rc = dev_alloc(e, &e->stage_dev, e->stage_bytes))) break;
        if (!cfg->p1_async) {
            for (int k = 0; k < nrsc5hip_engine::NSTAGE && !rc; k++) {
                if ((rc = dev_alloc(e, &e->stage_dev2[k], e->stage_ring_bytes + 16))) break;
                void *sp = nullptr;
                if (hipHostMalloc((void **)&e->stage_pin[k], e->stage_ring_bytes + 16, hipHostMallocMapped) != hipSuccess ||
                    hipHostGetDevicePointer(&sp, e->stage_pin[k], 0) != hipSuccess ||
                    hipEventCreateWithFlags(&e->stage_ev[k], hipEventDisableTiming) != hipSuccess) { rc = NRSC5HIP_ENOMEM; snprintf(g_err, sizeof(g_err), "pinned staging allocation failed"); }
                e->stage_pin_dev[k] = (uint8_t *)sp;
                e->stage_busy[k] = false;
            }
            if (rc) break;
            if (S * (size_t)cfg->p1_slots <= 64) {
                // the fast seam's P1 frames reach the host by the traceback's own stores (k_p1_traceback writes the pinned mirror beside the
                // device ring): nrsc5hip_p1_frame_packed then copies 18 KB of host memory instead of synchronising and issuing a D2H copy
                const size_t nw = S * (size_t)cfg->p1_slots * P1_WORDS;
                void *mp = nullptr;
                if (hipHostMalloc((void **)&e->frames_host, nw * sizeof(uint32_t), hipHostMallocMapped) != hipSuccess ||
                    hipHostGetDevicePointer(&mp, e->frames_host, 0) != hipSuccess) { rc = NRSC5HIP_ENOMEM; snprintf(g_err, sizeof(g_err), "pinned frame mirror allocation failed"); break; }
                memset(e->frames_host, 0, nw * sizeof(uint32_t));
                db.p1_mirror = (uint32_t *)mp;
            }
            // The ingest stream gets a queue priority of its own: HIP streams of one priority share a small pool of hardware queues, and
            // whether `ingest` and `main` landed on the same one -- which serialises the early chunks with the block step they are meant
            // to run beside -- depended on how many streams the process had created before (measured: the same drop-in build at 1110 x or
            // 820 x real time, from one process to the next).  Queues of different priorities are never shared.
            int prio_least = 0, prio_greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
            if (hipStreamCreateWithPriority(&e->ingest, hipStreamDefault, prio_greatest) != hipSuccess || hipEventCreateWithFlags(&e->ev_ingest, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&e->ev_main, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&e->ev_appended, hipEventDisableTiming) != hipSuccess) { rc = NRSC5HIP_EHIP; snprintf(g_err, sizeof(g_err), "ingest stream creation failed"); break; }
            e->ingest_dirty = false; e->main_stepped = false; e->main_appended = false; e->early_flush = 128u << 10;     // a block is 270 KB of cu8: 128 + 128 + a last chunk of ~14 KB
            if ((rc = dev_alloc(e, &e->decim_ticket, 1))) break;
            if (hipMemset(e->decim_ticket, 0, sizeof(unsigned)) != hipSuccess) { rc = NRSC5HIP_EHIP; break; }
            for (int k = 0; k < 2 && !rc; k++) {
                void *dp = nullptr;
                if (hipHostMalloc((void **)&e->report_host[k], sizeof(StreamReport), hipHostMallocMapped) != hipSuccess ||
                    hipHostGetDevicePointer(&dp, e->report_host[k], 0) != hipSuccess) { rc = NRSC5HIP_ENOMEM; snprintf(g_err, sizeof(g_err), "pinned report allocation failed"); break; }
                e->report_dev[k] = (StreamReport *)dp;
                memset(e->report_host[k], 0, sizeof(StreamReport));
            }
            if (rc) break;
        }
        e->hc_stream = -1; e->hc_abs0 = 0; e->hc_wr = 0; e->host_capture = false; e->hc_rebases = e->hc_attaches = e->hc_detaches = 0; e->reports_folded = 0;
        if (!cfg->p1_async) {
            e->hc_cap = 16u << 20;                             // 58 FM blocks between two rebases (~300 KB of host memmove each)
            void *hp = nullptr;
            if (hipHostMalloc((void **)&e->hc_pin, e->hc_cap, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&hp, e->hc_pin, 0) != hipSuccess) {
                rc = NRSC5HIP_ENOMEM; snprintf(g_err, sizeof(g_err), "pinned capture allocation failed"); break;
            }
            e->hc_dev = (uint8_t *)hp; e->host_capture = true;
        }
        e->hb_hist_host.assign(S, std::array<c16, 14>{});
        e->stage_slot = 0; e->staged_stream = -1; e->staged_bytes = 0; e->staged_q15 = 0; e->staged_cu8 = false;
        if ((rc = dev_alloc(e, &e->ids_dev, S))) break;
        if ((rc = dev_alloc(e, &e->nbytes_dev, S))) break;
        if ((rc = dev_alloc(e, &e->all_ids_dev, S))) break;
        std::vector<StreamState> init(S);
        std::vector<int> ident(S);
        for (size_t s = 0; s < S; s++) { init_state(init[s]); ident[s] = (int)s; }
        if (hipMemcpy(db.state, init.data(), S * sizeof(StreamState), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(e->all_ids_dev, ident.data(), S * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemset(db.records, 0, S * db.rec_cap * sizeof(BlockRecord)) != hipSuccess ||
            hipMemset(db.pm, 0, S * NPM * PM_FRAME) != hipSuccess) { rc = NRSC5HIP_EHIP; snprintf(g_err, sizeof(g_err), "state init copy failed"); break; }
        e->wr_host.assign(S, 0); e->base_host.assign(S, 0); e->drained.assign(S, 0);
        e->mode_host.assign(S, MODE_FM); e->raw_host.assign(S, 0); e->attached.assign(S, 0);
        e->rd_host.assign(S, 0); e->fetched.assign(S, 0); e->mirror_ok.assign(S, cfg->p1_async ? 0 : 1); e->pending.assign(S, {});
        e->pred_ok.assign(S, 0); e->pred_samperr.assign(S, 0); e->pred_bc.assign(S, 0); e->manual_step.assign(S, 0);
        e->inflight_stream = -1; e->report_seq = 0; e->inflight_rd_pred = -1; e->inflight_decoded = true; e->inflight_progress = false; e->inflight_seq = 0; e->ahead.valid = false;
        e->defer_wait = true; e->direct_decimate = true; e->counters_clean = false;
        e->lane.db = db; e->lane.counters_dev = db.counters;
        e->prof_on = false; e->prof_only = -1;
        for (int k = 0; k < NRSC5HIP_PROF_CLASSES; k++) { e->prof_ms[k] = 0; e->prof_launches[k] = 0; }
    } while (0);
    if (rc) { nrsc5hip_engine_destroy(e); return rc; }
    *out = e;
    return NRSC5HIP_OK;
}

extern "C" void nrsc5hip_engine_destroy(nrsc5hip_engine *e)
{
    if (!e) return;
    DeviceGuard guard(e->cfg.device);
    (void)hipDeviceSynchronize();
    vit_scratch_free(e->vit_scratch);
    for (void *p : e->allocs) (void)hipFree(p);
    for (hipEvent_t ev : e->dec_events) (void)hipEventDestroy(ev);
    if (e->dec_stream) (void)hipStreamDestroy(e->dec_stream);
    if (e->flow_err) (void)hipHostFree(e->flow_err);
    if (e->rec_host) (void)hipHostFree(e->rec_host);
    if (e->frames_host) (void)hipHostFree(e->frames_host);
    if (e->nblocks_host) (void)hipHostFree(e->nblocks_host);
    for (int k = 0; k < nrsc5hip_engine::NSTAGE; k++) { if (e->stage_pin[k]) (void)hipHostFree(e->stage_pin[k]); if (e->stage_ev[k]) (void)hipEventDestroy(e->stage_ev[k]); }
    for (int k = 0; k < 2; k++) if (e->report_host[k]) (void)hipHostFree(e->report_host[k]);
    if (e->hc_pin) (void)hipHostFree(e->hc_pin);
    if (e->ingest) (void)hipStreamDestroy(e->ingest);
    if (e->ev_ingest) (void)hipEventDestroy(e->ev_ingest);
    if (e->ev_main) (void)hipEventDestroy(e->ev_main);
    if (e->ev_appended) (void)hipEventDestroy(e->ev_appended);
    for (auto &sp : e->prof_spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    for (hipEvent_t ev : e->prof_pool) (void)hipEventDestroy(ev);
    {
        nrsc5hip_engine::Lane &ln = e->lane;
        if (ln.counters_host) (void)hipHostFree(ln.counters_host);
        for (int k = 0; k < NWIN; k++) { if (ln.ev_window[k]) (void)hipEventDestroy(ln.ev_window[k]); if (ln.ev_decoded[k]) (void)hipEventDestroy(ln.ev_decoded[k]); }
        if (ln.main) (void)hipStreamDestroy(ln.main);
        for (int k = 0; k < NAUX; k++) if (ln.aux[k]) (void)hipStreamDestroy(ln.aux[k]);
    }
    delete e;
}

extern "C" void *nrsc5hip_engine_hip_stream(nrsc5hip_engine *e) { return e ? (void *)e->main : nullptr; }

static int check_stream(nrsc5hip_engine *e, int s)
{
    if (!e) FAIL(NRSC5HIP_EINVAL, "null engine");
    if (s < 0 || s >= e->cfg.max_streams) FAIL(NRSC5HIP_EINVAL, "stream %d out of range", s);
    return 0;
}

// ---- block-step scheduler --------------------------------------------------------------------------
// One step = every listed stream whose 33-symbol window is complete advances by one block:
//   [acquisition kernels if any stream may be un-synchronised] -> prepare -> mix+FFT -> sync (+PIDS)
//   -> P1 de-interleave -> P1 Viterbi (in order, or deferred to the aux stream once per 16-step window).
// Decode stream of a window: round robin over the `naux` streams that keep the chip busy without starving the chain.  A THIN
// window (few streams advanced: the stragglers' tail of a pass) is latency- not throughput-bound -- a handful of one-wave
// trellis passes -- so when its regular stream is still busy it may take one of the spare streams instead of queueing.
static int pick_decode_lane(nrsc5hip_engine *e, nrsc5hip_engine::Lane &ln, long long window)
{
    const int lane = (int)(window % e->naux);
    auto busy = [&](int k) { return ln.lane_parity[k] >= 0 && hipEventQuery(ln.ev_decoded[ln.lane_parity[k]]) == hipErrorNotReady; };
    if (ln.thin && busy(lane))
        for (int k = e->naux; k < NAUX; k++) if (!busy(k)) return k;
    return lane;
}

// Waves per frame of the forward trellis pass: a window of a stream set of n streams holds ~n frames; give every frame as many
// segment waves as keeps the launch within one wave per SIMD (1024) -- up to 16.  Thin windows (the stragglers' tail) and
// small sets are latency-bound: 16.
static int fwd_segments_for(const nrsc5hip_engine *e, const nrsc5hip_engine::Lane &ln, int n)
{
    if (e->fwd_segments > 0) return e->fwd_segments;
    if (ln.thin) return 16;
    const int g = 1024 / (n > 0 ? n : 1);
    return g < 1 ? 1 : g > VIT3_GMAX ? VIT3_GMAX : g;          // a lone stream (the in-order seam, the drop-in): 64 segment waves
}

static int launch_window_decode(nrsc5hip_engine *e, nrsc5hip_engine::Lane &ln, int n, const int *ids_dev, int parity, int lane)
{
    // decode the window's PIDS frames and P1 frames on aux stream `lane`, overlapped with the next windows
    // (NAUX windows decode concurrently, each wave of the forward pass alone on a SIMD)
    hipStream_t ax = ln.aux[lane];
    HIPCHK(hipEventRecord(ln.ev_window[parity], ln.main));
    HIPCHK(hipStreamWaitEvent(ax, ln.ev_window[parity], 0));
    { ProfScope p(e, NRSC5HIP_PROF_PIDS, ax); launch_pids_decode(e->tb, ln.db, n, ids_dev, parity, 16, ax); }
    if (ln.px_needed) { ProfScope p(e, NRSC5HIP_PROF_PIDS, ax); launch_px_decode(e->tb, ln.db, n, ids_dev, parity, lane, ax); }
    { ProfScope p(e, NRSC5HIP_PROF_P1_DEINT, ax); launch_p1_deint(e->tb, ln.db, n, ids_dev, parity, lane, ax); }
    { ProfScope p(e, NRSC5HIP_PROF_P1_VITERBI, ax); launch_p1_forward(e->tb, ln.db, n, ids_dev, parity, lane, ax, fwd_segments_for(e, ln, n), e->fwd_warm); }
    { ProfScope p(e, NRSC5HIP_PROF_P1_TRACEBACK, ax); launch_p1_traceback(e->tb, ln.db, n, ids_dev, parity, lane, ax, e->cfg.l2_feedback ? 2 : 0, fwd_segments_for(e, ln, n), e->tb_walk); }
    HIPCHK(hipEventRecord(ln.ev_decoded[parity], ax));
    ln.decoded_pending[parity] = true;
    ln.lane_parity[lane] = parity;
    return 0;
}

// in-order mode: the P1 frames the step's blocks completed (the kernels leave at once for a stream without one)
static int launch_inorder_p1(nrsc5hip_engine *e, nrsc5hip_engine::Lane &ln, int n, const int *ids_dev)
{
    { ProfScope p(e, NRSC5HIP_PROF_P1_DEINT, ln.main); launch_p1_deint(e->tb, ln.db, n, ids_dev, 0, 0, ln.main); }
    { ProfScope p(e, NRSC5HIP_PROF_P1_VITERBI, ln.main); launch_p1_forward(e->tb, ln.db, n, ids_dev, 0, 0, ln.main, fwd_segments_for(e, ln, n), e->fwd_warm); }
    { ProfScope p(e, NRSC5HIP_PROF_P1_TRACEBACK, ln.main); launch_p1_traceback(e->tb, ln.db, n, ids_dev, 0, 0, ln.main, e->cfg.l2_feedback ? 1 : 0, fwd_segments_for(e, ln, n), e->tb_walk); }
    return 0;
}

// decode_p1 = false (fast streaming seam only): the caller KNOWS that no listed stream can complete a P1 frame in this step
// local_prepare (fast streaming seam, stream known to be FINE): no k_prepare launch -- the symbol kernel computes the block's
// bookkeeping for itself and the sync kernel commits it
struct StepReport { StreamReport *out; unsigned seq; int first_rec; bool folded; };   // fast seam: the report the step's last kernel may post itself (issue_step sets `folded` when k_sync did)
static int issue_step(nrsc5hip_engine *e, nrsc5hip_engine::Lane &ln, int n, const int *ids_dev, bool decode_p1 = true, bool decode_pids = true, bool local_prepare = false, StepReport *rep = nullptr)
{
    const bool async = e->cfg.p1_async != 0;
    const long long window = ln.step_count / 16;
    const int parity = async ? (int)(window % NWIN) : 0;       // buffer slot of this decode window
    if (async && (ln.step_count % 16) == 0 && ln.decoded_pending[parity]) {
        // the buffers of slot `parity` are about to be rewritten: the decoder launched NWIN windows ago must be done
        HIPCHK(hipStreamWaitEvent(ln.main, ln.ev_decoded[parity], 0));
        ln.decoded_pending[parity] = false;
    }
    if (e->dec_chunk) {
        // a stream starting from a fresh reset has read at most 70199 t + 71280 samples when step t begins
        const long long reach = 70199LL * ln.step_count + WIN_N;
        int need = (int)(reach / e->dec_chunk) + 1;
        if (need > (int)e->dec_events.size()) need = (int)e->dec_events.size();
        for (; ln.dec_waited < need; ln.dec_waited++) HIPCHK(hipStreamWaitEvent(ln.main, e->dec_events[ln.dec_waited], 0));
    }
    if (e->cfg.l2_feedback && !async) ln.acq_needed = true;   // an in-order P1 decode may send any stream back to NONE for the next block
    if (ln.acq_needed) { ProfScope p(e, NRSC5HIP_PROF_ACQUIRE, ln.main); launch_acquire(e->tb, ln.db, n, ids_dev, ln.main); }
    // prepare_block is idempotent for a stream the previous k_sync already prepared; a stream that is not FINE is only
    // prepared here, on a step that ran the acquisition kernels for its current window
    const bool fused_prepare = local_prepare && !ln.acq_needed && !async && ln.db.nco_policy != NCO_EXACT_ALWAYS;
    if (!fused_prepare && (!ln.prepared_by_sync || ln.acq_needed)) { ProfScope p(e, NRSC5HIP_PROF_PREPARE, ln.main); launch_prepare(ln.db, n, ids_dev, ln.acq_needed ? 1 : 0, ln.main); }
    // exact-oscillator blocks (a freshly reset stream up to its first lock): only a stream that is not FINE can be in that mode, and
    // those only advance on steps that run the acquisition kernels
    if (ln.db.nco_tab && (ln.acq_needed || ln.db.nco_policy == NCO_EXACT_ALWAYS)) { ProfScope p(e, NRSC5HIP_PROF_PREPARE, ln.main); launch_nco_exact(ln.db, n, ids_dev, ln.main); }
    { ProfScope p(e, NRSC5HIP_PROF_MIXFFT, ln.main); launch_mixfft(e->tb, ln.db, n, ids_dev, ln.main, e->mixfft_syms, fused_prepare ? 1 : 0); }
    const int slot = async ? (int)(ln.step_count % 16) : 0;
    // batch pipeline: once every stream of the set is FINE, the next block's bookkeeping rides in k_sync's tail
    const int fuse = (async && !ln.acq_needed) ? 1 : 0;
    // nothing runs behind k_sync on this step (no PX kernels, no separate PIDS decode, no in-order P1 decode): it posts the step's report itself
    const bool fold = rep && n == 1 && !async && !ln.px_needed && !decode_pids && !decode_p1 && e->sync_lanes != 256;     // (the wide form alone has the reporting twin: k_sync_report)
    { ProfScope p(e, NRSC5HIP_PROF_SYNC, ln.main); launch_sync(e->tb, ln.db, n, ids_dev, parity, slot, fuse, (int)window, ln.main, e->sync_lanes, decode_pids ? 0 : 1, fused_prepare ? 1 : 0, ln.px_needed ? 1 : 0,
                                                                fold ? rep->out : nullptr, fold ? rep->seq : 0u, fold ? rep->first_rec : 0); }
    if (rep) rep->folded = fold;
    ln.prepared_by_sync = fuse != 0;
    if (ln.px_needed) { ProfScope p(e, NRSC5HIP_PROF_PIDS, ln.main); launch_px_deint(e->tb, ln.db, n, ids_dev, parity, slot, ln.main); }
    if (!async) {
        if (decode_pids) { ProfScope p(e, NRSC5HIP_PROF_PIDS, ln.main); launch_pids_decode(e->tb, ln.db, n, ids_dev, parity, 1, ln.main); }
        if (ln.px_needed) { ProfScope p(e, NRSC5HIP_PROF_PIDS, ln.main); launch_px_decode(e->tb, ln.db, n, ids_dev, parity, 0, ln.main); }
        if (decode_p1) { int rc = launch_inorder_p1(e, ln, n, ids_dev); if (rc) return rc; }
    } else if ((ln.step_count % 16) == 15) {
        int rc = launch_window_decode(e, ln, n, ids_dev, parity, pick_decode_lane(e, ln, window)); if (rc) return rc;
    }
    ln.step_count++;
    HIPCHK(hipGetLastError());
    return 0;
}

// K block steps of a set whose streams are all FINE as ONE launch (k_flow, k_sync.hip): what issue_step does for each of them -- the wait for the decoder that used
// this window's buffers, the symbol and sync kernels with the next block's bookkeeping fused, the window decode behind step 15 -- with the K x 2 launches replaced
// by one grid whose workgroups hand over to each other.  The caller has checked the conditions (run_steps).
static int issue_flow_burst(nrsc5hip_engine *e, nrsc5hip_engine::Lane &ln, int n, const int *ids_dev, int K)
{
    const long long window = ln.step_count / 16;
    const int parity = (int)(window % NWIN), slot0 = (int)(ln.step_count % 16);
    if (slot0 + K > 16) FAIL(NRSC5HIP_EINVAL, "a dataflow burst does not cross a window boundary");
    if (slot0 == 0 && ln.decoded_pending[parity]) {
        HIPCHK(hipStreamWaitEvent(ln.main, ln.ev_decoded[parity], 0));
        ln.decoded_pending[parity] = false;
    }
    HIPCHK(hipMemsetAsync(e->flow_dev, 0, flow_words(n) * sizeof(unsigned), ln.main));
    { ProfScope p(e, NRSC5HIP_PROF_FLOW, ln.main); launch_flow(e->tb, ln.db, n, ids_dev, K, e->flow_dev, e->flow_err, parity, slot0, (int)window, ln.main); }
    // a poll that gave up (flow words [8], [9]): the host reads them with the burst's counters (run_steps)
    ln.step_count += K;
    e->flow_bursts++; e->flow_steps += K;
    if ((ln.step_count % 16) == 0) { int rc = launch_window_decode(e, ln, n, ids_dev, parity, pick_decode_lane(e, ln, window)); if (rc) return rc; }
    HIPCHK(hipGetLastError());
    return 0;
}

// finish a partially filled decode window (async mode) so that every produced frame gets decoded, and wait for all decodes
static int flush_p1(nrsc5hip_engine *e, nrsc5hip_engine::Lane &ln, int n, const int *ids_dev)
{
    if (!e->cfg.p1_async) return 0;
    if (ln.step_count % 16) {
        const long long window = ln.step_count / 16;
        int rc = launch_window_decode(e, ln, n, ids_dev, (int)(window % NWIN), pick_decode_lane(e, ln, window)); if (rc) return rc;
        ln.step_count += 16 - (ln.step_count % 16);            // the next steps start a fresh window
    }
    for (int k = 0; k < NAUX; k++) HIPCHK(hipStreamSynchronize(ln.aux[k]));
    for (int k = 0; k < NWIN; k++) ln.decoded_pending[k] = false;
    return 0;
}

// Runs block steps for the n streams listed at ids_dev until none of them has a complete window left (or max_steps).
// `set_sig` identifies the stream set: the acquisition / PX launch flags measured on one set say nothing about another.
static int run_steps(nrsc5hip_engine *e, int n, const int *ids_dev, unsigned long long set_sig, int max_steps, int check_every, int *steps_done)
{
    nrsc5hip_engine::Lane &ln = e->lane;
    if (set_sig != ln.set_sig) { ln.acq_needed = true; ln.px_needed = true; ln.set_sig = set_sig; }
    ln.prepared_by_sync = false;
    const bool replay = e->db.ckpt != nullptr;
    int done = 0;
    for (;;) {
        bool live = n > 0;
        while (live && done < max_steps) {
            HIPCHK(hipMemsetAsync(ln.counters_dev, 0, 4 * sizeof(int), ln.main));
            // While the acquisition kernels are being launched the host looks again after 4 steps instead of a whole window: they are
            // eight thin launches per step (~38 us of a ~135 us step) for as long as the LAST look saw a stream that was not FINE,
            // and every stream of a batch is past that point a few blocks after its (re-)acquisition.
            // Bursts end on window boundaries (the rollback below is launched there).
            int every = check_every;
            if (check_every == 16) {
                const int to_boundary = 16 - (int)(ln.step_count % 16);
                every = ln.acq_needed ? std::min(4, to_boundary) : to_boundary;   // 2 measured: the same
            }
            int burst = 0;
            // dataflow burst (k_flow): every stream of the set was FINE at the last look, MP1 routing only, zero-copy input, closed-form oscillator, the previous
            // step's k_sync prepared this one -- the whole burst (it ends on the window boundary) is one launch
            const bool flow = e->flow_min > 0 && n >= e->flow_min && check_every == 16 && e->cfg.p1_async && e->cfg.batch_zero_copy && !ln.acq_needed && !ln.px_needed
                              && ln.prepared_by_sync && !e->dec_chunk && ln.db.nco_policy != NCO_EXACT_ALWAYS && !ln.db.sync_phase_cycles && done + every <= max_steps && every >= 2;
            if (flow) { int rc = issue_flow_burst(e, ln, n, ids_dev, every); if (rc) return rc; burst = every; }
            for (; burst < every && done + burst < max_steps; burst++) { int rc = issue_step(e, ln, n, ids_dev); if (rc) return rc; }
            if (replay && (ln.step_count % 16) == 0) {
                // Window boundary: take the first-header verdicts of the deferred decodes that have finished.  The decode whose
                // job slot the next window reuses (launched NWIN windows before it) must be among them.
                const long long window = ln.step_count / 16;
                const int parity = (int)(window % NWIN);
                if (ln.decoded_pending[parity]) { HIPCHK(hipStreamWaitEvent(ln.main, ln.ev_decoded[parity], 0)); ln.decoded_pending[parity] = false; }
                ProfScope p(e, NRSC5HIP_PROF_PREPARE, ln.main);
                launch_rollback(ln.db, n, ids_dev, (int)window, e->verdict_lag, ln.main);
            }
            HIPCHK(hipMemcpyAsync(ln.counters_host, ln.counters_dev, 4 * sizeof(int), hipMemcpyDeviceToHost, ln.main));
            HIPCHK(hipStreamSynchronize(ln.main));
            if (flow && (e->flow_err[0] || e->flow_err[1]))
                FAIL(NRSC5HIP_EHIP, "dataflow burst: a hand-off was never seen (symbol item of stream position %d, block step of %d): the burst's results are void", (int)e->flow_err[0] - 1, (int)e->flow_err[1] - 1);
            ln.acq_needed = ln.counters_host[1] > 0;
            ln.thin = ln.counters_host[0] * 4 < burst * n;
            ln.px_needed = ln.counters_host[2] > 0;
            if (ln.counters_host[0] == 0) live = false;        // nothing was processed (or is pending) in this burst
            else done += burst;
        }
        if (e->dec_chunk) { HIPCHK(hipStreamSynchronize(e->dec_stream)); e->dec_chunk = 0; }
        { int rc = flush_p1(e, ln, n, ids_dev); if (rc) return rc; }
        if (!replay) break;
        // every decode has finished: apply what is left of their verdicts (also when the step budget is used up -- the caller
        // must never see records of blocks that ran behind a failed frame); a rewound stream has work again
        HIPCHK(hipMemsetAsync(ln.counters_dev, 0, 4 * sizeof(int), ln.main));
        launch_rollback(ln.db, n, ids_dev, (int)(ln.step_count / 16), 0, ln.main);
        HIPCHK(hipMemcpyAsync(ln.counters_host, ln.counters_dev, 4 * sizeof(int), hipMemcpyDeviceToHost, ln.main));
        HIPCHK(hipStreamSynchronize(ln.main));
        if (ln.counters_host[3] == 0) break;
        ln.acq_needed = true; ln.prepared_by_sync = false;
        if (done >= max_steps) break;                              // out of budget: the rewound streams resume on the next call
    }
    HIPCHK(hipStreamSynchronize(ln.main));
    if (e->prof_on) prof_collect(e);
    if (steps_done) *steps_done = done;
    return 0;
}

static unsigned long long set_signature(int n, const int *ids)
{
    unsigned long long h = 0xcbf29ce484222325ull ^ (unsigned long long)n;
    if (ids) for (int k = 0; k < n; k++) h = (h ^ (unsigned long long)(unsigned)ids[k]) * 0x100000001b3ull;
    return h | 1ull;                                           // never 0 (= "no set measured yet")
}

// AM streams: one fused kernel per block step (k_am.hip).  p1_async = 0: every frame decodes in order on the main stream
// (reference event timing).  p1_async = 1: window pipeline as for FM -- each 8-step window hands the L1 frames whose
// de-interleave fell into it to one of the decode streams, where their 8 P1 frames and P3 frame decode concurrently.
static int am_flush(nrsc5hip_engine *e, nrsc5hip_engine::Lane &ln, int n, const int *ids_dev)
{
    if (!e->cfg.p1_async) return 0;
    if (ln.am_step_count % 8) {
        const long long window = ln.am_step_count / 8;
        const int parity = (int)(window % NWIN), lane = (int)(window % e->naux_am);
        hipStream_t ax = ln.aux[lane];
        HIPCHK(hipEventRecord(ln.ev_window[parity], ln.main));
        HIPCHK(hipStreamWaitEvent(ax, ln.ev_window[parity], 0));
        { ProfScope p(e, NRSC5HIP_PROF_AM_DECODE, ax); launch_am_decode(e->tb, ln.db, n, ids_dev, parity, lane, e->cfg.l2_feedback, ax, e->am_segments, e->am_warm, e->am_runin); }
        ln.am_step_count += 8 - (ln.am_step_count % 8);
    }
    for (int k = 0; k < NAUX; k++) HIPCHK(hipStreamSynchronize(ln.aux[k]));
    for (int k = 0; k < NWIN; k++) ln.am_decoded_pending[k] = false;
    return 0;
}

static int run_steps_am(nrsc5hip_engine *e, int n, const int *ids_dev, int max_steps, int check_every, int *steps_done)
{
    nrsc5hip_engine::Lane &ln = e->lane;
    const bool pipe = e->cfg.p1_async != 0;
    const bool replay = ln.db.am_ckpt != nullptr;              // window pipeline with the on-device L2 feedback (k_replay.hip)
    int done = 0;
    for (;;) {
        bool live = n > 0;
        while (live && done < max_steps) {
            HIPCHK(hipMemsetAsync(ln.counters_dev, 0, 4 * sizeof(int), ln.main));
            int burst = 0;
            for (; burst < check_every && done + burst < max_steps; burst++) {
                const long long window = ln.am_step_count / 8;
                const int parity = pipe ? (int)(window % NWIN) : -1, lane = (int)(window % e->naux_am);
                if (pipe && (ln.am_step_count % 8) == 0) {
                    // window boundary: the decode that used this window's buffers NWIN windows ago must have finished; with the
                    // replay, take the first-header verdicts of every deferred decode that has (its job slot is reused next) --
                    // no host round trip: the burst runs on across window boundaries
                    if (ln.am_decoded_pending[parity]) { HIPCHK(hipStreamWaitEvent(ln.main, ln.ev_decoded[parity], 0)); ln.am_decoded_pending[parity] = false; }
                    if (replay && ln.am_step_count > 0) launch_rollback_am(ln.db, n, ids_dev, (int)window, e->verdict_lag, ln.main);
                }
                { ProfScope p(e, NRSC5HIP_PROF_AM, ln.main); launch_am_step(e->tb, ln.db, n, ids_dev, ln.main, e->cfg.l2_feedback, parity, (int)(ln.am_step_count % 8), (int)window); }
                if (pipe && (ln.am_step_count % 8) == 7) {
                    hipStream_t ax = ln.aux[lane];
                    HIPCHK(hipEventRecord(ln.ev_window[parity], ln.main));
                    HIPCHK(hipStreamWaitEvent(ax, ln.ev_window[parity], 0));
                    { ProfScope p(e, NRSC5HIP_PROF_AM_DECODE, ax); launch_am_decode(e->tb, ln.db, n, ids_dev, parity, lane, e->cfg.l2_feedback, ax, e->am_segments, e->am_warm, e->am_runin); }
                    HIPCHK(hipEventRecord(ln.ev_decoded[parity], ax));
                    ln.am_decoded_pending[parity] = true;
                }
                ln.am_step_count++;
            }
            HIPCHK(hipMemcpyAsync(ln.counters_host, ln.counters_dev, 4 * sizeof(int), hipMemcpyDeviceToHost, ln.main));
            HIPCHK(hipStreamSynchronize(ln.main));
            HIPCHK(hipGetLastError());
            if (ln.counters_host[0] == 0) live = false;
            else done += burst;
        }
        { int rc = am_flush(e, ln, n, ids_dev); if (rc) return rc; }
        if (!replay) break;
        // every decode has finished: apply what is left of their verdicts (also when the step budget is used up); a rewound
        // stream has work again
        HIPCHK(hipMemsetAsync(ln.counters_dev, 0, 4 * sizeof(int), ln.main));
        launch_rollback_am(ln.db, n, ids_dev, (int)(ln.am_step_count / 8), 0, ln.main);
        HIPCHK(hipMemcpyAsync(ln.counters_host, ln.counters_dev, 4 * sizeof(int), hipMemcpyDeviceToHost, ln.main));
        HIPCHK(hipStreamSynchronize(ln.main));
        if (ln.counters_host[3] == 0 || done >= max_steps) break;
    }
    HIPCHK(hipStreamSynchronize(ln.main));
    if (e->prof_on) prof_collect(e);
    if (steps_done) *steps_done = done;
    return 0;
}

// ---- FIFO space management (streaming) -----------------------------------------------------------------
__global__ void k_compact(DevBuffers db, int s)
{
    // move the unread tail [rd, wr) to the start of the stream's slab; forward copy, dst < src
    StreamState &st = db.state[s];
    c16 *buf = db.q15 + (size_t)s * db.q15_cap;
    const long long off = st.rd - st.base, n = st.wr - st.rd;
    __shared__ c16 tmp[1024];
    for (long long c = 0; c < n; c += 1024) {
        const long long k = c + threadIdx.x;
        if (k < n) tmp[threadIdx.x] = buf[off + k];
        __syncthreads();
        if (k < n) buf[k] = tmp[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) st.base = st.rd;
}

static int ensure_space(nrsc5hip_engine *e, int s, long long incoming, bool on_ingest = false)
{
    if (e->wr_host[s] - e->base_host[s] + incoming <= e->db.q15_cap) return 0;
    if (on_ingest && e->main_stepped) {                        // the compaction moves [rd, wr): the steps submitted so far must have left their final rd
        HIPCHK(hipEventRecord(e->ev_main, e->main));
        HIPCHK(hipStreamWaitEvent(e->ingest, e->ev_main, 0));
        e->main_stepped = false;
    }
    hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, on_ingest ? e->ingest : e->main, e->db, s);
    if (e->mirror_ok[s]) {
        e->base_host[s] = e->rd_host[s];                       // k_compact sets base = rd, and the mirror IS the device's rd
    } else {
        long long base = 0;
        HIPCHK(hipMemcpyAsync(&base, (const char *)(e->db.state + s) + offsetof(StreamState, base), sizeof(long long), hipMemcpyDeviceToHost, e->main));
        HIPCHK(hipStreamSynchronize(e->main));
        e->base_host[s] = base;
    }
    if (e->wr_host[s] - e->base_host[s] + incoming > e->db.q15_cap)
        FAIL(NRSC5HIP_EOVERFLOW, "stream %d: FIFO capacity %lld too small for %lld more samples", s, e->db.q15_cap, incoming);
    return 0;
}

// ---- streaming seam ---------------------------------------------------------------------------------------
// (the report itself is the tail of the step's last kernel: k_stream_tail, k_sync.hip)
static int window_of(const nrsc5hip_engine *e, int s) { return e->mode_host[s] == MODE_AM ? AM_WIN : WIN_N; }

static void forget_prediction(nrsc5hip_engine *e, int s) { e->pred_ok[s] = 0; }

// Wait for the report with sequence number `seq`: the kernel's last store is that number into mapped pinned memory, so the
// host spins on it (a stream synchronisation returns ~5-10 us after the kernel has ended); bounded, then the ordinary wait.
static int wait_report(nrsc5hip_engine *e, unsigned seq, bool block)
{
    const volatile unsigned *p = &e->report_host[seq & 1]->seq;
    if (__atomic_load_n(p, __ATOMIC_ACQUIRE) == seq) return 1;
    if (!block) return 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int spins = 0;; spins++) {
        if (__atomic_load_n(p, __ATOMIC_ACQUIRE) == seq) return 1;
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
        if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;   // a block step is ~50 us: past 2 ms something else holds the queue -- stop burning a core, block
    }
    HIPCHK(hipStreamSynchronize(e->lane.main));
    if (__atomic_load_n(p, __ATOMIC_ACQUIRE) != seq) FAIL(NRSC5HIP_EHIP, "stream report %u never arrived (have %u)", seq, *p);
    return 1;
}

// the next report's sequence number, buffer and first record
static StepReport next_report(nrsc5hip_engine *e, int s)
{
    e->report_seq++;
    if (e->report_seq == 0) e->report_seq = 2;                 // 0 = the freshly cleared report; 2, not 1: the step before the wrap posted into buffer 1 (seq & 1)
    // records to post: from the first one the host has not seen -- the block of a step still in flight is not this step's to report
    const int first_rec = e->fetched[s] + ((e->inflight_stream == s) ? 1 : 0);
    return StepReport{ e->report_dev[e->report_seq & 1], e->report_seq, first_rec, false };
}

static int launch_report(nrsc5hip_engine *e, int s, bool with_pids, const StepReport *prepared = nullptr)
{
    const StepReport r = prepared ? *prepared : next_report(e, s);
    launch_stream_tail(e->tb, e->lane.db, s, r.first_rec, r.out, r.seq, with_pids ? 1 : 0, e->lane.main);
    e->counters_clean = true;
    HIPCHK(hipGetLastError());
    return 0;
}

// Take the report of the step in flight (if any).  block = false: only if it has arrived.  Returns < 0 on error.
static int harvest(nrsc5hip_engine *e, bool block)
{
    const int s = e->inflight_stream;
    if (s < 0) return 0;
    nrsc5hip_engine::Lane &ln = e->lane;
    {
        const auto t_wait = std::chrono::steady_clock::now();
        const int got = wait_report(e, e->inflight_seq, block);
        if (got < 0) return got;
        if (!got) return 0;
        if (block) g_seam[3] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wait).count();
    }
    e->inflight_stream = -1;
    const StreamReport *rp = e->report_host[e->inflight_seq & 1];
    bool p1_missing = false;
    for (int k = 0; k < rp->nrec; k++) if ((rp->rec[k].flags & REC_P1) && e->mode_host[s] != MODE_AM && !e->inflight_decoded) p1_missing = true;
    if (p1_missing) {
        // the prediction said no P1 frame could complete in this block and one did: decode it now, take the record again
        g_seam[11] += 1;
        if (e->ahead.valid) {
            // cannot happen while the caller keeps the contract of nrsc5hip_stream_step_ahead (nothing that changes L1 state between it and the drain);
            // if it does, leave the engine in a state every later call understands: nothing in flight, the stream's mirror invalid (its next push
            // re-synchronises with the device), then report.  What is LOST in this case, and documented as such (include/nrsc5hip.h, nrsc5hip_stream_step_ahead): the events
            // of the two blocks already run on the device are not delivered through the seam -- their records stay in the device ring and are visible to
            // nrsc5hip_drain / nrsc5hip_batch_fetch, but the frame of the first lacks its decode (no P1 bits, no BER); the caller's session is over (error return).
            e->ahead.valid = false; e->inflight_rd_pred = -1;
            (void)hipStreamSynchronize(ln.main);
            e->mirror_ok[s] = 0; forget_prediction(e, s);
            FAIL(NRSC5HIP_EHIP, "stream %d: a block submitted without its P1 decode completed a frame, and the next block is already running", s);
        }
        int rc = launch_inorder_p1(e, ln, 1, e->all_ids_dev + s); if (rc) return rc;
        if ((rc = launch_report(e, s, false))) return rc;
        if ((rc = wait_report(e, e->report_seq, true)) < 0) return rc;
        rp = e->report_host[e->report_seq & 1];
    }
    ln.acq_needed = rp->counters[1] > 0;
    ln.px_needed = rp->counters[2] > 0;
    if (e->inflight_rd_pred >= 0 && e->inflight_rd_pred != rp->rd) g_seam[9] += 1;      // never seen; the mirror is put right below
    e->rd_host[s] = rp->rd;
    for (int k = 0; k < rp->nrec; k++) e->pending[s].push_back(rp->rec[k]);
    e->fetched[s] += rp->nrec;
    if (rp->nblocks != e->fetched[s]) FAIL(NRSC5HIP_EOVERFLOW, "stream %d: %d records behind the report", s, rp->nblocks - e->fetched[s]);
    if (rp->nrec > 0) {
        const BlockRecord &r = rp->rec[rp->nrec - 1];
        e->pred_ok[s] = (r.state_after == SYNC_FINE && !(r.flags & REC_LOST_SYNC)) ? 1 : 0;
        e->pred_samperr[s] = r.next_samperr; e->pred_bc[s] = r.bc;
    }
    e->inflight_progress = rp->counters[0] != 0;
    if (e->ahead.valid) {
        // the step submitted ahead becomes the step in flight; now that its predecessor's record is here, so does its prediction
        const int s2 = e->ahead.stream;
        e->ahead.valid = false;
        e->inflight_stream = s2; e->inflight_seq = e->ahead.seq; e->inflight_decoded = e->ahead.decoded; e->inflight_rd_pred = -1;
        if (e->pred_ok[s2] && !e->cfg.l2_feedback && e->defer_wait) {
            e->inflight_rd_pred = e->rd_host[s2] + WIN_N - SYM_N + e->pred_samperr[s2];
            e->rd_host[s2] = e->inflight_rd_pred;
            g_seam[8] += 1;
        } else {
            return harvest(e, true);                           // not predictable after all: wait for it now
        }
        return 0;
    }
    if (e->prof_on) { HIPCHK(hipStreamSynchronize(ln.main)); prof_collect(e); }
    return 0;
}

// Submit one block step of stream s (its window is complete by the mirror) and the report kernel behind it.
// ahead: a step of the same stream is still in flight (FINE at its start, no P1 decode): this one is queued behind it.
static int submit_step(nrsc5hip_engine *e, int s, bool ahead = false)
{
    nrsc5hip_engine::Lane &ln = e->lane;
    const int *ids_dev = e->all_ids_dev + s;                   // identity list: entry s is s
    const bool am = e->mode_host[s] == MODE_AM;
    const unsigned long long sig = set_signature(1, &s);
    if (sig != ln.set_sig) { ln.acq_needed = true; ln.px_needed = true; ln.set_sig = sig; }
    ln.prepared_by_sync = false;
    const auto t_enq = std::chrono::steady_clock::now();
    if (e->ingest_dirty) { HIPCHK(hipEventRecord(e->ev_ingest, e->ingest)); HIPCHK(hipStreamWaitEvent(ln.main, e->ev_ingest, 0)); e->ingest_dirty = false; }
    e->main_stepped = true;
    if (!e->counters_clean) HIPCHK(hipMemsetAsync(ln.counters_dev, 0, 4 * sizeof(int), ln.main));
    // A P1 frame completes only in a block that starts FINE with block count 15 (k_sync: started_pm && bc == 15; a block that
    // locks restarts the frame): when the stream's last record says otherwise the three decode launches are left out.
    const bool known = e->pred_ok[s] && !e->cfg.l2_feedback;
    bool decode = true, have_rep = false;
    StepReport rep{};
    if (am) {
        ProfScope p(e, NRSC5HIP_PROF_AM, ln.main);
        launch_am_step(e->tb, ln.db, 1, ids_dev, ln.main, e->cfg.l2_feedback, -1, (int)(ln.am_step_count % 8), (int)(ln.am_step_count / 8));
        ln.am_step_count++;
    } else {
        // (ahead: the block in flight runs FINE with block count pred_bc and ends no frame, so this one runs with pred_bc + 1)
        const int bc = ahead ? (e->pred_bc[s] + 1) % 16 : e->pred_bc[s];
        decode = !(known && bc != 15);
        if (!decode) g_seam[10] += 1;
        rep = next_report(e, s); have_rep = true;
        int rc = issue_step(e, ln, 1, ids_dev, decode, false, known && e->fuse_seam_prepare, e->fold_report ? &rep : nullptr); if (rc) return rc;   // PIDS frame: inside k_sync (pids_inline)
    }
    if (have_rep && rep.folded) { e->counters_clean = true; e->reports_folded++; }        // k_sync posted it
    else { int rc = launch_report(e, s, false, have_rep ? &rep : nullptr); if (rc) return rc; }    // FM: the PIDS frame was decoded inside k_sync; AM: inside its block kernel
    g_seam[2] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_enq).count();
    g_seam[6] += 1;
    if (ahead) { e->ahead.valid = true; e->ahead.stream = s; e->ahead.seq = e->report_seq; e->ahead.decoded = decode; g_seam[12] += 1; return 0; }
    e->inflight_stream = s; e->inflight_decoded = decode; e->inflight_rd_pred = -1; e->inflight_seq = e->report_seq;
    if (!am && known && e->defer_wait) {
        // the block starts FINE: samperr = 1080 + the previous block's feedback, keep = 2160 + (1080 - samperr), no keep_extra
        // (acquire.c:112,259; k_sync's tail): the mirror moves now, the report is taken when somebody needs it
        e->inflight_rd_pred = e->rd_host[s] + WIN_N - SYM_N + e->pred_samperr[s];
        e->rd_host[s] = e->inflight_rd_pred;
        g_seam[8] += 1;
    }
    return 0;
}

// Fast seam: block steps of one stream while the host mirror says a window is complete.  A step whose outcome the mirror can
// predict stays in flight when this returns (harvest takes it); any other is waited for here, as before.
static int stream_steps(nrsc5hip_engine *e, int s)
{
    int guard = 0;
    while (e->wr_host[s] - e->rd_host[s] >= window_of(e, s)) {
        int rc = 0;
        while (e->inflight_stream >= 0) if ((rc = harvest(e, true))) return rc;     // nothing in flight when a step is submitted here
        if (e->wr_host[s] - e->rd_host[s] < window_of(e, s)) break;
        if ((rc = submit_step(e, s))) return rc;
        if (e->inflight_rd_pred >= 0) continue;                // deferred: the mirror already shows the block consumed
        if ((rc = harvest(e, true))) return rc;
        if (!e->inflight_progress || ++guard > 64) break;      // nothing was processed or is pending
    }
    return 0;
}

static int settle(nrsc5hip_engine *e)
{
    if (!e) return 0;
    while (e->inflight_stream >= 0) { int rc = harvest(e, true); if (rc) return rc; }
    if (e->ingest_dirty) { HIPCHK(hipStreamSynchronize(e->ingest)); e->ingest_dirty = false; }    // whatever follows runs on `main` (or the host) alone
    return 0;
}

// how many input BYTES of this format complete the stream's next block (the drop-in pushes exactly that much, so that the L2
// feedback of the block's frames reaches the engine before the next block); -1: not known (the stream is not driven by the
// streaming seam alone, or p1_async)
extern "C" long long nrsc5hip_bytes_to_next_block(nrsc5hip_engine *e, int stream, int cu8)
{
    if (!e || stream < 0 || stream >= e->cfg.max_streams || !e->mirror_ok[stream]) return -1;
    if (e->ahead.valid) { DeviceGuard guard(e->cfg.device); if (harvest(e, true)) return -1; }     // the mirror lacks the step submitted ahead until its predecessor is harvested
    long long need = window_of(e, stream) - (e->wr_host[stream] - e->rd_host[stream]);     // decimated samples
    if (need < 1) need = 1;
    if (!cu8) return need * 4;                                                             // cs16: 4 bytes per complex sample
    if (e->mode_host[stream] != MODE_AM) return need * 4;                                  // FM cu8: 2 raw samples of 2 bytes each
    const long long raw = e->raw_host[stream];                                             // AM cu8: output k appears with raw sample 32 k + 31
    return 2 * ((raw / 32 + need) * 32 - raw);
}

// ---- streaming seam ---------------------------------------------------------------------------------------
// submit the staged samples of the fast seam: one async H2D from pinned memory ([count (u32), pad to 16][samples]) + the decimator
static int flush_staged(nrsc5hip_engine *e)
{
    const int s = e->staged_stream;
    if (s < 0 || e->staged_bytes == 0) { e->staged_stream = -1; return 0; }
    SeamClock clk(1); g_seam[5] += 1;
    const int slot = e->stage_slot;
    const bool cu8 = e->staged_cu8, am = e->mode_host[s] == MODE_AM;
    const size_t chunk = e->staged_bytes;
    const unsigned count = cu8 ? (unsigned)chunk : (unsigned)(chunk / 2);
    e->staged_stream = -1; e->staged_bytes = 0; e->staged_q15 = 0;
    e->stage_slot = (slot + 1) % nrsc5hip_engine::NSTAGE;     // the next pushes fill the next buffer
    const bool direct = cu8 && !am && e->direct_decimate;
    // (measured, profiles/r04_dropin_timeline.txt: with the block's last chunk on the step stream instead -- no dependency across two
    // queues in front of the step -- the decimator's own ~11 us of PCIe round trips sit on the chain and the drop-in is slower, 870 x
    // against 990 x; every chunk of the direct decimator goes on the ingest stream)
    const bool on_ingest = direct;
    if (!on_ingest && e->ingest_dirty) {                       // the FIFO is appended to in submission order whichever stream does it
        HIPCHK(hipEventRecord(e->ev_ingest, e->ingest)); HIPCHK(hipStreamWaitEvent(e->main, e->ev_ingest, 0)); e->ingest_dirty = false;
    }
    if (on_ingest && e->main_appended) {                       // ... and the next block's first chunk goes behind this block's last
        HIPCHK(hipStreamWaitEvent(e->ingest, e->ev_appended, 0)); e->main_appended = false;   // (recorded right behind that chunk: not behind the step that followed it)
    }
    int rc = ensure_space(e, s, 0, on_ingest); if (rc) return rc; // wr_host already counts the staged samples
    memcpy(e->stage_pin[slot], &count, sizeof(count));
    hipStream_t used = e->main;
    if (direct) {
        // FM cu8: the decimator reads the pinned buffer itself (one launch: no copy, no commit kernel)
        if (on_ingest) { used = e->ingest; e->ingest_dirty = true; }
        launch_decimate_fm_cu8_stream(e->tb, e->db, s, e->stage_pin_dev[slot] + 16, count, e->decim_ticket, used);
    } else {
        HIPCHK(hipMemcpyAsync(e->stage_dev2[slot], e->stage_pin[slot], chunk + 16, hipMemcpyHostToDevice, e->main));
        const int *ids_dev = e->all_ids_dev + s; const unsigned *count_dev = (const unsigned *)e->stage_dev2[slot]; const uint8_t *data_dev = e->stage_dev2[slot] + 16;
        if (cu8 && am) launch_am_decimate_cu8(e->tb, e->db, 1, ids_dev, data_dev, 0, count_dev, count, e->main);
        else if (cu8) launch_decimate_fm_cu8(e->tb, e->db, 1, ids_dev, data_dev, 0, count_dev, count, e->main);
        else launch_append_cs16(e->db, 1, ids_dev, (const int16_t *)data_dev, 0, count_dev, count, e->main);
    }
    if (!on_ingest) { HIPCHK(hipEventRecord(e->ev_appended, e->main)); e->main_appended = true; }
    HIPCHK(hipEventRecord(e->stage_ev[slot], used)); e->stage_busy[slot] = true;
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- host-resident capture (see nrsc5hip_engine::hc_*) ------------------------------------------------------------
static int push_common(nrsc5hip_engine *e, int s, const void *host, size_t nbytes_total, bool cu8);

// complex input sample k of the bound stream's session (k >= -14: its decimator history) as the Q15 pair the reference's decimator holds (U8_Q15, defines.h:93)
static c16 hc_sample_q15(const nrsc5hip_engine *e, long long k)
{
    const uint8_t *b = e->hc_pin + (nrsc5hip_engine::HC_PREFIX + 2 * k - e->hc_abs0);
    c16 v; v.r = (int16_t)(((int)b[0] - 127) * 64); v.i = (int16_t)(((int)b[1] - 127) * 64);
    return v;
}

// what decim[0]'s last compaction inside the first n input samples of the session leaves at the front of its window (StaleWindows, nrsc5_dev.h; the device-side
// form is hb_roll_history, k_decimate.hip): false = no compaction in that span, `out` untouched
static bool hc_stale_hb(const nrsc5hip_engine *e, long long n, c16 out[14])
{
    const long long p = stale_start(0, n, 14);
    if (p == STALE_NONE) return false;
    for (int k = 0; k < 14; k++) out[k] = hc_sample_q15(e, p + k);
    return true;
}

// a freshly reset FM stream's first cu8 push: bind the buffer to it if its decimator history is expressible as input bytes (always, unless an AM session's
// >> 4 samples were left in decim[0]'s window)
static int hc_try_attach(nrsc5hip_engine *e, int s)
{
    uint8_t pre[nrsc5hip_engine::HC_PREFIX];
    memset(pre, 0x7f, sizeof(pre));
    for (int k = 0; k < 14; k++) {
        const c16 h = e->hb_hist_host[s][k];
        if ((h.r & 63) || (h.i & 63)) return 0;
        const int r = h.r / 64 + 127, i = h.i / 64 + 127;
        if (r < 0 || r > 255 || i < 0 || i > 255) return 0;
        pre[nrsc5hip_engine::HC_PREFIX - 28 + 2 * k] = (uint8_t)r; pre[nrsc5hip_engine::HC_PREFIX - 28 + 2 * k + 1] = (uint8_t)i;
    }
    int rc = settle(e); if (rc) return rc;
    HIPCHK(hipStreamSynchronize(e->main));
    memcpy(e->hc_pin, pre, sizeof(pre));
    e->hc_abs0 = 0; e->hc_wr = nrsc5hip_engine::HC_PREFIX;
    // wr, rd, base, raw are the first four members of StreamState: the stream reads the capture from dword HC_OFF on, and the HOST decides when a window is complete
    // (the fast seam steps a stream only when its mirror says so), so the device-side end of data is set out of reach
    struct { long long wr, rd, base; const uint8_t *raw; } head = { 1ll << 60, nrsc5hip_engine::HC_OFF, 0, e->hc_dev };
    static_assert(offsetof(StreamState, wr) == 0 && offsetof(StreamState, rd) == 8 && offsetof(StreamState, base) == 16 && offsetof(StreamState, raw) == 24, "StreamState head layout");
    HIPCHK(hipMemcpy(e->db.state + s, &head, sizeof(head), hipMemcpyHostToDevice));
    e->wr_host[s] = e->rd_host[s] = nrsc5hip_engine::HC_OFF; e->base_host[s] = 0;
    e->hc_stream = s; e->hc_attaches++;
    return 0;
}

// the buffer is full: the live tail -- HC_KEEP bytes behind the read position (the decimator taps, and what a reset needs to tell the stale window) up to the write
// position -- moves to the front, and the stream's `raw` with it.  Nothing may be reading: every step is harvested first.
static int hc_rebase(nrsc5hip_engine *e)
{
    const int s = e->hc_stream;
    int rc = settle(e); if (rc) return rc;
    long long from = (4 * e->rd_host[s] - nrsc5hip_engine::HC_KEEP) & ~63ll;
    if (from <= e->hc_abs0) FAIL(NRSC5HIP_EOVERFLOW, "stream %d: the pinned capture (%zu bytes) cannot hold one window", s, e->hc_cap);
    memmove(e->hc_pin, e->hc_pin + (from - e->hc_abs0), (size_t)(e->hc_wr - from));
    e->hc_abs0 = from;
    const uint8_t *raw = e->hc_dev - from;                     // dword d of the stream's numbering lives at raw + 4 d
    HIPCHK(hipStreamSynchronize(e->main));
    HIPCHK(hipMemcpy((char *)(e->db.state + s) + offsetof(StreamState, raw), &raw, sizeof(raw), hipMemcpyHostToDevice));
    e->hc_rebases++;
    return 0;
}

// Turn the bound stream back into a FIFO stream: the device forgets the capture at its read position -- FIFO empty there, decimator history = the 14 input samples in
// front of it, decim[0]'s stale-window bookkeeping as the streaming decimator would have left it -- and the bytes behind that position go through the ordinary
// seam again (pinned staging, decimator).  For callers that leave what the capture can express: a cs16 push into the session, the batch entry points.
static int hc_detach(nrsc5hip_engine *e)
{
    const int s = e->hc_stream;
    if (s < 0) return 0;
    int rc = settle(e); if (rc) return rc;
    HIPCHK(hipStreamSynchronize(e->main));
    const long long rd = e->rd_host[s], consumed = 2 * (rd - nrsc5hip_engine::HC_OFF);   // input samples in front of the read position
    struct { long long wr, rd, base; const uint8_t *raw; c16 hb_hist[14]; } head = { rd, rd, rd, nullptr, {} };
    static_assert(offsetof(StreamState, hb_hist) == 32, "StreamState head layout");
    for (int k = 0; k < 14; k++) head.hb_hist[k] = hc_sample_q15(e, consumed - 14 + k);
    HIPCHK(hipMemcpy(e->db.state + s, &head, sizeof(head), hipMemcpyHostToDevice));
    c16 sw[14];
    if (hc_stale_hb(e, consumed, sw)) HIPCHK(hipMemcpy((char *)(e->db.state + s) + offsetof(StreamState, stale) + offsetof(StaleWindows, hb), sw, sizeof(sw), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy((char *)(e->db.state + s) + offsetof(StreamState, stale) + offsetof(StaleWindows, hb_pushed), &consumed, sizeof(consumed), hipMemcpyHostToDevice));
    e->wr_host[s] = rd; e->base_host[s] = rd;
    e->hc_stream = -1; e->hc_detaches++;
    const long long tail0 = 4 * rd, ntail = e->hc_wr - tail0;
    if (ntail > 0) {
        // (the source is the pinned capture itself: nothing writes it while the stream is unbound)
        const bool keep = e->host_capture; e->host_capture = false;
        const char manual = e->manual_step[s]; e->manual_step[s] = 1;      // the re-push only restores the FIFO: it completes at most the window the caller has not stepped yet
        rc = push_common(e, s, e->hc_pin + (tail0 - e->hc_abs0), (size_t)ntail, true);
        e->manual_step[s] = manual; e->host_capture = keep;
        if (rc) return rc;
    }
    return 0;
}

static int push_common(nrsc5hip_engine *e, int s, const void *host, size_t nbytes_total, bool cu8)
{
    int rc = check_stream(e, s); if (rc) return rc;
    const uint8_t *src = (const uint8_t *)host;
    const size_t unit = 4;                                     // cu8: 2 complex samples; cs16: 1 complex sample
    if (nbytes_total % unit) FAIL(NRSC5HIP_EINVAL, "length must be a multiple of %zu bytes", unit);
    const bool am = e->mode_host[s] == MODE_AM;
    if (e->attached[s]) FAIL(NRSC5HIP_EINVAL, "stream %d reads a zero-copy capture: reset it before pushing samples", s);
    const bool fast = !e->cfg.p1_async && e->mirror_ok[s];
    if (e->ahead.valid && (rc = harvest(e, true))) return rc;  // the mirror lacks a step submitted ahead until its predecessor is harvested
    while (e->inflight_stream >= 0 && (!fast || e->inflight_stream != s)) if ((rc = harvest(e, true))) return rc;
    if (e->hc_stream == s && (!fast || !cu8 || am) && (rc = hc_detach(e))) return rc;      // the capture holds FM cu8 input of the fast seam, nothing else
    if (fast && cu8 && !am && e->host_capture && e->hc_stream < 0 && e->wr_host[s] == 0 && e->rd_host[s] == 0 && e->staged_stream != s && nbytes_total &&
        (rc = hc_try_attach(e, s))) return rc;
    const bool hc = e->hc_stream == s;
    if (fast && e->staged_stream >= 0 && (e->staged_stream != s || e->staged_cu8 != cu8) && (rc = flush_staged(e))) return rc;
    if (fast && e->manual_step[s] && e->wr_host[s] - e->rd_host[s] >= window_of(e, s) && (rc = stream_steps(e, s))) return rc;   // the caller did not step
    while (nbytes_total) {
        if (hc) {
            // host-resident capture: the bytes stay where this copy puts them; a block is stepped when the mirror says its window is complete
            size_t chunk = nbytes_total;
            const long long to_block = nrsc5hip_bytes_to_next_block(e, s, 1);      // (as below: block by block, whatever the size of the push)
            if (to_block > 0 && (size_t)to_block < chunk) chunk = (size_t)to_block;
            if ((size_t)(e->hc_wr - e->hc_abs0) + chunk > e->hc_cap && (rc = hc_rebase(e))) return rc;
            if ((size_t)(e->hc_wr - e->hc_abs0) + chunk > e->hc_cap) FAIL(NRSC5HIP_EOVERFLOW, "stream %d: the pinned capture (%zu bytes) is too small", s, e->hc_cap);
            { SeamClock clk(0); memcpy(e->hc_pin + (e->hc_wr - e->hc_abs0), src, chunk); }
            g_seam[4] += 1; g_seam[13] += 1;
            e->hc_wr += (long long)chunk; e->wr_host[s] += (long long)chunk / 4;
            src += chunk; nbytes_total -= chunk;
            if (e->wr_host[s] - e->rd_host[s] >= window_of(e, s)) {
                if (e->manual_step[s] && nbytes_total == 0) break;     // nrsc5hip_stream_step runs the block
                if ((rc = stream_steps(e, s))) return rc;
            }
            continue;
        }
        if (fast) {
            // stage in pinned memory; submit when the block completes (the mirror knows) or the buffer is full
            const int slot = e->stage_slot;
            if (e->staged_bytes == 0 && e->stage_busy[slot]) { SeamClock clk(1); HIPCHK(hipEventSynchronize(e->stage_ev[slot])); e->stage_busy[slot] = false; }
            const size_t room = e->stage_ring_bytes - e->staged_bytes;
            size_t chunk = nbytes_total > room ? room : nbytes_total;
            // never stage past the sample that completes the stream's next block: a large push is then processed block by block and
            // the FIFO never holds more than one window plus the carry of the last block, whatever q15_capacity is (>= 2 windows)
            const long long to_block = nrsc5hip_bytes_to_next_block(e, s, cu8 ? 1 : 0);
            if (to_block > 0 && (size_t)to_block < chunk) chunk = (size_t)to_block;
            long long nq15 = (long long)chunk / 4;
            if (am && cu8) nq15 = (e->raw_host[s] + (long long)chunk / 2) / 32 - e->raw_host[s] / 32;
            { SeamClock clk(0); memcpy(e->stage_pin[slot] + 16 + e->staged_bytes, src, chunk); }
            g_seam[4] += 1;
            e->staged_stream = s; e->staged_cu8 = cu8; e->staged_bytes += chunk; e->staged_q15 += nq15;
            e->wr_host[s] += nq15;
            if (am && cu8) e->raw_host[s] += (long long)chunk / 2;
            src += chunk; nbytes_total -= chunk;
            if (e->wr_host[s] - e->rd_host[s] >= window_of(e, s) || e->staged_bytes == e->stage_ring_bytes) {
                if ((rc = flush_staged(e))) return rc;
                if (e->manual_step[s] && nbytes_total == 0) break;     // samples are on their way to the FIFO; nrsc5hip_stream_step runs the block
                if ((rc = stream_steps(e, s))) return rc;
            } else if (e->early_flush && e->staged_bytes >= e->early_flush && cu8 && !am && e->direct_decimate) {
                if ((rc = flush_staged(e))) return rc;         // ahead of the block's end, beside the step that is running
            }
            continue;
        }
        const size_t chunk = nbytes_total > e->stage_bytes ? e->stage_bytes : nbytes_total;
        long long nq15 = (long long)chunk / 4;                  // FM cu8: 2:1; cs16: one complex sample per 4 bytes
        if (am && cu8) nq15 = (e->raw_host[s] + (long long)chunk / 2) / 32 - e->raw_host[s] / 32;
        if ((rc = ensure_space(e, s, nq15))) return rc;
        const unsigned count = cu8 ? (unsigned)chunk : (unsigned)(chunk / 2);
        e->mirror_ok[s] = 0; e->pending[s].clear(); e->fetched[s] = e->drained[s]; forget_prediction(e, s); e->counters_clean = false;
        HIPCHK(hipMemcpyAsync(e->stage_dev, src, chunk, hipMemcpyHostToDevice, e->main));
        HIPCHK(hipMemcpyAsync(e->ids_dev, &s, sizeof(int), hipMemcpyHostToDevice, e->main));
        HIPCHK(hipMemcpyAsync(e->nbytes_dev, &count, sizeof(unsigned), hipMemcpyHostToDevice, e->main));
        HIPCHK(hipStreamSynchronize(e->main));                 // &s / &count are stack temporaries
        if (cu8 && am) { launch_am_decimate_cu8(e->tb, e->db, 1, e->ids_dev, e->stage_dev, 0, e->nbytes_dev, count, e->main); e->raw_host[s] += (long long)chunk / 2; }
        else if (cu8) launch_decimate_fm_cu8(e->tb, e->db, 1, e->ids_dev, e->stage_dev, 0, e->nbytes_dev, count, e->main);
        else launch_append_cs16(e->db, 1, e->ids_dev, (const int16_t *)e->stage_dev, 0, e->nbytes_dev, count, e->main);
        e->wr_host[s] += nq15;
        int steps = 0;
        if (am) { if ((rc = run_steps_am(e, 1, e->ids_dev, 1 << 30, 1, &steps))) return rc; }
        else if ((rc = run_steps(e, 1, e->ids_dev, set_signature(1, &s), 1 << 30, 1, &steps))) return rc;
        src += chunk; nbytes_total -= chunk;
    }
    return 0;
}

extern "C" int nrsc5hip_push_cu8(nrsc5hip_engine *e, int stream, const uint8_t *iq, uint32_t nbytes)
{
    ON_ENGINE_DEVICE_FAST(e);
    return push_common(e, stream, iq, nbytes, true);
}
extern "C" int nrsc5hip_push_cs16(nrsc5hip_engine *e, int stream, const int16_t *iq, uint32_t n)
{
    ON_ENGINE_DEVICE_FAST(e);
    if (n % 2) FAIL(NRSC5HIP_EINVAL, "cs16 length must be even");
    return push_common(e, stream, iq, (size_t)n * 2, false);
}

// input_reset (input.c:126-138).  keep_windows: the reference's reset of a USED session -- firdecim_q15_reset rewinds the index of every FIR window
// and leaves its samples (firdecim_q15.c:53-56), so decim[0]'s first outputs and the acquisition filter's first 31 see what the last compaction of
// their windows left there (StaleWindows, nrsc5_dev.h).  Otherwise a fresh session (nrsc5_open_pipe: calloc'd windows).
static int reset_stream(nrsc5hip_engine *e, int stream, bool keep_windows)
{
    ON_ENGINE_DEVICE(e);
    int rc = check_stream(e, stream); if (rc) return rc;
    if (e->staged_stream == stream) {
        // samples not yet submitted die with the session -- but the reference's decimator has seen them (decimate_samples runs inside the push): when the
        // windows are kept they go through the decimator first (no block step: wr alone moves, and the reset below forgets it)
        if (keep_windows && e->staged_bytes && (rc = flush_staged(e))) return rc;
        e->staged_stream = -1; e->staged_bytes = 0; e->staged_q15 = 0;
    }
    // this engine's queues only (another session of the process keeps running)
    if (e->ingest) HIPCHK(hipStreamSynchronize(e->ingest));
    HIPCHK(hipStreamSynchronize(e->main));
    if (e->cfg.p1_async) { for (int k = 0; k < NAUX; k++) HIPCHK(hipStreamSynchronize(e->lane.aux[k])); HIPCHK(hipStreamSynchronize(e->dec_stream)); }
    StreamState st; init_state(st, e->mode_host[stream]);
    const bool was_hc = e->hc_stream == stream;
    if (keep_windows && !e->cfg.batch_zero_copy) {             // (zero-copy engines: every attach is an independent recording, read in place with byte-valued history)
        HIPCHK(hipMemcpy(&st.stale, (const char *)(e->db.state + stream) + offsetof(StreamState, stale), sizeof(st.stale), hipMemcpyDeviceToHost));
        // sync_reset (sync.c:810-830) leaves sync_t.samperr, .angle and .bc alone.  The FM path overwrites all three in the block that locks, before anything reads
        // them; the AM path never writes .angle, so the first synchronised block of an AM session after an FM one turns by the FM session's last angle
        // (acquire.c:115-118) -- and the block records of the un-synchronised blocks in between show the old values
        HIPCHK(hipMemcpy(&st.samperr, (const char *)(e->db.state + stream) + offsetof(StreamState, samperr), sizeof(st.samperr), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&st.angle, (const char *)(e->db.state + stream) + offsetof(StreamState, angle), sizeof(st.angle), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&st.bc, (const char *)(e->db.state + stream) + offsetof(StreamState, bc), sizeof(st.bc), hipMemcpyDeviceToHost));
        // a stream that read the pinned capture never ran the streaming decimator: what decim[0]'s window holds after every byte pushed in that session is told from the capture
        if (was_hc) (void)hc_stale_hb(e, (e->hc_wr - nrsc5hip_engine::HC_PREFIX) / 2, st.stale.hb);
        st.stale.hb_pushed = 0; st.stale.fir_pushed[0] = 0; st.stale.fir_pushed[1] = 0;
        memcpy(st.hb_hist, st.stale.hb, sizeof(st.hb_hist));
        memcpy(st.fir_hist, st.stale.fir[st.mode == MODE_AM ? MODE_AM : MODE_FM], sizeof(st.fir_hist));
    }
    HIPCHK(hipMemcpy(e->db.state + stream, &st, sizeof(st), hipMemcpyHostToDevice));
    if (e->db.am) {
        AmStream am; init_am_state(am);
        memcpy(am.seed[0], st.stale.hb, sizeof(am.seed[0]));   // zeros unless the windows were kept
        memcpy(am.seed[1], st.stale.am_stage, sizeof(st.stale.am_stage));
        HIPCHK(hipMemcpy(e->db.am + stream, &am, sizeof(am), hipMemcpyHostToDevice));
        HIPCHK(hipMemset(e->db.am_job + (size_t)stream * NWIN, 0, NWIN * sizeof(AmJob)));
        HIPCHK(hipMemset(e->db.am_pids_rec + (size_t)stream * NWIN * 8, 0xff, NWIN * 8 * sizeof(int)));
    }
    if (was_hc) e->hc_stream = -1;
    memcpy(e->hb_hist_host[stream].data(), st.hb_hist, sizeof(st.hb_hist));
    e->wr_host[stream] = 0; e->base_host[stream] = 0; e->drained[stream] = 0; e->raw_host[stream] = 0; e->attached[stream] = 0;
    e->rd_host[stream] = 0; e->fetched[stream] = 0; e->pending[stream].clear(); e->mirror_ok[stream] = e->cfg.p1_async ? 0 : 1;
    forget_prediction(e, stream);
    e->lane.acq_needed = true; e->lane.px_needed = true; e->lane.set_sig = 0;
    return 0;
}

extern "C" int nrsc5hip_stream_reset(nrsc5hip_engine *e, int stream) { return reset_stream(e, stream, true); }
extern "C" int nrsc5hip_stream_fresh(nrsc5hip_engine *e, int stream) { return reset_stream(e, stream, false); }

// nrsc5_set_mode -> input_set_mode (input.c:158-162): switch the stream's waveform and reset it
extern "C" int nrsc5hip_stream_set_mode(nrsc5hip_engine *e, int stream, int mode)
{
    ON_ENGINE_DEVICE(e);
    int rc = check_stream(e, stream); if (rc) return rc;
    if (mode != NRSC5HIP_MODE_FM && mode != NRSC5HIP_MODE_AM) FAIL(NRSC5HIP_EINVAL, "unknown mode %d", mode);
    if (mode == NRSC5HIP_MODE_AM && !e->db.am) FAIL(NRSC5HIP_EINVAL, "engine was created without am_enable");
    if (mode == NRSC5HIP_MODE_AM && e->db.q15_cap < 2 * AM_WIN) FAIL(NRSC5HIP_EINVAL, "q15_capacity too small");
    if (e->staged_stream == stream && e->staged_bytes && (rc = flush_staged(e))) return rc;     // bytes pushed in the old mode pass through the old mode's decimator (reset_stream)
    e->mode_host[stream] = mode;
    return nrsc5hip_stream_reset(e, stream);
}

__global__ void k_force_none(DevBuffers db, int s) { db.state[s].sync_state = SYNC_NONE; }

extern "C" int nrsc5hip_force_resync(nrsc5hip_engine *e, int stream)
{
    ON_ENGINE_DEVICE(e);
    int rc = check_stream(e, stream); if (rc) return rc;
    hipLaunchKernelGGL(k_force_none, dim3(1), dim3(1), 0, e->main, e->db, stream);
    forget_prediction(e, stream);
    e->lane.acq_needed = true; e->lane.px_needed = true; e->lane.set_sig = 0;
    HIPCHK(hipGetLastError());
    return 0;
}

// ---- batch path ----------------------------------------------------------------------------------------------
// the batch entry points move a stream's FIFO without the host mirror of the fast streaming seam: records are read from the device again
static int leave_mirror(nrsc5hip_engine *e, int n, const int *ids)
{
    for (int k = 0; k < n; k++) if (e->hc_stream >= 0 && (ids ? ids[k] : k) == e->hc_stream) { int rc = hc_detach(e); if (rc) return rc; }   // the batch kernels read a FIFO (or a capture of known length)
    if (e->staged_stream >= 0) { int rc = flush_staged(e); if (rc) return rc; }   // whatever a push left in the pinned buffer goes to the FIFO first
    for (int k = 0; k < n; k++) {
        const int s = ids ? ids[k] : k;
        if (s < 0 || s >= e->cfg.max_streams || !e->mirror_ok[s]) continue;
        e->mirror_ok[s] = 0; e->pending[s].clear(); e->fetched[s] = e->drained[s];
        forget_prediction(e, s);
    }
    e->counters_clean = false;
    return 0;
}

static int upload_ids(nrsc5hip_engine *e, int n, const int *ids, const uint32_t *counts, const int **ids_dev)
{
    if (n < 1 || n > e->cfg.max_streams) FAIL(NRSC5HIP_EINVAL, "nstreams %d out of range", n);
    if (ids) {
        for (int k = 0; k < n; k++) if (ids[k] < 0 || ids[k] >= e->cfg.max_streams) FAIL(NRSC5HIP_EINVAL, "stream id %d out of range", ids[k]);
        HIPCHK(hipMemcpy(e->ids_dev, ids, n * sizeof(int), hipMemcpyHostToDevice));
        *ids_dev = e->ids_dev;
    } else {
        // the identity set: every kernel resolves `ids ? ids[i] : i` (stream_of), and without the list the stream index costs no trip to memory in
        // front of the stream-state loads that depend on it (k_mixfft / k_sync begin with exactly that chain)
        *ids_dev = nullptr;
    }
    if (counts) HIPCHK(hipMemcpy(e->nbytes_dev, counts, n * sizeof(unsigned), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int nrsc5hip_batch_append_cu8(nrsc5hip_engine *e, int nstreams, const int *stream_ids,
                                         const uint8_t *dev_iq, long long stride_bytes, const uint32_t *nbytes)
{
    ON_ENGINE_DEVICE(e);
    if (!e || !dev_iq || !nbytes) FAIL(NRSC5HIP_EINVAL, "null argument");
    const int *ids_dev; int rc = upload_ids(e, nstreams, stream_ids, nbytes, &ids_dev); if (rc) return rc;
    if ((rc = leave_mirror(e, nstreams, stream_ids))) return rc;
    unsigned mx = 0;
    {
        int nam = 0;
        for (int k = 0; k < nstreams; k++) nam += e->mode_host[stream_ids ? stream_ids[k] : k] == MODE_AM;
        if (nam && nam != nstreams) FAIL(NRSC5HIP_EINVAL, "one append call must list streams of one mode (FM or AM)");
        if (nam) {
            for (int k = 0; k < nstreams; k++) {
                const int s = stream_ids ? stream_ids[k] : k;
                if (nbytes[k] % 4) FAIL(NRSC5HIP_EINVAL, "chunk %d: nbytes %% 4 != 0", k);
                const long long nout = (e->raw_host[s] + nbytes[k] / 2) / 32 - e->raw_host[s] / 32;
                if (e->wr_host[s] - e->base_host[s] + nout > e->db.q15_cap) FAIL(NRSC5HIP_EOVERFLOW, "stream %d: q15_capacity %lld too small for this batch", s, e->db.q15_cap);
                if (nbytes[k] > mx) mx = nbytes[k];
            }
            { ProfScope p(e, NRSC5HIP_PROF_DECIMATE, e->main); launch_am_decimate_cu8(e->tb, e->db, nstreams, ids_dev, dev_iq, stride_bytes, e->nbytes_dev, mx, e->main); }
            for (int k = 0; k < nstreams; k++) {
                const int s = stream_ids ? stream_ids[k] : k;
                e->wr_host[s] += (e->raw_host[s] + nbytes[k] / 2) / 32 - e->raw_host[s] / 32;
                e->raw_host[s] += nbytes[k] / 2;
            }
            HIPCHK(hipGetLastError());
            return 0;
        }
    }
    for (int k = 0; k < nstreams; k++) {
        const int s = stream_ids ? stream_ids[k] : k;
        if (e->attached[s]) FAIL(NRSC5HIP_EINVAL, "stream %d already reads a zero-copy capture (one append per reset)", s);
    }
    if (e->cfg.batch_zero_copy) {
        bool all_fresh = true;
        for (int k = 0; k < nstreams; k++) all_fresh = all_fresh && e->wr_host[stream_ids ? stream_ids[k] : k] == 0;
        if (all_fresh) {
            // zero-copy: the capture stays where it is; the block steps decimate what they read (k_mixfft, k_acq_decimate)
            if (((uintptr_t)dev_iq | (uintptr_t)stride_bytes) & 3) FAIL(NRSC5HIP_EINVAL, "zero-copy captures must be 4-byte aligned");
            for (int k = 0; k < nstreams; k++) if (nbytes[k] % 4) FAIL(NRSC5HIP_EINVAL, "chunk %d: nbytes %% 4 != 0", k);
            launch_attach_raw(e->db, nstreams, ids_dev, dev_iq, stride_bytes, e->nbytes_dev, e->main);
            for (int k = 0; k < nstreams; k++) { const int s = stream_ids ? stream_ids[k] : k; e->wr_host[s] += nbytes[k] / 4; e->attached[s] = 1; }
            HIPCHK(hipGetLastError());
            return 0;
        }
    }
    for (int k = 0; k < nstreams; k++) {
        const int s = stream_ids ? stream_ids[k] : k;
        if (nbytes[k] % 4) FAIL(NRSC5HIP_EINVAL, "chunk %d: nbytes %% 4 != 0", k);
        if (e->wr_host[s] - e->base_host[s] + nbytes[k] / 4 > e->db.q15_cap)
            FAIL(NRSC5HIP_EOVERFLOW, "stream %d: q15_capacity %lld too small for this batch", s, e->db.q15_cap);
        if (nbytes[k] > mx) mx = nbytes[k];
    }
    bool fresh = e->cfg.p1_async != 0 && stream_ids == nullptr && nstreams == e->cfg.max_streams;
    for (int k = 0; k < nstreams && fresh; k++) fresh = e->wr_host[k] == 0;
    fresh = fresh && e->lane.step_count == 0;
    const long long CH = 16 * 70199LL;                         // one decode window's worth of output samples
    if (fresh && (long long)mx / 4 > 3 * CH) {
        // Fresh batch in the pipelined mode: decimate window-sized chunks on a side stream so that K1 (HBM-bound)
        // overlaps the issue-bound block steps; the scheduler waits for the chunk a step can reach (run_steps_lanes).
        const int nch = (int)(((long long)mx / 4 + CH - 1) / CH);
        if (e->chunk_cap < nch * nstreams) {
            unsigned *p = nullptr;
            if (dev_alloc(e, &p, (size_t)nch * nstreams)) return NRSC5HIP_ENOMEM;
            e->chunk_nbytes_dev = p; e->chunk_cap = nch * nstreams;
        }
        std::vector<unsigned> cb((size_t)nch * nstreams);
        for (int c = 0; c < nch; c++)
            for (int k = 0; k < nstreams; k++) {
                const long long lo = 4 * CH * c, left = (long long)nbytes[k] - lo;
                cb[(size_t)c * nstreams + k] = (unsigned)(left <= 0 ? 0 : (left > 4 * CH ? 4 * CH : left));
            }
        HIPCHK(hipMemcpy(e->chunk_nbytes_dev, cb.data(), cb.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        while ((int)e->dec_events.size() < nch) { hipEvent_t ev; HIPCHK(hipEventCreate(&ev)); e->dec_events.push_back(ev); }
        hipEvent_t start; HIPCHK(hipEventCreate(&start));
        HIPCHK(hipEventRecord(start, e->main));                // after whatever the caller/engine queued before (reset)
        HIPCHK(hipStreamWaitEvent(e->dec_stream, start, 0));
        (void)hipEventDestroy(start);
        for (int c = 0; c < nch; c++) {
            ProfScope p(e, NRSC5HIP_PROF_DECIMATE, e->dec_stream);
            launch_decimate_fm_cu8(e->tb, e->db, nstreams, ids_dev, dev_iq + 4 * CH * c, stride_bytes,
                                   e->chunk_nbytes_dev + (size_t)c * nstreams, (unsigned)(4 * CH), e->dec_stream);
            HIPCHK(hipEventRecord(e->dec_events[c], e->dec_stream));
        }
        e->dec_chunk = CH;
        e->lane.dec_waited = 0;
    } else {
        ProfScope p(e, NRSC5HIP_PROF_DECIMATE, e->main);
        launch_decimate_fm_cu8(e->tb, e->db, nstreams, ids_dev, dev_iq, stride_bytes, e->nbytes_dev, mx, e->main);
    }
    for (int k = 0; k < nstreams; k++) e->wr_host[stream_ids ? stream_ids[k] : k] += nbytes[k] / 4;
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int nrsc5hip_batch_append_cs16(nrsc5hip_engine *e, int nstreams, const int *stream_ids,
                                          const int16_t *dev_iq, long long stride_elems, const uint32_t *nelems)
{
    ON_ENGINE_DEVICE(e);
    if (!e || !dev_iq || !nelems) FAIL(NRSC5HIP_EINVAL, "null argument");
    const int *ids_dev; int rc = upload_ids(e, nstreams, stream_ids, nelems, &ids_dev); if (rc) return rc;
    if ((rc = leave_mirror(e, nstreams, stream_ids))) return rc;
    unsigned mx = 0;
    for (int k = 0; k < nstreams; k++) {
        const int s = stream_ids ? stream_ids[k] : k;
        if (nelems[k] % 2) FAIL(NRSC5HIP_EINVAL, "chunk %d: odd cs16 length", k);
        if (e->attached[s]) FAIL(NRSC5HIP_EINVAL, "stream %d reads a zero-copy capture: reset it before appending samples", s);
        if (e->wr_host[s] - e->base_host[s] + nelems[k] / 2 > e->db.q15_cap)
            FAIL(NRSC5HIP_EOVERFLOW, "stream %d: q15_capacity %lld too small for this batch", s, e->db.q15_cap);
        if (nelems[k] > mx) mx = nelems[k];
    }
    launch_append_cs16(e->db, nstreams, ids_dev, dev_iq, stride_elems, e->nbytes_dev, mx, e->main);
    for (int k = 0; k < nstreams; k++) e->wr_host[stream_ids ? stream_ids[k] : k] += nelems[k] / 2;
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int nrsc5hip_batch_process(nrsc5hip_engine *e, int nstreams, const int *stream_ids, int max_steps, int *steps_done)
{
    ON_ENGINE_DEVICE(e);
    if (!e) FAIL(NRSC5HIP_EINVAL, "null engine");
    if (nstreams < 1 || nstreams > e->cfg.max_streams) FAIL(NRSC5HIP_EINVAL, "nstreams %d out of range", nstreams);
    if (stream_ids) for (int k = 0; k < nstreams; k++) if (stream_ids[k] < 0 || stream_ids[k] >= e->cfg.max_streams) FAIL(NRSC5HIP_EINVAL, "stream id %d out of range", stream_ids[k]);
    { int rc = leave_mirror(e, nstreams, stream_ids); if (rc) return rc; }
    {   // AM streams advance through their own fused block kernel; split a mixed list by mode
        std::vector<int> fm, am;
        for (int k = 0; k < nstreams; k++) {
            const int s = stream_ids ? stream_ids[k] : k;
            if (s < 0 || s >= e->cfg.max_streams) FAIL(NRSC5HIP_EINVAL, "stream id %d out of range", s);
            (e->mode_host[s] == MODE_AM ? am : fm).push_back(s);
        }
        if (!am.empty()) {
            int done_am = 0, done_fm = 0;
            HIPCHK(hipMemcpy(e->ids_dev, am.data(), am.size() * sizeof(int), hipMemcpyHostToDevice));
            int rc = run_steps_am(e, (int)am.size(), e->ids_dev, max_steps > 0 ? max_steps : (1 << 30), e->cfg.p1_async ? 32 : 8, &done_am);
            if (rc) return rc;
            if (!fm.empty()) { rc = nrsc5hip_batch_process(e, (int)fm.size(), fm.data(), max_steps, &done_fm); if (rc) return rc; }
            if (steps_done) *steps_done = done_am > done_fm ? done_am : done_fm;
            return 0;
        }
    }
    const int *ids_dev; int rc = upload_ids(e, nstreams, stream_ids, nullptr, &ids_dev); if (rc) return rc;
    return run_steps(e, nstreams, ids_dev, set_signature(nstreams, stream_ids), max_steps > 0 ? max_steps : (1 << 30), e->cfg.p1_async ? 16 : 8, steps_done);
}

// ---- results ------------------------------------------------------------------------------------------------------
static int fetch_nblocks(nrsc5hip_engine *e, int s, int *nblocks)
{
    HIPCHK(hipMemcpy(nblocks, (const char *)(e->db.state + s) + offsetof(StreamState, nblocks), sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

// Window pipeline, AM: the BER of an L1 frame is known when the last of its nine deferred decodes finishes, after the
// record of its block 7 was written -- it is kept per ring slot and merged into the records handed to the caller.
static int patch_am_ber(nrsc5hip_engine *e, int stream, nrsc5hip_record *recs, int n, const float *ber_row)
{
    std::vector<float> tmp;
    if (!ber_row) {
        tmp.resize(e->db.p1_slots);
        HIPCHK(hipMemcpy(tmp.data(), e->db.am_ber + (size_t)stream * e->db.p1_slots, tmp.size() * sizeof(float), hipMemcpyDeviceToHost));
        ber_row = tmp.data();
    }
    for (int k = 0; k < n; k++)
        if ((recs[k].flags & NRSC5HIP_REC_P1) && recs[k].bc_decoded == 7 && recs[k].p1_slot >= 0 && recs[k].p1_slot < e->db.p1_slots)
            recs[k].ber = ber_row[recs[k].p1_slot];
    return 0;
}

extern "C" int nrsc5hip_drain(nrsc5hip_engine *e, int stream, nrsc5hip_record *out, int max, int *n_out)
{
    ON_ENGINE_DEVICE_FAST(e);
    int rc = check_stream(e, stream); if (rc) return rc;
    if (!out || !n_out) FAIL(NRSC5HIP_EINVAL, "null argument");
    // the block step in flight is waited for; samples that are still being decimated on the ingest stream are not (a drop-in session
    // drains right after it has submitted the last chunk of the next block: waiting for that kernel was ~20 us per block)
    if (e->inflight_stream >= 0 && (rc = harvest(e, true))) return rc;
    if (!e->mirror_ok[stream] && (rc = settle(e))) return rc;
    if (e->mirror_ok[stream]) {
        // fast streaming seam: every record of a finished block step is on the host already (k_stream_report)
        std::deque<BlockRecord> &q = e->pending[stream];
        int n = 0;
        for (; n < max && !q.empty(); n++) { memcpy(&out[n], &q.front(), sizeof(BlockRecord)); q.pop_front(); }
        e->drained[stream] += n;
        *n_out = n;
        return 0;
    }
    HIPCHK(hipStreamSynchronize(e->main));
    int nb = 0;
    if ((rc = fetch_nblocks(e, stream, &nb))) return rc;
    if (nb - e->drained[stream] > e->db.rec_cap) FAIL(NRSC5HIP_EOVERFLOW, "stream %d: %d records overwrote the ring (capacity %d)", stream, nb - e->drained[stream], e->db.rec_cap);
    const bool replay = e->db.ckpt || e->db.am_ckpt;
    int n = 0;
    // Replay: blocks that ran behind a failed P1 frame are void (k_replay.hip) and never delivered.  A rewind can void up to
    // NWIN * 16 records in a row, so keep reading until `max` valid records are collected or the ring is empty -- a caller that
    // loops "until fewer than max came back" must not stop at a chunk of void records.
    while (n < max && e->drained[stream] < nb) {
        const int want = std::min(max - n, nb - e->drained[stream]);
        const int first = e->drained[stream] % e->db.rec_cap;      // at most two contiguous pieces of the ring
        const int n1 = (first + want <= e->db.rec_cap) ? want : e->db.rec_cap - first;
        const BlockRecord *ring = e->db.records + (size_t)stream * e->db.rec_cap;
        HIPCHK(hipMemcpy(out + n, ring + first, (size_t)n1 * sizeof(BlockRecord), hipMemcpyDeviceToHost));
        if (want - n1 > 0) HIPCHK(hipMemcpy(out + n + n1, ring, (size_t)(want - n1) * sizeof(BlockRecord), hipMemcpyDeviceToHost));
        e->drained[stream] += want;
        int m = n;
        for (int k = n; k < n + want; k++) if (!replay || !(out[k].flags & NRSC5HIP_REC_DISCARDED)) { if (m != k) out[m] = out[k]; m++; }
        n = m;
    }
    *n_out = n;
    if (e->cfg.p1_async && e->db.am && e->mode_host[stream] == MODE_AM && n > 0) return patch_am_ber(e, stream, out, n, nullptr);
    return 0;
}

// The drop-in's form of drain: whatever has been reported so far, without waiting for a block step that is still running
extern "C" int nrsc5hip_drain_ready(nrsc5hip_engine *e, int stream, nrsc5hip_record *out, int max, int *n_out)
{
    ON_ENGINE_DEVICE_FAST(e);
    int rc = check_stream(e, stream); if (rc) return rc;
    if (!out || !n_out) FAIL(NRSC5HIP_EINVAL, "null argument");
    if (!e->mirror_ok[stream]) return nrsc5hip_drain(e, stream, out, max, n_out);
    if (e->inflight_stream >= 0 && (rc = harvest(e, false))) return rc;
    std::deque<BlockRecord> &q = e->pending[stream];
    int n = 0;
    for (; n < max && !q.empty(); n++) { memcpy(&out[n], &q.front(), sizeof(BlockRecord)); q.pop_front(); }
    e->drained[stream] += n;
    *n_out = n;
    return 0;
}

extern "C" int nrsc5hip_stream_set_manual_step(nrsc5hip_engine *e, int stream, int on)
{
    ON_ENGINE_DEVICE(e);
    int rc = check_stream(e, stream); if (rc) return rc;
    e->manual_step[stream] = on ? 1 : 0;
    return 0;
}

// manual-step streams: run the block(s) whose window the pushes so far completed (the step may stay in flight: drain waits for it,
// drain_ready does not)
extern "C" int nrsc5hip_stream_step(nrsc5hip_engine *e, int stream)
{
    ON_ENGINE_DEVICE_FAST(e);
    int rc = check_stream(e, stream); if (rc) return rc;
    if (e->cfg.p1_async || !e->mirror_ok[stream]) FAIL(NRSC5HIP_EINVAL, "stream %d is not driven by the fast streaming seam", stream);
    if (e->inflight_stream >= 0 && e->inflight_stream != stream && (rc = harvest(e, true))) return rc;
    if (e->staged_stream == stream && (rc = flush_staged(e))) return rc;
    return stream_steps(e, stream);
}

// manual-step streams: submit the block the pushes so far completed BEHIND the step still in flight, if that is safe; *submitted
// tells.  0: the caller drains, feeds L2 and calls nrsc5hip_stream_step as usual.
extern "C" int nrsc5hip_stream_step_ahead(nrsc5hip_engine *e, int stream, int *submitted)
{
    ON_ENGINE_DEVICE_FAST(e);
    int rc = check_stream(e, stream); if (rc) return rc;
    if (!submitted) FAIL(NRSC5HIP_EINVAL, "null argument");
    *submitted = 0;
    if (e->cfg.p1_async || !e->mirror_ok[stream] || !e->manual_step[stream] || !e->defer_wait || e->cfg.l2_feedback || e->prof_on) return 0;
    if (e->mode_host[stream] == MODE_AM || e->ahead.valid) return 0;
    // the step in flight: same stream, started FINE (predicted), no P1 decode -> its delivery cannot send the stream back to NONE
    if (e->inflight_stream != stream || e->inflight_rd_pred < 0 || e->inflight_decoded || !e->pred_ok[stream]) return 0;
    if (e->staged_stream == stream && (rc = flush_staged(e))) return rc;
    if (e->wr_host[stream] - e->rd_host[stream] < window_of(e, stream)) return 0;
    if ((rc = submit_step(e, stream, true))) return rc;
    *submitted = 1;
    return 0;
}

extern "C" int nrsc5hip_p1_frame_packed(nrsc5hip_engine *e, int stream, int slot, uint32_t *words)
{
    ON_ENGINE_DEVICE_FAST(e);
    SeamClock clk(7);
    int rc = check_stream(e, stream); if (rc) return rc;
    if (slot < 0 || slot >= e->db.p1_slots || !words) FAIL(NRSC5HIP_EINVAL, "bad slot/argument");
    if (e->inflight_stream >= 0 && (rc = harvest(e, true))) return rc;
    if (e->mirror_ok[stream] && e->db.p1_mirror && e->frames_host && e->mode_host[stream] != MODE_AM) {
        // fast seam (FM): the step that decoded the frame has been harvested, and its traceback wrote the frame into the pinned
        // mirror before the report kernel that the harvest waited for
        memcpy(words, e->frames_host + ((size_t)stream * e->db.p1_slots + slot) * P1_WORDS, P1_WORDS * sizeof(uint32_t));
        return 0;
    }
    if ((rc = settle(e))) return rc;
    HIPCHK(hipStreamSynchronize(e->main));
    HIPCHK(hipMemcpy(words, e->db.p1_ring + ((size_t)stream * e->db.p1_slots + slot) * P1_WORDS, P1_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" void nrsc5hip_unpack_bits(const uint32_t *words, int nbits, uint8_t *bits)
{
    // one byte of packed bits -> eight bytes through a table (a P1 frame is 146 176 bits: bit by bit this was ~0.1 ms of the drop-in's
    // host time per frame)
    static const struct Lut { uint64_t v[256]; Lut() { for (int b = 0; b < 256; b++) { uint64_t x = 0; for (int k = 0; k < 8; k++) x |= (uint64_t)((b >> k) & 1) << (8 * k); v[b] = x; } } } lut;
    const uint8_t *src = (const uint8_t *)words;               // little-endian host: bit i of the frame = bit i % 8 of byte i / 8
    int i = 0;
    for (; i + 8 <= nbits; i += 8) memcpy(bits + i, &lut.v[src[i >> 3]], 8);
    for (; i < nbits; i++) bits[i] = (words[i >> 5] >> (i & 31)) & 1u;
}

extern "C" int nrsc5hip_p1_frame_bits(nrsc5hip_engine *e, int stream, int slot, uint8_t *bits)
{
    ON_ENGINE_DEVICE_FAST(e);
    std::vector<uint32_t> w(P1_WORDS);
    int rc = nrsc5hip_p1_frame_packed(e, stream, slot, w.data()); if (rc) return rc;
    nrsc5hip_unpack_bits(w.data(), P1_LEN, bits);
    return 0;
}

// FM extended sidebands: P3 (channel 0) / P4 (channel 1) frame of a REC_P3 / REC_P4 record; slot = record.sis
extern "C" int nrsc5hip_px_frame_bits(nrsc5hip_engine *e, int stream, int slot, int channel, int nbits, uint8_t *bits)
{
    ON_ENGINE_DEVICE(e);
    int rc = check_stream(e, stream); if (rc) return rc;
    if (slot < 0 || slot >= e->db.px_slots || channel < 0 || channel > 1 || !bits || (nbits != 2304 && nbits != 4608)) FAIL(NRSC5HIP_EINVAL, "bad slot/argument");
    uint32_t w[PX_WORDS];
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(w, e->db.px_ring + (((size_t)stream * e->db.px_slots + slot) * 2 + channel) * PX_WORDS, (nbits / 32) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    nrsc5hip_unpack_bits(w, nbits, bits);
    return 0;
}

// bulk variant: all P3/P4 slots of the listed streams, [nstreams][8 * p1_slots][2][144] words
extern "C" int nrsc5hip_batch_fetch_px(nrsc5hip_engine *e, int nstreams, const int *stream_ids, uint32_t *frames)
{
    ON_ENGINE_DEVICE(e);
    if (!e || !frames) FAIL(NRSC5HIP_EINVAL, "null argument");
    HIPCHK(hipDeviceSynchronize());
    const size_t per = (size_t)e->db.px_slots * 2 * PX_WORDS;
    for (int k = 0; k < nstreams; k++) {
        const int s = stream_ids ? stream_ids[k] : k;
        int rc = check_stream(e, s); if (rc) return rc;
        HIPCHK(hipMemcpy(frames + (size_t)k * per, e->db.px_ring + (size_t)s * per, per * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    return 0;
}

// AM: frames of one L1 frame share a ring slot: P1 frame of block b at word b * 118, the P3 frame at word 944
extern "C" int nrsc5hip_am_frame_bits(nrsc5hip_engine *e, int stream, int slot, int which, int nbits, uint8_t *bits)
{
    ON_ENGINE_DEVICE(e);
    int rc = check_stream(e, stream); if (rc) return rc;
    if (slot < 0 || slot >= e->db.p1_slots || !bits || which < 0 || which > 8) FAIL(NRSC5HIP_EINVAL, "bad slot/argument");
    const int maxbits = which < 8 ? AM_P1_LEN : AM_P3_LEN_MA3;
    if (nbits < 1 || nbits > maxbits) FAIL(NRSC5HIP_EINVAL, "nbits %d out of range", nbits);
    const int word0 = which < 8 ? which * AM_P1_WORDS : AM_P3_WORD0, words = (nbits + 31) / 32;
    std::vector<uint32_t> w(words);
    HIPCHK(hipStreamSynchronize(e->main));
    HIPCHK(hipMemcpy(w.data(), e->db.p1_ring + ((size_t)stream * e->db.p1_slots + slot) * P1_WORDS + word0, words * sizeof(uint32_t), hipMemcpyDeviceToHost));
    nrsc5hip_unpack_bits(w.data(), nbits, bits);
    return 0;
}

// ---- L2 audio transport index ---------------------------------------------------------------------------------------
struct DevTmp { void *p = nullptr; ~DevTmp() { if (p) (void)hipFree(p); } };

static int l2_run(nrsc5hip_engine *e, const std::vector<L2Job> &jobs, nrsc5hip_l2_frame *out, uint8_t *pdu_bytes, long long stride)
{
    const int n = (int)jobs.size();
    if (pdu_bytes && stride < L2_MAX_BYTES) {
        for (const L2Job &j : jobs) if ((j.nbits - 22) / 8 > stride) FAIL(NRSC5HIP_EINVAL, "stride %lld too small for a %d-bit frame", stride, j.nbits);
    }
    DevTmp tj, to, tb;                                  // freed on every return path
    HIPCHK(hipMalloc(&tj.p, sizeof(L2Job) * n));
    HIPCHK(hipMalloc(&to.p, sizeof(nrsc5hip_l2_frame) * n));
    if (pdu_bytes) HIPCHK(hipMalloc(&tb.p, (size_t)stride * n));
    L2Job *djobs = (L2Job *)tj.p; nrsc5hip_l2_frame *dout = (nrsc5hip_l2_frame *)to.p; uint8_t *dbytes = (uint8_t *)tb.p;
    HIPCHK(hipMemcpy(djobs, jobs.data(), sizeof(L2Job) * n, hipMemcpyHostToDevice));
    launch_l2_index(djobs, n, dout, dbytes, stride, e->main);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->main));
    HIPCHK(hipMemcpy(out, dout, sizeof(nrsc5hip_l2_frame) * n, hipMemcpyDeviceToHost));
    if (pdu_bytes) HIPCHK(hipMemcpy(pdu_bytes, dbytes, (size_t)stride * n, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int nrsc5hip_l2_index(nrsc5hip_engine *e, int njobs, const nrsc5hip_l2_job *jobs, nrsc5hip_l2_frame *out, uint8_t *pdu_bytes, long long stride)
{
    ON_ENGINE_DEVICE(e);
    if (!e || !jobs || !out || njobs < 1) FAIL(NRSC5HIP_EINVAL, "bad argument");
    std::vector<L2Job> dj((size_t)njobs);
    for (int k = 0; k < njobs; k++) {
        const nrsc5hip_l2_job &j = jobs[k];
        int rc = check_stream(e, j.stream); if (rc) return rc;
        const uint32_t *words = nullptr;
        if (j.kind == NRSC5HIP_L2_FM_P1) {
            if (j.slot < 0 || j.slot >= e->db.p1_slots || j.nbits != P1_LEN) FAIL(NRSC5HIP_EINVAL, "job %d: bad P1 slot / length", k);
            words = e->db.p1_ring + ((size_t)j.stream * e->db.p1_slots + j.slot) * P1_WORDS;
        } else if (j.kind == NRSC5HIP_L2_FM_PX) {
            if (j.slot < 0 || j.slot >= e->db.px_slots || j.which < 0 || j.which > 1 || (j.nbits != 2304 && j.nbits != 4608)) FAIL(NRSC5HIP_EINVAL, "job %d: bad P3/P4 slot / channel / length", k);
            words = e->db.px_ring + (((size_t)j.stream * e->db.px_slots + j.slot) * 2 + j.which) * PX_WORDS;
        } else if (j.kind == NRSC5HIP_L2_AM) {
            const bool p1 = j.which >= 0 && j.which < 8 && j.nbits == AM_P1_LEN;
            const bool p3 = j.which == 8 && (j.nbits == AM_P3_LEN_MA1 || j.nbits == AM_P3_LEN_MA3);
            if (j.slot < 0 || j.slot >= e->db.p1_slots || !(p1 || p3)) FAIL(NRSC5HIP_EINVAL, "job %d: bad AM slot / frame / length", k);
            words = e->db.p1_ring + ((size_t)j.stream * e->db.p1_slots + j.slot) * P1_WORDS + (p1 ? j.which * AM_P1_WORDS : AM_P3_WORD0);
        } else FAIL(NRSC5HIP_EINVAL, "job %d: unknown kind %d", k, j.kind);
        dj[k] = L2Job{words, j.nbits, 0};
    }
    HIPCHK(hipDeviceSynchronize());                 // the frames may still be in flight on a decode stream
    return l2_run(e, dj, out, pdu_bytes, stride);
}

extern "C" int nrsc5hip_l2_frame_get(nrsc5hip_engine *e, int stream, int slot, nrsc5hip_l2_frame *out)
{
    ON_ENGINE_DEVICE(e);
    int rc = check_stream(e, stream); if (rc) return rc;
    if (!e->db.l2_ring) FAIL(NRSC5HIP_EINVAL, "engine was created without l2_index");
    if (slot < 0 || slot >= e->db.p1_slots || !out) FAIL(NRSC5HIP_EINVAL, "bad slot/argument");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, e->db.l2_ring + (size_t)stream * e->db.p1_slots + slot, sizeof(*out), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int nrsc5hip_batch_fetch_l2(nrsc5hip_engine *e, int nstreams, const int *stream_ids, nrsc5hip_l2_frame *out)
{
    ON_ENGINE_DEVICE(e);
    if (!e || !out || nstreams < 1) FAIL(NRSC5HIP_EINVAL, "bad argument");
    if (!e->db.l2_ring) FAIL(NRSC5HIP_EINVAL, "engine was created without l2_index");
    HIPCHK(hipDeviceSynchronize());
    const size_t per = (size_t)e->db.p1_slots;
    bool contiguous = true;
    for (int k = 0; k < nstreams; k++) {
        const int s = stream_ids ? stream_ids[k] : k;
        int rc = check_stream(e, s); if (rc) return rc;
        if (s != (stream_ids ? stream_ids[0] : 0) + k) contiguous = false;
    }
    if (contiguous) {
        HIPCHK(hipMemcpy(out, e->db.l2_ring + (size_t)(stream_ids ? stream_ids[0] : 0) * per, (size_t)nstreams * per * sizeof(*out), hipMemcpyDeviceToHost));
    } else {
        for (int k = 0; k < nstreams; k++)
            HIPCHK(hipMemcpy(out + (size_t)k * per, e->db.l2_ring + (size_t)stream_ids[k] * per, per * sizeof(*out), hipMemcpyDeviceToHost));
    }
    return 0;
}

static int fetch_l2_ring(nrsc5hip_engine *e, const nrsc5hip_l2_frame *ring, size_t per, int nstreams, const int *stream_ids, nrsc5hip_l2_frame *out, const char *what)
{
    if (!e || !out || nstreams < 1) FAIL(NRSC5HIP_EINVAL, "bad argument");
    if (!ring) FAIL(NRSC5HIP_EINVAL, "engine was created without l2_index%s", what);
    HIPCHK(hipDeviceSynchronize());
    for (int k = 0; k < nstreams; k++) {
        const int s = stream_ids ? stream_ids[k] : k;
        int rc = check_stream(e, s); if (rc) return rc;
        HIPCHK(hipMemcpy(out + (size_t)k * per, ring + (size_t)s * per, per * sizeof(*out), hipMemcpyDeviceToHost));
    }
    return 0;
}
extern "C" int nrsc5hip_batch_fetch_l2_px(nrsc5hip_engine *e, int nstreams, const int *stream_ids, nrsc5hip_l2_frame *out)
{
    ON_ENGINE_DEVICE(e);
    return fetch_l2_ring(e, e ? e->db.l2_px_ring : nullptr, e ? (size_t)e->db.px_slots * 2 : 0, nstreams, stream_ids, out, "");
}
extern "C" int nrsc5hip_batch_fetch_l2_am(nrsc5hip_engine *e, int nstreams, const int *stream_ids, nrsc5hip_l2_frame *out)
{
    ON_ENGINE_DEVICE(e);
    return fetch_l2_ring(e, e ? e->db.l2_am_ring : nullptr, e ? (size_t)e->db.p1_slots * 9 : 0, nstreams, stream_ids, out, " and am_enable");
}

extern "C" int nrsc5hip_stage_l2_index(nrsc5hip_engine *e, const uint8_t *bits, int nbits, int nframes, nrsc5hip_l2_frame *out, uint8_t *pdu_bytes, long long stride)
{
    ON_ENGINE_DEVICE(e);
    if (!e || !bits || !out || nbits < 1 || nframes < 1) FAIL(NRSC5HIP_EINVAL, "bad argument");
    const int words = (nbits + 31) / 32;
    std::vector<uint32_t> w((size_t)words * nframes, 0u);
    for (int f = 0; f < nframes; f++)
        for (int i = 0; i < nbits; i++) w[(size_t)f * words + (i >> 5)] |= (uint32_t)(bits[(size_t)f * nbits + i] & 1u) << (i & 31);
    DevTmp tw;
    HIPCHK(hipMalloc(&tw.p, w.size() * sizeof(uint32_t)));
    uint32_t *dw = (uint32_t *)tw.p;
    HIPCHK(hipMemcpy(dw, w.data(), w.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    std::vector<L2Job> dj((size_t)nframes);
    for (int f = 0; f < nframes; f++) dj[f] = L2Job{dw + (size_t)f * words, nbits, 0};
    return l2_run(e, dj, out, pdu_bytes, stride);
}

extern "C" int nrsc5hip_stage_first_header(nrsc5hip_engine *e, const uint8_t *bits, int nbits, int nframes, int threads, int *ok)
{
    ON_ENGINE_DEVICE(e);
    if (!e || !bits || !ok || nframes < 1 || (nbits != P1_LEN && nbits != AM_P1_LEN) || threads < 64 || threads > 1024 || (threads & 63)) FAIL(NRSC5HIP_EINVAL, "bad argument");
    const int words = (nbits + 31) / 32;
    std::vector<uint32_t> w((size_t)nframes * words, 0u);
    for (int f = 0; f < nframes; f++)
        for (int i = 0; i < nbits; i++) if (bits[(size_t)f * nbits + i] & 1) w[(size_t)f * words + (i >> 5)] |= 1u << (i & 31);
    uint32_t *dw = nullptr; int *dok = nullptr;
    HIPCHK(hipMalloc((void **)&dw, w.size() * sizeof(uint32_t)));
    HIPCHK(hipMalloc((void **)&dok, (size_t)nframes * sizeof(int)));
    HIPCHK(hipMemcpy(dw, w.data(), w.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    launch_stage_first_header(dw, words, nframes, nbits == AM_P1_LEN ? 1 : 0, threads, dok, e->main);
    HIPCHK(hipStreamSynchronize(e->main));
    HIPCHK(hipMemcpy(ok, dok, (size_t)nframes * sizeof(int), hipMemcpyDeviceToHost));
    (void)hipFree(dw); (void)hipFree(dok);
    return 0;
}

extern "C" int nrsc5hip_stage_viterbi_k9(nrsc5hip_engine *e, const int8_t *soft, int len, int nframes, const unsigned gens[3], uint8_t *bits)
{
    ON_ENGINE_DEVICE(e);
    if (!e || !soft || !bits || !gens || len < 64 || nframes < 1) FAIL(NRSC5HIP_EINVAL, "bad argument");
    int8_t *dsoft = nullptr; unsigned long long *ddec = nullptr; uint32_t *dout = nullptr;
    const int words = (len + 31) / 32;
    HIPCHK(hipMalloc((void **)&dsoft, (size_t)nframes * 3 * len));
    HIPCHK(hipMalloc((void **)&ddec, (size_t)nframes * 4 * (len + 64) * sizeof(unsigned long long)));
    HIPCHK(hipMalloc((void **)&dout, (size_t)nframes * words * sizeof(uint32_t)));
    HIPCHK(hipMemcpy(dsoft, soft, (size_t)nframes * 3 * len, hipMemcpyHostToDevice));
    K9Meta *dmeta = nullptr;                                   // segment waves (the window pipeline's form) unless tuned down to one
    if (e->am_segments > 1 && len > 80) HIPCHK(hipMalloc((void **)&dmeta, (size_t)nframes * sizeof(K9Meta)));
    launch_viterbi_k9_frames(dsoft, len, nframes, gens[0], gens[1], gens[2], ddec, dout, e->main, 3, dmeta, e->am_segments, e->am_warm, e->am_runin, e->db.am_k9stats);
    HIPCHK(hipStreamSynchronize(e->main));
    if (dmeta) (void)hipFree(dmeta);
    std::vector<uint32_t> w((size_t)nframes * words);
    HIPCHK(hipMemcpy(w.data(), dout, w.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int f = 0; f < nframes; f++) nrsc5hip_unpack_bits(w.data() + (size_t)f * words, len, bits + (size_t)f * len);
    (void)hipFree(dsoft); (void)hipFree(ddec); (void)hipFree(dout);
    return 0;
}

extern "C" int nrsc5hip_batch_fetch(nrsc5hip_engine *e, int nstreams, const int *stream_ids, nrsc5hip_record *records,
                                    int max_records, int *counts, uint32_t *frames)
{
    ON_ENGINE_DEVICE(e);
    if (!e || !records || !counts) FAIL(NRSC5HIP_EINVAL, "null argument");
    HIPCHK(hipStreamSynchronize(e->main));
    std::vector<StreamState> *dummy = nullptr; (void)dummy;
    for (int k = 0; k < nstreams; k++) {
        const int s = stream_ids ? stream_ids[k] : k;
        int rc = check_stream(e, s); if (rc) return rc;
        int n = 0;
        rc = nrsc5hip_drain(e, s, records + (size_t)k * max_records, max_records, &n); if (rc) return rc;
        counts[k] = n;
        if (frames)
            HIPCHK(hipMemcpy(frames + (size_t)k * e->db.p1_slots * P1_WORDS, e->db.p1_ring + (size_t)s * e->db.p1_slots * P1_WORDS,
                             (size_t)e->db.p1_slots * P1_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    return 0;
}

// ---- stage-level entry points ----------------------------------------------------------------------------------------
extern "C" int nrsc5hip_stage_halfband_fm_cu8(nrsc5hip_engine *e, const uint8_t *iq, uint32_t nbytes, int16_t *out)
{
    ON_ENGINE_DEVICE(e);
    // runs the production K1 kernel on stream 0 of a scratch state: requires a freshly reset stream 0
    int rc = check_stream(e, 0); if (rc) return rc;
    if (nbytes % 4 || nbytes > e->stage_bytes || nbytes / 4 > e->db.q15_cap) FAIL(NRSC5HIP_EINVAL, "bad length");
    if ((rc = nrsc5hip_stream_fresh(e, 0))) return rc;
    const int s = 0; const unsigned count = nbytes;
    HIPCHK(hipMemcpy(e->stage_dev, iq, nbytes, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->ids_dev, &s, sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->nbytes_dev, &count, sizeof(unsigned), hipMemcpyHostToDevice));
    launch_decimate_fm_cu8(e->tb, e->db, 1, e->ids_dev, e->stage_dev, 0, e->nbytes_dev, count, e->main);
    HIPCHK(hipStreamSynchronize(e->main));
    HIPCHK(hipMemcpy(out, e->db.q15, (size_t)(nbytes / 4) * sizeof(c16), hipMemcpyDeviceToHost));
    return nrsc5hip_stream_fresh(e, 0);
}

extern "C" int nrsc5hip_stage_fft2048(nrsc5hip_engine *e, const float *in, float *out, int n)
{
    ON_ENGINE_DEVICE(e);
    if (!e || !in || !out || n < 1) FAIL(NRSC5HIP_EINVAL, "bad argument");
    float2 *din = nullptr, *dout = nullptr;
    const size_t bytes = (size_t)n * FFT_N * sizeof(float2);
    HIPCHK(hipMalloc((void **)&din, bytes));
    HIPCHK(hipMalloc((void **)&dout, bytes));
    HIPCHK(hipMemcpy(din, in, bytes, hipMemcpyHostToDevice));
    launch_fft2048(e->tb, din, dout, n, e->main, e->mixfft_syms);
    HIPCHK(hipStreamSynchronize(e->main));
    HIPCHK(hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost));
    (void)hipFree(din); (void)hipFree(dout);
    return 0;
}

extern "C" int nrsc5hip_stage_viterbi_k7(nrsc5hip_engine *e, const int8_t *soft, int len, int nframes, uint8_t *bits)
{
    ON_ENGINE_DEVICE(e);
    if (!e || !soft || !bits || len < 64 || nframes < 1) FAIL(NRSC5HIP_EINVAL, "bad argument");
    int8_t *dsoft = nullptr; unsigned long long *ddec = nullptr; uint32_t *dout = nullptr;
    const int words = (len + 31) / 32;
    HIPCHK(hipMalloc((void **)&dsoft, (size_t)nframes * 3 * len));
    HIPCHK(hipMalloc((void **)&ddec, (size_t)nframes * (len + 64) * sizeof(unsigned long long)));
    HIPCHK(hipMalloc((void **)&dout, (size_t)nframes * words * sizeof(uint32_t)));
    HIPCHK(hipMemcpy(dsoft, soft, (size_t)nframes * 3 * len, hipMemcpyHostToDevice));
    if (launch_viterbi_frames(e->vit_scratch, dsoft, len, nframes, ddec, dout, e->main, 3 | (e->tb_walk ? 0 : 16), e->fwd_segments > 0 ? e->fwd_segments : 16, e->db.fwd_stats, e->fwd_warm)) FAIL(NRSC5HIP_EINVAL, "frame length %d not supported or out of device memory", len);
    HIPCHK(hipStreamSynchronize(e->main));
    std::vector<uint32_t> w((size_t)nframes * words);
    HIPCHK(hipMemcpy(w.data(), dout, w.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (int f = 0; f < nframes; f++) nrsc5hip_unpack_bits(w.data() + (size_t)f * words, len, bits + (size_t)f * len);
    (void)hipFree(dsoft); (void)hipFree(ddec); (void)hipFree(dout);
    return 0;
}

extern "C" int nrsc5hip_debug_fetch(nrsc5hip_engine *e, int stream, int8_t *pm, float *bins)
{
    ON_ENGINE_DEVICE(e);
    int rc = check_stream(e, stream); if (rc) return rc;
    HIPCHK(hipDeviceSynchronize());
    if (pm) {
        int slot = 0;
        HIPCHK(hipMemcpy(&slot, (const char *)(e->db.state + stream) + offsetof(StreamState, last_pm_slot), sizeof(int), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(pm, e->db.pm + ((size_t)stream * NPM + slot) * PM_FRAME, PM_FRAME, hipMemcpyDeviceToHost));
    }
    if (bins) HIPCHK(hipMemcpy(bins, e->db.bins + (size_t)stream * NSYM * LIVE_N, (size_t)NSYM * LIVE_N * sizeof(float2), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int nrsc5hip_debug_fetch_costas(nrsc5hip_engine *e, int stream, float *freq, float *phase)
{
    ON_ENGINE_DEVICE(e);
    int rc = check_stream(e, stream); if (rc) return rc;
    if (!freq || !phase) FAIL(NRSC5HIP_EINVAL, "null argument");
    if (e->staged_stream >= 0 && (rc = flush_staged(e))) return rc;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(freq, (const char *)(e->db.state + stream) + offsetof(StreamState, costas_freq), LIVE_N * sizeof(float), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(phase, (const char *)(e->db.state + stream) + offsetof(StreamState, costas_phase), LIVE_N * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int nrsc5hip_debug_fetch_px(nrsc5hip_engine *e, int stream, int8_t *pair)
{
    ON_ENGINE_DEVICE(e);
    int rc = check_stream(e, stream); if (rc) return rc;
    if (!pair) FAIL(NRSC5HIP_EINVAL, "null argument");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(pair, e->db.px_pair + (size_t)stream * 4 * PX_MAX, 4 * PX_MAX, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int nrsc5hip_debug_fetch_q15(nrsc5hip_engine *e, int stream, long long n, int16_t *out)
{
    ON_ENGINE_DEVICE(e);
    int rc = check_stream(e, stream); if (rc) return rc;
    if (n < 0 || n > e->db.q15_cap || !out) FAIL(NRSC5HIP_EINVAL, "bad argument");
    if (e->hc_stream == stream && (rc = hc_detach(e))) return rc;      // a stream that reads the pinned capture has no FIFO to show: it gets one (from its read position on)
    if (e->staged_stream >= 0 && (rc = flush_staged(e))) return rc;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, e->db.q15 + (size_t)stream * e->db.q15_cap, (size_t)n * sizeof(c16), hipMemcpyDeviceToHost));
    return 0;
}

// ---- device helpers for C hosts that do not link the HIP runtime themselves (integration/batch_shard.c) ------------------------------
extern "C" int nrsc5hip_device_count(int *n)
{
    if (!n) FAIL(NRSC5HIP_EINVAL, "null argument");
    HIPCHK(hipGetDeviceCount(n));
    return 0;
}
extern "C" int nrsc5hip_device_upload(int device, const void *host, size_t nbytes, void **dev_out)
{
    if (!dev_out) FAIL(NRSC5HIP_EINVAL, "null argument");
    *dev_out = nullptr;
    DeviceGuard guard(device);
    void *d = nullptr;
    if (hipMalloc(&d, nbytes ? nbytes : 1) != hipSuccess) FAIL(NRSC5HIP_ENOMEM, "hipMalloc(%zu bytes) on device %d failed", nbytes, device);
    if (host) { hipError_t err = hipMemcpy(d, host, nbytes, hipMemcpyHostToDevice); if (err != hipSuccess) { (void)hipFree(d); FAIL(NRSC5HIP_EHIP, "upload failed: %s", hipGetErrorString(err)); } }
    *dev_out = d;
    return 0;
}
extern "C" int nrsc5hip_device_free(int device, void *dev)
{
    DeviceGuard guard(device);
    HIPCHK(hipFree(dev));
    return 0;
}

extern "C" void *nrsc5hip_debug_alloc_copy(const void *host, size_t nbytes)
{
    void *d = nullptr;
    if (hipMalloc(&d, nbytes ? nbytes : 1) != hipSuccess) return nullptr;
    if (host && hipMemcpy(d, host, nbytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    return d;
}

extern "C" void nrsc5hip_debug_free(void *dev) { (void)hipFree(dev); }

extern "C" int nrsc5hip_reset_all(nrsc5hip_engine *e)
{
    ON_ENGINE_DEVICE(e);
    if (!e) FAIL(NRSC5HIP_EINVAL, "null engine");
    HIPCHK(hipDeviceSynchronize());
    e->dec_chunk = 0;
    e->staged_stream = -1; e->staged_bytes = 0; e->staged_q15 = 0;
    e->hc_stream = -1; std::fill(e->hb_hist_host.begin(), e->hb_hist_host.end(), std::array<c16, 14>{});
    const size_t S = e->cfg.max_streams;
    std::vector<StreamState> init(S);
    for (size_t s = 0; s < S; s++) init_state(init[s], e->mode_host[s]);
    HIPCHK(hipMemcpy(e->db.state, init.data(), S * sizeof(StreamState), hipMemcpyHostToDevice));
    if (e->db.am) {
        std::vector<AmStream> ainit(S);
        for (size_t s = 0; s < S; s++) init_am_state(ainit[s]);
        HIPCHK(hipMemcpy(e->db.am, ainit.data(), S * sizeof(AmStream), hipMemcpyHostToDevice));
        HIPCHK(hipMemset(e->db.am_job, 0, S * NWIN * sizeof(AmJob)));
        HIPCHK(hipMemset(e->db.am_pids_rec, 0xff, S * NWIN * 8 * sizeof(int)));
    }
    std::fill(e->raw_host.begin(), e->raw_host.end(), 0);
    std::fill(e->attached.begin(), e->attached.end(), 0);
    std::fill(e->wr_host.begin(), e->wr_host.end(), 0);
    std::fill(e->base_host.begin(), e->base_host.end(), 0);
    std::fill(e->drained.begin(), e->drained.end(), 0);
    std::fill(e->rd_host.begin(), e->rd_host.end(), 0);
    std::fill(e->fetched.begin(), e->fetched.end(), 0);
    std::fill(e->mirror_ok.begin(), e->mirror_ok.end(), (char)(e->cfg.p1_async ? 0 : 1));
    std::fill(e->pred_ok.begin(), e->pred_ok.end(), 0); e->counters_clean = false;
    for (auto &q : e->pending) q.clear();
    HIPCHK(hipMemset(e->db.pids_rec, 0xff, S * NWIN * 16 * sizeof(int)));
    HIPCHK(hipMemset(e->db.px_job, 0xff, S * NWIN * 16 * sizeof(PxJob)));
    e->lane.acq_needed = true; e->lane.px_needed = true; e->lane.set_sig = 0; e->lane.step_count = 0; e->lane.am_step_count = 0;
    for (int k = 0; k < NWIN; k++) { e->lane.am_decoded_pending[k] = false; e->lane.decoded_pending[k] = false; }
    return 0;
}

// Test / bench hygiene: overwrite every result buffer a pass writes (decoded-frame rings on the device, their pinned host mirror,
// the record rings) with a pattern no decode produces, so that a check after the next pass can only pass on bits written by it.
extern "C" int nrsc5hip_debug_poison_results(nrsc5hip_engine *e)
{
    ON_ENGINE_DEVICE(e);
    HIPCHK(hipDeviceSynchronize());
    const size_t S = e->cfg.max_streams;
    HIPCHK(hipMemset(e->db.p1_ring, 0xA5, S * e->db.p1_slots * (size_t)P1_WORDS * sizeof(uint32_t)));
    HIPCHK(hipMemset(e->db.records, 0, S * e->db.rec_cap * sizeof(BlockRecord)));
    HIPCHK(hipMemset(e->db.px_ring, 0xA5, S * (size_t)e->db.px_slots * 2 * PX_WORDS * sizeof(uint32_t)));
    if (e->frames_host) memset(e->frames_host, 0xA5, S * e->db.p1_slots * (size_t)P1_WORDS * sizeof(uint32_t));
    if (e->rec_host) memset(e->rec_host, 0, S * e->db.rec_cap * sizeof(BlockRecord));
    return 0;
}

extern "C" int nrsc5hip_profile(nrsc5hip_engine *e, int enable, double *total_ms, long long *launches)
{
    ON_ENGINE_DEVICE(e);
    if (!e) FAIL(NRSC5HIP_EINVAL, "null engine");
    HIPCHK(hipDeviceSynchronize());
    if (e->prof_on) prof_collect(e);
    for (int k = 0; k < NRSC5HIP_PROF_CLASSES; k++) {
        if (total_ms) total_ms[k] = e->prof_ms[k];
        if (launches) launches[k] = e->prof_launches[k];
        if (enable >= 0) { e->prof_ms[k] = 0; e->prof_launches[k] = 0; }
    }
    if (enable >= 0) { e->prof_on = enable != 0; e->prof_only = (enable & 0x100) ? (enable & 0xff) : -1; }
    return 0;
}

extern "C" int nrsc5hip_stage_selftest(nrsc5hip_engine *e, int *failures)
{
    ON_ENGINE_DEVICE(e);
    if (!e || !failures) FAIL(NRSC5HIP_EINVAL, "null argument");
    HIPCHK(hipMemsetAsync(e->db.counters + 2, 0, sizeof(int), e->main));
    launch_selftest(e->db.counters + 2, e->main);
    HIPCHK(hipMemcpyAsync(failures, e->db.counters + 2, sizeof(int), hipMemcpyDeviceToHost, e->main));
    HIPCHK(hipStreamSynchronize(e->main));
    return 0;
}

extern "C" int nrsc5hip_stage_viterbi_k7_debug(nrsc5hip_engine *e, const int8_t *soft, int len, uint8_t *bits, unsigned long long *dec_out)
{
    ON_ENGINE_DEVICE(e);
    if (!e || !soft || !bits || !dec_out || len < 64) FAIL(NRSC5HIP_EINVAL, "bad argument");
    int8_t *dsoft = nullptr; unsigned long long *ddec = nullptr; uint32_t *dout = nullptr;
    const int words = (len + 31) / 32;
    HIPCHK(hipMalloc((void **)&dsoft, (size_t)3 * len));
    HIPCHK(hipMalloc((void **)&ddec, (size_t)(len + 64) * sizeof(unsigned long long)));
    HIPCHK(hipMalloc((void **)&dout, (size_t)words * sizeof(uint32_t)));
    HIPCHK(hipMemcpy(dsoft, soft, (size_t)3 * len, hipMemcpyHostToDevice));
    if (launch_viterbi_frames(e->vit_scratch, dsoft, len, 1, ddec, dout, e->main)) FAIL(NRSC5HIP_EINVAL, "frame length %d not supported or out of device memory", len);
    HIPCHK(hipStreamSynchronize(e->main));
    std::vector<uint32_t> w(words);
    HIPCHK(hipMemcpy(w.data(), dout, w.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(dec_out, ddec, (size_t)(len + 64) * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    nrsc5hip_unpack_bits(w.data(), len, bits);
    (void)hipFree(dsoft); (void)hipFree(ddec); (void)hipFree(dout);
    return 0;
}

// micro-benchmark: nframes random frames, `phases` bit0 = forward pass, bit1 = traceback; ms per launch
extern "C" int nrsc5hip_stage_viterbi_bench(nrsc5hip_engine *e, int len, int nframes, int phases, int reps, float *ms_per_launch)
{
    ON_ENGINE_DEVICE(e);
    if (!e || !ms_per_launch || len < 64 || nframes < 1 || reps < 1) FAIL(NRSC5HIP_EINVAL, "bad argument");
    int8_t *dsoft = nullptr; unsigned long long *ddec = nullptr; uint32_t *dout = nullptr;
    const int words = (len + 31) / 32;
    std::vector<int8_t> h((size_t)nframes * 3 * len);
    unsigned x = 12345;
    for (auto &v : h) { x = x * 1664525u + 1013904223u; v = (int8_t)((int)(x >> 24) - 128); if (v == -128) v = -127; }
    HIPCHK(hipMalloc((void **)&dsoft, h.size()));
    HIPCHK(hipMalloc((void **)&ddec, (size_t)nframes * (len + 64) * sizeof(unsigned long long)));
    HIPCHK(hipMalloc((void **)&dout, (size_t)nframes * words * sizeof(uint32_t)));
    HIPCHK(hipMemcpy(dsoft, h.data(), h.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(ddec, 0x55, (size_t)nframes * (len + 64) * sizeof(unsigned long long)));
    hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    const int seg = e->fwd_segments > 0 ? e->fwd_segments : 1;
    if (launch_viterbi_frames(e->vit_scratch, dsoft, len, nframes, ddec, dout, e->main, phases | 1 | (e->tb_walk ? 0 : 16), seg)) FAIL(NRSC5HIP_EINVAL, "frame length %d not supported or out of device memory", len);      // warm-up; packs the soft words and leaves decisions behind
    HIPCHK(hipEventRecord(a, e->main));
    for (int r = 0; r < reps; r++) {
        // a traceback-only measurement consumes the decisions in place: re-run the (untimed-irrelevant) forward pass is not possible
        // without timing it, so phases == 2 measures forward + traceback minus nothing -- callers subtract the forward figure
        (void)launch_viterbi_frames(e->vit_scratch, dsoft, len, nframes, ddec, dout, e->main, ((phases & 2) ? (phases | 1) : phases) | 8 | (e->tb_walk ? 0 : 16), seg);
    }
    HIPCHK(hipEventRecord(b, e->main));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, a, b));
    *ms_per_launch = ms / reps;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    (void)hipFree(dsoft); (void)hipFree(ddec); (void)hipFree(dout);
    return 0;
}

// micro-benchmark of the K=9 trellis kernel (E2 code) on random hard-decision frames: phases bit0 = forward, bit1 = traceback
extern "C" int nrsc5hip_stage_viterbi_k9_bench(nrsc5hip_engine *e, int len, int nframes, int phases, int reps, float *ms_per_launch)
{
    ON_ENGINE_DEVICE(e);
    if (!e || !ms_per_launch || len < 128 || nframes < 1 || reps < 1) FAIL(NRSC5HIP_EINVAL, "bad argument");
    int8_t *dsoft = nullptr; unsigned long long *ddec = nullptr; uint32_t *dout = nullptr;
    const int words = (len + 31) / 32;
    // tail-biting code words of random payloads (generators 0561 / 0753 / 0711, bit 8 - k of the register = payload bit i - k),
    // one sign in 16 flipped: what the decoder sees on a healthy channel (pure noise would make every segment speculation fail)
    std::vector<int8_t> h((size_t)nframes * 3 * len);
    std::vector<uint8_t> pay((size_t)len);
    const unsigned gens[3] = { 0561, 0753, 0711 };
    unsigned x = 4321;
    for (int f = 0; f < nframes; f++) {
        for (auto &v : pay) { x = x * 1664525u + 1013904223u; v = (uint8_t)((x >> 24) & 1u); }
        for (int i = 0; i < len; i++) {
            unsigned r = 0;
            for (int k = 0; k < 9; k++) r |= (unsigned)pay[(size_t)((i - k + len) % len)] << (8 - k);
            for (int j = 0; j < 3; j++) {
                x = x * 1664525u + 1013904223u;
                int v = (__builtin_popcount(r & gens[j]) & 1) ? 1 : -1;
                if (((x >> 20) & 15u) == 0) v = -v;
                h[((size_t)f * len + i) * 3 + j] = (int8_t)v;
            }
        }
    }
    HIPCHK(hipMalloc((void **)&dsoft, h.size()));
    HIPCHK(hipMalloc((void **)&ddec, (size_t)nframes * 4 * (len + 64) * sizeof(unsigned long long)));
    HIPCHK(hipMalloc((void **)&dout, (size_t)nframes * words * sizeof(uint32_t)));
    HIPCHK(hipMemcpy(dsoft, h.data(), h.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(ddec, 0x55, (size_t)nframes * 4 * (len + 64) * sizeof(unsigned long long)));
    K9Meta *dmeta = nullptr;
    if (e->am_segments > 1) HIPCHK(hipMalloc((void **)&dmeta, (size_t)nframes * sizeof(K9Meta)));
    hipEvent_t a, b; HIPCHK(hipEventCreate(&a)); HIPCHK(hipEventCreate(&b));
    launch_viterbi_k9_frames(dsoft, len, nframes, 0561, 0753, 0711, ddec, dout, e->main, 3, dmeta, e->am_segments, e->am_warm, e->am_runin, e->db.am_k9stats);
    HIPCHK(hipEventRecord(a, e->main));
    for (int r = 0; r < reps; r++) launch_viterbi_k9_frames(dsoft, len, nframes, 0561, 0753, 0711, ddec, dout, e->main, phases, dmeta, e->am_segments, e->am_warm, e->am_runin, e->db.am_k9stats);
    HIPCHK(hipEventRecord(b, e->main));
    HIPCHK(hipEventSynchronize(b));
    if (dmeta) (void)hipFree(dmeta);
    float ms = 0; HIPCHK(hipEventElapsedTime(&ms, a, b));
    *ms_per_launch = ms / reps;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    (void)hipFree(dsoft); (void)hipFree(ddec); (void)hipFree(dout);
    return 0;
}

// Tuning knobs and test hooks: an explicit entry point, nothing is read from the environment.
extern "C" int nrsc5hip_debug_tune(nrsc5hip_engine *e, int knob, int value)
{
    ON_ENGINE_DEVICE(e);
    if (!e) FAIL(NRSC5HIP_EINVAL, "null engine");
    HIPCHK(hipDeviceSynchronize());
    switch (knob) {
    case NRSC5HIP_TUNE_DECODE_STREAMS:    e->naux = std::min(std::max(value, 1), NAUX); break;
    case NRSC5HIP_TUNE_AM_DECODE_STREAMS: e->naux_am = std::min(std::max(value, 1), NAUX); break;
    case NRSC5HIP_TUNE_VERDICT_LAG:       e->verdict_lag = std::min(std::max(value, 0), NWIN); break;
    case NRSC5HIP_TUNE_FWD_SEGMENTS:      e->fwd_segments = std::min(std::max(value, 0), VIT3_GMAX); break;
    case NRSC5HIP_TUNE_FWD_WARM:          e->fwd_warm = value > 0 ? 2 : 0; break;
    case NRSC5HIP_TUNE_DECODE_CUS: {
        // decode streams confined to value / 32 of the CUs (the pattern keeps that share of every XCD whichever way mask bits map to CUs)
        const int k = std::min(std::max(value, 8), 32) & ~7;
        hipDeviceProp_t prop; HIPCHK(hipGetDeviceProperties(&prop, e->cfg.device));
        const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
        std::vector<uint32_t> mask((size_t)words, 0u);
        for (int i = 0; i < ncu; i++) if ((i % 32) < k) mask[(size_t)i / 32] |= 1u << (i % 32);
        for (int a = 0; a < NAUX; a++) {                       // the new stream first; the old one is destroyed only once it exists
            hipStream_t fresh = nullptr;
            if (k >= 32) HIPCHK(hipStreamCreate(&fresh));
            else HIPCHK(hipExtStreamCreateWithCUMask(&fresh, (uint32_t)words, mask.data()));
            (void)hipStreamDestroy(e->lane.aux[a]);
            e->lane.aux[a] = fresh;
        }
        break;
    }
    case NRSC5HIP_TUNE_DECODE_PRIORITY: {
        int least = 0, greatest = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        for (int a = 0; a < NAUX; a++) {
            hipStream_t fresh = nullptr;
            if (value) HIPCHK(hipStreamCreateWithPriority(&fresh, hipStreamDefault, least));
            else HIPCHK(hipStreamCreate(&fresh));
            (void)hipStreamDestroy(e->lane.aux[a]);
            e->lane.aux[a] = fresh;
        }
        break;
    }
    case NRSC5HIP_TUNE_TRACEBACK_WALK:    e->tb_walk = std::min(std::max(value, 0), 16384); break;
    case NRSC5HIP_TUNE_SYNC_LANES:        e->sync_lanes = (value == 256 || value == 768) ? value : 0; break;
    case NRSC5HIP_TUNE_SEAM_PREPARE:      e->fuse_seam_prepare = value != 0; break;
    case NRSC5HIP_TUNE_FOLD_REPORT:       e->fold_report = value != 0; break;
    case NRSC5HIP_TUNE_NCO_EXACT:         e->db.nco_policy = e->lane.db.nco_policy = e->db.nco_tab ? std::min(std::max(value, 0), (int)NCO_EXACT_ALWAYS) : (int)NCO_CLOSED_FORM; break;
    case NRSC5HIP_TUNE_FLOW_MIN:          e->flow_min = std::max(value, 0); break;
    case NRSC5HIP_TUNE_LOOP_EXACT:        e->db.loop_exact = e->lane.db.loop_exact = std::min(std::max(value, 0), 2); break;
    case NRSC5HIP_TUNE_EARLY_FLUSH_KB:    e->early_flush = (size_t)std::max(value, 0) << 10; break;
    case NRSC5HIP_TUNE_DEFER_WAIT:        e->defer_wait = value != 0; break;
    case NRSC5HIP_TUNE_DIRECT_DECIMATE:   e->direct_decimate = value != 0; break;
    case NRSC5HIP_TUNE_HOST_CAPTURE: {
        if (e->hc_stream >= 0) { int rc = hc_detach(e); if (rc) return rc; }
        if (!e->hc_pin) break;                                 // window-pipeline engines have no fast seam
        e->host_capture = value != 0;
        if (value >= 512) {                                    // that many KiB of pinned capture instead of the default 16 MiB (tests: small values exercise hc_rebase)
            uint8_t *np = nullptr; void *dp = nullptr;
            if (hipHostMalloc((void **)&np, (size_t)value << 10, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&dp, np, 0) != hipSuccess) FAIL(NRSC5HIP_ENOMEM, "pinned capture allocation failed");
            (void)hipHostFree(e->hc_pin);
            e->hc_pin = np; e->hc_dev = (uint8_t *)dp; e->hc_cap = (size_t)value << 10;
        }
        break;
    }
    case NRSC5HIP_TUNE_MIXFFT_SYMS: {
        e->mixfft_syms = (value == 2 || value == 4 || value == 8 || value == 16 || value == 32 || (value >= 100 && value <= 140)) ? value : 1;
        if (e->mixfft_syms >= 100) {                           // DIAGNOSTIC LDS padding: never beyond what a workgroup may have beside the kernel's own ~20 KB (an oversized request failed the
            int lds_max = 65536;                               // launch, and the failure surfaced at some later hipGetLastError)
#ifndef HIPEMU
            HIPCHK(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, e->cfg.device));
#endif
            const int room_kib = (lds_max - 24 * 1024) / 1024;
            if (e->mixfft_syms - 100 > room_kib) e->mixfft_syms = 100 + std::max(room_kib, 0);
        }
        break;
    }
    case NRSC5HIP_TUNE_AM_SEGMENTS:       e->am_segments = std::min(std::max(value, 1), K9_GMAX); break;
    case NRSC5HIP_TUNE_AM_WARM:           e->am_warm = value > 0 ? K9_WARM : 0; e->am_runin = value > 0 ? K9_TB_RUNIN : 0; break;
    case NRSC5HIP_TUNE_SYNC_PHASES:
        if (value && !e->db.sync_phase_cycles) {
            int rc = dev_alloc(e, &e->db.sync_phase_cycles, 16); if (rc) return rc;
            HIPCHK(hipMemset(e->db.sync_phase_cycles, 0, 16 * sizeof(long long)));
        }
        e->lane.db.sync_phase_cycles = value ? e->db.sync_phase_cycles : nullptr;
        break;
    default: FAIL(NRSC5HIP_EINVAL, "unknown knob %d", knob);
    }
    return 0;
}

extern "C" int nrsc5hip_abi_version(void) { return NRSC5HIP_ABI_VERSION; }

extern "C" int nrsc5hip_debug_flow_stats(nrsc5hip_engine *e, long long stats[2])
{
    if (!e || !stats) return NRSC5HIP_EINVAL;
    stats[0] = e->flow_bursts; stats[1] = e->flow_steps;
    return 0;
}

extern "C" int nrsc5hip_debug_host_capture_stats(nrsc5hip_engine *e, long long stats[5])
{
    if (!e || !stats) return NRSC5HIP_EINVAL;
    stats[0] = e->hc_attaches; stats[1] = e->hc_detaches; stats[2] = e->hc_rebases; stats[3] = e->hc_stream; stats[4] = e->reports_folded;
    return 0;
}

extern "C" int nrsc5hip_debug_fwd_stats(nrsc5hip_engine *e, int stats[2])
{
    ON_ENGINE_DEVICE(e);
    if (!e || !stats) FAIL(NRSC5HIP_EINVAL, "null argument");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(stats, e->db.fwd_stats, 2 * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int nrsc5hip_debug_tb_stats(nrsc5hip_engine *e, int stats[2])
{
    ON_ENGINE_DEVICE(e);
    if (!stats) FAIL(NRSC5HIP_EINVAL, "null argument");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(stats, e->db.tb_stats, 2 * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int nrsc5hip_debug_k9_stats(nrsc5hip_engine *e, int stats[4])
{
    ON_ENGINE_DEVICE(e);
    if (!e || !stats) FAIL(NRSC5HIP_EINVAL, "null argument");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(stats, e->db.am_k9stats, 4 * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int nrsc5hip_debug_sync_phases(nrsc5hip_engine *e, long long *cycles16)
{
    ON_ENGINE_DEVICE(e);
    if (!e || !cycles16) FAIL(NRSC5HIP_EINVAL, "null argument");
    if (!e->db.sync_phase_cycles) FAIL(NRSC5HIP_EINVAL, "turn the instrumentation on first: nrsc5hip_debug_tune(e, NRSC5HIP_TUNE_SYNC_PHASES, 1)");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(cycles16, e->db.sync_phase_cycles, 16 * sizeof(long long), hipMemcpyDeviceToHost));
    return 0;
}

// Zero-copy variant of nrsc5hip_batch_fetch for streams 0..nstreams-1: three bulk D2H copies into engine-owned
// pinned buffers; the returned pointers stay valid until the next fetch/reset.  records: [nstreams][record_capacity],
// frames: [nstreams][p1_slots][4568].  Requires that nothing was drained since the last reset.
extern "C" int nrsc5hip_batch_fetch_view(nrsc5hip_engine *e, int nstreams, const nrsc5hip_record **records, int *counts, const uint32_t **frames)
{
    ON_ENGINE_DEVICE(e);
    if (!e || !records || !counts) FAIL(NRSC5HIP_EINVAL, "null argument");
    if (nstreams < 1 || nstreams > e->cfg.max_streams) FAIL(NRSC5HIP_EINVAL, "nstreams out of range");
    { int rc = leave_mirror(e, nstreams, nullptr); if (rc) return rc; }
    const size_t S = e->cfg.max_streams;
    if (!e->rec_host) {
        HIPCHK(hipHostMalloc((void **)&e->rec_host, S * e->db.rec_cap * sizeof(BlockRecord), hipHostMallocDefault));
        HIPCHK(hipHostMalloc((void **)&e->nblocks_host, S * sizeof(int), hipHostMallocDefault));
    }
    HIPCHK(hipStreamSynchronize(e->main));
    HIPCHK(hipMemcpy2DAsync(e->nblocks_host, sizeof(int), (const char *)e->db.state + offsetof(StreamState, nblocks), sizeof(StreamState),
                            sizeof(int), nstreams, hipMemcpyDeviceToHost, e->main));
    HIPCHK(hipStreamSynchronize(e->main));
    int maxn = 0;
    bool any_am = false;
    for (int s = 0; s < nstreams; s++) { if (e->nblocks_host[s] > maxn) maxn = e->nblocks_host[s]; any_am |= e->mode_host[s] == MODE_AM; }
    if (maxn > e->db.rec_cap) maxn = e->db.rec_cap;
    // records: only the used head of every stream's ring (the view needs unwrapped rings anyway, checked below)
    if (maxn > 0)
        HIPCHK(hipMemcpy2DAsync(e->rec_host, (size_t)e->db.rec_cap * sizeof(BlockRecord), e->db.records, (size_t)e->db.rec_cap * sizeof(BlockRecord),
                                (size_t)maxn * sizeof(BlockRecord), nstreams, hipMemcpyDeviceToHost, e->main));
    if (frames) {
        // P1 frames: the first view copies the ring and hands the pinned buffer to the FM traceback as a mirror (DevBuffers::
        // p1_mirror); from then on every frame reaches the host while the pass is still running and nothing is left to copy here.
        // AM frames are written by other kernels: a batch with AM streams keeps copying.
        const size_t nwords = (size_t)e->db.p1_slots * P1_WORDS;
        if (!e->frames_host) {
            HIPCHK(hipHostMalloc((void **)&e->frames_host, S * nwords * sizeof(uint32_t), hipHostMallocMapped));
            HIPCHK(hipMemcpyAsync(e->frames_host, e->db.p1_ring, S * nwords * sizeof(uint32_t), hipMemcpyDeviceToHost, e->main));
            void *dp = nullptr;
            HIPCHK(hipHostGetDevicePointer(&dp, e->frames_host, 0));
            e->db.p1_mirror = (uint32_t *)dp;
            e->lane.db.p1_mirror = (uint32_t *)dp;
        } else if (any_am) {
            HIPCHK(hipMemcpyAsync(e->frames_host, e->db.p1_ring, (size_t)nstreams * nwords * sizeof(uint32_t), hipMemcpyDeviceToHost, e->main));
        }
    }
    HIPCHK(hipStreamSynchronize(e->main));
    if (e->cfg.p1_async && e->db.am) {
        std::vector<float> ber((size_t)nstreams * e->db.p1_slots);
        bool any = false;
        for (int s = 0; s < nstreams; s++) any |= e->mode_host[s] == MODE_AM;
        if (any) {
            HIPCHK(hipMemcpy(ber.data(), e->db.am_ber, ber.size() * sizeof(float), hipMemcpyDeviceToHost));
            for (int s = 0; s < nstreams; s++)
                if (e->mode_host[s] == MODE_AM) {
                    const int nrec = e->nblocks_host[s] < e->db.rec_cap ? e->nblocks_host[s] : e->db.rec_cap;
                    patch_am_ber(e, s, (nrsc5hip_record *)e->rec_host + (size_t)s * e->db.rec_cap, nrec, ber.data() + (size_t)s * e->db.p1_slots);
                }
        }
    }
    for (int s = 0; s < nstreams; s++) {
        if (e->drained[s] != 0 || e->nblocks_host[s] > e->db.rec_cap)
            FAIL(NRSC5HIP_EOVERFLOW, "stream %d: view needs an undrained, unwrapped record ring (%d records, capacity %d)", s, e->nblocks_host[s], e->db.rec_cap);
        int n = e->nblocks_host[s];
        e->drained[s] = n;
        if (e->db.ckpt || e->db.am_ckpt) {                     // replay: squeeze the void records out, in place in the pinned buffer
            BlockRecord *r = e->rec_host + (size_t)s * e->db.rec_cap;
            int m = 0;
            for (int k = 0; k < n; k++) if (!(r[k].flags & REC_DISCARDED)) { if (m != k) r[m] = r[k]; m++; }
            n = m;
        }
        counts[s] = n;
    }
    *records = (const nrsc5hip_record *)e->rec_host;
    if (frames) *frames = e->frames_host;
    return 0;
}
