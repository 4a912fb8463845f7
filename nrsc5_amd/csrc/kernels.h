// Host-callable launchers for the gfx950 kernels (one .hip file per pipeline stage).
#pragma once
#include "nrsc5hip.h"
#include "nrsc5_dev.h"

namespace nrsc5 {

// Read-only device tables built once per engine (engine.hip: build_tables()).
struct DevTables {
    const uint16_t *deint_lut;       // [384] byte q of a 384-byte depunctured run -> block*720 + part*36, 0xffff = erasure
    const uint16_t *pids_gather;     // [16][PIDS_CODED] index inside block bc (decode.c:324-342)
    const uint32_t *eq_cell;         // [11520] MP1 data cell c = ((side * 10 + part) * 32 + n) * 18 + k - 1, packed: live bin | n << 10 | low ref << 15 |
                                     //         high ref << 20 | k << 25 | side << 30 (adjust_data's operands, sync.c:263-282)
    const uint16_t *eq_out;          // [11520] where the cell's soft-bit pair goes inside the block's interleaver rows (sync.c:514-536)
    const uint32_t *scr_p1;          // [P1_WORDS] packed scrambler stream (decode.c:279-294)
    const uint32_t *scr_pids;        // [3]
    const float2 *twiddle;           // [2048] e^{-2 pi i k / 2048}
    const float2 *twiddle_a;         // [7][256] the same values in the order the FFT's first exchange reads them: [k1 - 1][r] = twiddle[(k1 r) & 2047]
    const float *shape;              // [2160] pulse shape (acquire.c:322-331)
    const int16_t *hb_q15;           // [4]  half-band taps, window order
    const int16_t *acq_q15;          // [17] acquisition FIR taps, [1..16] used
    const uint32_t *px_delay_wide;   // [9216] interleaver IV write-to-read delay per position of a block pair, MP3 / MP11
    const uint32_t *px_delay_narrow; // [4608] same for MP2
    const uint32_t *am_deint_p1;        // [90000] source of every P1 trellis input (interleaver_ma1 folded into a table, see build_tables)
    const uint32_t *am_deint_p3_ma3;    // [90000] same for the MA3 P3 code word
    const uint32_t *am_deint_p3_ma1;    // [72000] same for the MA1 P3 code word
    const int16_t *am_acq_q15;       // [17] AM acquisition FIR taps (acquire.c:63-96)
    const float *am_shape;           // [270] AM pulse shape (acquire.c:333-342)
    const float2 *am_twiddle;        // [256] e^{-2 pi i k / 256}
};

// Engine-wide device buffers (slabs indexed by stream).
struct DevBuffers {
    StreamState *state;              // [S]
    StreamState *ckpt;               // [S][NWIN] replay: the stream's state right after the block that completed the P1 frame of decode
                                     // window w (slot w % NWIN), before the next block's bookkeeping; null unless p1_async && l2_feedback
    c16 *q15;                        // [S][q15_cap]
    long long q15_cap;
    c16 *acq_win;                    // [S][WIN_N]   zero-copy batch only: decimated acquisition window of streams that are not FINE
    c16 *acq_filt;                   // [S][WIN_N]   acquisition FIR output
    int *acq_list;                   // [S + 1]      streams that need the acquisition kernels this step (k_acq_list); [S] = how many
    float2 *acq_sums;                // [S][SYM_N]
    float2 *bins;                    // [S][NSYM][LIVE_N]
    float2 *cfo_snap;                // [S][LIVE_N][11]  the CFO search's loop-state snapshots (k_sync: one per visit of a live bin)
    float *cfo_phase;                // [S][NSYM][LIVE_N]  exact CFO search (loop_exact): phases[][] of the visit in progress, one column per live bin
    int loop_exact;                  // 0: fast loop arithmetic, 1: the reference's own operations in blocks that start un-synchronised, 2: in every block (k_sync.hip)
    float2 *nco_tab;                 // [S][NSYM][SYM_N]  the reference's oscillator sample by sample for a block that runs in exact mode (k_nco_exact -> k_mixfft); null: closed form only
    int nco_policy;                  // NCO_*: which blocks of a freshly reset stream advance the oscillator by the reference's float recurrence
    int8_t *pm;                      // [S][NPM][PM_FRAME]  soft-bit interleaver matrices (one per frame in flight)
    int8_t *pids_stage;              // [S][NWIN][16][240]  depunctured PIDS soft bits awaiting k_pids_decode
    int *pids_rec;                   // [S][NWIN][16]     record index of each staged PIDS frame, -1 = empty
    int *coded;                      // [NAUX][S][P1_LEN]  depunctured P1 trellis input, one dword per step (s0 | s1 << 8 | s2 << 16), one per decode lane
    uint32_t *dec;                   // [NAUX][S][2 * (P1_LEN + 64)]  survivor decisions: per 32 steps one history word per lane (viterbi_v3.h)
    int nstreams_alloc;              // S
    uint8_t *tbmap;                  // [NAUX][S][2285 * 64]  traceback chunk maps (start lane per end lane)
    int *fwd_meta;                   // [NAUX][S][VIT3_GMAX][512]  segmented forward pass: per segment the metric snapshot [64], end metrics [64] and the
                                     //                     history scratch of its speculative warm-up [384] (viterbi_v3.h)
    int *fwd_stats;                  // [2] segment boundaries checked / segments repaired since the engine was created
    int *tb_stats;                   // [2] single-path traceback: chunk boundaries checked / chunks re-walked since the engine was created
    uint32_t *p1_ring;               // [S][p1_slots][P1_WORDS]
    uint32_t *p1_mirror;             // same layout in pinned host memory (device-visible), written beside p1_ring by the FM traceback once
                                     // nrsc5hip_batch_fetch_view has set it up; null before
    int p1_slots;
    BlockRecord *records;            // [S][rec_cap]
    int rec_cap;
    int *counters;                   // per burst: [0] blocks prepared, [1] not-FINE streams seen (host: keep launching the acquisition kernels),
                                     // [2] streams that need the PX kernels, [3] streams rewound by k_rollback
    long long *sync_phase_cycles;    // [16] optional: accumulated shader cycles per k_sync phase (stream 0 only) in [0..7]; [8..15]: k_mixfft's phases in the diagnostic build; or null
    // extended sidebands
    int8_t *px_mem;                  // [S][2][PX_MEM]           interleaver IV memories of PX1, PX2
    int8_t *px_pair;                 // [S][2][2 * PX_MAX]       soft bits of the current block pair
    int8_t *px_stage;                // [S][NWIN][8][2][PX_DEPUNCT]  depunctured trellis inputs awaiting k_px_decode
    PxJob *px_job;                   // [S][NWIN][8][2]
    unsigned long long *px_dec;      // [NAUX][S][16][PX_MAX + 64] survivor decisions
    uint32_t *px_ring;               // [S][px_slots][2][PX_WORDS]
    int px_slots;
    // AM (null unless the engine was created with am_enable)
    AmStream *am;                    // [S]
    uint8_t *am_sym;                 // [S][4][AM_SYMS]   hard symbols of the current L1 frame: pl, pu, s, t
    uint8_t *am_q;                   // [S][3][2][AM_VIT]  3-frame diversity delay of the ml / mu (/ eml / emu) bits: one cell per trellis input of the P1 / P3 code word, keyed by the input's index (coalesced; the reference's line / position numbering is only a numbering)
    int8_t *am_vit;                  // [S][am_nvit][2][AM_VIT]  depunctured trellis inputs: 8 x P1, P3 (am_nvit = NWIN in the window pipeline, else 1)
    int am_nvit;
    unsigned long long *am_dec;      // [am_ndec][S][8 * AM_DEC_P1 + AM_DEC_P3]  survivor decisions (one set per decode stream)
    K9Meta *am_k9meta;               // [am_ndec][S]  window pipeline: segment boundaries of the P3 frame, end states of the P1 frames
    unsigned *am_k9stats;            // [4] forward boundaries checked / re-run, traceback boundaries checked / re-walked
    AmJob *am_job;                   // [S][NWIN]
    AmCkpt *am_ckpt;                 // [S][NWIN][8]  replay: state after the block that delivered P1 PDU j of the job's frame; null unless
                                     // p1_async && l2_feedback
    float *am_ber;                   // [S][p1_slots]  window pipeline: BER of the L1 frame in each ring slot
    int8_t *am_pids_stage;           // [S][NWIN][8][240]  window pipeline: PIDS trellis inputs awaiting k_am_decode
    int *am_pids_rec;                // [S][NWIN][8]       record index of each staged PIDS frame, -1 = empty
    nrsc5hip_l2_frame *l2_ring;      // [S][p1_slots]  engine option l2_index: audio-transport index of each P1 frame slot (else null)
    nrsc5hip_l2_frame *l2_px_ring;   // [S][px_slots][2]  same for the P3 / P4 frames of the extended sidebands (else null)
    nrsc5hip_l2_frame *l2_am_ring;   // [S][p1_slots][9]  same for the 8 P1 frames + the P3 frame of each AM L1 frame slot (else null)
};

// ---- K1 -------------------------------------------------------------------------------
// cu8 -> Q15 half-band 2:1 for one chunk per stream.  iq[s] = base + s*stride, nbytes[s] each.
// what a block step of the fast streaming seam posts into pinned host memory (engine.hip: harvest)
struct StreamReport { int counters[4]; long long rd; int nblocks; int nrec; BlockRecord rec[4]; unsigned seq; unsigned pad; };
// streaming seam, ONE stream: the chunk is read where the host staged it (pinned, device-visible: no copy engine, no second buffer),
// every input byte once; the workgroup that finishes last rolls the decimator history and publishes the new write position
void launch_decimate_fm_cu8_stream(const DevTables &tb, const DevBuffers &db, int s, const uint8_t *iq, unsigned nbytes, unsigned *ticket, hipStream_t st);
// streaming seam, ONE stream: the block's PIDS frame (do_pids) and then the report
void launch_stream_tail(const DevTables &tb, const DevBuffers &db, int s, int first_rec, StreamReport *out, unsigned seq, int do_pids, hipStream_t st);

void launch_decimate_fm_cu8(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids,
                            const uint8_t *iq_base, long long iq_stride, const unsigned *nbytes, unsigned max_nbytes,
                            hipStream_t st);
// engine option batch_zero_copy: attach the caller's cu8 captures to freshly reset streams (no copy, no decimation)
void launch_attach_raw(const DevBuffers &db, int nstreams, const int *stream_ids, const uint8_t *iq_base, long long iq_stride,
                       const unsigned *nbytes, hipStream_t st);
void launch_append_cs16(const DevBuffers &db, int nstreams, const int *stream_ids,
                        const int16_t *iq_base, long long iq_stride, const unsigned *nsamples, unsigned max_n, hipStream_t st);

// ---- one block step for a set of streams ----------------------------------------------------
void launch_acquire(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, hipStream_t st);
void launch_prepare(const DevBuffers &db, int nstreams, const int *stream_ids, int acq_on, hipStream_t st);
// streams whose current block runs in exact-oscillator mode (StreamState::nco_mode): the reference's 69 120-step float recurrence, one lane per stream
void launch_nco_exact(const DevBuffers &db, int nstreams, const int *stream_ids, hipStream_t st);
void launch_mixfft(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, hipStream_t st, int syms_per_wg = 1, int local_prepare = 0);
size_t flow_words(int n);
void launch_flow(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, int K, unsigned *flow, unsigned *err_host, int parity, int slot0, int window, hipStream_t st);
void launch_sync(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, int parity, int slot, int fuse_prepare, int window, hipStream_t st, int lanes = 0, int pids_inline = 0, int do_prepare = 0, int ext_refs = 1,
                 StreamReport *report = nullptr, unsigned report_seq = 0, int report_first = 0);   // report: one-stream launch of the fast seam posts the step's report itself (k_stream_tail's job)
// replay (k_replay.hip): apply the first-header verdicts of finished deferred P1 decodes -- rewind the stream to the failed frame
void launch_rollback(const DevBuffers &db, int nstreams, const int *stream_ids, int cur_window, int min_age, hipStream_t st);
void launch_rollback_am(const DevBuffers &db, int nstreams, const int *stream_ids, int cur_window, int min_age, hipStream_t st);
void launch_pids_decode(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, int parity, int nslots, hipStream_t st);
// extended sidebands: interleaver IV for streams whose block pair just completed (after k_sync), and the staged P3/P4 decodes
void launch_px_deint(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, int parity, int slot, hipStream_t st);
void launch_px_decode(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, int parity, int lane_id, hipStream_t st);
// a window's P1 decode: de-interleave -> forward trellis pass -> traceback + BER + descramble + first-header verdict (+ fused L2 index)
void launch_p1_deint(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, int parity, int lane_id, hipStream_t st);
// segments: waves per frame of the forward pass (1..16; clamped to what the frame length allows), see viterbi_v3.h
void launch_p1_forward(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, int parity, int lane_id, hipStream_t st, int segments, int warm = 2);
// parts: workgroups per frame of the traceback's first pass (k_p1_tbmap), 1..16
void launch_p1_traceback(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, int parity, int lane_id, hipStream_t st, int l2_mode = 0, int parts = 4, int walk = 1);

// ---- AM path (k_am.hip) -------------------------------------------------------------------------
// cu8 -> five cascaded half-bands 32:1, any nbytes % 4 == 0 per stream (stage phases carry over)
void launch_am_decimate_cu8(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids,
                            const uint8_t *iq_base, long long iq_stride, const unsigned *nbytes, unsigned max_nbytes, hipStream_t st);
// one block step: acquire (or track) -> 2 x 32 FFT-256 -> sync_process_am -> PIDS; then this block's P1 / P3 decodes
// and, after block 7, the bit de-interleaver of the finished L1 frame
void launch_am_step(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, hipStream_t st, int l2_feedback = 0, int pipeline_parity = -1, int slot = 0,
                    int window = 0);
// entry of the AM de-interleave tables: cell | bit << 13 | matrix << 16 | delayed << 18 | punctured << 19
constexpr unsigned AMT_DELAYED = 1u << 18, AMT_PUNCT = 1u << 19;
// window pipeline: the 8 P1 frames and the P3 frame of every L1 frame whose de-interleave happened in window `parity`
void launch_am_decode(const DevTables &tb, const DevBuffers &db, int nstreams, const int *stream_ids, int parity, int lane_id, int l2_feedback, hipStream_t st,
                      int segments, int warm, int runin);
void launch_stage_first_header(const uint32_t *words, int nwords, int nframes, int am, int threads, int *ok, hipStream_t st);
void launch_viterbi_k9_frames(const int8_t *coded, int len, int nframes, unsigned g0, unsigned g1, unsigned g2,
                              unsigned long long *dec, uint32_t *out, hipStream_t st, int phases = 3,
                              K9Meta *meta = nullptr, int segments = 1, int warm = K9_WARM, int runin = K9_TB_RUNIN, unsigned *stats = nullptr);

// ---- L2 audio transport index (k_l2.hip): one workgroup per decoded frame --------------------------------------
constexpr int L2_MAX_BYTES = 18269;
struct L2Job { const uint32_t *words; int nbits; int pad; };        // packed frame (bit i at words[i / 32] bit i % 32)
void launch_l2_index(const L2Job *jobs, int njobs, nrsc5hip_l2_frame *frames, uint8_t *bytes_out, long long stride, hipStream_t st);
// engine option l2_index: index the P1 frames k_p1_traceback finished in decode window `parity` (called by launch_p1_traceback)
void launch_l2_index_window(const DevBuffers &db, int nstreams, const int *stream_ids, int parity, hipStream_t st);
// ... the P3 / P4 frames k_px_decode just finished (called by launch_px_decode), and an AM L1 frame's frames: window pipeline
// (called by launch_am_decode) or in order (called by launch_am_step: the frames k_am_viterbi delivered this step)
void launch_l2_index_px_window(const DevBuffers &db, int nstreams, const int *stream_ids, int parity, hipStream_t st);
void launch_l2_index_am_window(const DevBuffers &db, int nstreams, const int *stream_ids, int parity, hipStream_t st);
void launch_l2_index_am_step(const DevBuffers &db, int nstreams, const int *stream_ids, hipStream_t st);

// ---- stage-level entry points (parity tests) ---------------------------------------------------
// scratch of the stage-level K=7 decode (end lanes, chunk maps, packed soft words, segment metadata): owned by the engine that calls it
struct VitScratch { int *endlane = nullptr; int cap = 0; uint8_t *gmap = nullptr; size_t gcap = 0; int *soft = nullptr; size_t scap = 0; int *meta = nullptr; int mcap = 0; };
void vit_scratch_free(VitScratch &sc);
// -> 0, or -1: frame too long for the block-parallel traceback (viterbi3_traceback_block: at most 64 segments of TB_SEG chunks) / out of device memory
int launch_viterbi_frames(VitScratch &sc, const int8_t *coded, int len, int nframes, unsigned long long *dec, uint32_t *out, hipStream_t st, int phases = 3, int segments = 1, int *stats = nullptr, int warm = 2);
void launch_selftest(int *fail_count, hipStream_t st);
void launch_fft2048(const DevTables &tb, const float2 *in, float2 *out, int nffts, hipStream_t st, int form = 1);   // form 32: the 256-lane FFT

}  // namespace nrsc5
