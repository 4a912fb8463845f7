// Shared definitions for the gfx950 kernels and the host engine of libnrsc5hip.
// Domain constants follow the reference's src/defines.h:12-81; layouts are this design's own.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nrsc5 {

constexpr int FFT_N = 2048;            // defines.h:12
constexpr int CP_N = 112;              // defines.h:15
constexpr int SYM_N = FFT_N + CP_N;    // 2160 samples per OFDM symbol @744187.5 Hz
constexpr int NSYM = 32;               // symbols per L1 block (BLKSZ, defines.h:20)
constexpr int WIN_N = SYM_N * (NSYM + 1);   // 71280: acquire window (acquire.h:12)
constexpr int LB0 = FFT_N / 2 - 546;   // 478  first lower-sideband bin (defines.h:24)
constexpr int UB1 = FFT_N / 2 + 546;   // 1570 last upper-sideband bin (defines.h:26)
constexpr int PW = 19;                 // carriers per partition incl. reference
constexpr int LIVE_HALF = 14 * PW + 1; // 267 bins kept per sideband (sync.c:785-789)
constexpr int LIVE_N = 2 * LIVE_HALF;  // 534
constexpr int UB0 = UB1 - (LIVE_HALF - 1);  // 1304 first kept upper bin
constexpr int PM_PART = 10;             // partitions per primary-main sideband (defines.h:79)
constexpr int PM_BLOCK = 23040;        // soft bits per block (defines.h:81)
constexpr int PM_FRAME = 16 * PM_BLOCK;
constexpr int P1_LEN = 146176;         // defines.h:42
constexpr int P1_CODED = 365440;
constexpr int P1_DEPUNCT = 3 * P1_LEN; // 438528
constexpr int PIDS_LEN = 80;
constexpr int PIDS_CODED = 200;
constexpr int VIT_EXTRA = 32;          // TAIL_BITING_EXTRA, conv_dec.c:43
// segmented forward trellis pass (viterbi_v3.h)
constexpr int VIT3_GMAX = 64;          // segments per frame at most (64 x ~12 trips + warm-up: a lone P1 frame's forward pass in ~40 us)
constexpr int VIT3_META = 128 + 384;   // ints per segment: snapshot [64], end metrics [64], warm-up history scratch [384]
constexpr int P1_WORDS = P1_LEN / 32;  // packed output words per P1 frame (4568)
constexpr int NWIN = 8;                // decode windows (16 block steps each) that may be in flight: buffers indexed w % NWIN
constexpr int NPM = NWIN + 3;           // soft-bit matrices per stream: a frame's matrix must outlive its (deferred) decode
constexpr int NAUX = 5;                // HIP streams that decode windows concurrently (each with its own decision scratch)

// ---- extended sidebands PX1 / PX2 -> P3 / P4 (defines.h:50-52, decode.h:9-17) ----
constexpr int PX_MAX = 4608;                    // soft bits per block from 2 extended partitions per sideband = P3 frame bits (MP3 / MP11)
constexpr int PX_MEM = 32 * PX_MAX;             // interleaver IV memory (147456)
constexpr int PX_WORDS = PX_MAX / 32;           // 144 packed words per P3 / P4 frame
constexpr int PX_DEPUNCT = 3 * PX_MAX;          // 13824

// ---- AM (defines.h:13-38,44-60) ----
constexpr int AM_FFT = 256;
constexpr int AM_CP = 14;
constexpr int AM_SYM = AM_FFT + AM_CP;          // 270 samples per OFDM symbol @46511.71875 Hz
constexpr int AM_WIN = AM_SYM * (NSYM + 1);     // 8910
constexpr int AM_C = AM_FFT / 2;                // carrier bin after fftshift
constexpr int AM_PW = 25;                       // carriers per partition
constexpr int AM_IDX_MAX = 81;
constexpr int AM_MA3 = 2;                       // SERVICE_MODE_MA3
constexpr int AM_P1_LEN = 3750;
constexpr int AM_P3_LEN_MA1 = 24000, AM_P3_LEN_MA3 = 30000;
constexpr int AM_RAW_HIST = 480;                // raw cu8 samples kept for the 5-stage decimator's dependency cone (434)
constexpr int AM_SYMS = 8 * NSYM * AM_PW;       // 6400 hard symbols per partition per L1 frame
constexpr int AM_VIT = 3 * AM_P3_LEN_MA3;       // 90000: depunctured trellis input (8 P1 frames, or one P3 frame)
constexpr int AM_P1_WORDS = 118;                // packed words per 3750-bit P1 frame
constexpr int AM_P3_WORD0 = 8 * AM_P1_WORDS;    // 944: first word of the P3 frame inside an AM frame slot
constexpr int AM_DEC_P1 = (AM_P1_LEN + 64) * 4; // survivor-decision words (4 x u64 per step)
constexpr int AM_DEC_P3 = (AM_P3_LEN_MA3 + 64) * 4 + 9 * 4 * (PIDS_LEN + 64);   // + scratch for the window's 8 deferred PIDS decodes

enum { MODE_FM = 0, MODE_AM = 1 };                        // nrsc5.h:70-74
enum { SYNC_NONE = 0, SYNC_COARSE = 1, SYNC_FINE = 2 };   // input.h:18

// live-bin index <-> FFT bin (after fftshift, bin 1024 = DC)
__host__ __device__ inline int live_to_bin(int l) { return l < LIVE_HALF ? LB0 + l : UB0 + (l - LIVE_HALF); }
__host__ __device__ inline int bin_to_live(int b) { return b < FFT_N / 2 ? b - LB0 : LIVE_HALF + (b - UB0); }

struct c16 { int16_t r, i; };

// ---- what a reset leaves in the reference's FIR windows ------------------------------------------------------------
// firdecim_q15_reset only rewinds the window index to ntaps - 1 (firdecim_q15.c:53-56); the samples below it stay.  push()
// (firdecim_q15.c:58-67) copies the newest ntaps - 1 samples to the front of the 2048-sample window whenever the index reaches
// its end, i.e. at push number k (2048 - (ntaps - 1)), k >= 1, counted from the filter's last reset -- so after input_reset
// (input.c:126-138: the five decimator stages; acquire_reset, acquire.c:290-293: filter_fm and filter_am) the first outputs of a
// USED session see, as their history, the ntaps - 1 samples that preceded the filter's LAST compaction (zeros for a fresh
// session: calloc).  The engine keeps those samples per stream and nrsc5hip_stream_reset seeds hb_hist / fir_hist / AmStream::seed with them.
constexpr int FIRDECIM_WINDOW = 2048;           // WINDOW_SIZE, firdecim_q15.c:16
struct StaleWindows {
    c16 hb[14];                                 // decim[0] (both modes push it: FM samples, AM samples >> 4)
    c16 fir[2][31];                             // filter_fm, filter_am (acquire.c:312-313)
    c16 am_stage[4][14];                        // decim[1..4]: the AM cu8 cascade's later stages (input.c:76-88); their push counts follow from hb_pushed
    long long hb_pushed;                        // samples pushed since the reset
    long long fir_pushed[2];
};
// a window that had taken `a` samples since its reset takes n more: position, relative to the first of the n, of the first of the
// `hist` samples its last compaction inside this span moves to the front (>= -hist); STALE_NONE if the span holds no compaction
constexpr long long STALE_NONE = -(1ll << 40);
__host__ __device__ inline long long stale_start(long long a, long long n, int hist)
{
    const long long period = FIRDECIM_WINDOW - hist, b = a + n;
    if (n <= 0) return STALE_NONE;
    const long long kb = (b - 1) / period * period;            // the compaction runs inside push number kb (0-based), before it stores
    if (kb < period || kb < a) return STALE_NONE;
    return kb - a - hist;
}

// record flags
enum : uint32_t {
    REC_PROCESSED   = 1u << 0,   // a block was processed in this slot
    REC_TO_COARSE   = 1u << 1,   // sync_state changed to COARSE at the top of the block
    REC_TO_FINE     = 1u << 2,   // EVENT_SYNC fired in this block
    REC_MER         = 1u << 3,   // EVENT_MER fired
    REC_PIDS        = 1u << 4,   // a PIDS frame was decoded
    REC_P1          = 1u << 5,   // this block completed an L1 frame: P1 frame slot valid
    REC_LOST_SYNC   = 1u << 6,   // the block started from a host-forced NONE while FINE (input.c:177)
    REC_P3          = 1u << 7,   // a P3 frame completed (FM: odd blocks once the PX1 interleaver is primed, slot in `sis`; AM: block 7)
    REC_P4          = 1u << 8,   // FM MP11: a P4 frame completed (same slot)
    REC_PIDS_CRC    = 1u << 9,   // the PIDS frame passes pids_frame_push's CRC-12 (pids.c:52-86)
    REC_DISCARDED   = 1u << 10,  // window pipeline + l2_feedback: the block ran speculatively behind a P1 frame whose first L2 header
                                 // failed (frame.c:535-540); the stream was rewound to that frame and this record is void (k_replay.hip)
};

// One per (stream, processed block).  Plain-old-data, mirrored by include/nrsc5hip.h.
struct BlockRecord {
    uint32_t flags;
    int32_t state_before, state_after;
    int32_t samperr, cfo, keep, bc, psmi, cfo_wait, next_samperr;
    float prev_angle, phase_re, phase_im, next_angle;
    float freq_offset;          // EVENT_SYNC payload (input.c:181-184)
    float mer_lb, mer_ub;       // EVENT_MER payload
    float ber;                  // EVENT_BER payload (filled by the P1 decoder)
    int32_t p1_slot;            // index into the stream's P1 frame ring, or -1
    int32_t bc_decoded;         // block count the soft bits were filed under
    uint32_t pids[3];           // 80 descrambled PIDS bits, bit i at word i/32 bit i%32
    uint32_t sis;               // AM: pli | hppi << 1 | aabi << 2 | rdbi << 3 (EVENT_SYNC payload), bit 4 = valid; FM: P3/P4 frame slot
};

// one staged P3 / P4 decode (filled by k_px_deint, consumed by k_px_decode)
struct PxJob { int rec, slot, len, pad; };                      // rec < 0: empty

enum { NCO_CLOSED_FORM = 0, NCO_EXACT_FIRST_BLOCK = 1, NCO_EXACT_UNTIL_FINE = 2, NCO_EXACT_ALWAYS = 3 };   // DevBuffers::nco_policy (include/nrsc5hip.h: NRSC5HIP_TUNE_NCO_EXACT)

// Per-stream device-resident state ("the checkpoint").
struct StreamState {
    // decimated Q15 FIFO: absolute sample counters; q15[(abs - base)] addresses the slab
    long long wr, rd, base;
    // zero-copy batch (engine option batch_zero_copy): the stream's whole cu8 capture in the caller's device buffer; decimated
    // sample a is the half-band output over raw complex samples 2a-14 .. 2a (zeros before the start).  null: use the FIFO
    const uint8_t *raw;
    // K1 state
    c16 hb_hist[14];
    // acquisition state (acquire.h:25-29)
    c16 fir_hist[31];
    float prev_angle;
    double theta;               // NCO phase (the reference keeps a unit complex `phase`)
    int keep_extra, cfo;
    int sync_state;
    // sync state (sync.h:13-31)
    int psmi, cfo_wait, bc, samperr;
    float angle;
    int mer_cnt;
    float error_lb, error_ub;
    float costas_freq[LIVE_N], costas_phase[LIVE_N];
    // decode state (decode.h:23-24)
    int started_pm;
    // bookkeeping
    int nblocks;                // processed blocks so far (record index)
    int p1_count;               // P1 frames produced so far
    int fine_epoch;             // number of transitions into SYNC_FINE so far: a request from an earlier lock is stale
    // per-step scratch written by k_prepare / acquisition
    int active;                 // this step processes a block for this stream
    int samperr_cur; int pad0;
    double dtheta;              // effective NCO step (rad/sample) for the current block
    double growth;              // |phase_increment| - 1 of the current block: the per-sample amplitude drift of the reference's oscillator (prepare_block.h)
    // The reference's oscillator as the float complex it is (acquire_t.phase, acquire.h:28), bit for bit, for as long as every block since the
    // stream's reset has advanced it by the reference's own recurrence (k_nco_exact): nco_exact = 1.  The first block that runs on the
    // closed-form phasor clears it for good (until the next reset).  nco_mode: the CURRENT block takes its phasors from db.nco_tab.
    float nco_re, nco_im, inc_re, inc_im;
    int nco_exact, nco_mode;
    int coarse_samperr; float coarse_re, coarse_im;
    // P1 hand-off, one slot per in-flight decode window (see engine.hip: P1 pipeline); `parity` = window % NWIN
    int p1_pending[NWIN];          // 1: frame completed this step (gather it), 2: gathered into coded[s][parity]
    int p1_slot[NWIN];             // slot of the stream's P1 ring the decoder must fill
    int p1_record[NWIN];           // record index that gets the BER
    int p1_epoch[NWIN];            // fine_epoch the frame was received in
    int p1_l2slot[NWIN];           // slot + 1 of a freshly decoded frame awaiting k_l2_index_window, 0 = none
    int p1_endlane[NWIN];
    int p1_pmslot[NWIN];        // which of the stream's NPM soft-bit matrices holds the frame
    // replay (k_replay.hip): verdict of the frame's first L2 header once its deferred decode is done (0: none yet / consumed,
    // 1: ok, 2: failed), absolute index of the record that announced the frame, absolute decode window it belongs to
    int p1_verdict[NWIN], p1_recabs[NWIN], p1_window[NWIN];
    int ndiscard;               // records marked REC_DISCARDED so far
    int pm_slot;                // matrix being filled; advances after every block 15
    int last_pm_slot;           // matrix that received the most recent block (debug fetch)
    int mode;                   // MODE_FM / MODE_AM (nrsc5_set_mode)
    // extended-sideband interleavers (interleaver_iv_t, decode.h:9-17): both channels advance in lock step
    int px_pos, px_ready, px_started;
    int px_count;               // block pairs that produced frames so far (ring slot)
    int px_go;                  // this step: soft bits per block of a completed pair (0: nothing to de-interleave)
    int px_nch, px_slot, px_record;
    c16 hb_next[14];            // streaming decimator: the chunk's last 14 input samples, parked by the workgroup that holds them until the chunk's last workgroup rolls hb_hist
    StaleWindows stale;         // survives nrsc5hip_stream_reset (the reference's rewound windows); cleared by a fresh session
};

// AM-only per-stream state (allocated when the engine is created with am_enable)
struct AmStream {
    // K1-AM: raw cu8 samples consumed so far and the newest AM_RAW_HIST of them (I,Q bytes, oldest first)
    long long raw_count;
    uint8_t raw_hist[2 * AM_RAW_HIST];
    c16 seed[5][14];            // what the five stages' windows held in front of the first sample after the reset (StaleWindows; zeros for a fresh session)
    // system control bits latched from the reference carrier at block 0 (sync.h:17-21), bc history (sync.c:648-653)
    int pli, hppi, aabi, rdbi;
    unsigned offset_history;
    // decode.h:31-32
    unsigned am_errors; int am_diversity_wait;
    int q_head;                 // oldest third of the 3-frame diversity delay lines
    // hand-off from k_am_block to the decode kernels of the same step
    int dec_bc;                 // block count the symbols were filed under, -1: nothing to decode
    int dec_record;             // record index of that block
    int dec_rdbi, dec_psmi;
    int frame_slot;             // slot of the frame ring that receives this L1 frame's P1/P3 frames
    int next_slot;              // slot assigned (by the de-interleaver) to the frame whose trellis inputs it just produced
    int vit_parity;             // window pipeline: decode job (window parity) of the L1 frame being delivered
    int next_job;               // ... and of the frame whose trellis inputs the de-interleaver just produced
    unsigned il_done;           // slices of k_am_interleave that have finished this frame (the last one commits); 0 between launches
};

// window pipeline: one L1 frame's worth of decodes (8 x P1 + P3) handed to k_am_decode
struct AmJob {
    int valid, slot, psmi, rdbi; unsigned errors; int done; int epoch; int pad;
    // replay (window pipeline + l2_feedback): verdict of frame_process's first-header check for each of the frame's eight P1 PDUs
    // (0 unknown, 1 good, 2 failed, 3 failed and applied), the absolute index of the block record that delivered it (-1: not
    // yet) and the decode window the job was filed in
    int verdict[8], deliver_abs[8], window, pad2;
};
// K=9 decode in segment waves (k_am.hip): what the segment waves of one P3 frame leave for the wave that checks their boundaries,
// and the end states of the L1 frame's eight P1 frames (forward and traceback are separate launches)
constexpr int K9_GMAX = 8;            // segment waves per frame at most
constexpr int K9_WARM = 3;            // forward warm-up of a segment wave: chunks of 64 step pairs (384 trellis steps)
constexpr int K9_TB_RUNIN = 4;        // traceback run-in of a segment wave: chunks of 32 step pairs (256 trellis steps)
struct K9Meta {
    int snap[K9_GMAX][256];            // forward: metrics segment g enters its first own chunk with (after its warm-up)
    int uend[K9_GMAX][256];            // ... and leaves its last chunk with
    unsigned end_state;                // of the P3 frame (k_am_decode_fix)
    unsigned arrive[K9_GMAX], leave[K9_GMAX];   // traceback: state segment g enters / leaves its own chunks with
    unsigned p1_end[8];
    unsigned pad[7];
};
// replay checkpoint of an AM stream: everything k_rollback_am rewinds, as of the end of a block that delivered a P1 PDU
struct AmCkpt { StreamState st; AmStream am; };

}  // namespace nrsc5
