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
constexpr int P1_WORDS = P1_LEN / 32;  // packed output words per P1 frame (4568)
constexpr int NWIN = 8;                // decode windows (16 block steps each) that may be in flight: buffers indexed w % NWIN
constexpr int NPM = NWIN + 3;           // soft-bit matrices per stream: a frame's matrix must outlive its (deferred) decode
constexpr int NAUX = 5;                // HIP streams that decode windows concurrently (each with its own decision scratch)

enum { SYNC_NONE = 0, SYNC_COARSE = 1, SYNC_FINE = 2 };   // input.h:18

// live-bin index <-> FFT bin (after fftshift, bin 1024 = DC)
__host__ __device__ inline int live_to_bin(int l) { return l < LIVE_HALF ? LB0 + l : UB0 + (l - LIVE_HALF); }
__host__ __device__ inline int bin_to_live(int b) { return b < FFT_N / 2 ? b - LB0 : LIVE_HALF + (b - UB0); }

struct c16 { int16_t r, i; };

// record flags
enum : uint32_t {
    REC_PROCESSED   = 1u << 0,   // a block was processed in this slot
    REC_TO_COARSE   = 1u << 1,   // sync_state changed to COARSE at the top of the block
    REC_TO_FINE     = 1u << 2,   // EVENT_SYNC fired in this block
    REC_MER         = 1u << 3,   // EVENT_MER fired
    REC_PIDS        = 1u << 4,   // a PIDS frame was decoded
    REC_P1          = 1u << 5,   // this block completed an L1 frame: P1 frame slot valid
    REC_LOST_SYNC   = 1u << 6,   // the block started from a host-forced NONE while FINE (input.c:177)
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
    uint32_t pad;
};

// Per-stream device-resident state ("the checkpoint").
struct StreamState {
    // decimated Q15 FIFO: absolute sample counters; q15[(abs - base)] addresses the slab
    long long wr, rd, base;
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
    int force_none;             // host request: drop to SYNC_NONE before the next block
    // per-step scratch written by k_prepare / acquisition
    int active;                 // this step processes a block for this stream
    int samperr_cur; int pad0;
    double dtheta;              // effective NCO step (rad/sample) for the current block
    int coarse_samperr; float coarse_re, coarse_im;
    // P1 hand-off, one slot per in-flight decode window (see engine.hip: P1 pipeline); `parity` = window % NWIN
    int p1_pending[NWIN];          // 1: frame completed this step (gather it), 2: gathered into coded[s][parity]
    int p1_slot[NWIN];             // slot of the stream's P1 ring the decoder must fill
    int p1_record[NWIN];           // record index that gets the BER
    int p1_endlane[NWIN];
    int p1_pmslot[NWIN];        // which of the stream's NPM soft-bit matrices holds the frame
    int pm_slot;                // matrix being filled; advances after every block 15
    int last_pm_slot;           // matrix that received the most recent block (debug fetch)          // forward pass -> traceback hand-off (lane of the winning end state)
};

}  // namespace nrsc5
