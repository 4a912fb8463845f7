/* libnrsc5hip -- C ABI of the MI355X (gfx950) NRSC-5 demodulate/decode engine.
 *
 * This is the drop-in boundary for the reference's IQ -> L1-frame hot path.  It replaces the
 * internal seam of theori-io/nrsc5 between L4 (src/nrsc5.c) and L2 (src/frame.c, src/pids.c):
 *
 *   down-calls replaced                      reference interface (file:line)
 *   ---------------------------------------  ----------------------------------------------
 *   nrsc5hip_push_cu8                        input_push_cu8      src/input.h:42, input.c:96-117
 *   nrsc5hip_push_cs16                       input_push_cs16     src/input.h:43, input.c:119-124
 *   nrsc5hip_stream_reset                    input_reset         src/input.h:39, input.c:126-138
 *   nrsc5hip_stream_fresh                    nrsc5_close + nrsc5_open_pipe of one slot   src/nrsc5.c
 *   nrsc5hip_force_resync                    input_set_sync_state(st, SYNC_STATE_NONE) as called
 *                                            by frame_process    src/frame.c:535-540
 *   up-calls delivered as ordered records    output_advance (acquire.c:108), nrsc5_report_sync /
 *   (nrsc5hip_drain)                         _lost_sync (input.c:177-185), nrsc5_report_mer
 *                                            (sync.c:490-501), nrsc5_report_ber (decode.c:458),
 *                                            pids_frame_push (decode.c:471, pids.h:98),
 *                                            frame_push (decode.c:460, frame.h:53)
 *
 * plus an additive batch API (device-resident captures, many independent streams per call) that the
 * reference has no counterpart for.  All functions return 0 on success and a negative NRSC5HIP_E*
 * code on failure; nothing throws across the boundary; no C++/torch types appear in signatures.
 * Buffers passed in are borrowed for the duration of the call.  INTEGRATION.md shows the binding a
 * maintainer of the reference would add (src/input.c replacement + CMake lines).
 */
#ifndef NRSC5HIP_H_
#define NRSC5HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NRSC5HIP_OK            0
#define NRSC5HIP_EINVAL       (-1)
#define NRSC5HIP_ENOMEM       (-2)
#define NRSC5HIP_EHIP         (-3)   /* a HIP runtime call failed; see nrsc5hip_last_error() */
#define NRSC5HIP_EOVERFLOW    (-4)   /* FIFO / record ring / frame ring capacity exceeded */

#define NRSC5HIP_P1_FRAME_BITS   146176   /* P1_FRAME_LEN_FM, defines.h:42 */
#define NRSC5HIP_P1_FRAME_WORDS  4568
#define NRSC5HIP_PIDS_FRAME_BITS 80       /* PIDS_FRAME_LEN, defines.h:47 */

/* waveform of a stream: NRSC5_MODE_FM / NRSC5_MODE_AM, nrsc5.h:70-74 */
enum { NRSC5HIP_MODE_FM = 0, NRSC5HIP_MODE_AM = 1 };
#define NRSC5HIP_AM_P1_FRAME_BITS  3750    /* P1_FRAME_LEN_AM, defines.h:43 */
#define NRSC5HIP_AM_P3_BITS_MA1    24000   /* P3_FRAME_LEN_MA1, defines.h:53 */
#define NRSC5HIP_AM_P3_BITS_MA3    30000   /* P3_FRAME_LEN_MA3, defines.h:54 */

/* sync states, input.h:18 */
enum { NRSC5HIP_SYNC_NONE = 0, NRSC5HIP_SYNC_COARSE = 1, NRSC5HIP_SYNC_FINE = 2 };

/* nrsc5hip_record.flags */
enum {
    NRSC5HIP_REC_PROCESSED = 1u << 0,  /* one acquire block was processed: call output_advance() first */
    NRSC5HIP_REC_TO_COARSE = 1u << 1,  /* sync state NONE -> COARSE at the top of this block */
    NRSC5HIP_REC_TO_FINE   = 1u << 2,  /* sync achieved in this block: nrsc5_report_sync(freq_offset, psmi) */
    NRSC5HIP_REC_MER       = 1u << 3,  /* nrsc5_report_mer(mer_lb, mer_ub) */
    NRSC5HIP_REC_PIDS      = 1u << 4,  /* pids_frame_push(pids) */
    NRSC5HIP_REC_P1        = 1u << 5,  /* nrsc5_report_ber(ber); frame_push(P1 frame in slot p1_slot) */
    NRSC5HIP_REC_LOST_SYNC = 1u << 6,  /* l2_feedback only: after this block's frames the engine dropped the stream to NONE (nrsc5_report_lost_sync) */
    NRSC5HIP_REC_P3        = 1u << 7,  /* FM (MP2/MP3/MP11, odd blocks): frame_push(P3 frame of PX slot `sis`, 2304 or 4608 bits);
                                          AM, block 7: frame_push(P3 frame of slot p1_slot), then nrsc5_report_ber(ber) */
    NRSC5HIP_REC_P4        = 1u << 8,  /* FM MP11: frame_push(P4 frame of PX slot `sis`, 4608 bits) */
    NRSC5HIP_REC_PIDS_CRC  = 1u << 9,  /* with REC_PIDS: the frame passes pids_frame_push's CRC-12 (pids.c:52-86), i.e. sis_decode will see it */
    NRSC5HIP_REC_DISCARDED = 1u << 10  /* internal (p1_async + l2_feedback): a block that ran speculatively behind a P1 frame whose first L2 header
                                          failed; the stream was rewound to that frame.  nrsc5hip_drain / _batch_fetch* never deliver such records */
};

/* One record per processed 32-symbol block, in stream order.  Events implied by one record fire in
 * the order of the flag bits above (which is the reference's order inside acquire_process). */
typedef struct nrsc5hip_record {
    uint32_t flags;
    int32_t state_before, state_after;   /* sync state entering / leaving the block */
    int32_t samperr;                     /* timing pick used for this block, 0..2159 (acquire.c:112,149) */
    int32_t cfo;                         /* integer carrier offset in bins after this block */
    int32_t keep;                        /* samples carried to the next window (acquire.c:259) */
    int32_t bc;                          /* block count after this block */
    int32_t psmi;
    int32_t cfo_wait;
    int32_t next_samperr;                /* tracking feedback for the next block (sync.c:455) */
    float prev_angle;                    /* CP-lag phase estimate, rad per 2048 samples */
    float phase_re, phase_im;            /* NCO phase after the block */
    float next_angle;                    /* residual CFO feedback (sync.c:458) */
    float freq_offset;                   /* Hz, valid with REC_TO_FINE (input.c:181-184) */
    float mer_lb, mer_ub;                /* dB, valid with REC_MER */
    float ber;                           /* valid with REC_P1 */
    int32_t p1_slot;                     /* valid with REC_P1 */
    int32_t bc_decoded;                  /* block count the PIDS frame / soft bits belong to, or -1 */
    uint32_t pids[3];                    /* valid with REC_PIDS: bit i of the frame at pids[i/32] bit i%32 */
    uint32_t sis;                        /* AM with REC_TO_FINE: pli | hppi << 1 | aabi << 2 | rdbi << 3 | 16 (sync.c:231-235);
                                            FM with REC_P3 / REC_P4: slot of the P3/P4 frame ring (8 * p1_slots slots) */
} nrsc5hip_record;

typedef struct nrsc5hip_config {
    int device;                /* HIP device ordinal */
    int max_streams;           /* independent IQ streams resident in this engine */
    long long q15_capacity;    /* decimated (744187.5 S/s) samples of FIFO per stream; >= 2 * 71280.
                                  Batch use: capture length / 2 + 64. */
    int record_capacity;       /* block records retained per stream between drains (>= 64) */
    int p1_slots;              /* decoded P1 frames retained per stream between drains (>= 2) */
    int p1_async;              /* 0: decode each P1 frame before the next block of any stream (exact
                                  reference event timing; required for nrsc5hip_force_resync feedback);
                                  1: decode the frames of each 16-block window on a second HIP stream,
                                  overlapped with the next window (throughput mode) */
    int l2_feedback;           /* 1: the engine itself applies the L2 -> L1 feedback of frame_process (frame.c:516-540): a P1 frame
                                  whose first L2 header fails the RS(255,247) check drops the stream to SYNC_NONE (REC_LOST_SYNC)
                                  before its next block, as in the reference.  With p1_async = 1 (FM and AM) the verdict of a deferred
                                  decode arrives windows later: the stream is then rewound to the end of the block that delivered the
                                  frame and re-run from there, so the delivered records and frames are the reference's all the same
                                  (needs the samples since that block still in the FIFO: batch use, or pushes whose records are
                                  drained after each call; and record_capacity >= 256).
                                  0: the host does it through nrsc5hip_force_resync (the drop-in shim). */
    int am_enable;             /* allocate the AM buffers (1.4 MB per stream) so that streams may be switched to
                                  NRSC5HIP_MODE_AM */
    int batch_zero_copy;       /* 1: nrsc5hip_batch_append_cu8 on freshly reset FM streams does NOT decimate into the FIFO: the engine keeps a
                                  reference to the caller's device buffer and the block steps read the cu8 samples directly (half-band
                                  fused into the symbol kernel: the capture is read from HBM once, no 4-byte-per-sample Q15 copy).
                                  The buffer must stay valid and unchanged until those streams are reset; one append per stream and
                                  reset; q15_capacity may then be the minimum (2 * 71280).  0: samples are copied (decimated) into the
                                  engine's FIFO during the call, as the streaming seam does. */
    int l2_index;              /* 1: every FM P1 frame is also indexed on the decode stream right after its traceback
                                  (nrsc5hip_l2_frame per ring slot, read with nrsc5hip_l2_frame_get / nrsc5hip_batch_fetch_l2), and
                                  so are the P3 / P4 frames and the frames of AM streams (nrsc5hip_batch_fetch_l2_px / _am);
                                  0: indexes only on request (nrsc5hip_l2_index) */
} nrsc5hip_config;

typedef struct nrsc5hip_engine nrsc5hip_engine;

/* Devices and threads.  An engine lives on cfg.device; EVERY entry point that takes an engine switches the calling thread to that
 * device for the duration of the call and restores the thread's previous current device on return, so one process may own one
 * engine per GPU (integration/batch_shard.c: N host threads x one engine per visible GPU) whatever hipSetDevice the caller last
 * made.  An engine is NOT re-entrant: at most one thread may be inside calls on the SAME engine at a time (give each thread its
 * own engine, or serialise); different engines -- on the same or on different devices -- may be driven concurrently from different
 * threads: nothing is shared between engines (no process-global scratch, per-thread error string and seam counters).
 * nrsc5hip_last_error() returns the calling thread's last message.  Device pointers handed to the batch entry points must belong
 * to the engine's device.  The window pipeline (p1_async = 1) needs GPU_MAX_HW_QUEUES >= 8 in the environment before the HIP runtime
 * initialises (one hardware queue per chain / decode stream); engine creation warns on stderr when it is lower. */
int nrsc5hip_engine_create(const nrsc5hip_config *cfg, nrsc5hip_engine **out);
void nrsc5hip_engine_destroy(nrsc5hip_engine *e);
const char *nrsc5hip_last_error(void);
/* fingerprint of the device sources this library was built from (nrsc5_amd/build.py: source_sha) -- measurements are only taken
 * with a library that matches the tree */
const char *nrsc5hip_source_sha(void);
/* hipStream_t the engine launches on, as void* (so that callers can order their own work) */
void *nrsc5hip_engine_hip_stream(nrsc5hip_engine *e);

/* Device helpers for hosts that do not link the HIP runtime themselves: number of visible GPUs; a device buffer on `device` filled
 * from host memory (host may be NULL: allocation only); its release.  The pointers are what the batch entry points take. */
int nrsc5hip_device_count(int *n);
int nrsc5hip_device_upload(int device, const void *host, size_t nbytes, void **dev_out);
int nrsc5hip_device_free(int device, void *dev);

/* ---- streaming seam (host buffers), one stream at a time --------------------------------------- */
/* input_push_cu8 (input.c:96-117): nbytes % 4 == 0.  Decimates, appends, and processes every block
 * whose 33-symbol window is complete; records are then available through nrsc5hip_drain.  With p1_async = 0 the host keeps a
 * mirror of the stream's read position.  FM cu8 (round 6, NRSC5HIP_TUNE_HOST_CAPTURE): the bytes stay, as pushed, in a pinned device-mapped capture
 * that the stream reads in place -- a push that completes no block is ONE host memcpy and nothing else; the block step is the symbol kernel
 * (half-band fused, loading across PCIe) + the sync kernel, which posts the block's record into pinned host memory itself.  Other input (cs16, AM,
 * or with the knob at 0): a push that completes no block is one memcpy into pinned staging + the decimator launch (no synchronisation).  Either way a
 * push that completes a block ends with the host spinning on the report's sequence number, the block's record already in host memory. */
int nrsc5hip_push_cu8(nrsc5hip_engine *e, int stream, const uint8_t *iq, uint32_t nbytes);
/* input_push_cs16 (input.c:119-124): n = number of int16 values, n % 2 == 0 */
int nrsc5hip_push_cs16(nrsc5hip_engine *e, int stream, const int16_t *iq, uint32_t n);
/* input_reset (input.c:126-138) on a session that may have been used: like the reference's, the reset rewinds the FIR windows without clearing them
 * (firdecim_q15_reset, firdecim_q15.c:53-56) -- the first 7 samples out of the FM half-band and the acquisition filter's first 31 outputs (filter_fm or
 * filter_am, acquire.c:290-293) see the samples the window's last compaction left at its front, exactly as a second capture on one nrsc5_t does
 * (tests: engine_checks.check_reset_keeps_fir_windows vs the unmodified reference).  Bytes of a partial push still staged on the host pass through the
 * decimator first.  The five stages of the AM cu8 cascade (input.c:70-88) are covered as well, and so is what sync_reset (sync.c:810-830) leaves alone:
 * sync_t.samperr, .angle and .bc keep their values -- visible in the block records of the next capture's un-synchronised blocks, and, because the AM path never
 * writes .angle, in the angle of the first synchronised block of an AM session that follows an FM one (acquire.c:115-118).  A LOST_SYNC that the reference
 * fires inside the reset (input_set_sync_state) is the caller's own doing and no record.  Engines with batch_zero_copy treat every reset as a fresh session. */
int nrsc5hip_stream_reset(nrsc5hip_engine *e, int stream);
/* ABI NOTE (NRSC5HIP_ABI_VERSION >= 5): until round 4 nrsc5hip_stream_reset gave a FRESH session; since round 5 it is the reference's input_reset as described above (stale FIR
 * windows, samperr / angle / bc kept) and the fresh session is nrsc5hip_stream_fresh.  A caller that used reset to start an independent capture on a slot must call
 * nrsc5hip_stream_fresh now (on engines with batch_zero_copy both are the fresh form).  nrsc5hip_abi_version() lets a binding check what it was linked against. */
#define NRSC5HIP_ABI_VERSION 7   /* 7: + NRSC5HIP_TUNE_HOST_CAPTURE / _FOLD_REPORT, nrsc5hip_debug_host_capture_stats; 6: + nrsc5hip_abi_version, nrsc5hip_debug_flow_stats, NRSC5HIP_TUNE_FLOW_MIN / _LOOP_EXACT, NRSC5HIP_PROF_FLOW; 5: the reset semantics above */
int nrsc5hip_abi_version(void);
/* nrsc5_close + nrsc5_open_pipe on this slot: a fresh session (calloc'd windows), what nrsc5hip_reset_all does for every stream */
int nrsc5hip_stream_fresh(nrsc5hip_engine *e, int stream);
/* nrsc5_set_mode -> input_set_mode (nrsc5.h:754, input.c:158-162): NRSC5HIP_MODE_FM (default) or _AM; resets the stream.
 * AM: cu8 pushes go through the 5-stage 32:1 decimator, cs16 pushes are 46511.71875 S/s samples (input.c:70-91,119-124);
 * every FINE block yields a PIDS frame and (after the 4-frame diversity start-up) one 3750-bit P1 frame, block 7 also the
 * P3 frame and the BER.  Event order inside an AM record: TO_FINE, PIDS, P1 frame, P3 frame, BER (decode.c:507-554). */
int nrsc5hip_stream_set_mode(nrsc5hip_engine *e, int stream, int mode);
/* L2 feedback (frame.c:535-540): the stream drops to SYNC_NONE before its next block */
int nrsc5hip_force_resync(nrsc5hip_engine *e, int stream);
/* Input bytes of this format (cu8 != 0: cu8, else cs16) whose push completes the stream's next 32-symbol block.  A caller that
 * applies the L2 feedback itself (frame.c through nrsc5hip_force_resync: the drop-in) pushes at most this much per call, so the
 * frames of one block reach L2 before the next block is processed -- the reference's event order for any push size.  -1: not
 * known (p1_async engines, or a stream that the batch entry points touched since its reset): push <= 17280 bytes per call. */
long long nrsc5hip_bytes_to_next_block(nrsc5hip_engine *e, int stream, int cu8);
/* Deferred wait (p1_async = 0).  A block that starts in SYNC_FINE consumes a number of samples the host can compute from the
 * previous block's record (keep = 2160 - next_samperr, acquire.c:112,259), so the push that completes such a block returns with
 * the block step still running: the device works on block n while the caller reads and pushes the samples of block n + 1.
 * nrsc5hip_bytes_to_next_block stays exact.  Every other entry point (nrsc5hip_drain first of all) waits for the step before it
 * does anything, so a caller that never heard of this sees the old behaviour; a caller that wants the overlap polls with
 * nrsc5hip_drain_ready, which hands out what has been reported so far and never waits.
 *
 * Manual stepping lets the L2 feedback of block n (frame.c -> nrsc5hip_force_resync) reach the engine before block n + 1 is
 * STEPPED while the samples of block n + 1 are already in the capture / on their way into the FIFO: with it set, a push that completes a block
 * copies (or submits) the samples and returns; the caller then drains block n (nrsc5hip_drain, waits), feeds L2, and calls
 * nrsc5hip_stream_step.  integration/input_hip.c does exactly that.  (A push that finds an unstepped complete block steps it
 * itself: forgetting the call costs speed, never samples.) */
int nrsc5hip_drain_ready(nrsc5hip_engine *e, int stream, nrsc5hip_record *out, int max, int *n_out);
int nrsc5hip_stream_set_manual_step(nrsc5hip_engine *e, int stream, int on);
int nrsc5hip_stream_step(nrsc5hip_engine *e, int stream);
/* One step further: when the block step still in flight started in SYNC_FINE and cannot complete a P1 frame (15 of 16 blocks), nothing
 * its delivery tells the host can change what the next block does -- frame_process's only way back into L1 is the first header of a
 * P1 frame (frame.c:535-540) -- so the next block's step may be queued BEHIND it before the host has even looked at it:
 * nrsc5hip_stream_step_ahead does that if it is safe and says so in *submitted; the nrsc5hip_drain that follows waits for the older
 * step only.  The device then runs block n + 1 while the host hands block n to L2.
 * CONTRACT: between a nrsc5hip_stream_step_ahead that reported *submitted = 1 and the nrsc5hip_drain that follows it, the caller must not issue
 * anything that changes the stream's L1 state (nrsc5hip_force_resync, nrsc5hip_stream_reset, nrsc5hip_stream_set_mode): the block queued ahead has
 * already been submitted with the old state, so the change would land one block late.  (integration/input_hip.c cannot violate it: those calls
 * only come from frame.c during a delivery, and a delivery that can produce them -- a block that may end a P1 frame -- is never stepped ahead of.)
 * Should the engine find its own prediction violated (a block submitted without its P1 decode completed a frame while the next one is running)
 * it drops both steps' bookkeeping, marks the stream's host mirror invalid (the next push re-synchronises) and returns NRSC5HIP_EHIP.  The events of those two
 * blocks are then NOT delivered through the seam (their records remain in the device ring for nrsc5hip_drain; the first block's P1 frame has no decode): the
 * session has failed and says so -- it never continues silently. */
int nrsc5hip_stream_step_ahead(nrsc5hip_engine *e, int stream, int *submitted);
/* EVENT LATENCY of the drop-in built on these calls (integration/input_hip.c; INTEGRATION.md, first section).  DEFAULT since round 6: the shim waits inside the call that
 * completes a block and delivers that block's events before it returns -- the reference's contract (src/input.c:41-50).  With NRSC5HIP_OVERLAP_DELIVERY=1 (deferred waits,
 * steps queued ahead: ~1.5 x the throughput) the records of block n reach the caller during the first call after the device has finished it -- at the latest in the call
 * that completes block n + 1, i.e. up to one block (92.9 ms of signal) later than src/input.c fires them -- or in a zero-length push (flush), nrsc5hip_drain (waits),
 * stream reset / engine destruction paths of the shim. */

/* ---- batch path (device buffers) ------------------------------------------------------------------ */
/* Decimate + append one cu8 chunk per listed stream.  dev_iq: device pointer, chunk k at
 * dev_iq + k * stride_bytes (16-byte aligned), nbytes[k] bytes each (host array, % 4 == 0). */
int nrsc5hip_batch_append_cu8(nrsc5hip_engine *e, int nstreams, const int *stream_ids,
                              const uint8_t *dev_iq, long long stride_bytes, const uint32_t *nbytes);
int nrsc5hip_batch_append_cs16(nrsc5hip_engine *e, int nstreams, const int *stream_ids,
                               const int16_t *dev_iq, long long stride_elems, const uint32_t *nelems);
/* Run block steps (every listed stream advances by at most one block per step) until no listed stream
 * has a complete window or max_steps is reached; *steps_done gets the number of steps that did work. */
int nrsc5hip_batch_process(nrsc5hip_engine *e, int nstreams, const int *stream_ids, int max_steps, int *steps_done);

/* ---- results ----------------------------------------------------------------------------------------- */
/* Copies up to max records of `stream`, oldest first, that were produced since the previous drain. */
int nrsc5hip_drain(nrsc5hip_engine *e, int stream, nrsc5hip_record *out, int max, int *n_out);
/* P1 frame of a REC_P1 record, packed (bit i at words[i/32] bit i%32) ... */
int nrsc5hip_p1_frame_packed(nrsc5hip_engine *e, int stream, int slot, uint32_t *words /* [4568] */);
/* ... or one bit per byte, the layout frame_push() takes (frame.h:53) */
int nrsc5hip_p1_frame_bits(nrsc5hip_engine *e, int stream, int slot, uint8_t *bits /* [146176] */);
/* FM extended sidebands (decode_push_px1/px2, decode.c:393-437): P3 (channel 0) / P4 (channel 1) frame of a REC_P3 / REC_P4
 * record, slot = record.sis, nbits = 2304 (MP2) or 4608 (MP3 / MP11), one bit per byte as frame_push() takes it */
int nrsc5hip_px_frame_bits(nrsc5hip_engine *e, int stream, int slot, int channel, int nbits, uint8_t *bits);
/* every P3/P4 slot of the listed streams, packed: frames[nstreams][8 * p1_slots][2][144] */
int nrsc5hip_batch_fetch_px(nrsc5hip_engine *e, int nstreams, const int *stream_ids, uint32_t *frames);
/* AM frames of a REC_P1 / REC_P3 record, one bit per byte as frame_push() takes them: which = 0..7 selects the P1 frame
 * of that block (nbits 3750), which = 8 the P3 frame (nbits 24000 for MA1, 30000 for MA3).  In the packed slot
 * (nrsc5hip_p1_frame_packed / batch_fetch) P1 frame b starts at word 118 b and the P3 frame at word 944. */
int nrsc5hip_am_frame_bits(nrsc5hip_engine *e, int stream, int slot, int which, int nbits, uint8_t *bits);
/* Bulk D2H of every record/frame produced by a batch: records[nstreams][max_records],
 * counts[nstreams]; frames may be NULL, else frames[nstreams][p1_slots][4568] */
int nrsc5hip_batch_fetch(nrsc5hip_engine *e, int nstreams, const int *stream_ids, nrsc5hip_record *records,
                         int max_records, int *counts, uint32_t *frames);
/* Zero-copy variant for streams 0..nstreams-1: bulk D2H into engine-owned pinned buffers; *records points at
 * [nstreams][record_capacity] records, *frames (may be NULL) at [nstreams][p1_slots][4568] words; valid until the
 * next fetch / reset.  Requires an undrained, unwrapped record ring (batch use after nrsc5hip_reset_all). */
int nrsc5hip_batch_fetch_view(nrsc5hip_engine *e, int nstreams, const nrsc5hip_record **records, int *counts, const uint32_t **frames);
/* unpack helper (host only) */
void nrsc5hip_unpack_bits(const uint32_t *words, int nbits, uint8_t *bits);

/* ---- L2 audio transport index (frame.c:516-714), an additive post-pass over decoded frames that are still in HBM ----
 * frame_push (PCI extraction, bit order) and the audio walk of frame_process run on the device: every audio PDU's
 * RS(255,247)-corrected header, its locators, header expansion fields, PSD span and the CRC-8 verdict of each packet,
 * i.e. what frame_process hands to output_align / parse_hdlc / output_push, as offsets into the frame's PDU bytes.
 * A host that takes the index need not touch the 146 176 bits of a P1 frame one by one (frame.c:688-709). */
#define NRSC5HIP_L2_MAX_PDUS     16
#define NRSC5HIP_L2_MAX_PACKETS  64      /* MAX_AUDIO_PACKETS, frame.c:30 */
#define NRSC5HIP_L2_MAX_BYTES    18269   /* MAX_PDU_LEN, defines.h:63 */
enum {                                   /* nrsc5hip_l2_frame.status: why the walk ended */
    NRSC5HIP_L2_END = 0,                 /* ran to the end of the frame (frame.c:525) */
    NRSC5HIP_L2_NO_AUDIO,                /* !has_audio (frame.c:522) */
    NRSC5HIP_L2_FIXED_DATA,              /* (unused since round 2: frames with fixed-data sub-channels ARE indexed, with audio_end = nbytes - 1, the
                                            largest value process_fixed_data can return; the consumer cuts the index back with the true value:
                                            nrsc5hip_l2_apply_audio_end, done inside nrsc5hip_hdc_push_frame) */
    NRSC5HIP_L2_HEADER_RS,               /* fix_header failed (frame.c:534-541); lost_sync tells whether this drops the receiver to SYNC_STATE_NONE */
    NRSC5HIP_L2_BAD_LOCATORS,            /* one of the returns of frame.c:547-556 (typical: zero padding after the last PDU) */
    NRSC5HIP_L2_TOO_MANY_PDUS,           /* more than NRSC5HIP_L2_MAX_PDUS audio PDUs */
    NRSC5HIP_L2_HEF_OVERRUN,             /* header expansion runs past la_location: the reference's parse_hdlc length wraps (undefined) */
    NRSC5HIP_L2_BAD_STREAM,              /* stream_id >= MAX_STREAMS with nop == 0: the reference reads locations[-1] */
    NRSC5HIP_L2_BAD_LENGTH,              /* not one of frame_push's six frame lengths (frame.c:683) */
    NRSC5HIP_L2_AUDIO_END                /* set by nrsc5hip_l2_apply_audio_end: the walk ended where the fixed-data region begins (frame.c:525,547-556) */
};
/* PCI values that announce fixed-data sub-channels next to audio (has_fixed && has_audio, frame.c:138-151) */
#define NRSC5HIP_L2_PCI_HAS_FIXED(pci) ((((pci) & 0xFFFFFCu) == (0xE3634Cu & 0xFFFFFCu)) || (((pci) & 0xFFFFFCu) == (0x8D8D33u & 0xFFFFFCu)))
typedef struct nrsc5hip_l2_pdu {
    uint32_t start;                      /* offset of the PDU (its RS parity bytes) in the frame's PDU bytes */
    uint32_t psd_off; int32_t psd_len;   /* the span frame_process gives parse_hdlc (frame.c:611) */
    uint32_t audio_off;                  /* first byte of packet 0 = start + la_location + 1 */
    uint32_t crc_bad_lo, crc_bad_hi;     /* bit j: packet j fails its CRC-8 (PACKET_FLAG_CRC_ERROR, frame.c:616-627) */
    uint32_t pdu_marker;                 /* HEF class 4 */
    uint16_t hef_pdu_len;                /* HEF class 1 */
    uint16_t loc[NRSC5HIP_L2_MAX_PACKETS]; /* offset of packet j's CRC byte; packet j = [loc[j-1] + 1 (audio_off for j = 0), loc[j]) */
    uint8_t codec_mode, stream_id, pdu_seq, blend_control, per_stream_delay, common_delay, latency, pfirst, plast, seq, nop,
            hef, la_location;            /* parse_header, frame.c:181-196 */
    uint8_t rs_corrections;              /* symbols fix_header corrected in this header */
    uint8_t class_ind, prog_num, access, prog_type, applied_services;   /* parse_hef, frame.c:198-265 */
    uint8_t elastic_seq;                 /* `seq` of packet 0 in the elastic buffer, frame.c:593 */
    uint8_t align_offset;                /* output_align's offset, frame.c:595-600 */
    uint8_t skipped;                     /* stream_id >= MAX_STREAMS: packets skipped as frame.c:559-564 does */
} nrsc5hip_l2_pdu;
typedef struct nrsc5hip_l2_frame {
    uint32_t pci;                        /* protocol control information bits, frame.c:695-699 */
    uint32_t nbytes;                     /* PDU bytes of the frame */
    uint32_t n_pdu, status;
    uint32_t end_offset;                 /* where the walk stopped */
    uint32_t lost_sync;                  /* 1: frame_process calls input_set_sync_state(SYNC_STATE_NONE) on this frame (frame.c:537-538) */
    nrsc5hip_l2_pdu pdu[NRSC5HIP_L2_MAX_PDUS];
} nrsc5hip_l2_frame;
enum { NRSC5HIP_L2_FM_P1 = 0, NRSC5HIP_L2_FM_PX = 1, NRSC5HIP_L2_AM = 2 };
typedef struct nrsc5hip_l2_job {
    int32_t stream, slot;                /* as for nrsc5hip_p1_frame_bits / _px_frame_bits / _am_frame_bits */
    int32_t kind;                        /* NRSC5HIP_L2_FM_P1 | _FM_PX (which = channel) | _AM (which = 0..7 P1 of that block, 8 = P3) */
    int32_t which, nbits;                /* nbits: 146176 | 2304 / 4608 | 3750 / 24000 / 30000 */
} nrsc5hip_l2_job;
/* Index njobs decoded frames in one launch.  out[njobs]; pdu_bytes may be NULL, else job k's PDU bytes (RS-corrected
 * headers) go to pdu_bytes + k * stride (stride >= (nbits - pci bits) / 8).  Host buffers. */
int nrsc5hip_l2_index(nrsc5hip_engine *e, int njobs, const nrsc5hip_l2_job *jobs, nrsc5hip_l2_frame *out,
                      uint8_t *pdu_bytes, long long stride);
/* engine option l2_index: the index computed in the pipeline for the FM P1 frame in `slot` (the slot a REC_P1 record names) */
int nrsc5hip_l2_frame_get(nrsc5hip_engine *e, int stream, int slot, nrsc5hip_l2_frame *out);
/* ... and of every P1 slot of the listed streams: out[nstreams][p1_slots] */
int nrsc5hip_batch_fetch_l2(nrsc5hip_engine *e, int nstreams, const int *stream_ids, nrsc5hip_l2_frame *out);
/* The same option also indexes, in the pipeline, the P3 / P4 frames of the extended sidebands -- out[nstreams][px_slots = 8 *
 * p1_slots][2], entry [slot][0] = the P3 frame a REC_P3 record names in `sis`, [slot][1] its P4 frame (MP11) -- and, in engines
 * created with am_enable, the frames of every AM L1 frame slot: out[nstreams][p1_slots][9], entries 0..7 = the P1 frame delivered
 * at block 0..7 (REC_P1, p1_slot), 8 = the P3 frame (REC_P3).  Entries of slots no record names are stale or zero. */
int nrsc5hip_batch_fetch_l2_px(nrsc5hip_engine *e, int nstreams, const int *stream_ids, nrsc5hip_l2_frame *out);
int nrsc5hip_batch_fetch_l2_am(nrsc5hip_engine *e, int nstreams, const int *stream_ids, nrsc5hip_l2_frame *out);
/* stage-level twin: nframes logical frames given as frame_push takes them (one bit per byte, nbits each) */
int nrsc5hip_stage_l2_index(nrsc5hip_engine *e, const uint8_t *bits, int nbits, int nframes, nrsc5hip_l2_frame *out,
                            uint8_t *pdu_bytes, long long stride);

/* ---- batch HDC hand-off (host side) --------------------------------------------------------------------
 * The reference keeps a 22.9 MB nrsc5_t per session, 18.7 MB of it the elastic buffers of output_t (output.h:104-122).  A
 * batch of 2048 streams cannot; this consumer keeps ~40 KB per stream and program and reproduces, from the L2 index
 * (nrsc5hip_l2_frame + the PDU bytes nrsc5hip_l2_index returns), exactly the NRSC5_EVENT_HDC sequence of the reference:
 *   nrsc5hip_hdc_push_frame = frame_process's output_align / output_push calls for one frame   (frame.c:590-640, output.c:31-92)
 *   nrsc5hip_hdc_advance    = output_advance: call it once per block record (NRSC5HIP_REC_PROCESSED), BEFORE handing over
 *                             the frames that record announces, as acquire.c:108 does; every complete packet goes to `cb`
 *                             with the event's fields (program, data, count, flags; nrsc5.c:709-728).  Returns the count.
 *   nrsc5hip_hdc_adts       = dump_hdc's framing (main.c:182-212): 7-byte ADTS header + payload into out[count + 7]
 * No device work: plain host functions, usable from any thread (one consumer object per thread). */
typedef struct nrsc5hip_hdc nrsc5hip_hdc;
typedef void (*nrsc5hip_hdc_cb)(void *opaque, int stream, unsigned program, const uint8_t *data, unsigned count, unsigned flags);
int nrsc5hip_hdc_create(int nstreams, nrsc5hip_hdc **out);
void nrsc5hip_hdc_destroy(nrsc5hip_hdc *h);
int nrsc5hip_hdc_reset(nrsc5hip_hdc *h, int stream);                 /* output_reset (output.c:204-218) */
/* lc: logical channel of the frame (0 = P1, 1 = P3, 2 = P4), which selects the fixed-data (CCC) state it advances */
int nrsc5hip_hdc_push_frame(nrsc5hip_hdc *h, int stream, int lc, const nrsc5hip_l2_frame *ix, const uint8_t *pdu_bytes);
/* frames with fixed-data sub-channels: process_fixed_data's audio_end for this frame (frame.c:458-514: sync-byte tracking,
 * CCC HDLC frames, sub-channel lengths) -- advances the stream's CCC state; and the cut of an index that was built with
 * audio_end = nbytes - 1 back to it (the loop / locator conditions of frame.c:525,547-556).  nrsc5hip_hdc_push_frame calls both
 * for such frames; they are exported for consumers that keep their own elastic buffers.  apply returns the PDUs kept, or -1
 * when a header expansion ran into the fixed-data region (the reference's parse is cut there: walk that frame on the host). */
unsigned nrsc5hip_hdc_fixed_audio_end(nrsc5hip_hdc *h, int stream, int lc, const uint8_t *pdu_bytes, unsigned nbytes);
int nrsc5hip_l2_apply_audio_end(nrsc5hip_l2_frame *ix, unsigned audio_end);
/* sync.c:405-409: frame_reset when a stream enters FINE (NRSC5HIP_REC_TO_FINE) clears the CCC state; the elastic buffers stay */
int nrsc5hip_hdc_frame_reset(nrsc5hip_hdc *h, int stream);
int nrsc5hip_hdc_advance(nrsc5hip_hdc *h, int stream, int mode /* NRSC5HIP_MODE_FM | _AM */, nrsc5hip_hdc_cb cb, void *opaque);
size_t nrsc5hip_hdc_adts(const uint8_t *data, unsigned count, uint8_t *out);
size_t nrsc5hip_hdc_host_bytes(const nrsc5hip_hdc *h);                /* host memory held by the consumer */

/* ---- stage-level entry points (host buffers): parity tests of single kernels against the oracle ---- */
int nrsc5hip_stage_halfband_fm_cu8(nrsc5hip_engine *e, const uint8_t *iq, uint32_t nbytes, int16_t *out /* [nbytes/4][2] */);
int nrsc5hip_stage_fft2048(nrsc5hip_engine *e, const float *in /* [n][2048][2] */, float *out, int n);
int nrsc5hip_stage_viterbi_k7(nrsc5hip_engine *e, const int8_t *soft /* [nframes][3*len] */, int len, int nframes,
                              uint8_t *bits /* [nframes][len] */);
/* K=9 codes of the AM path (nrsc5_conv_decode_e1 / _e2_e3, conv_dec.c:469-478): gens = {0561,0657,0711} or {0561,0753,0711} */
/* frame_process's first-header check (frame.c:516-540: RS(255,247) over the first 96 PDU bytes unless the PCI says "no audio") as the
 * decode kernels run it on a finished P1 frame: `bits` = nframes descrambled frames of nbits (146176 FM / 3750 AM) bits, one per byte;
 * `threads` = workgroup size (64 ... 1024); ok[f] = 1 when the reference would keep the receiver synchronised */
int nrsc5hip_stage_first_header(nrsc5hip_engine *e, const uint8_t *bits, int nbits, int nframes, int threads, int *ok);
int nrsc5hip_stage_viterbi_k9(nrsc5hip_engine *e, const int8_t *soft /* [nframes][3*len] */, int len, int nframes,
                              const unsigned gens[3], uint8_t *bits /* [nframes][len] */);
/* device check of the DPP / v_permlane / v_writelane / v_dot4 helpers against generic shuffles: *failures == 0 */
int nrsc5hip_stage_selftest(nrsc5hip_engine *e, int *failures);
/* one frame, also returning the len+64 survivor-decision words of the forward pass */
int nrsc5hip_stage_viterbi_k7_debug(nrsc5hip_engine *e, const int8_t *soft, int len, uint8_t *bits, unsigned long long *dec_out);
/* micro-benchmark of the Viterbi kernel on random frames: phases bit0 = forward, bit1 = traceback */
int nrsc5hip_stage_viterbi_bench(nrsc5hip_engine *e, int len, int nframes, int phases, int reps, float *ms_per_launch);
int nrsc5hip_stage_viterbi_k9_bench(nrsc5hip_engine *e, int len, int nframes, int phases, int reps, float *ms_per_launch);
/* Tuning knobs and test hooks (the library reads nothing from the environment).  Call on an idle engine. */
enum {
    NRSC5HIP_TUNE_DECODE_STREAMS = 0,    /* FM window pipeline: HIP streams that decode windows concurrently (1..5; default 1 since round 5, 3 before) */
    NRSC5HIP_TUNE_AM_DECODE_STREAMS,     /* same for the AM window pipeline (default 3) */
    NRSC5HIP_TUNE_VERDICT_LAG,           /* TEST HOOK: the replay takes first-header verdicts this many windows late (0..8): deep speculation */
    NRSC5HIP_TUNE_SYNC_PHASES,           /* 1: k_sync accumulates shader cycles per phase for stream 0 (nrsc5hip_debug_sync_phases) */
    NRSC5HIP_TUNE_FWD_SEGMENTS           /* waves per frame of the K=7 forward trellis pass (1..64; 0 = chosen from the size of the stream set).  Any value
                                            gives the sequential decoder's decisions bit for bit: segments start speculatively and are verified /
                                            repaired (viterbi_v3.h).  Also used by nrsc5hip_stage_viterbi_k7 / _bench. */
    , NRSC5HIP_TUNE_FWD_WARM               /* TEST HOOK: 0 = the segments start cold (no speculative warm-up), so that the speculation fails wherever the
                                            input carries information and every segment takes the repair path; 1 = normal */
    , NRSC5HIP_TUNE_AM_SEGMENTS            /* waves per P3 frame of the K=9 decode in the AM window pipeline (1..8, default 8); forward pass AND traceback
                                            run in segment waves, both verified / repaired: any value gives the sequential decoder's bits.  Also used by
                                            nrsc5hip_stage_viterbi_k9 / _bench (1 = the single-wave form). */
    , NRSC5HIP_TUNE_DECODE_CUS             /* decode streams confined to value / 32 of every XCD's CUs (8, 16, 24; 32 = all, the default) */
    , NRSC5HIP_TUNE_DECODE_PRIORITY        /* 1: decode streams at the lowest queue priority (default 0: all queues equal) */
    , NRSC5HIP_TUNE_AM_WARM                /* TEST HOOK: 0 = no forward warm-up and no traceback run-in (every boundary takes the repair path); 1 = normal */
    , NRSC5HIP_TUNE_MIXFFT_SYMS            /* OFDM symbols per k_mixfft workgroup: 1 (default), 2, 4, 8 -- any value gives identical bins; 16 = two symbols side by side in a
                                             256-lane workgroup (identical bins); 32 = the 256-lane x 8-point kernel k_mixfft8 (bins within float tolerance); 100 .. 140 =
                                             DIAGNOSTIC: the default kernel with (value - 100) KiB of unused dynamic LDS per workgroup (fewer workgroups per CU; the engine
                                             clamps the padding to what the device's per-workgroup LDS limit leaves); anything else = 1 */
    , NRSC5HIP_TUNE_DEFER_WAIT             /* fast streaming seam: 1 (default) = a block step whose FIFO consumption the host can compute in advance stays in
                                             flight when the push returns; 0 = every step is waited for at once (round 3's behaviour) */
    , NRSC5HIP_TUNE_TRACEBACK_WALK         /* > 0: single-path traceback of the K=7 frames (one speculative walk per chunk, verified, re-walked where wrong): 1 (default) = one workgroup per
                                             (frame, part); N > 1 = a persistent grid of N one-wave workgroups (opt-in: measured slower, profiles/r04_traceback_walk.txt);
                                             0: round 3's block-parallel traceback (all 64 candidates per chunk).  Identical output */
    , NRSC5HIP_TUNE_SYNC_LANES             /* work-items per stream of the sync kernel: 256, 768, 0 = chosen from the size of the stream set (default) */
    , NRSC5HIP_TUNE_DIRECT_DECIMATE        /* fast streaming seam, FM cu8: 1 (default) = the decimator reads the pinned staging buffer across PCIe itself (one
                                             launch per chunk); 0 = hipMemcpyAsync into a device buffer, decimator, commit kernel (round 3's chain) */
    , NRSC5HIP_TUNE_EARLY_FLUSH_KB         /* fast streaming seam with the direct decimator: staged KiB at which a chunk is submitted (on the engine's ingest stream,
                                             beside the running block step) before its block is complete; 0 = only at the block's end.  Default 128 */
    , NRSC5HIP_TUNE_SEAM_PREPARE           /* fast streaming seam, FINE stream: 1 (default) = the block's bookkeeping is computed by the symbol kernel for itself and
                                             committed by the sync kernel; 0 = k_prepare as a launch of its own in front of them */
    , NRSC5HIP_TUNE_NCO_EXACT              /* which blocks of a freshly reset FM stream advance the NCO by the reference's own float recurrence (acquire.c:237-252: 69 120
                                             dependent complex multiplications per block, ~0.4 ms of one lane per stream whatever the number of streams) instead of the
                                             closed-form phasor: 0 = none, 1 (default since round 6) = the first block after a reset (the block the CFO search runs on), 2 = every block until the
                                             stream is FINE, 3 = every block (diagnostic).  The float oscillator state is the reference's bit for bit for as long as every
                                             block since the reset ran in this mode; the first closed-form block ends that until the next reset */
    , NRSC5HIP_TUNE_FLOW_MIN               /* dataflow bursts (k_flow): a zero-copy batch with the window pipeline runs the block steps of a burst in which every stream is
                                             FINE (MP1 routing, no acquisition kernels needed) as ONE launch whose workgroups -- pairs of symbol transforms and per-stream block
                                             steps -- hand over to each other, for stream sets of at least this many streams; 0 = never (two launches per step) */
    , NRSC5HIP_TUNE_LOOP_EXACT             /* FM Costas loops / CFO search (sync.c:90-136, 292-337) by the reference's own operations -- glibc's sincosf and atan2f restated
                                             bit for bit (fastmath.h), its float complex products, adjust_ref / reset_ref in place -- instead of the fast forms (v_sin / v_cos,
                                             a 4-term arc tangent, ~5e-7): 0 = never, 1 (default) = in every block that starts un-synchronised (the tracking pass over garbage and the
                                             CFO search, where a last-bit difference can be amplified into a different loop state), 2 = in every block */
    , NRSC5HIP_TUNE_HOST_CAPTURE           /* fast streaming seam, FM cu8 (round 6): 1 (default) = the pushes of a session stay, as they arrive, in a pinned device-mapped capture that
                                             the stream reads in place across PCIe (the zero-copy batch's kernels: half-band fused into the symbol transform) -- a push is one host
                                             memcpy, no decimator launch; 0 = pinned staging + decimator kernel into the device FIFO (rounds 3 - 5).  Identical records either way.
                                             >= 512: on, with that many KiB of capture buffer instead of 16 MiB (the live tail moves to the front when it is full).  One stream per
                                             engine reads such a capture (the first to push cu8 after its reset); cs16 / AM input and the batch entry points use the FIFO */
    , NRSC5HIP_TUNE_FOLD_REPORT            /* fast streaming seam: 1 (default) = a block step with nothing to launch behind the sync kernel (no P1 frame to decode, no extended sidebands: 15 of 16
                                             MP1 blocks) has the sync kernel post the step's report into pinned host memory itself; 0 = the report kernel as a launch of its own.  Identical records */
};
/* process-wide wall-clock totals of the streaming seam with p1_async = 0 (what the drop-in uses): [0] s copying pushes into pinned
 * staging, [1] s enqueueing H2D + decimator, [2] s enqueueing block steps, [3] s waiting for the device (one sync per block),
 * [4] pushes, [5] submissions, [6] block steps, [7] s fetching P1 frames */
void nrsc5hip_debug_seam_totals(double out[8], int reset);   /* totals of the CALLING THREAD's sessions */
/* ... [0] block steps left in flight (deferred wait), [1] read positions the host predicted wrongly (expected: 0), [2] steps submitted
 * without the P1 decode launches (no frame could complete), [3] P1 decodes launched after the fact (expected: 0) */
void nrsc5hip_debug_seam_counts(double out[6], int reset);   /* ... [4] block steps submitted ahead of the previous block's delivery, [5] pushes copied into a host-resident capture (NRSC5HIP_TUNE_HOST_CAPTURE) */
/* test / bench hygiene: overwrite every result buffer a pass writes (frame rings on the device and their pinned host mirror, record
 * rings) with a pattern no decode produces -- a check after the next pass can then only pass on bits written by that pass */
int nrsc5hip_debug_poison_results(nrsc5hip_engine *e);
/* segmented forward pass: segment boundaries checked / segments that had to be re-run since the engine was created */
int nrsc5hip_debug_fwd_stats(nrsc5hip_engine *e, int stats[2]);
/* dataflow bursts (NRSC5HIP_TUNE_FLOW_MIN): [0] bursts, [1] block steps issued as k_flow launches since the engine was created */
int nrsc5hip_debug_flow_stats(nrsc5hip_engine *e, long long stats[2]);
/* host-resident capture of the fast seam (NRSC5HIP_TUNE_HOST_CAPTURE): [0] sessions that bound the pinned capture, [1] captures turned back into the FIFO (cs16 push,
 * batch entry point, debug fetch), [2] times the live tail moved to the front of a full buffer, [3] the stream bound now (-1: none); and of NRSC5HIP_TUNE_FOLD_REPORT:
 * [4] block steps whose report the sync kernel posted itself */
int nrsc5hip_debug_host_capture_stats(nrsc5hip_engine *e, long long stats[5]);
/* K=9 decode in segment waves: [0] forward boundaries checked, [1] segments re-run, [2] traceback boundaries checked, [3] segments re-walked */
/* single-path traceback: [0] chunk boundaries checked, [1] chunks re-walked since the engine was created */
int nrsc5hip_debug_tb_stats(nrsc5hip_engine *e, int stats[2]);
int nrsc5hip_debug_k9_stats(nrsc5hip_engine *e, int stats[4]);
int nrsc5hip_debug_tune(nrsc5hip_engine *e, int knob, int value);
/* accumulated shader cycles per phase of the sync kernel for stream 0 (after nrsc5hip_debug_tune(e, NRSC5HIP_TUNE_SYNC_PHASES, 1)) in [0..7];
 * [8..15]: phases of the symbol kernel, filled only by a diagnostic build of the library (-DNRSC5HIP_MIXFFT_PHASES), else zero */
int nrsc5hip_debug_sync_phases(nrsc5hip_engine *e, long long *cycles16);
/* debugging aid: soft-bit matrix (16 x 23040 int8) and live FFT bins of a stream's latest block */
int nrsc5hip_debug_fetch(nrsc5hip_engine *e, int stream, int8_t *pm /* [368640] or NULL */, float *bins /* [32][534][2] or NULL */);

/* debugging aid: Costas loop state (sync_t.costas_freq / costas_phase) of the 534 live bins (2 x 267: lower sideband bins 478..744,
 * upper 1304..1570) of a stream */
int nrsc5hip_debug_fetch_costas(nrsc5hip_engine *e, int stream, float *freq /* [534] */, float *phase /* [534] */);

/* debugging aid: PX1 / PX2 soft bits of the stream's current block pair, [2 channels][2 blocks][4608] int8 */
int nrsc5hip_debug_fetch_px(nrsc5hip_engine *e, int stream, int8_t *pair /* [18432] */);

/* fresh-session state for every stream (input_reset on all of them) */
int nrsc5hip_reset_all(nrsc5hip_engine *e);

/* Per-kernel-class device timing, measured with HIP events recorded on the stream each kernel is
 * launched on.  enable: 1 start (and zero the accumulators), 0x100 | class start but time that class only (the events around
 * every launch of the block-step chain cost ~9 % of a batch pass), 0 stop (and zero), -1 just read.
 * total_ms / launches: arrays of NRSC5HIP_PROF_CLASSES entries (may be NULL). */
enum {
    NRSC5HIP_PROF_DECIMATE = 0, NRSC5HIP_PROF_ACQUIRE, NRSC5HIP_PROF_PREPARE, NRSC5HIP_PROF_MIXFFT,
    NRSC5HIP_PROF_SYNC, NRSC5HIP_PROF_P1_DEINT, NRSC5HIP_PROF_P1_VITERBI, NRSC5HIP_PROF_PIDS, NRSC5HIP_PROF_AM /* block steps */, NRSC5HIP_PROF_AM_DECODE /* window pipeline: the deferred decodes */,
    NRSC5HIP_PROF_P1_TRACEBACK /* P1_DEINT / P1_VITERBI (the forward trellis pass) / P1_TRACEBACK: the three stages of a P1 decode */,
    NRSC5HIP_PROF_FLOW /* dataflow bursts: up to 16 block steps of a stream set (symbol transforms + block steps) as one launch of k_flow */,
    NRSC5HIP_PROF_CLASSES
};
int nrsc5hip_profile(nrsc5hip_engine *e, int enable, double *total_ms, long long *launches);

/* test helpers: first n samples of a stream's Q15 FIFO slab; raw device allocation for batch tests */
int nrsc5hip_debug_fetch_q15(nrsc5hip_engine *e, int stream, long long n, int16_t *out /* [n][2] */);
void *nrsc5hip_debug_alloc_copy(const void *host, size_t nbytes);
void nrsc5hip_debug_free(void *dev);

#ifdef __cplusplus
}
#endif
#endif /* NRSC5HIP_H_ */
