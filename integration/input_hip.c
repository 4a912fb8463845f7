/* Drop-in replacement for the reference's src/input.c: same functions, same `input_t` (src/input.h:20-35),
 * but the whole L3a-L3d hot path (decimate, acquire, sync, decode, Viterbi) runs in libnrsc5hip on an
 * MI355X.  A maintainer of theori-io/nrsc5 compiles THIS file instead of
 *     src/input.c src/acquire.c src/sync.c src/decode.c src/conv_dec.c src/firdecim_q15.c
 * and links libnrsc5hip.so; include/nrsc5.h, src/nrsc5.c, frame.c, pids.c, output.c ... stay untouched.
 *
 * Mapping (reference line -> here):
 *   input_push_cu8  input.c:96-117  -> nrsc5hip_push_cu8 + deliver()
 *   input_push_cs16 input.c:119-124 -> nrsc5hip_push_cs16 + deliver()
 *   input_reset     input.c:126-138 -> nrsc5hip_stream_reset + pids_init/frame_reset
 *   input_set_sync_state input.c:172-188: called by frame_process (frame.c:539) with SYNC_STATE_NONE
 *                                   -> nrsc5hip_force_resync + nrsc5_report_lost_sync
 *   up-calls, in the reference's order inside acquire_process: output_advance (acquire.c:108),
 *   nrsc5_report_sync (input.c:185), decode_reset/frame_reset (sync.c:405-409), nrsc5_report_mer
 *   (sync.c:497), pids_frame_push (decode.c:471), nrsc5_report_ber (decode.c:458), frame_push (decode.c:460).
 *   MP2/MP3/MP11: frame_push of the P3 / P4 frames of the extended sidebands (decode.c:409,432).
 *   AM (NRSC5_MODE_AM): input_set_mode -> nrsc5hip_stream_set_mode; per FINE block pids_frame_push (decode.c:504),
 *   frame_push of the 3750-bit P1 frame (decode.c:519), after block 7 frame_push of the P3 frame and
 *   nrsc5_report_ber (decode.c:528-543).
 */
#include "config.h"

#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "defines.h"
#include "input.h"
#include "private.h"

#include "nrsc5hip.h"

/* one engine per session (any number of sessions per process and GPU; each has its own HIP streams and buffers); stream 0.
 * acquire_t's otherwise unused FFTW slots carry the handle and the failure mark so that input_t keeps the reference's exact
 * layout. */
#define ENGINE(st) ((nrsc5hip_engine *)(st)->acq.fftin)
#define FAILED(st) ((st)->acq.fftout != NULL)

/* The pipe API cannot report errors (nrsc5.c:624 always returns 0) and a library must not take the host application down:
 * a HIP / engine failure is reported once on stderr and the session goes inert -- it accepts samples and delivers nothing
 * more, like a receiver that lost its signal -- until it is closed. */
static void fail(input_t *st, const char *what)
{
    if (!FAILED(st))
        fprintf(stderr, "nrsc5hip: %s failed: %s -- this session delivers no further events\n", what, nrsc5hip_last_error());
    st->acq.fftout = (void *)st;
}

/* wait = 1: every block the engine has been given is delivered (nrsc5hip_drain waits for a block step that is still running);
 * wait = 0: only what has been reported so far (nrsc5hip_drain_ready) -- the device keeps working on block n while the caller
 * fetches and pushes the samples of block n + 1. */
static void deliver(input_t *st, int wait)
{
    /* frame bits as frame_push takes them: the session's own descrambler buffer (decode.h, P1_FRAME_LEN_FM bytes, unused
     * here because descrambling happens on the device) -- per session, so concurrent sessions of one process do not share it */
    uint8_t *bits = st->decode.scrambler_p1;
    nrsc5hip_record rec[64];
    int n = 0;
    const int am = st->radio->mode == NRSC5_MODE_AM;

    do
    {
        if ((wait ? nrsc5hip_drain(ENGINE(st), 0, rec, 64, &n) : nrsc5hip_drain_ready(ENGINE(st), 0, rec, 64, &n)) != 0) { fail(st, "drain"); return; }
        for (int k = 0; k < n; k++)
        {
            const nrsc5hip_record *r = &rec[k];
            if (r->flags & NRSC5HIP_REC_PROCESSED)
                output_advance(st->output);
            if (r->flags & NRSC5HIP_REC_TO_COARSE)
                st->sync_state = SYNC_STATE_COARSE;
            if (r->flags & NRSC5HIP_REC_TO_FINE)
            {
                st->sync.psmi = r->psmi;
                st->sync_state = SYNC_STATE_FINE;
                if (am)
                    nrsc5_report_sync(st->radio, r->freq_offset, r->psmi, r->sis & 1, (r->sis >> 1) & 1, (r->sis >> 2) & 1, (r->sis >> 3) & 1);
                else
                    nrsc5_report_sync(st->radio, r->freq_offset, r->psmi, -1, -1, -1, -1);
                pids_init(&st->decode.pids, st);     /* decode_reset (decode.c:563-572) */
                frame_reset(&st->frame);
            }
            if (r->flags & NRSC5HIP_REC_MER)
                nrsc5_report_mer(st->radio, r->mer_lb, r->mer_ub);
            if (r->flags & NRSC5HIP_REC_PIDS)      /* (NRSC5HIP_REC_PIDS_CRC tells whether its CRC-12 passes; pids_frame_push re-checks) */
            {
                uint8_t pids[PIDS_FRAME_LEN];
                nrsc5hip_unpack_bits(r->pids, PIDS_FRAME_LEN, pids);
                pids_frame_push(&st->decode.pids, pids);
            }
            if (am)
            {
                if (r->flags & NRSC5HIP_REC_P1)
                {
                    if (nrsc5hip_am_frame_bits(ENGINE(st), 0, r->p1_slot, r->bc_decoded, P1_FRAME_LEN_AM, bits) != 0) { fail(st, "am_frame_bits"); return; }
                    frame_push(&st->frame, bits, P1_FRAME_LEN_AM, P1_LOGICAL_CHANNEL);   /* may call input_set_sync_state(NONE) */
                }
                if (r->flags & NRSC5HIP_REC_P3)
                {
                    const int n3 = (r->psmi == SERVICE_MODE_MA3) ? P3_FRAME_LEN_MA3 : P3_FRAME_LEN_MA1;
                    if (nrsc5hip_am_frame_bits(ENGINE(st), 0, r->p1_slot, 8, n3, bits) != 0) { fail(st, "am_frame_bits"); return; }
                    frame_push(&st->frame, bits, n3, P3_LOGICAL_CHANNEL);
                }
                if ((r->flags & NRSC5HIP_REC_P1) && r->bc_decoded == 7)
                    nrsc5_report_ber(st->radio, r->ber);
            }
            else if (r->flags & NRSC5HIP_REC_P1)
            {
                nrsc5_report_ber(st->radio, r->ber);
                if (nrsc5hip_p1_frame_bits(ENGINE(st), 0, r->p1_slot, bits) != 0) { fail(st, "p1_frame_bits"); return; }
                frame_push(&st->frame, bits, P1_FRAME_LEN_FM, P1_LOGICAL_CHANNEL);   /* may call input_set_sync_state(NONE) */
            }
            if (!am && (r->flags & (NRSC5HIP_REC_P3 | NRSC5HIP_REC_P4)))
            {
                /* extended sidebands: decode_push_px1 / px2 (decode.c:393-437) */
                const int nbits = (r->psmi == 2) ? P3_FRAME_LEN_MP2 : P3_FRAME_LEN_MP3_MP11;
                for (int ch = 0; ch < 2; ch++)
                {
                    if (!(r->flags & (ch ? NRSC5HIP_REC_P4 : NRSC5HIP_REC_P3))) continue;
                    if (nrsc5hip_px_frame_bits(ENGINE(st), 0, (int)r->sis, ch, nbits, bits) != 0) { fail(st, "px_frame_bits"); return; }
                    frame_push(&st->frame, bits, nbits, ch ? P4_LOGICAL_CHANNEL : P3_LOGICAL_CHANNEL);
                }
            }
        }
    } while (n == 64);
}

/* Feed the engine up to the end of the next block, deliver the events of the block before it (frame_push may answer with
 * input_set_sync_state(NONE)), step, go on: the L2 feedback of a frame reaches the engine before the next block is PROCESSED,
 * exactly as in the reference (FINE -> NONE only comes from frame_process), for any push size -- while the samples of
 * that next block are already being copied and decimated, and while the device is still busy with the block the previous call
 * submitted (deferred wait, include/nrsc5hip.h).  Events of a block therefore reach the application during the first call after
 * the device has finished it -- at the latest in the call that completes the following block, in a zero-length
 * nrsc5_pipe_samples_* call (a flush), or at input_free / input_reset;
 * that is the OVERLAPPED mode, opt-in since round 6 (NRSC5HIP_OVERLAP_DELIVERY=1).  The DEFAULT is the reference's own contract (src/input.c:41-50, 96-117: every
 * callback fires inside the nrsc5_pipe_samples_* call that completes its block): the call that completes a block waits for the device and delivers that block's
 * events before it returns.  NRSC5HIP_SYNC_DELIVERY (rounds 4 - 5: 1 = strict, 0 = overlapped) is still honoured when set. */
static int sync_delivery(void)
{
    /* read per call (a getenv is ~50 ns against a block's ~80 us): a host can switch between two sessions, and bench.py times both modes in one process */
    const char *e = getenv("NRSC5HIP_SYNC_DELIVERY");
    if (e && *e) return atoi(e) > 0 ? 1 : 0;
    e = getenv("NRSC5HIP_OVERLAP_DELIVERY");
    return (e && atoi(e) > 0) ? 0 : 1;
}

static void push_pieces(input_t *st, const uint8_t *buf, uint32_t nbytes, int cu8)
{
    uint32_t consumed = 0;
    const int strict = sync_delivery();
    if (nbytes == 0 && !FAILED(st)) deliver(st, 1);             /* nrsc5_pipe_samples_*(radio, buf, 0): flush -- every pending event, now */
    while (consumed < nbytes && !FAILED(st))
    {
        long long room = nrsc5hip_bytes_to_next_block(ENGINE(st), 0, cu8);
        const int known = room >= 0;
        if (!known) room = 4 * 4320;                            /* engine cannot tell: two OFDM symbols never complete two blocks */
        const uint32_t left = nbytes - consumed;
        const uint32_t piece = left < (uint32_t)room ? left : (uint32_t)room;
        const int completes = known && (long long)piece >= room;
        int rc = cu8 ? nrsc5hip_push_cu8(ENGINE(st), 0, buf + consumed, piece) : nrsc5hip_push_cs16(ENGINE(st), 0, (const int16_t *)(buf + consumed), piece / 2);
        if (rc != 0) { fail(st, cu8 ? "push_cu8" : "push_cs16"); return; }
        consumed += piece;
        if (completes)
        {
            /* 15 of 16 blocks cannot end a P1 frame: nothing their delivery tells frame.c can send the receiver back to NONE, so the
             * step of THIS block is queued behind them before they are even looked at (the engine decides; include/nrsc5hip.h) */
            int ahead = 0;
            if (!strict && nrsc5hip_stream_step_ahead(ENGINE(st), 0, &ahead) != 0) { fail(st, "stream_step_ahead"); return; }
            deliver(st, 1);                                     /* the block before: its frames reach frame.c now ... */
            if (FAILED(st)) return;
            if (!ahead && nrsc5hip_stream_step(ENGINE(st), 0) != 0) { fail(st, "stream_step"); return; }   /* ... and only then is this one processed */
            deliver(st, strict);
        }
        else
            deliver(st, known ? 0 : 1);
    }
}

void input_push_cu8(input_t *st, const uint8_t *buf, const uint32_t len)
{
    nrsc5_report_iq(st->radio, buf, len);
    assert(len % 4 == 0);
    push_pieces(st, buf, len, 1);
}

void input_push_cs16(input_t *st, const int16_t *buf, const uint32_t len)
{
    assert(len % 2 == 0);
    push_pieces(st, (const uint8_t *)buf, len * 2, 0);
}

void input_set_sync_state(input_t *st, unsigned int new_state)
{
    if (st->sync_state == new_state)
        return;
    if (st->sync_state == SYNC_STATE_FINE)
        nrsc5_report_lost_sync(st->radio);
    if (new_state == SYNC_STATE_NONE && ENGINE(st) && !FAILED(st))
        if (nrsc5hip_force_resync(ENGINE(st), 0) != 0) fail(st, "force_resync");
    st->sync_state = new_state;
}

void input_reset(input_t *st)
{
    if (ENGINE(st) && !FAILED(st)) deliver(st, 1);              /* a block still in flight belongs to the session that ends here */
    input_set_sync_state(st, SYNC_STATE_NONE);
    if (ENGINE(st) && !FAILED(st) && nrsc5hip_stream_reset(ENGINE(st), 0) != 0) fail(st, "stream_reset");
    pids_init(&st->decode.pids, st);
    frame_reset(&st->frame);
    st->sync.psmi = 1;
}

void input_init(input_t *st, nrsc5_t *radio, output_t *output)
{
    const char *dev = getenv("NRSC5HIP_DEVICE");
    nrsc5hip_config cfg = { dev ? atoi(dev) : 0, 1, 1 << 20, 256, 4, 0 /* in-order P1: reference event timing */, 0 /* L2 feedback comes from frame.c through input_set_sync_state */, 1 /* AM too */ };
    nrsc5hip_engine *e = NULL;

    memset(&st->acq, 0, sizeof(st->acq));
    st->radio = radio;
    st->output = output;
    st->sync_state = SYNC_STATE_NONE;
    if (nrsc5hip_engine_create(&cfg, &e) != 0) { e = NULL; fail(st, "engine_create"); }
    else if (nrsc5hip_stream_set_manual_step(e, 0, 1) != 0) fail(st, "stream_set_manual_step");
    /* NRSC5HIP_NCO_EXACT=1 | 2 | 3 in the environment: the first block after a reset (the block the CFO search runs on) / every block until the first lock /
     * every block with the reference's own oscillator recurrence (k_nco_exact: ~1.5 ms per such block; DESIGN.md (c) limit 2).  Default: the closed form with
     * the oscillator's amplitude ramp, as in the batch API -- measured on the MI355X the exact form changes no event and costs a 20-s capture 5 % of its speed */
    else { const char *x = getenv("NRSC5HIP_NCO_EXACT"); if (x && atoi(x) > 0 && nrsc5hip_debug_tune(e, NRSC5HIP_TUNE_NCO_EXACT, atoi(x)) != 0) fail(st, "debug_tune"); }
    /* NRSC5HIP_HOST_CAPTURE=0: the FIFO seam of rounds 3 - 5 (pinned staging + decimator kernel) instead of the pinned capture read in place (A/B; identical events) */
    /* NRSC5HIP_FOLD_REPORT=0: A/B switch of the seam's launch plan (include/nrsc5hip.h, NRSC5HIP_TUNE_*); identical events */
    if (e && !FAILED(st)) { const char *x = getenv("NRSC5HIP_FOLD_REPORT"); if (x && *x && nrsc5hip_debug_tune(e, NRSC5HIP_TUNE_FOLD_REPORT, atoi(x)) != 0) fail(st, "debug_tune"); }
    if (e && !FAILED(st)) { const char *x = getenv("NRSC5HIP_HOST_CAPTURE"); if (x && *x && nrsc5hip_debug_tune(e, NRSC5HIP_TUNE_HOST_CAPTURE, atoi(x)) != 0) fail(st, "debug_tune"); }
    st->acq.fftin = (void *)e;
    st->decode.input = st;
    frame_init(&st->frame, st);
    input_reset(st);
}

void input_set_mode(input_t *st)
{
    if (ENGINE(st) && !FAILED(st)) deliver(st, 1);              /* before the stream is reset: the events of a block still in flight belong to the session that ends here */
    if (ENGINE(st) && !FAILED(st) && nrsc5hip_stream_set_mode(ENGINE(st), 0, st->radio->mode == NRSC5_MODE_AM ? NRSC5HIP_MODE_AM : NRSC5HIP_MODE_FM) != 0)
        fail(st, "stream_set_mode");
    input_reset(st);
}

void input_free(input_t *st)
{
    if (ENGINE(st) && !FAILED(st)) deliver(st, 1);              /* the last block's events */
    frame_free(&st->frame);
    nrsc5hip_engine_destroy(ENGINE(st));
    st->acq.fftin = NULL;
}
