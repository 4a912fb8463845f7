/* TEST INFRASTRUCTURE. Drives the UNMODIFIED reference (compiled from
 * /root/reference/src by oracle/Makefile into oracle/_ref/) through its public
 * pipe API (nrsc5.h:712,754,847,859,871) and taps stage boundaries with
 * -Wl,--wrap so the restatement (oracle/nrsc5_oracle.c) and the HIP path can be
 * pinned against the real thing. Nothing here ships. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <nrsc5.h>
#include "private.h"   /* struct nrsc5_t, input_t, acquire_t, sync_t (read-only peeking) */

enum {
    REFH_TAP_Q15   = 1 << 0,   /* cint16 stream entering acquire_push (K1 out) */
    REFH_TAP_FFT   = 1 << 1,   /* fftshifted bins handed to sync_push */
    REFH_TAP_SOFT  = 1 << 2,   /* decode_push_pm soft bits */
    REFH_TAP_VIT   = 1 << 3,   /* conv decoder in/out */
    REFH_TAP_HDC   = 1 << 4,   /* HDC packet payloads in the log */
    REFH_TAP_L2    = 1 << 5,   /* output_align / output_push calls of frame_process (frame.c:600,631) */
};

/* ordered log record kinds */
enum {
    REC_BLOCK = 1, REC_STATE, REC_SOFT, REC_PIDS, REC_FRAME, REC_SYNC, REC_LOST_SYNC,
    REC_MER, REC_BER, REC_HDC, REC_VIT, REC_AMSYM, REC_PXSOFT, REC_STATION, REC_L2PKT, REC_L2ALIGN, REC_L2AAS, REC_L2SVC
};

typedef struct { uint8_t *p; size_t len, cap; } gbuf;
static gbuf g_log, g_q15, g_fft;
static unsigned g_taps;
static nrsc5_t *g_radio;
static int g_last_adj;
static unsigned g_fft_limit_blocks = 4, g_fft_syms;

static void gb_put(gbuf *b, const void *src, size_t n)
{
    if (b->len + n > b->cap) {
        size_t nc = b->cap ? b->cap * 2 : (1 << 20);
        while (nc < b->len + n) nc *= 2;
        b->p = realloc(b->p, nc); b->cap = nc;
    }
    memcpy(b->p + b->len, src, n); b->len += n;
}
static void log_rec(uint32_t kind, const void *payload, uint32_t n)
{
    uint32_t hdr[2] = { kind, n };
    gb_put(&g_log, hdr, sizeof(hdr));
    if (n) gb_put(&g_log, payload, n);
    uint32_t pad = (4 - (n & 3)) & 3, z = 0;
    if (pad) gb_put(&g_log, &z, pad);
}

/* ---- wrapped internal seams -------------------------------------------------- */
unsigned int __real_acquire_push(acquire_t *st, const cint16_t *buf, unsigned int length);
unsigned int __wrap_acquire_push(acquire_t *st, const cint16_t *buf, unsigned int length)
{
    unsigned int n = __real_acquire_push(st, buf, length);
    if (g_taps & REFH_TAP_Q15) gb_put(&g_q15, buf, sizeof(cint16_t) * n);
    return n;
}

void __real_sync_adjust(sync_t *st, int sample_adj);
void __wrap_sync_adjust(sync_t *st, int sample_adj) { g_last_adj = sample_adj; __real_sync_adjust(st, sample_adj); }

void __real_acquire_process(acquire_t *st);
void __wrap_acquire_process(acquire_t *st)
{
    int will = (st->idx == (unsigned int)st->fftcp * (ACQUIRE_SYMBOLS + 1));
    int32_t before = st->input->sync_state;
    __real_acquire_process(st);
    if (will) {
        struct { int32_t state_before, state_after, samperr, cfo, keep, bc, psmi, cfo_wait, next_samperr;
                 float prev_angle, phase_re, phase_im, next_angle; } r;
        r.state_before = before; r.state_after = st->input->sync_state;
        r.samperr = st->fftcp / 2 - g_last_adj; r.cfo = st->cfo; r.keep = (int32_t)st->idx;
        r.bc = st->input->sync.bc; r.psmi = st->input->sync.psmi; r.cfo_wait = st->input->sync.cfo_wait;
        r.next_samperr = st->input->sync.samperr; r.prev_angle = st->prev_angle;
        r.phase_re = crealf(st->phase); r.phase_im = cimagf(st->phase); r.next_angle = st->input->sync.angle;
        log_rec(REC_BLOCK, &r, sizeof(r));
    }
}

void __real_sync_push(sync_t *st, float complex *fftout);
void __wrap_sync_push(sync_t *st, float complex *fftout)
{
    if ((g_taps & REFH_TAP_FFT) && g_fft_syms < g_fft_limit_blocks * 32) {
        gb_put(&g_fft, fftout, sizeof(float complex) * st->input->acq.fft);
        g_fft_syms++;
    }
    __real_sync_push(st, fftout);
}

void __real_decode_push_pm(decode_t *st, const int8_t *sbit, unsigned int bc);
void __wrap_decode_push_pm(decode_t *st, const int8_t *sbit, unsigned int bc)
{
    if (g_taps & REFH_TAP_SOFT) {
        static uint8_t tmp[4 + PM_BLOCK_SIZE];
        uint32_t b = bc; memcpy(tmp, &b, 4); memcpy(tmp + 4, sbit, PM_BLOCK_SIZE);
        log_rec(REC_SOFT, tmp, sizeof(tmp));
    }
    __real_decode_push_pm(st, sbit, bc);
}

void __real_decode_push_pl_pu_s_t(decode_t *st, const uint8_t *pl, const uint8_t *pu, const uint8_t *s, const uint8_t *t, unsigned int bc);
void __wrap_decode_push_pl_pu_s_t(decode_t *st, const uint8_t *pl, const uint8_t *pu, const uint8_t *s, const uint8_t *t, unsigned int bc)
{
    if (g_taps & REFH_TAP_SOFT) {
        const unsigned n = BLKSZ * PARTITION_WIDTH_AM;
        uint8_t tmp[4 + 4 * BLKSZ * PARTITION_WIDTH_AM];
        uint32_t b = bc; memcpy(tmp, &b, 4);
        memcpy(tmp + 4, pl, n); memcpy(tmp + 4 + n, pu, n); memcpy(tmp + 4 + 2 * n, s, n); memcpy(tmp + 4 + 3 * n, t, n);
        log_rec(REC_AMSYM, tmp, sizeof(tmp));
    }
    __real_decode_push_pl_pu_s_t(st, pl, pu, s, t, bc);
}

static void log_px(uint32_t ch, const int8_t *sbit, unsigned int len, unsigned int bc)
{
    if (!(g_taps & REFH_TAP_SOFT)) return;
    uint8_t *tmp = malloc(12 + len);
    uint32_t h[3] = { ch, bc, len };
    memcpy(tmp, h, 12); memcpy(tmp + 12, sbit, len);
    log_rec(REC_PXSOFT, tmp, 12 + len);
    free(tmp);
}
void __real_decode_push_px1(decode_t *st, const int8_t *sbit, unsigned int len, unsigned int bc);
void __wrap_decode_push_px1(decode_t *st, const int8_t *sbit, unsigned int len, unsigned int bc) { log_px(0, sbit, len, bc); __real_decode_push_px1(st, sbit, len, bc); }
void __real_decode_push_px2(decode_t *st, const int8_t *sbit, unsigned int len, unsigned int bc);
void __wrap_decode_push_px2(decode_t *st, const int8_t *sbit, unsigned int len, unsigned int bc) { log_px(1, sbit, len, bc); __real_decode_push_px2(st, sbit, len, bc); }

void __real_pids_frame_push(pids_t *st, const uint8_t *bits);
void __wrap_pids_frame_push(pids_t *st, const uint8_t *bits)
{
    log_rec(REC_PIDS, bits, PIDS_FRAME_LEN);
    __real_pids_frame_push(st, bits);
}

void __real_frame_push(frame_t *st, uint8_t *bits, size_t length, logical_channel_t lc);
void __wrap_frame_push(frame_t *st, uint8_t *bits, size_t length, logical_channel_t lc)
{
    uint8_t *tmp = malloc(8 + length);
    uint32_t h[2] = { (uint32_t)lc, (uint32_t)length };
    memcpy(tmp, h, 8); memcpy(tmp + 8, bits, length);
    log_rec(REC_FRAME, tmp, 8 + (uint32_t)length);
    free(tmp);
    __real_frame_push(st, bits, length, lc);
}

void __real_input_set_sync_state(input_t *st, unsigned int new_state);
void __wrap_input_set_sync_state(input_t *st, unsigned int new_state)
{
    if (st->sync_state != new_state) {
        int32_t r[2] = { (int32_t)st->sync_state, (int32_t)new_state };
        log_rec(REC_STATE, r, sizeof(r));
    }
    __real_input_set_sync_state(st, new_state);
}

int __real_nrsc5_conv_decode_p1(const int8_t *in, uint8_t *out);
int __wrap_nrsc5_conv_decode_p1(const int8_t *in, uint8_t *out)
{
    int rc = __real_nrsc5_conv_decode_p1(in, out);
    if (g_taps & REFH_TAP_VIT) {
        uint8_t *tmp = malloc(4 + P1_FRAME_LEN_FM * 4);
        uint32_t len = P1_FRAME_LEN_FM; memcpy(tmp, &len, 4);
        memcpy(tmp + 4, in, P1_FRAME_LEN_FM * 3); memcpy(tmp + 4 + P1_FRAME_LEN_FM * 3, out, P1_FRAME_LEN_FM);
        log_rec(REC_VIT, tmp, 4 + P1_FRAME_LEN_FM * 4);
        free(tmp);
    }
    return rc;
}

/* L2 audio transport taps: what frame_process hands to the elastic buffer */
void __real_output_push(output_t *st, const packet_ref_t *ref);
void __wrap_output_push(output_t *st, const packet_ref_t *ref)
{
    if (g_taps & REFH_TAP_L2) {
        uint8_t *tmp = malloc(24 + ref->size + 1);
        uint32_t h[6] = { ref->program, ref->stream_id, ref->seq, ref->size, ref->flags, ref->shape };
        memcpy(tmp, h, 24); memcpy(tmp + 24, ref->data, ref->size + 1);      /* payload + its CRC byte */
        log_rec(REC_L2PKT, tmp, 24 + ref->size + 1);
        free(tmp);
    }
    __real_output_push(st, ref);
}
void __real_output_align(output_t *st, unsigned int program, unsigned int stream_id, unsigned int offset);
void __wrap_output_align(output_t *st, unsigned int program, unsigned int stream_id, unsigned int offset)
{
    if (g_taps & REFH_TAP_L2) { uint32_t h[3] = { program, stream_id, offset }; log_rec(REC_L2ALIGN, h, sizeof(h)); }
    __real_output_align(st, program, stream_id, offset);
}

void __real_output_aas_push(output_t *st, uint8_t *psd, unsigned int len);
void __wrap_output_aas_push(output_t *st, uint8_t *psd, unsigned int len)
{
    if (g_taps & REFH_TAP_L2) log_rec(REC_L2AAS, psd, len);
    __real_output_aas_push(st, psd, len);
}

/* ---- public-API event callback ------------------------------------------------ */
static void on_event(const nrsc5_event_t *evt, void *opaque)
{
    (void)opaque;
    switch (evt->event) {
    case NRSC5_EVENT_SYNC: {
        struct { float freq_offset; int32_t psmi, pli, hppi, aabi, rdbi; } r =
            { evt->sync.freq_offset, evt->sync.psmi, evt->sync.pli, evt->sync.hppi, evt->sync.aabi, evt->sync.rdbi };
        log_rec(REC_SYNC, &r, sizeof(r)); break; }
    case NRSC5_EVENT_LOST_SYNC: log_rec(REC_LOST_SYNC, NULL, 0); break;
    case NRSC5_EVENT_MER: { float r[2] = { evt->mer.lower, evt->mer.upper }; log_rec(REC_MER, r, sizeof(r)); break; }
    case NRSC5_EVENT_BER: { float r = evt->ber.cber; log_rec(REC_BER, &r, sizeof(r)); break; }
    case NRSC5_EVENT_STATION_ID: {
        struct { int32_t fcc; char cc[4]; } r = { evt->station_id.fcc_facility_id, { 0, 0, 0, 0 } };
        strncpy(r.cc, evt->station_id.country_code, 3);
        log_rec(REC_STATION, &r, sizeof(r)); break; }
    case NRSC5_EVENT_AUDIO_SERVICE: {
        if (g_taps & REFH_TAP_L2) {
            int32_t r[8] = { (int32_t)evt->audio_service.program, (int32_t)evt->audio_service.access, (int32_t)evt->audio_service.type,
                             (int32_t)evt->audio_service.codec_mode, (int32_t)evt->audio_service.blend_control,
                             (int32_t)evt->audio_service.digital_audio_gain, (int32_t)evt->audio_service.common_delay,
                             (int32_t)evt->audio_service.latency };
            log_rec(REC_L2SVC, r, sizeof(r));
        }
        break; }
    case NRSC5_EVENT_HDC: {
        size_t n = (g_taps & REFH_TAP_HDC) ? evt->hdc.count : 0;
        uint8_t *tmp = malloc(12 + n);
        uint32_t h[3] = { evt->hdc.program, (uint32_t)evt->hdc.count, evt->hdc.flags };
        memcpy(tmp, h, 12); if (n) memcpy(tmp + 12, evt->hdc.data, n);
        log_rec(REC_HDC, tmp, 12 + (uint32_t)n); free(tmp); break; }
    default: break;
    }
}

/* ---- harness API (ctypes) ------------------------------------------------------ */
int refh_open(int mode, unsigned taps, unsigned fft_limit_blocks)
{
    if (g_radio) return -1;
    g_log.len = g_q15.len = g_fft.len = 0;
    g_taps = taps; g_fft_limit_blocks = fft_limit_blocks; g_fft_syms = 0;
    if (nrsc5_open_pipe(&g_radio) != 0) return -2;
    nrsc5_set_mode(g_radio, mode);
    nrsc5_set_callback(g_radio, on_event, NULL);
    return 0;
}
/* nrsc5_set_mode on the live session (-> input_set_mode -> input_reset, input.c:126-162); the taps keep accumulating.
 * Returns the length of the log so far so that the caller can split it at the switch. */
size_t refh_set_mode(int mode) { nrsc5_set_mode(g_radio, mode); return g_log.len; }
size_t refh_q15_len(void) { return g_q15.len; }
int refh_push_cu8(const uint8_t *iq, unsigned nbytes) { return nrsc5_pipe_samples_cu8(g_radio, iq, nbytes); }
int refh_push_cs16(const int16_t *iq, unsigned n) { return nrsc5_pipe_samples_cs16(g_radio, iq, n); }
/* feed a whole capture in `chunk`-byte calls, as src/main.c:1097-1120 does with 32768 */
int refh_run_cu8(const uint8_t *iq, size_t nbytes, unsigned chunk)
{
    for (size_t off = 0; off < nbytes; off += chunk) {
        unsigned n = (nbytes - off < chunk) ? (unsigned)(nbytes - off) : chunk;
        nrsc5_pipe_samples_cu8(g_radio, iq + off, n);
    }
    return 0;
}
int refh_run_cs16(const int16_t *iq, size_t n, unsigned chunk)
{
    for (size_t off = 0; off < n; off += chunk) {
        unsigned k = (n - off < chunk) ? (unsigned)(n - off) : chunk;
        nrsc5_pipe_samples_cs16(g_radio, iq + off, k);
    }
    return 0;
}
/* hand one logical frame straight to the reference's L2 (frame.c:645): pins oracle/nrsc5_oracle_l2.c without a demod run.
 * The session is put in FINE first so that the sync-loss feedback of frame_process shows up as a REC_STATE record. */
void refh_frame_push(const uint8_t *bits, size_t length, int lc)
{
    uint8_t *tmp = malloc(length);
    memcpy(tmp, bits, length);
    g_radio->input.sync_state = SYNC_STATE_FINE;
    __real_frame_push(&g_radio->input.frame, tmp, length, (logical_channel_t)lc);
    free(tmp);
}
/* the same frame through the entry point a maintainer would add for the device's L2 index (ref_shim/frame_indexed.c) */
struct nrsc5hip_l2_frame;
void frame_push_indexed(frame_t *st, const struct nrsc5hip_l2_frame *ix, const uint8_t *pdu_bytes, logical_channel_t lc);
void refh_frame_push_indexed(const void *ix, const uint8_t *pdu_bytes, int lc)
{
    g_radio->input.sync_state = SYNC_STATE_FINE;
    frame_push_indexed(&g_radio->input.frame, (const struct nrsc5hip_l2_frame *)ix, pdu_bytes, (logical_channel_t)lc);
}
void refh_force_resync(void) { input_set_sync_state(&g_radio->input, SYNC_STATE_NONE); }
void refh_close(void) { if (g_radio) { nrsc5_close(g_radio); g_radio = NULL; } }
size_t refh_buf(int which, const uint8_t **p)
{
    gbuf *b = which == 0 ? &g_log : which == 1 ? &g_q15 : &g_fft;
    *p = b->p; return b->len;
}
size_t refh_sizeof_session(void) { return sizeof(struct nrsc5_t); }
