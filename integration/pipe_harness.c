/* TEST INFRASTRUCTURE: drives any libnrsc5 build (the reference's, or the HIP drop-in) through the PUBLIC
 * pipe API only (nrsc5.h:712-871) and logs the events both must agree on. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <nrsc5.h>

typedef struct { uint8_t *p; size_t len, cap; } gbuf;
static gbuf g_log, g_log2;

static void put(gbuf *g, const void *src, size_t n)
{
    if (g->len + n > g->cap) { size_t nc = g->cap ? g->cap * 2 : 1 << 20; while (nc < g->len + n) nc *= 2; g->p = realloc(g->p, nc); g->cap = nc; }
    memcpy(g->p + g->len, src, n); g->len += n;
}
static void rec(gbuf *g, uint32_t kind, const void *payload, uint32_t n)
{
    uint32_t hdr[2] = { kind, n }, z = 0, pad = (4 - (n & 3)) & 3;
    put(g, hdr, 8); if (n) put(g, payload, n); if (pad) put(g, &z, pad);
}
static void on_event(const nrsc5_event_t *evt, void *opaque)
{
    gbuf *g = (gbuf *)opaque;
    switch (evt->event) {
    case NRSC5_EVENT_SYNC: { struct { float f; int32_t a[5]; } r = { evt->sync.freq_offset, { evt->sync.psmi, evt->sync.pli, evt->sync.hppi, evt->sync.aabi, evt->sync.rdbi } }; rec(g, 6, &r, sizeof(r)); break; }
    case NRSC5_EVENT_LOST_SYNC: rec(g, 7, NULL, 0); break;
    case NRSC5_EVENT_MER: { float r[2] = { evt->mer.lower, evt->mer.upper }; rec(g, 8, r, sizeof(r)); break; }
    case NRSC5_EVENT_BER: { float r = evt->ber.cber; rec(g, 9, &r, sizeof(r)); break; }
    case NRSC5_EVENT_HDC: {
        uint8_t *tmp = malloc(12 + evt->hdc.count);
        uint32_t h[3] = { evt->hdc.program, (uint32_t)evt->hdc.count, evt->hdc.flags };
        memcpy(tmp, h, 12); if (evt->hdc.count) memcpy(tmp + 12, evt->hdc.data, evt->hdc.count);
        rec(g, 10, tmp, 12 + (uint32_t)evt->hdc.count); free(tmp); break; }
    default: break;
    }
}

static double g_feed_seconds;
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
/* wall time of the nrsc5_pipe_samples_* loop of the last pipe_run (session open / close excluded): what `nrsc5 -r` spends per capture */
double pipe_last_feed_seconds(void) { return g_feed_seconds; }

/* mode: NRSC5_MODE_FM / NRSC5_MODE_AM; cs16 != 0: iq holds n int16 values, else n bytes of cu8.
 * flags & PIPE_NO_FLUSH: no zero-length call after the loop -- the way src/main.c:1095-1121 ends a file: it just calls nrsc5_close.  With the
 * drop-in's default (overlapped) delivery the last block's events are then delivered by nrsc5_close (input_free); the log is read after the
 * close either way.  events_at_loop_end (optional): how many bytes of the log existed when the feeding loop (incl. the flush, if any) ended. */
#define PIPE_NO_FLUSH 1
size_t pipe_run_opts(const void *iq, size_t n, unsigned chunk, int mode, int cs16, int flags, size_t *log_bytes_at_loop_end, const uint8_t **out);
size_t pipe_run(const void *iq, size_t n, unsigned chunk, int mode, int cs16, const uint8_t **out)
{
    return pipe_run_opts(iq, n, chunk, mode, cs16, 0, NULL, out);
}

size_t pipe_run_opts(const void *iq, size_t n, unsigned chunk, int mode, int cs16, int flags, size_t *log_bytes_at_loop_end, const uint8_t **out)
{
    nrsc5_t *radio = NULL;
    g_log.len = 0;
    if (nrsc5_open_pipe(&radio) != 0) return 0;
    nrsc5_set_mode(radio, mode);
    nrsc5_set_callback(radio, on_event, &g_log);
    const double t0 = now_s();
    for (size_t off = 0; off < n; off += chunk) {
        unsigned k = (n - off < chunk) ? (unsigned)(n - off) : chunk;
        if (cs16) nrsc5_pipe_samples_cs16(radio, (const int16_t *)iq + off, k);
        else nrsc5_pipe_samples_cu8(radio, (const uint8_t *)iq + off, k);
    }
    /* a zero-length call: with the drop-in, "deliver what is still pending" (the last block may be on the device when the loop
     * ends); the plain reference has nothing pending and returns at once.  Inside the timed region: no work escapes the clock. */
    if (!(flags & PIPE_NO_FLUSH)) {
        if (cs16) nrsc5_pipe_samples_cs16(radio, (const int16_t *)iq, 0);
        else nrsc5_pipe_samples_cu8(radio, (const uint8_t *)iq, 0);
    }
    g_feed_seconds = now_s() - t0;
    if (log_bytes_at_loop_end) *log_bytes_at_loop_end = g_log.len;
    nrsc5_close(radio);
    *out = g_log.p;
    return g_log.len;
}

/* ONE session, two captures: A, then nrsc5_set_mode(mode_b) on the live session (input_set_mode -> input_reset, input.c:126-162: the FIR windows are
 * rewound, not cleared), then B.  *split = bytes of the log that belong to A (everything delivered before nrsc5_set_mode returned). */
size_t pipe_run_two(const void *iq_a, size_t na, int mode_a, int cs16_a, const void *iq_b, size_t nb, int mode_b, int cs16_b, unsigned chunk,
                    size_t *split, const uint8_t **out)
{
    nrsc5_t *radio = NULL;
    g_log.len = 0;
    if (nrsc5_open_pipe(&radio) != 0) return 0;
    nrsc5_set_mode(radio, mode_a);
    nrsc5_set_callback(radio, on_event, &g_log);
    for (size_t off = 0; off < na; off += chunk) {
        unsigned k = (na - off < chunk) ? (unsigned)(na - off) : chunk;
        if (cs16_a) nrsc5_pipe_samples_cs16(radio, (const int16_t *)iq_a + off, k);
        else nrsc5_pipe_samples_cu8(radio, (const uint8_t *)iq_a + off, k);
    }
    nrsc5_set_mode(radio, mode_b);
    if (split) *split = g_log.len;
    for (size_t off = 0; off < nb; off += chunk) {
        unsigned k = (nb - off < chunk) ? (unsigned)(nb - off) : chunk;
        if (cs16_b) nrsc5_pipe_samples_cs16(radio, (const int16_t *)iq_b + off, k);
        else nrsc5_pipe_samples_cu8(radio, (const uint8_t *)iq_b + off, k);
    }
    nrsc5_close(radio);
    *out = g_log.p;
    return g_log.len;
}

size_t pipe_run_cu8(const uint8_t *iq, size_t nbytes, unsigned chunk, const uint8_t **out)
{
    return pipe_run(iq, nbytes, chunk, NRSC5_MODE_FM, 0, out);
}

/* Two sessions of ONE process fed alternately, chunk by chunk (cu8 FM): each must see exactly the events it sees alone. */
int pipe_run_pair(const uint8_t *iq0, size_t n0, const uint8_t *iq1, size_t n1, unsigned chunk,
                  const uint8_t **out0, size_t *len0, const uint8_t **out1, size_t *len1)
{
    nrsc5_t *r0 = NULL, *r1 = NULL;
    g_log.len = 0; g_log2.len = 0;
    if (nrsc5_open_pipe(&r0) != 0) return -1;
    if (nrsc5_open_pipe(&r1) != 0) { nrsc5_close(r0); return -1; }
    nrsc5_set_callback(r0, on_event, &g_log);
    nrsc5_set_callback(r1, on_event, &g_log2);
    for (size_t off = 0; off < n0 || off < n1; off += chunk) {
        if (off < n0) nrsc5_pipe_samples_cu8(r0, iq0 + off, (n0 - off < chunk) ? (unsigned)(n0 - off) : chunk);
        if (off < n1) nrsc5_pipe_samples_cu8(r1, iq1 + off, (n1 - off < chunk) ? (unsigned)(n1 - off) : chunk);
    }
    nrsc5_close(r0); nrsc5_close(r1);
    *out0 = g_log.p; *len0 = g_log.len; *out1 = g_log2.p; *len1 = g_log2.len;
    return 0;
}
