/* TEST INFRASTRUCTURE -- see nrsc5_oracle.h.  Plain-C restatement of the reference's
 * cu8/cs16 IQ -> decimate -> acquire -> sync -> de-interleave -> Viterbi -> descramble path,
 * organised as one function per pipeline stage (the same cut the HIP kernels use) instead
 * of the reference's object-per-file layout.  Floating-point expressions keep the
 * reference's operand types and evaluation order so that, built with the same compiler
 * flags, the float trace is reproducible against oracle/_ref. */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "nrsc5_oracle.h"
#include "cpu_fft.h"

#define FFT_N      2048
#define CP_N       112
#define SYM_N      (FFT_N + CP_N)            /* 2160 */
#define NSYM       32
#define WIN_N      (SYM_N * (NSYM + 1))      /* 71280 samples per acquire window */
#define LB0        (FFT_N / 2 - 546)         /* 478 */
#define UB1        (FFT_N / 2 + 546)         /* 1570 */
#define PW         19                        /* carriers per partition incl. reference */
#define PM_PART    10
#define PM_BLOCK   23040
#define P1_LEN     146176
#define P1_CODED   365440
#define PIDS_LEN   80
#define PIDS_CODED 200
#define FS_FM      744187.5
#define PX_MAX     4608                      /* soft bits per block from 2 extended partitions per sideband = P3 frame bits (MP3/MP11) */

/* ------------------------------------------------------------------------------------ */
/* constants that are part of the algorithm's contract                                    */

/* half-band prototype, input.c:27-40 (GNU Radio design, 4 unique taps + unity centre) */
static const float HB_TAPS[4] = { 0.6062333583831787, -0.13481467962265015, 0.032919470220804214, -0.00410953676328063 };
/* acquisition band-select FIR, acquire.c:28-61 (32 entries, last one zero) */
static const float ACQ_TAPS_FM[32] = {
    -0.000685643230099231, 0.005636964458972216, 0.009015781804919243, -0.015486305579543114,
    -0.035108357667922974, 0.017446253448724747, 0.08155813068151474, 0.007995186373591423,
    -0.13311293721199036, -0.0727422907948494, 0.15914097428321838, 0.16498781740665436,
    -0.1324498951435089, -0.2484012246131897, 0.051773931831121445, 0.2821577787399292,
    0.051773931831121445, -0.2484012246131897, -0.1324498951435089, 0.16498781740665436,
    0.15914097428321838, -0.0727422907948494, -0.13311293721199036, 0.007995186373591423,
    0.08155813068151474, 0.017446253448724747, -0.035108357667922974, -0.015486305579543114,
    0.009015781804919243, 0.005636964458972216, -0.000685643230099231, 0 };
/* primary-main partition order, decode.c:34-37 */
static const int8_t PM_V[20] = { 10, 2, 18, 6, 14, 8, 16, 0, 12, 4, 11, 3, 19, 7, 15, 9, 17, 1, 13, 5 };
/* PSMI -> compatibility mode, sync.c:29-35 */
static const int COMPAT[64] = {
    0, 1, 2, 3, 1, 5, 6, 5, 6, 1, 2, 11, 1, 5, 6, 5, 6, 1, 2, 3, 1, 5, 6, 5, 6, 1, 2, 11, 1, 5, 6, 5,
    6, 1, 2, 3, 1, 5, 6, 5, 6, 1, 2, 11, 1, 5, 6, 5, 6, 1, 2, 3, 1, 5, 6, 5, 6, 1, 2, 11, 1, 5, 6, 5 };

static int16_t hb_q15[4];      /* Q15 of HB_TAPS in window order: index 0 pairs a[0]+a[14] */
static int16_t acq_q15[17];    /* acq_q15[i], i=1..16: tap applied to a[i]+a[32-i] (i=16: centre) */
static float shape_fm[SYM_N];
static int32_t p1_gather[P1_CODED];
static int32_t pids_gather[16][PIDS_CODED];
static uint8_t scr_seq[2047];
static int tables_ready;

static void build_tables(void)
{
    if (tables_ready) return;
    /* firdecim_q15_create: q15 = (int16)(tap * 32767.0f), stored reversed (firdecim_q15.c:37-42) */
    for (int i = 0; i < 4; i++) hb_q15[i] = (int16_t)(HB_TAPS[3 - i] * 32767.0f);
    for (int i = 1; i <= 16; i++) acq_q15[i] = (int16_t)(ACQ_TAPS_FM[31 - i] * 32767.0f);
    /* pulse shape, acquire.c:322-331 */
    for (int i = 0; i < SYM_N; i++) {
        if (i < CP_N) shape_fm[i] = sinf(M_PI / 2 * i / CP_N);
        else if (i < FFT_N) shape_fm[i] = 1;
        else shape_fm[i] = cosf(M_PI / 2 * (i - FFT_N) / CP_N);
    }
    /* interleaver I (decode.c:296-322 with J=20,B=16,C=36,M=1) and II (decode.c:324-342) as tables */
    for (unsigned i = 0; i < P1_CODED; i++) {
        unsigned part = PM_V[i % 20], block = ((i / 20) + part * 7) % 16, k = i / 320;
        unsigned row = (k * 11) % 32, col = (k * 11 + k / 288) % 36;
        p1_gather[i] = (block * 32 + row) * 720 + part * 36 + col;
    }
    for (unsigned bc = 0; bc < 16; bc++)
        for (unsigned n = 0; n < PIDS_CODED; n++) {
            unsigned i = bc * PIDS_CODED + n, part = PM_V[i % 20], block = i / PIDS_CODED;
            unsigned k = ((i / 20) % 10) + P1_CODED / 320;
            unsigned row = (k * 11) % 32, col = (k * 11 + k / 288) % 36;
            pids_gather[bc][n] = (block * 32 + row) * 720 + part * 36 + col;
        }
    /* scrambler, decode.c:279-294: 11-bit LFSR, period 2047, restarted per frame */
    unsigned val = 0x3ff;
    for (int i = 0; i < 2047; i++) {
        unsigned bit = ((val >> 9) ^ val) & 1;
        val |= bit << 11; val >>= 1;
        scr_seq[i] = bit;
    }
    tables_ready = 1;
}

/* ------------------------------------------------------------------------------------ */
/* K1: cu8 -> Q15 half-band decimator                                                     */

static inline int16_t u8_q15(uint8_t x) { return (int16_t)(((int16_t)x - 127) * 64); }   /* defines.h:93 */

/* one half-band output from the 15-sample window a[0..14] (firdecim_q15.c:137-151, generic branch):
 * every product is shifted before accumulation and the accumulator is an int16 that wraps */
static inline int16_t hb_dot(const int16_t a[15])
{
    int16_t acc = 0;
    for (int i = 0; i < 4; i++)
        acc = (int16_t)(acc + (((a[2 * i] + a[14 - 2 * i]) * hb_q15[i]) >> 15));
    return (int16_t)(acc + a[7]);
}

size_t orc_halfband_fm_cu8(orc_c16 hist[14], const uint8_t *iq, size_t nbytes, orc_c16 *out)
{
    build_tables();
    size_t nout = nbytes / 4;
    int16_t wr[16], wi[16];   /* 14 history + 2 new */
    for (int k = 0; k < 14; k++) { wr[k] = hist[k].r; wi[k] = hist[k].i; }
    for (size_t m = 0; m < nout; m++) {
        wr[14] = u8_q15(iq[4 * m]);     wi[14] = u8_q15(iq[4 * m + 1]);
        wr[15] = u8_q15(iq[4 * m + 2]); wi[15] = u8_q15(iq[4 * m + 3]);
        out[m].r = hb_dot(wr);          /* window ends at the first sample of the pair */
        out[m].i = hb_dot(wi);
        memmove(wr, wr + 2, sizeof(int16_t) * 14);
        memmove(wi, wi + 2, sizeof(int16_t) * 14);
    }
    for (int k = 0; k < 14; k++) { hist[k].r = wr[k]; hist[k].i = wi[k]; }
    return nout;
}

/* ------------------------------------------------------------------------------------ */
/* K2a: 32-tap acquisition FIR (firdecim_q15.c:95-109)                                     */

void orc_fir32_fm(orc_c16 hist[31], const orc_c16 *in, size_t n, orc_c16 *out)
{
    build_tables();
    orc_c16 *w = malloc(sizeof(orc_c16) * (n + 31));
    memcpy(w, hist, sizeof(orc_c16) * 31);
    memcpy(w + 31, in, sizeof(orc_c16) * n);
    for (size_t t = 0; t < n; t++) {
        const orc_c16 *a = w + t;        /* a[31] is the newest sample, a[0] is unused (tap 0) */
        int16_t sr = 0, si = 0;
        for (int i = 1; i < 16; i++) {
            sr = (int16_t)(sr + (((a[i].r + a[32 - i].r) * acq_q15[i]) >> 15));
            si = (int16_t)(si + (((a[i].i + a[32 - i].i) * acq_q15[i]) >> 15));
        }
        sr = (int16_t)(sr + ((a[16].r * acq_q15[16]) >> 15));
        si = (int16_t)(si + ((a[16].i * acq_q15[16]) >> 15));
        out[t].r = sr; out[t].i = si;
    }
    memcpy(hist, w + n, sizeof(orc_c16) * 31);
    free(w);
}

/* ------------------------------------------------------------------------------------ */
/* K2b: cyclic-prefix correlation (acquire.c:129-151)                                      */

static inline float complex q15_to_cf_conj(orc_c16 v)   /* defines.h:111 */
{
    return CMPLXF((float)v.r / 32767.0f, (float)v.i / -32767.0f);
}
static inline float norm2(float complex v) { float a = crealf(v), b = cimagf(v); return a * a + b * b; }

static void cp_correlate(const float complex *buf, int *samperr_out, float complex *peak)
{
    static __thread float complex sums[SYM_N];
    float complex max_v = 0;
    float max_mag = -1.0f;
    int samperr = 0;
    memset(sums, 0, sizeof(sums));
    for (int i = 0; i < SYM_N; ++i)
        for (int j = 0; j < NSYM; ++j)
            sums[i] += buf[i + j * SYM_N] * conjf(buf[i + j * SYM_N + FFT_N]);
    for (int i = 0; i < SYM_N; ++i) {
        float complex v = 0;
        for (int j = 0; j < CP_N; ++j)
            v += sums[(i + j) % SYM_N] * shape_fm[j] * shape_fm[j + FFT_N];
        float mag = norm2(v);
        if (mag > max_mag) { max_mag = mag; max_v = v; samperr = (i + SYM_N - 15) % SYM_N; }
    }
    *samperr_out = samperr; *peak = max_v;
}

void orc_cp_correlate_fm(const orc_c16 *filtered, int *samperr, float peak[2])
{
    build_tables();
    float complex *buf = malloc(sizeof(float complex) * WIN_N), pk;
    for (int i = 0; i < WIN_N; i++) buf[i] = q15_to_cf_conj(filtered[i]);
    cp_correlate(buf, samperr, &pk);
    peak[0] = crealf(pk); peak[1] = cimagf(pk);
    free(buf);
}

/* ------------------------------------------------------------------------------------ */
/* K6: de-interleavers                                                                    */

void orc_deinterleave_p1(const int8_t *pm, int8_t *out)
{
    build_tables();
    unsigned o = 0;
    for (unsigned i = 0; i < P1_CODED; i++) {
        out[o++] = pm[p1_gather[i]];
        if ((o % 6) == 5) out[o++] = 0;     /* depuncture [1,1,1,1,1,0] */
    }
}

void orc_deinterleave_pids(const int8_t *pm, unsigned bc, int8_t *out)
{
    build_tables();
    unsigned o = 0;
    for (unsigned n = 0; n < PIDS_CODED; n++) {
        out[o++] = pm[pids_gather[bc][n]];
        if ((o % 6) == 5) out[o++] = 0;
    }
}

/* ------------------------------------------------------------------------------------ */
/* K7: tail-biting Viterbi (conv_dec.c:217-249 trellis, :402-427 schedule, :304-339 traceback;  */
/* ACS rule conv_gen.h:32-63).  Decisions are kept one bit per state instead of an int16.      */

int orc_viterbi(const int8_t *in, int len, int k, const unsigned gens[3], uint8_t *out)
{
    const int ns = 1 << (k - 1), half = ns / 2, extra = 32, steps = len + 2 * extra;
    const int interval = 32767 / (3 * 127) - k;          /* conv_dec.c:370 */
    const unsigned smask = (unsigned)ns - 2;             /* vstate_lshift mask: 0x3e / 0xfe */
    int8_t (*sgn)[3] = malloc(sizeof(int8_t[3]) * half);
    int16_t *pm = calloc(ns, sizeof(int16_t)), *nm = malloc(sizeof(int16_t) * ns);
    uint8_t *dec = malloc((size_t)steps * ns / 8);      /* bit s of step t: 1 = came from 2b+1 */
    if (!sgn || !pm || !nm || !dec) return -1;

    for (int b = 0; b < half; b++) {                     /* expected outputs on edge 2b --0--> b */
        unsigned reg = ((unsigned)b << 1) & smask;
        for (int g = 0; g < 3; g++) sgn[b][g] = __builtin_parity(reg & gens[g]) ? 1 : -1;
    }
    int j = len - extra;
    for (int t = 0; t < steps; t++, j++) {
        if (j == len) j = 0;
        const int8_t *sq = in + 3 * j;
        uint8_t *d = dec + (size_t)t * ns / 8;
        memset(d, 0, ns / 8);
        for (int b = 0; b < half; b++) {
            int m = sq[0] * sgn[b][0] + sq[1] * sgn[b][1] + sq[2] * sgn[b][2];
            int e = pm[2 * b], o = pm[2 * b + 1];
            int s0 = e + m, s1 = o - m, s2 = e - m, s3 = o + m;
            if (s0 > s1) nm[b] = (int16_t)s0; else { nm[b] = (int16_t)s1; d[b >> 3] |= 1u << (b & 7); }
            if (s2 > s3) nm[b + half] = (int16_t)s2; else { nm[b + half] = (int16_t)s3; d[(b + half) >> 3] |= 1u << ((b + half) & 7); }
        }
        if (t % interval == 0) {
            int16_t mn = nm[0];
            for (int s = 1; s < ns; s++) if (nm[s] < mn) mn = nm[s];
            for (int s = 0; s < ns; s++) nm[s] = (int16_t)(nm[s] - mn);
        }
        int16_t *tmp = pm; pm = nm; nm = tmp;
    }
    int best = -1, second = -1; unsigned state = 0;
    for (int s = 0; s < ns; s++) if (pm[s] > best) { second = best; best = pm[s]; state = s; }
    int rc = best - second;
    if (best < 0) rc = -71;                              /* -EPROTO */
    else {
        for (int t = steps - 1; t >= 0; t--) {
            const uint8_t *d = dec + (size_t)t * ns / 8;
            unsigned bit = (d[state >> 3] >> (state & 7)) & 1;
            if (t >= extra && t < len + extra) out[t - extra] = (state >> (k - 2)) & 1;
            state = ((state << 1) & smask) | bit;
        }
    }
    free(sgn); free(pm); free(nm); free(dec);
    return rc;
}

int orc_viterbi_k7(const int8_t *in, int len, uint8_t *out)
{
    static const unsigned g[3] = { 0133, 0171, 0165 };   /* decode.c:39-45 */
    return orc_viterbi(in, len, 7, g, out);
}

/* ------------------------------------------------------------------------------------ */
/* K8: descrambler + re-encode BER                                                         */

void orc_descramble(uint8_t *bits, unsigned len)
{
    build_tables();
    for (unsigned i = 0; i < len; i++) bits[i] ^= scr_seq[i % 2047];
}

int orc_bit_errors_k7(const int8_t *coded, const uint8_t *decoded, int len)
{
    static const unsigned g[3] = { 0133, 0171, 0165 };
    unsigned r = 0;
    int errors = 0;
    for (int i = 0; i < 6; i++) r = (r >> 1) | ((unsigned)decoded[len - 6 + i] << 6);
    for (int i = 0, j = 0; i < len; i++, j += 3) {
        r = (r >> 1) | ((unsigned)decoded[i] << 6);
        for (int q = 0; q < 3; q++)
            if (((j + q) % 6) != 5 && ((coded[j + q] > 0) != __builtin_parity(r & g[q]))) errors++;
    }
    return errors;
}

/* ------------------------------------------------------------------------------------ */
/* stream object                                                                          */

typedef struct { uint8_t *p; size_t len, cap; } gbuf;

struct orc_stream {
    /* front end */
    orc_c16 hb_hist[14];
    orc_c16 fir_hist[31];
    orc_c16 ring[WIN_N];
    unsigned ring_fill;
    unsigned sync_state;
    /* acquire (acquire.h:7-33) */
    float prev_angle;
    float complex phase;
    int keep_extra, cfo;
    /* sync (sync.h:7-32) */
    float complex (*bins)[NSYM];          /* [2048][32] */
    float (*phases)[NSYM];
    float costas_freq[FFT_N], costas_phase[FFT_N];
    unsigned sym_idx;
    int psmi, cfo_wait;
    unsigned bc;
    int samperr;
    float angle, alpha, beta;
    int mer_cnt;
    float error_lb, error_ub;
    /* decode (decode.h:19-62) */
    int8_t pm[16 * PM_BLOCK];
    int started_pm;
    struct px_chan { int8_t pair[2 * PX_MAX]; int8_t mem[32 * PX_MAX]; unsigned pos, taken[4]; int ready, started; } px[2];
    /* plumbing */
    gbuf log, q15, fft;
    unsigned taps, fft_limit, fft_syms;
    orc_p1_hook hook; void *hook_user;
};

static void gb_put(gbuf *b, const void *src, size_t n)
{
    if (b->len + n > b->cap) {
        size_t nc = b->cap ? b->cap * 2 : (1 << 20);
        while (nc < b->len + n) nc *= 2;
        b->p = realloc(b->p, nc); b->cap = nc;
    }
    memcpy(b->p + b->len, src, n); b->len += n;
}
static void log_rec(orc_stream *s, uint32_t kind, const void *payload, uint32_t n)
{
    uint32_t hdr[2] = { kind, n }, z = 0, pad = (4 - (n & 3)) & 3;
    gb_put(&s->log, hdr, sizeof(hdr));
    if (n) gb_put(&s->log, payload, n);
    if (pad) gb_put(&s->log, &z, pad);
}

/* input.c:172-188 */
static void set_sync_state(orc_stream *s, unsigned new_state)
{
    if (s->sync_state == new_state) return;
    int32_t r[2] = { (int32_t)s->sync_state, (int32_t)new_state };
    log_rec(s, ORC_REC_STATE, r, sizeof(r));
    if (s->sync_state == ORC_SYNC_FINE) log_rec(s, ORC_REC_LOST_SYNC, NULL, 0);
    if (new_state == ORC_SYNC_FINE) {
        float freq_offset = (s->prev_angle - 2 * M_PI * s->cfo) * FS_FM / (2 * M_PI * FFT_N);
        struct { float f; int32_t psmi, pli, hppi, aabi, rdbi; } ev = { freq_offset, s->psmi, -1, -1, -1, -1 };
        log_rec(s, ORC_REC_SYNC, &ev, sizeof(ev));
    }
    s->sync_state = new_state;
}

void orc_force_resync(orc_stream *s) { set_sync_state(s, ORC_SYNC_NONE); }

/* ---- decode side (decode.c:378-391, 451-472) ---------------------------------------- */
static void decode_block(orc_stream *s, const int8_t *soft, unsigned bc)
{
    static __thread int8_t coded_p1[P1_LEN * 3];
    static __thread uint8_t bits_p1[P1_LEN];
    int8_t coded[PIDS_LEN * 3];
    uint8_t bits[PIDS_LEN];

    memcpy(s->pm + PM_BLOCK * bc, soft, PM_BLOCK);
    orc_deinterleave_pids(s->pm, bc, coded);
    orc_viterbi_k7(coded, PIDS_LEN, bits);
    orc_descramble(bits, PIDS_LEN);
    log_rec(s, ORC_REC_PIDS, bits, PIDS_LEN);

    if (bc == 0) s->started_pm = 1;
    if (s->started_pm && bc == 15) {
        orc_deinterleave_p1(s->pm, coded_p1);
        orc_viterbi_k7(coded_p1, P1_LEN, bits_p1);
        float cber = (float)orc_bit_errors_k7(coded_p1, bits_p1, P1_LEN) / P1_CODED;
        log_rec(s, ORC_REC_BER, &cber, sizeof(cber));
        orc_descramble(bits_p1, P1_LEN);
        uint8_t *tmp = malloc(8 + P1_LEN);
        uint32_t h[2] = { 0, P1_LEN };
        memcpy(tmp, h, 8); memcpy(tmp + 8, bits_p1, P1_LEN);
        log_rec(s, ORC_REC_FRAME, tmp, 8 + P1_LEN);
        free(tmp);
        if (s->hook && s->hook(s->hook_user, bits_p1, P1_LEN))
            set_sync_state(s, ORC_SYNC_NONE);           /* frame.c:535-540 */
    }
}

/* ---- extended sidebands: PX1 -> P3, PX2 -> P4 (decode.c:344-376, 393-437) -------------------- */

/* interleaver_iv: one call consumes the 2 x frame_len soft bits of a block pair and emits one depunctured
 * (rate 2/3 -> [1,0,1,1,0,1]) trellis input of 3 x frame_len.  Convolutional: a bit read now was written up to
 * 32 blocks earlier; `mem` persists across calls, `pos` / `taken` restart every 32 blocks. */
void orc_interleave_px(int8_t *mem, unsigned *pos, unsigned taken[4], int *ready, const int8_t *pair, unsigned frame_len, int8_t *out)
{
    const unsigned wide = frame_len == PX_MAX;
    const unsigned J = wide ? 4 : 2, C = 36, M = wide ? 2 : 4, N = 32 * frame_len;
    const unsigned bk_bits = 32 * C, bk_adj = 32 * C - 1;
    if (*pos == N) { *pos = 0; memset(taken, 0, sizeof(unsigned) * 4); *ready = 1; }
    unsigned o = 0;
    for (unsigned i = 0; i < frame_len * 2; i++) {
        const unsigned part = ((*pos + 2 * (M / 4)) / M) % J;
        const unsigned pti = taken[part]++;
        const unsigned block = (pti + (part * 7) - (bk_adj * (pti / bk_bits))) % 32;
        const unsigned row = ((11 * pti) % bk_bits) / C, col = (pti * 11) % C;
        out[o++] = mem[(block * 32 + row) * (J * C) + part * C + col];
        if ((o % 6) == 1 || (o % 6) == 4) out[o++] = 0;
        mem[*pos] = pair[i];
        (*pos)++;
    }
}

static void px_push(orc_stream *s, int ch, const int8_t *soft, unsigned len, unsigned bc)
{
    struct px_chan *x = &s->px[ch];
    if (s->taps & ORC_TAP_SOFT) {
        uint8_t *tmp = malloc(12 + len);
        uint32_t h[3] = { (uint32_t)ch, bc, len };
        memcpy(tmp, h, 12); memcpy(tmp + 12, soft, len);
        log_rec(s, ORC_REC_PXSOFT, tmp, 12 + len);
        free(tmp);
    }
    if (bc % 2 == 0) x->started = 1;
    if (!x->started) return;
    memcpy(x->pair + len * (bc % 2), soft, len);
    if (bc % 2 == 0) return;
    int8_t coded[3 * PX_MAX];
    uint8_t bits[PX_MAX];
    orc_interleave_px(x->mem, &x->pos, x->taken, &x->ready, x->pair, len, coded);
    if (!x->ready) return;
    orc_viterbi_k7(coded, (int)len, bits);
    orc_descramble(bits, len);
    uint8_t *tmp = malloc(8 + len);
    uint32_t h[2] = { 1 + (uint32_t)ch, len };               /* P3_LOGICAL_CHANNEL / P4_LOGICAL_CHANNEL */
    memcpy(tmp, h, 8); memcpy(tmp + 8, bits, len);
    log_rec(s, ORC_REC_FRAME, tmp, 8 + len);
    free(tmp);
}

/* ---- sync side (sync.c) ---------------------------------------------------------------- */

/* sync.c:90-130: one reference carrier through its 2nd-order Costas loop for a block */
static void costas_ref(orc_stream *s, unsigned ref, int cfo)
{
    static const signed char pat[NSYM] = {
        -1, 1, -1, -1, -1, 1, 1, 0, 1, -1, 0, 0, 0, -1, -1, 0,
        0, 0, 0, 0, -1, 1, -1, 0, 0, 0, 0, 0, 0, 0, 0, -1 };
    float cfo_freq = 2 * M_PI * cfo * CP_N / FFT_N;
    float complex *z = s->bins[ref];
    for (unsigned n = 0; n < NSYM; n++) {
        float error = cargf(z[n] * z[n] * cexpf(-I * 2 * s->costas_phase[ref])) * 0.5;
        s->phases[ref][n] = s->costas_phase[ref];
        z[n] *= cexpf(-I * s->costas_phase[ref]);
        s->costas_freq[ref] += s->beta * error;
        if (s->costas_freq[ref] > 0.5) s->costas_freq[ref] = 0.5;
        if (s->costas_freq[ref] < -0.5) s->costas_freq[ref] = -0.5;
        s->costas_phase[ref] += s->costas_freq[ref] + cfo_freq + (s->alpha * error);
        if (s->costas_phase[ref] > M_PI) s->costas_phase[ref] -= 2 * M_PI;
        if (s->costas_phase[ref] < -M_PI) s->costas_phase[ref] += 2 * M_PI;
    }
    float x = 0;
    for (unsigned n = 0; n < NSYM; n++) x += crealf(z[n]) * pat[n];
    if (x < 0) {                                         /* resolve the pi ambiguity */
        for (unsigned n = 0; n < NSYM; n++) { s->phases[ref][n] += M_PI; z[n] *= -1; }
        s->costas_phase[ref] += M_PI;
    }
}

/* sync.c:132-136 */
static void uncostas_ref(orc_stream *s, unsigned ref)
{
    for (unsigned n = 0; n < NSYM; n++) s->bins[ref][n] *= cexpf(I * s->phases[ref][n]);
}

static void ref_needle(signed char nd[NSYM], unsigned rsid)
{
    static const signed char base[NSYM] = {
        0, 1, 0, 0, 0, 1, 1, -1, 1, 0, 0, 0, -1, 0, 0, -1,
        -1, -1, -1, -1, 0, 1, 0, -1, -1, -1, -1, -1, -1, -1, -1, 0 };
    memcpy(nd, base, NSYM);
    nd[10] = rsid >> 1; nd[11] = (rsid >> 1) ^ (rsid & 1);
}

/* sync.c:169-186 */
static int read_ref(orc_stream *s, unsigned ref, unsigned rsid, unsigned *bc, unsigned *psmi)
{
    signed char nd[NSYM];
    unsigned char d[NSYM], prev = 0;
    ref_needle(nd, rsid);
    for (int n = 0; n < NSYM; n++)
        if (nd[n] >= 0 && nd[n] != (crealf(s->bins[ref][n]) > 0)) return -1;
    for (int n = 0; n < NSYM; n++) {                     /* DBPSK, sync.c:138-148 */
        unsigned char bit = crealf(s->bins[ref][n]) <= 0 ? 0 : 1;
        d[n] = bit ^ prev; prev = bit;
    }
    *bc = (d[16] << 3) | (d[17] << 2) | (d[18] << 1) | d[19];
    *psmi = (d[25] << 5) | (d[26] << 4) | (d[27] << 3) | (d[28] << 2) | (d[29] << 1) | d[30];
    return 0;
}

/* sync.c:150-167 */
static int cyclic_match(const signed char nd[NSYM], const unsigned char *d)
{
    for (int n = 0; n < NSYM; n++) {
        int i;
        for (i = 0; i < NSYM; i++) {
            if (nd[i] < 0) continue;
            if (nd[i] != d[(n + i) % NSYM]) break;
        }
        if (i == NSYM) return n;
    }
    return -1;
}

/* sync.c:188-207 */
static int locate_ref(orc_stream *s, unsigned ref, unsigned rsid)
{
    signed char nd[NSYM];
    unsigned char d[NSYM];
    ref_needle(nd, rsid);
    for (int n = 0; n < NSYM; n++) d[n] = crealf(s->bins[ref][n]) <= 0 ? 0 : 1;
    int m = cyclic_match(nd, d);
    if (m >= 0) return m;
    for (int n = 0; n < NSYM; n++) d[n] ^= 1;
    return cyclic_match(nd, d);
}

/* sync.c:292-337 */
static void coarse_cfo_search(orc_stream *s)
{
    for (int cfo = -2 * PW; cfo < 2 * PW; cfo++) {
        unsigned count[NSYM] = { 0 }, best_count = 0;
        int best = -1;
        for (int i = 0; i <= PM_PART; i++) {
            unsigned lo = cfo + LB0 + i * PW, hi = cfo + UB1 - i * PW, rsid = (30 - i) & 3;
            costas_ref(s, lo, cfo);
            int off = locate_ref(s, lo, rsid);
            uncostas_ref(s, lo);
            if (off >= 0) count[off]++;
            costas_ref(s, hi, cfo);
            off = locate_ref(s, hi, rsid);
            uncostas_ref(s, hi);
            if (off >= 0) count[off]++;
        }
        for (int off = 0; off < NSYM; off++)
            if (count[off] > best_count) { best = off; best_count = count[off]; }
        if (best >= 0 && best_count >= 3) {
            s->keep_extra = ((NSYM - best) % NSYM) * SYM_N;     /* acquire_keep_extra */
            s->cfo += cfo;                                       /* acquire_cfo_adjust */
            s->cfo_wait = 8;
            break;
        }
    }
}

/* sync.c:254-282 */
static void equalize_partition(orc_stream *s, unsigned lower, unsigned upper)
{
    float smag0 = 0, smag19 = 0;
    for (int n = 0; n < NSYM; n++) smag0 += fabsf(crealf(s->bins[lower][n]));
    smag0 = smag0 / NSYM;
    for (int n = 0; n < NSYM; n++) smag19 += fabsf(crealf(s->bins[upper][n]));
    smag19 = smag19 / NSYM;
    for (int n = 0; n < NSYM; n++) {
        float complex up = cexpf(s->phases[upper][n] * I);
        float complex lp = cexpf(s->phases[lower][n] * I);
        for (int k = 1; k < PW; k++) {
            float complex c = CMPLXF(PW, PW) / (k * smag19 * up + (PW - k) * smag0 * lp);
            s->bins[lower + k][n] *= c;
        }
    }
}

/* sync.c:284-290 */
static float half_turn_diff(float a, float b)
{
    float d = a - b;
    while (d > M_PI / 2) d -= M_PI;
    while (d < -M_PI / 2) d += M_PI;
    return d;
}

/* sync.c:69-73 */
static inline int8_t soft_bit(float x, float mult)
{
    float c = fmaxf(fminf(x, 1), -1);
    return lroundf(c * mult);
}

/* sync.c:339-610 (primary-main part) */
static void sync_block(orc_stream *s)
{
    int i, ppb;
    switch (COMPAT[s->psmi]) {
    case 2: ppb = 11; break;
    case 3: ppb = 12; break;
    case 5: case 6: case 11: ppb = 14; break;
    default: ppb = 10;
    }
    for (i = 0; i < ppb * PW + 1; i += PW) {
        costas_ref(s, LB0 + i, 0);
        costas_ref(s, UB1 - i, 0);
    }

    if (s->sync_state == ORC_SYNC_COARSE) {
        unsigned good = 0, seen_bc[16] = { 0 }, seen_psmi[64] = { 0 };
        for (i = 0; i <= ppb; i++) {
            unsigned bc, psmi;
            if (read_ref(s, LB0 + i * PW, (30 - i) & 3, &bc, &psmi) == 0) { good++; seen_bc[bc]++; seen_psmi[psmi]++; }
            if (read_ref(s, UB1 - i * PW, (30 - i) & 3, &bc, &psmi) == 0) { good++; seen_bc[bc]++; seen_psmi[psmi]++; }
        }
        if (good >= 4) {
            int maj_bc = -1, maj_psmi = -1;
            for (unsigned v = 0; v < 16; v++) if (seen_bc[v] > good / 2) maj_bc = v;
            for (unsigned v = 0; v < 16; v++) if (seen_psmi[v] > good / 2) maj_psmi = v;   /* 0..15 only, sync.c:396 */
            if (maj_bc >= 0 && maj_psmi >= 0) {
                s->bc = maj_bc; s->psmi = maj_psmi;
                set_sync_state(s, ORC_SYNC_FINE);
                s->started_pm = 0;                       /* decode_reset (decode.c:563-572) */
                for (int ch = 0; ch < 2; ch++) { s->px[ch].pos = 0; memset(s->px[ch].taken, 0, sizeof(s->px[ch].taken)); s->px[ch].started = 0; s->px[ch].ready = 0; }
            }
        } else if (s->cfo_wait == 0) {
            coarse_cfo_search(s);
        } else {
            s->cfo_wait--;
        }
    }

    if (s->sync_state != ORC_SYNC_FINE) return;

    float samperr = 0, angle = 0, sum_xy = 0, sum_x2 = 0;
    for (i = 0; i < ppb * PW; i += PW) {
        equalize_partition(s, LB0 + i, LB0 + i + PW);
        equalize_partition(s, UB1 - i - PW, UB1 - i);
        samperr += half_turn_diff(s->phases[LB0 + i][0], s->phases[LB0 + i + PW][0]);
        samperr += half_turn_diff(s->phases[UB1 - i - PW][0], s->phases[UB1 - i][0]);
    }
    samperr = samperr / (ppb * 2) * FFT_N / PW / (2 * M_PI);
    for (i = 0; i < ppb * PW + 1; i += PW) {
        float x, y;
        x = LB0 + i - (FFT_N / 2); y = s->costas_freq[LB0 + i];
        angle += y; sum_xy += x * y; sum_x2 += x * x;
        x = UB1 - i - (FFT_N / 2); y = s->costas_freq[UB1 - i];
        angle += y; sum_xy += x * y; sum_x2 += x * x;
    }
    samperr -= (sum_xy / sum_x2) * FFT_N / (2 * M_PI) * NSYM;
    s->samperr = roundf(samperr);
    angle /= (ppb + 1) * 2;
    s->angle = angle;
    for (i = 0; i < ppb * PW + 1; i += PW) {
        s->costas_freq[LB0 + i] -= angle;
        s->costas_freq[UB1 - i] -= angle;
    }

    float error_lb = 0, error_ub = 0;
    for (int n = 0; n < NSYM; n++)
        for (i = 0; i < ppb * PW; i += PW)
            for (int j = 1; j < PW; j++) {
                float complex c = s->bins[LB0 + i + j][n];
                float complex ideal = CMPLXF(crealf(c) >= 0 ? 1 : -1, cimagf(c) >= 0 ? 1 : -1);
                error_lb += norm2(ideal - c);
                c = s->bins[UB1 - i - PW + j][n];
                ideal = CMPLXF(crealf(c) >= 0 ? 1 : -1, cimagf(c) >= 0 ? 1 : -1);
                error_ub += norm2(ideal - c);
            }
    s->error_lb += error_lb;
    s->error_ub += error_ub;
    if (++s->mer_cnt == 16) {
        float signal = 2 * NSYM * (ppb * 18) * s->mer_cnt;
        float mer[2] = { 10 * log10f(signal / s->error_lb), 10 * log10f(signal / s->error_ub) };
        log_rec(s, ORC_REC_MER, mer, sizeof(mer));
        s->mer_cnt = 0; s->error_lb = 0; s->error_ub = 0;
    }
    const float mer_lb = 2.0f * NSYM * (float)(ppb * 18) / error_lb;
    const float mer_ub = 2.0f * NSYM * (float)(ppb * 18) / error_ub;
    const float mult_lb = fmaxf(fminf(mer_lb * 10, 127), 1);
    const float mult_ub = fmaxf(fminf(mer_ub * 10, 127), 1);

    int8_t soft[PM_BLOCK], px1[PX_MAX], px2[PX_MAX];
    int o = 0, o1 = 0, o2 = 0;
    const int compat = COMPAT[s->psmi];
    for (int n = 0; n < NSYM; n++) {
        for (i = LB0; i < LB0 + PM_PART * PW; i += PW)
            for (int j = 1; j < PW; j++) {
                float complex c = s->bins[i + j][n];
                soft[o++] = soft_bit(crealf(c), mult_lb);
                soft[o++] = soft_bit(cimagf(c), mult_lb);
            }
        for (i = UB1 - PM_PART * PW; i < UB1; i += PW)
            for (int j = 1; j < PW; j++) {
                float complex c = s->bins[i + j][n];
                soft[o++] = soft_bit(crealf(c), mult_ub);
                soft[o++] = soft_bit(cimagf(c), mult_ub);
            }
        /* extended partitions (sync.c:537-596): 1 per sideband in MP2, 2 in MP3/MP11 -> PX1; 2 more in MP11 -> PX2,
         * where BOTH sidebands use the lower sideband's gain (the reference's quirk, sync.c:591-592) */
        const int nx1 = compat == 2 ? 1 : (compat == 3 || compat == 11) ? 2 : 0, nx2 = compat == 11 ? 2 : 0;
        for (int q = 0; q < nx1; q++)
            for (int j = 1; j < PW; j++) {
                float complex c = s->bins[LB0 + (PM_PART + q) * PW + j][n];
                px1[o1++] = soft_bit(crealf(c), mult_lb); px1[o1++] = soft_bit(cimagf(c), mult_lb);
            }
        for (int q = 0; q < nx1; q++)
            for (int j = 1; j < PW; j++) {
                float complex c = s->bins[UB1 - (PM_PART + nx1 - q) * PW + j][n];
                px1[o1++] = soft_bit(crealf(c), mult_ub); px1[o1++] = soft_bit(cimagf(c), mult_ub);
            }
        for (int q = 0; q < nx2; q++)
            for (int j = 1; j < PW; j++) {
                float complex c = s->bins[LB0 + (PM_PART + 2 + q) * PW + j][n];
                px2[o2++] = soft_bit(crealf(c), mult_lb); px2[o2++] = soft_bit(cimagf(c), mult_lb);
            }
        for (int q = 0; q < nx2; q++)
            for (int j = 1; j < PW; j++) {
                float complex c = s->bins[UB1 - (PM_PART + 2 + nx2 - q) * PW + j][n];
                px2[o2++] = soft_bit(crealf(c), mult_lb); px2[o2++] = soft_bit(cimagf(c), mult_lb);
            }
    }
    if (s->taps & ORC_TAP_SOFT) {
        uint8_t *tmp = malloc(4 + PM_BLOCK);
        uint32_t b = s->bc; memcpy(tmp, &b, 4); memcpy(tmp + 4, soft, PM_BLOCK);
        log_rec(s, ORC_REC_SOFT, tmp, 4 + PM_BLOCK);
        free(tmp);
    }
    {
        const unsigned bc = s->bc;
        decode_block(s, soft, bc);
        if (o1 > 0) px_push(s, 0, px1, (unsigned)o1, bc);
        if (o2 > 0) px_push(s, 1, px2, (unsigned)o2, bc);
    }
    s->bc = (s->bc + 1) % 16;
}

/* sync.c:779-808 (FM): keep the 2 x 267 live bins of a symbol; a full block triggers sync */
static void push_symbol(orc_stream *s, const float complex *shifted)
{
    for (unsigned i = 0; i < 14 * PW + 1; i++) {
        s->bins[LB0 + i][s->sym_idx] = shifted[LB0 + i];
        s->bins[UB1 - i][s->sym_idx] = shifted[UB1 - i];
    }
    if (++s->sym_idx == NSYM) { s->sym_idx = 0; sync_block(s); }
}

/* ---- acquire (acquire.c:98-263, FM) ---------------------------------------------------- */
static void process_window(orc_stream *s)
{
    static __thread float complex buf[WIN_N];
    static __thread orc_c16 filt[WIN_N];
    float complex fftin[FFT_N], fftout[FFT_N], shifted[FFT_N], phase_inc;
    float angle, angle_diff, angle_factor;
    int samperr = 0;
    const unsigned state_before = s->sync_state;

    if (s->sync_state == ORC_SYNC_FINE) {
        samperr = SYM_N / 2 + s->samperr;  s->samperr = 0;
        angle_diff = -s->angle;            s->angle = 0;
        angle = s->prev_angle + angle_diff;
        s->prev_angle = angle;
    } else {
        float complex peak;
        orc_fir32_fm(s->fir_hist, s->ring, WIN_N, filt);
        for (int i = 0; i < WIN_N; i++) buf[i] = q15_to_cf_conj(filt[i]);
        cp_correlate(buf, &samperr, &peak);
        angle_diff = cargf(peak * cexpf(I * -s->prev_angle));
        angle_factor = (s->prev_angle) ? 0.25 : 1.0;
        angle = s->prev_angle + (angle_diff * angle_factor);
        s->prev_angle = angle;
        set_sync_state(s, ORC_SYNC_COARSE);
    }

    for (int i = 0; i < WIN_N; i++) buf[i] = q15_to_cf_conj(s->ring[i]);

    /* sync_adjust, sync.c:769-777 */
    const int adj = SYM_N / 2 - samperr;
    for (int i = 0; i < 14 * PW + 1; i++) {
        s->costas_phase[LB0 + i] -= adj * (LB0 + i - (FFT_N / 2)) * 2 * M_PI / FFT_N;
        s->costas_phase[UB1 - i] -= adj * (UB1 - i - (FFT_N / 2)) * 2 * M_PI / FFT_N;
    }
    angle -= 2 * M_PI * s->cfo;
    s->phase *= cexpf(-(SYM_N / 2 - samperr) * angle / FFT_N * I);
    phase_inc = cexpf(angle / FFT_N * I);

    for (int i = 0; i < NSYM; ++i) {
        for (int j = 0; j < SYM_N; ++j) {
            float complex sample = s->phase * buf[i * SYM_N + j + samperr];
            if (j < CP_N) fftin[j] = shape_fm[j] * sample;
            else if (j < FFT_N) fftin[j] = sample;
            else fftin[j - FFT_N] += shape_fm[j] * sample;
            s->phase *= phase_inc;
        }
        s->phase /= cabsf(s->phase);
        oracle_fft_forward(FFT_N, (const float *)fftin, (float *)fftout);
        memcpy(shifted, fftout + FFT_N / 2, sizeof(float complex) * FFT_N / 2);     /* fftshift */
        memcpy(shifted + FFT_N / 2, fftout, sizeof(float complex) * FFT_N / 2);
        if ((s->taps & ORC_TAP_FFT) && s->fft_syms < s->fft_limit * NSYM) {
            gb_put(&s->fft, shifted, sizeof(shifted)); s->fft_syms++;
        }
        push_symbol(s, shifted);
    }

    int keep = SYM_N + (SYM_N / 2 - samperr) + s->keep_extra;
    s->keep_extra = 0;
    memmove(s->ring, s->ring + WIN_N - keep, sizeof(orc_c16) * keep);
    s->ring_fill = keep;

    struct { int32_t state_before, state_after, samperr, cfo, keep, bc, psmi, cfo_wait, next_samperr;
             float prev_angle, phase_re, phase_im, next_angle; } r = {
        (int32_t)state_before, (int32_t)s->sync_state, samperr, s->cfo, keep, (int32_t)s->bc, s->psmi, s->cfo_wait,
        s->samperr, s->prev_angle, crealf(s->phase), cimagf(s->phase), s->angle };
    log_rec(s, ORC_REC_BLOCK, &r, sizeof(r));
}

/* input.c:41-50 + acquire.c:275-288 */
static void feed_q15(orc_stream *s, const orc_c16 *x, size_t n)
{
    if (s->taps & ORC_TAP_Q15) gb_put(&s->q15, x, sizeof(orc_c16) * n);
    while (n) {
        size_t take = WIN_N - s->ring_fill;
        if (take > n) take = n;
        memcpy(s->ring + s->ring_fill, x, sizeof(orc_c16) * take);
        s->ring_fill += take; x += take; n -= take;
        if (s->ring_fill == WIN_N) process_window(s);
    }
}

void orc_push_cu8(orc_stream *s, const uint8_t *iq, uint32_t nbytes)
{
    orc_c16 out[4096];
    while (nbytes) {
        uint32_t take = nbytes > 4 * 4096 ? 4 * 4096 : nbytes;
        size_t n = orc_halfband_fm_cu8(s->hb_hist, iq, take, out);
        feed_q15(s, out, n);
        iq += take; nbytes -= take;
    }
}

void orc_push_cs16(orc_stream *s, const int16_t *iq, uint32_t n)
{
    feed_q15(s, (const orc_c16 *)iq, n / 2);
}

void orc_reset(orc_stream *s)
{
    memset(s->hb_hist, 0, sizeof(s->hb_hist));
    memset(s->fir_hist, 0, sizeof(s->fir_hist));
    s->ring_fill = 0;
    set_sync_state(s, ORC_SYNC_NONE);
    s->prev_angle = 0; s->phase = 1; s->keep_extra = 0; s->cfo = 0;        /* acquire_reset */
    memset(s->costas_freq, 0, sizeof(s->costas_freq));                      /* sync_reset */
    memset(s->costas_phase, 0, sizeof(s->costas_phase));
    s->sym_idx = 0; s->psmi = 1; s->cfo_wait = 0; s->mer_cnt = 0; s->error_lb = 0; s->error_ub = 0;
    s->started_pm = 0;                                                      /* decode_reset */
    for (int ch = 0; ch < 2; ch++) { s->px[ch].pos = 0; memset(s->px[ch].taken, 0, sizeof(s->px[ch].taken)); s->px[ch].started = 0; s->px[ch].ready = 0; }
}

orc_stream *orc_open(void)
{
    build_tables();
    orc_stream *s = calloc(1, sizeof(*s));
    s->bins = calloc(FFT_N, sizeof(*s->bins));
    s->phases = calloc(FFT_N, sizeof(*s->phases));
    /* loop gains, sync.c:832-841 */
    float loop_bw = 0.05, damping = 0.70710678;
    float denom = 1 + (2 * damping * loop_bw) + (loop_bw * loop_bw);
    s->alpha = (4 * damping * loop_bw) / denom;
    s->beta = (4 * loop_bw * loop_bw) / denom;
    s->sync_state = ORC_SYNC_NONE;
    orc_reset(s);
    return s;
}

void orc_close(orc_stream *s)
{
    if (!s) return;
    free(s->bins); free(s->phases); free(s->log.p); free(s->q15.p); free(s->fft.p); free(s);
}

void orc_set_taps(orc_stream *s, unsigned mask, unsigned fft_limit_blocks) { s->taps = mask; s->fft_limit = fft_limit_blocks; }
void orc_set_p1_hook(orc_stream *s, orc_p1_hook hook, void *user) { s->hook = hook; s->hook_user = user; }
size_t orc_buf(orc_stream *s, int which, const uint8_t **p)
{
    gbuf *b = which == 0 ? &s->log : which == 1 ? &s->q15 : &s->fft;
    *p = b->p; return b->len;
}
void orc_clear_bufs(orc_stream *s) { s->log.len = s->q15.len = s->fft.len = 0; s->fft_syms = 0; }

void orc_snapshot(const orc_stream *s, orc_sync_snapshot *o)
{
    o->sync_state = s->sync_state; o->bc = s->bc; o->psmi = s->psmi; o->cfo_wait = s->cfo_wait;
    o->samperr_next = s->samperr; o->mer_cnt = s->mer_cnt; o->acq_cfo = s->cfo; o->keep_extra = s->keep_extra;
    o->angle_next = s->angle; o->prev_angle = s->prev_angle;
    o->phase_re = crealf(s->phase); o->phase_im = cimagf(s->phase);
    o->error_lb = s->error_lb; o->error_ub = s->error_ub;
    for (int i = 0; i < 15; i++) {
        o->costas_freq[2 * i] = s->costas_freq[LB0 + i * PW]; o->costas_freq[2 * i + 1] = s->costas_freq[UB1 - i * PW];
        o->costas_phase[2 * i] = s->costas_phase[LB0 + i * PW]; o->costas_phase[2 * i + 1] = s->costas_phase[UB1 - i * PW];
    }
}
