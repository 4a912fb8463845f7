/* TEST INFRASTRUCTURE -- see nrsc5_oracle.h.  Plain-C restatement of the reference's AM (hybrid MA1 /
 * all-digital MA3) path: cu8 -> 5-stage /32 decimator (or cs16 straight in) -> acquire (256-point OFDM,
 * carrier-phase regression) -> sync_process_am (training-cell equalisation, hard QAM decisions) ->
 * bit de-interleave with the 3-frame diversity delay -> K=9 tail-biting Viterbi -> descrambled
 * P1 / P3 / PIDS frames.  One function per pipeline stage (the cut the HIP kernels use).  Floating-point
 * expressions keep the reference's operand types and evaluation order so that, built with the same
 * compiler flags, the float trace is reproducible against oracle/_ref. */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "nrsc5_oracle.h"
#include "cpu_fft.h"

#define FFT_A   256
#define CP_A    14
#define SYM_A   (FFT_A + CP_A)            /* 270 */
#define NSYM    32
#define WIN_A   (SYM_A * (NSYM + 1))      /* 8910 */
#define C_A     (FFT_A / 2)               /* carrier bin after fftshift */
#define PW_A    25                        /* carriers per partition */
#define IDX_REF 1
#define IDX_PIDS_IN 27
#define IDX_PIDS_OUT 53
#define IDX_INNER 2
#define IDX_MIDDLE 28
#define IDX_OUTER 57
#define IDX_MAX 81
#define MA3     2                         /* SERVICE_MODE_MA3, defines.h:39 */
#define P1_LEN_A 3750
#define P3_LEN_MA1 24000
#define P3_LEN_MA3 30000
#define PIDS_LEN 80
#define DIV_DELAY (18000 * 3)             /* decode.h:7 */
#define FS_AM   46511.71875

/* acquisition band-select FIR for AM, acquire.c:63-96 (passes the primary sidebands, removes the carrier) */
static const float ACQ_TAPS_AM[32] = {
    -0.00038464731187559664, -0.00021618751634377986, 0.0026779419276863337, -0.00029802651260979474,
    -0.0012626448879018426, -0.0013182522961869836, -0.012252614833414555, 0.015980124473571777,
    0.037112727761268616, -0.05451361835002899, -0.05804193392395973, 0.11320608854293823,
    0.055298302322626114, -0.16878043115139008, -0.022917453199625015, 0.19178225100040436,
    -0.022917453199625015, -0.16878043115139008, 0.055298302322626114, 0.11320608854293823,
    -0.05804193392395973, -0.05451361835002899, 0.037112727761268616, 0.015980124473571777,
    -0.012252614833414555, -0.0013182522961869836, -0.0012626448879018426, -0.00029802651260979474,
    0.0026779419276863337, -0.00021618751634377986, -0.00038464731187559664, 0 };
static const float HB_TAPS[4] = { 0.6062333583831787, -0.13481467962265015, 0.032919470220804214, -0.00410953676328063 };
/* decode.c:26-32, 63-64: positions inside each 12- / 6- / 24-bit group of the convolutional code word */
static const int BL_POS[3] = { 2, 1, 5 }, ML_POS[3] = { 11, 6, 7 }, BU_POS[3] = { 10, 8, 9 }, MU_POS[3] = { 4, 3, 0 };
static const int EL_POS[2] = { 0, 1 }, EU_POS[4] = { 2, 3, 5, 4 };
static const int PIDS_IL_POS[12] = { 0, 1, 12, 13, 6, 5, 18, 17, 11, 7, 23, 19 };
static const int PIDS_IU_POS[12] = { 2, 4, 14, 16, 3, 8, 15, 20, 9, 10, 21, 22 };
static const unsigned GEN_E1[3] = { 0561, 0657, 0711 };       /* decode.c:47-53 */
static const unsigned GEN_E2[3] = { 0561, 0753, 0711 };       /* decode.c:55-61 */
static const uint8_t PUNCT_E1[15] = { 1, 0, 1, 1, 0, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1 };
static const uint8_t PUNCT_E2[6] = { 1, 0, 1, 1, 0, 0 };

static int16_t hb_q15[4], acq_q15[17];
static float shape_am[SYM_A];
static int tables_ready;

static void build_tables(void)
{
    if (tables_ready) return;
    for (int i = 0; i < 4; i++) hb_q15[i] = (int16_t)(HB_TAPS[3 - i] * 32767.0f);          /* firdecim_q15.c:37-42 */
    for (int i = 1; i <= 16; i++) acq_q15[i] = (int16_t)(ACQ_TAPS_AM[31 - i] * 32767.0f);
    for (int i = 0; i < SYM_A; i++) {                                                       /* acquire.c:333-342 */
        if (i < CP_A) shape_am[i] = sinf(M_PI / 2 * i / CP_A);
        else if (i < FFT_A) shape_am[i] = 1;
        else shape_am[i] = cosf(M_PI / 2 * (i - FFT_A) / CP_A);
    }
    tables_ready = 1;
}

/* ------------------------------------------------------------------------------------ */
/* K1-AM: cu8 -> Q15 >> 4 -> five cascaded 2:1 half-bands (input.c:52-94)                  */

struct orc_am_decim {
    int16_t wr[5][16], wi[5][16];       /* per stage: 14 samples of history + the pair being assembled */
    unsigned have[5];                   /* samples of the current pair already present (0 or 1) */
};

static inline int16_t hb_dot(const int16_t a[15])
{
    int16_t acc = 0;
    for (int i = 0; i < 4; i++)
        acc = (int16_t)(acc + (((a[2 * i] + a[14 - 2 * i]) * hb_q15[i]) >> 15));
    return (int16_t)(acc + a[7]);
}

/* feed one sample to stage k; returns 1 and the stage's output when the pair is complete */
static int hb_stage(struct orc_am_decim *d, int k, orc_c16 x, orc_c16 *y)
{
    d->wr[k][14 + d->have[k]] = x.r; d->wi[k][14 + d->have[k]] = x.i;
    if (++d->have[k] < 2) return 0;
    y->r = hb_dot(d->wr[k]); y->i = hb_dot(d->wi[k]);
    memmove(d->wr[k], d->wr[k] + 2, sizeof(int16_t) * 14);
    memmove(d->wi[k], d->wi[k] + 2, sizeof(int16_t) * 14);
    d->have[k] = 0;
    return 1;
}

orc_am_decim *orc_am_decim_new(void) { build_tables(); return calloc(1, sizeof(struct orc_am_decim)); }
void orc_am_decim_free(orc_am_decim *d) { free(d); }

size_t orc_am_decimate_cu8(orc_am_decim *d, const uint8_t *iq, size_t nbytes, orc_c16 *out)
{
    size_t n = 0;
    for (size_t i = 0; i + 1 < nbytes; i += 2) {
        orc_c16 x = { (int16_t)((((int16_t)iq[i] - 127) * 64) >> 4), (int16_t)((((int16_t)iq[i + 1] - 127) * 64) >> 4) };
        for (int k = 0; k < 5; k++) {
            orc_c16 y;
            if (!hb_stage(d, k, x, &y)) break;
            if (k == 4) out[n++] = y;
            x = y;
        }
    }
    return n;
}

/* ------------------------------------------------------------------------------------ */
/* K6-AM: bit de-interleavers                                                              */

/* decode.c:66-71: bit p of the cell that holds index k of block b */
static inline int cell_bit(const uint8_t *m, int b, int k, int p)
{
    const int col = (9 * k) % 25;
    const int row = (11 * col + 16 * (k / 25) + 11 * (k / 50)) % 32;
    return (m[PW_A * (b * NSYM + row) + col] >> p) & 1;
}

/* interleaver_ma1 (decode.c:74-231).  pl/pu/s/t: 8 blocks x 32 x 25 hard symbols of one L1 frame.
 * ml_q / mu_q (and eml_q / emu_q in MA3): the 54000-bit diversity-delay lines, updated in place.
 * vit_p1: 8 x 11250 depunctured +-1/0 inputs; vit_p3: 72000 (MA1) or 90000 (MA3). */
void orc_am_deinterleave(int psmi, const uint8_t *pl, const uint8_t *pu, const uint8_t *s, const uint8_t *t,
                         uint8_t *ml_q, uint8_t *mu_q, uint8_t *eml_q, uint8_t *emu_q, int8_t *vit_p1, int8_t *vit_p3)
{
    static __thread uint8_t bl[18000], bu[18000], ml[18000], mu[18000], p1[72000], p3[72000];
    static __thread uint8_t el[12000], eu[24000], ebl[18000], ebu[18000], eml[18000], emu[18000];
    for (int n = 0; n < 18000; n++) {
        bl[n] = cell_bit(pl, n / 2250, (n + n / 750 + 1) % 750, n % 3);
        ml[n] = cell_bit(pl, (3 * n + 3) % 8, (n + n / 3000 + 3) % 750, 3 + (n % 3));
        bu[n] = cell_bit(pu, n / 2250, (n + n / 750) % 750, n % 3);
        mu[n] = cell_bit(pu, (3 * n) % 8, (n + n / 3000 + 2) % 750, 3 + (n % 3));
    }
    if (psmi != MA3) {
        for (int n = 0; n < 12000; n++) el[n] = cell_bit(t, (3 * n + n / 3000) % 8, (n + n / 6000) % 750, n % 2);
        for (int n = 0; n < 24000; n++) eu[n] = cell_bit(s, (3 * n + n / 3000 + 2 * (n / 12000)) % 8, (n + n / 6000) % 750, n % 4);
    } else {
        for (int n = 0; n < 18000; n++) {
            ebl[n] = cell_bit(t, (3 * n + 3) % 8, (n + n / 3000 + 3) % 750, n % 3);
            eml[n] = cell_bit(t, (3 * n + 3) % 8, (n + n / 3000 + 3) % 750, 3 + (n % 3));
            ebu[n] = cell_bit(s, (3 * n) % 8, (n + n / 3000 + 2) % 750, n % 3);
            emu[n] = cell_bit(s, (3 * n) % 8, (n + n / 3000 + 2) % 750, 3 + (n % 3));
        }
    }
    /* the main (m*) bits enter a 3-frame delay line: what leaves it now pairs with this frame's backup bits */
    for (int i = 0; i < 6000; i++) {
        for (int j = 0; j < 3; j++) {
            p1[i * 12 + BL_POS[j]] = bl[i * 3 + j];
            p1[i * 12 + ML_POS[j]] = ml_q[i * 3 + j];
            p1[i * 12 + BU_POS[j]] = bu[i * 3 + j];
            p1[i * 12 + MU_POS[j]] = mu_q[i * 3 + j];
        }
        if (psmi != MA3) {
            for (int j = 0; j < 2; j++) p3[i * 6 + EL_POS[j]] = el[i * 2 + j];
            for (int j = 0; j < 4; j++) p3[i * 6 + EU_POS[j]] = eu[i * 4 + j];
        } else {
            for (int j = 0; j < 3; j++) {
                p3[i * 12 + BL_POS[j]] = ebl[i * 3 + j];
                p3[i * 12 + ML_POS[j]] = eml_q[i * 3 + j];
                p3[i * 12 + BU_POS[j]] = ebu[i * 3 + j];
                p3[i * 12 + MU_POS[j]] = emu_q[i * 3 + j];
            }
        }
    }
    memmove(ml_q, ml_q + 18000, DIV_DELAY - 18000); memcpy(ml_q + DIV_DELAY - 18000, ml, 18000);
    memmove(mu_q, mu_q + 18000, DIV_DELAY - 18000); memcpy(mu_q + DIV_DELAY - 18000, mu, 18000);
    if (psmi == MA3) {
        memmove(eml_q, eml_q + 18000, DIV_DELAY - 18000); memcpy(eml_q + DIV_DELAY - 18000, eml, 18000);
        memmove(emu_q, emu_q + 18000, DIV_DELAY - 18000); memcpy(emu_q + DIV_DELAY - 18000, emu, 18000);
    }
    int o = 0;
    for (int i = 0; i < 8 * P1_LEN_A * 3; i++) vit_p1[i] = PUNCT_E1[i % 15] ? (p1[o++] ? 1 : -1) : 0;
    o = 0;
    if (psmi != MA3) for (int i = 0; i < P3_LEN_MA1 * 3; i++) vit_p3[i] = PUNCT_E2[i % 6] ? (p3[o++] ? 1 : -1) : 0;
    else for (int i = 0; i < P3_LEN_MA3 * 3; i++) vit_p3[i] = PUNCT_E1[i % 15] ? (p3[o++] ? 1 : -1) : 0;
}

/* decode_process_pids_am's bit gather (decode.c:476-501): sym[2n] / sym[2n+1] = the two PIDS carriers' QAM16 symbols */
void orc_am_deinterleave_pids(const uint8_t sym[64], int pids1_disabled, int8_t out[240])
{
    uint8_t il[120], iu[120];
    for (int n = 0; n < 120; n++) {
        int p = n % 4, k = (n + (n / 60) + 11) % 30, row = (11 * (k + (k / 15)) + 3) % 32;
        il[n] = (sym[row * 2] >> p) & 1;
        k = (n + (n / 60)) % 30; row = (11 * (k + (k / 15)) + 3) % 32;
        iu[n] = (sym[row * 2 + 1] >> p) & 1;
    }
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 12; j++) {
            out[i * 24 + PIDS_IL_POS[j]] = pids1_disabled ? 0 : (il[i * 12 + j] ? 1 : -1);
            out[i * 24 + PIDS_IU_POS[j]] = iu[i * 12 + j] ? 1 : -1;
        }
}

/* decode.c:234-261, any constraint length / puncture pattern */
int orc_bit_errors(const int8_t *coded, const uint8_t *decoded, int k, int len, const unsigned gens[3],
                   const uint8_t *puncture, int plen)
{
    unsigned r = 0;
    int errors = 0;
    for (int i = 0; i < k - 1; i++) r = (r >> 1) | ((unsigned)decoded[len - (k - 1) + i] << (k - 1));
    for (int i = 0, j = 0; i < len; i++, j += 3) {
        r = (r >> 1) | ((unsigned)decoded[i] << (k - 1));
        for (int q = 0; q < 3; q++)
            if (puncture[(j + q) % plen] && ((coded[j + q] > 0) != __builtin_parity(r & gens[q]))) errors++;
    }
    return errors;
}

/* ------------------------------------------------------------------------------------ */
/* hard-decision slicers, sync.c:37-88                                                    */

static uint8_t slice4(float f) { return f < -1 ? 0 : f < 0 ? 2 : f < 1 ? 3 : 1; }
static uint8_t slice8(float f) { return f < -3 ? 0 : f < -2 ? 4 : f < -1 ? 6 : f < 0 ? 2 : f < 1 ? 3 : f < 2 ? 7 : f < 3 ? 5 : 1; }
static uint8_t sym_qpsk(float complex c) { return (crealf(c) < 0 ? 0 : 1) | (cimagf(c) < 0 ? 0 : 2); }
static uint8_t sym_qam16(float complex c) { return slice4(crealf(c)) | (slice4(cimagf(c)) << 2); }
static uint8_t sym_qam64(float complex c) { return slice8(crealf(c)) | (slice8(cimagf(c)) << 3); }

/* ------------------------------------------------------------------------------------ */
/* stream object                                                                          */

typedef struct { uint8_t *p; size_t len, cap; } gbuf;

struct orc_am_stream {
    orc_am_decim *decim;
    orc_c16 fir_hist[31];
    orc_c16 ring[WIN_A];
    unsigned ring_fill;
    unsigned sync_state;
    float prev_angle;
    float complex phase;
    int keep_extra, cfo;
    float complex bins[FFT_A][NSYM];
    unsigned sym_idx;
    int psmi, pli, hppi, aabi, rdbi, cfo_wait;
    unsigned offset_history, bc;
    int samperr;
    uint8_t sym_pl[8 * 800], sym_pu[8 * 800], sym_s[8 * 800], sym_t[8 * 800];
    uint8_t ml_q[DIV_DELAY], mu_q[DIV_DELAY], eml_q[DIV_DELAY], emu_q[DIV_DELAY];
    int8_t vit_p1[8 * P1_LEN_A * 3], vit_p3[P3_LEN_MA3 * 3];
    unsigned am_errors, am_diversity_wait;
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
static void log_rec(orc_am_stream *s, uint32_t kind, const void *payload, uint32_t n)
{
    uint32_t hdr[2] = { kind, n }, z = 0, pad = (4 - (n & 3)) & 3;
    gb_put(&s->log, hdr, sizeof(hdr));
    if (n) gb_put(&s->log, payload, n);
    if (pad) gb_put(&s->log, &z, pad);
}

/* input.c:172-188 */
static void set_sync_state(orc_am_stream *s, unsigned new_state)
{
    if (s->sync_state == new_state) return;
    int32_t r[2] = { (int32_t)s->sync_state, (int32_t)new_state };
    log_rec(s, ORC_REC_STATE, r, sizeof(r));
    if (s->sync_state == ORC_SYNC_FINE) log_rec(s, ORC_REC_LOST_SYNC, NULL, 0);
    if (new_state == ORC_SYNC_FINE) {
        float freq_offset = (s->prev_angle - 2 * M_PI * s->cfo) * FS_AM / (2 * M_PI * FFT_A);
        struct { float f; int32_t psmi, pli, hppi, aabi, rdbi; } ev = { freq_offset, s->psmi, s->pli, s->hppi, s->aabi, s->rdbi };
        log_rec(s, ORC_REC_SYNC, &ev, sizeof(ev));
    }
    s->sync_state = new_state;
}
void orc_am_force_resync(orc_am_stream *s) { set_sync_state(s, ORC_SYNC_NONE); }

static void emit_frame(orc_am_stream *s, const uint8_t *bits, uint32_t len, uint32_t lc)
{
    uint8_t *tmp = malloc(8 + len);
    uint32_t h[2] = { lc, len };
    memcpy(tmp, h, 8); memcpy(tmp + 8, bits, len);
    log_rec(s, ORC_REC_FRAME, tmp, 8 + len);
    free(tmp);
}

/* ---- decode side ------------------------------------------------------------------------ */

/* decode_process_pids_am, decode.c:474-505 */
static void decode_pids(orc_am_stream *s, const uint8_t sym[64])
{
    int8_t coded[PIDS_LEN * 3];
    uint8_t bits[PIDS_LEN];
    orc_am_deinterleave_pids(sym, (s->psmi == 1) && s->rdbi, coded);
    orc_viterbi(coded, PIDS_LEN, 9, GEN_E2, bits);
    orc_descramble(bits, PIDS_LEN);
    log_rec(s, ORC_REC_PIDS, bits, PIDS_LEN);
}

/* decode_push_pl_pu_s_t + decode_process_p1_p3_am, decode.c:439-449, 507-554 */
static void decode_block(orc_am_stream *s, const uint8_t *pl, const uint8_t *pu, const uint8_t *sy, const uint8_t *t, unsigned bc)
{
    static __thread uint8_t bits[P3_LEN_MA3];
    memcpy(s->sym_pl + bc * 800, pl, 800); memcpy(s->sym_pu + bc * 800, pu, 800);
    memcpy(s->sym_s + bc * 800, sy, 800); memcpy(s->sym_t + bc * 800, t, 800);
    if (s->taps & ORC_TAP_SOFT) {
        uint8_t tmp[4 + 3200];
        uint32_t b = bc; memcpy(tmp, &b, 4);
        memcpy(tmp + 4, pl, 800); memcpy(tmp + 804, pu, 800); memcpy(tmp + 1604, sy, 800); memcpy(tmp + 2404, t, 800);
        log_rec(s, ORC_REC_AMSYM, tmp, sizeof(tmp));
    }

    if (bc == 0) s->am_errors = 0;
    if (s->am_diversity_wait == 0) {
        const int8_t *in = s->vit_p1 + bc * P1_LEN_A * 3;
        orc_viterbi(in, P1_LEN_A, 9, GEN_E1, bits);
        s->am_errors += orc_bit_errors(in, bits, 9, P1_LEN_A, GEN_E1, PUNCT_E1, 15);
        orc_descramble(bits, P1_LEN_A);
        emit_frame(s, bits, P1_LEN_A, 0);
        if (s->hook && s->hook(s->hook_user, bits, P1_LEN_A)) set_sync_state(s, ORC_SYNC_NONE);    /* frame.c:535-540 */
        if (bc == 7) {
            unsigned total = 8 * (P1_LEN_A * 12 / 5);
            if (!s->rdbi) {
                if (s->psmi != MA3) {
                    total += P3_LEN_MA1 * 3 / 2;
                    orc_viterbi(s->vit_p3, P3_LEN_MA1, 9, GEN_E2, bits);
                    s->am_errors += orc_bit_errors(s->vit_p3, bits, 9, P3_LEN_MA1, GEN_E2, PUNCT_E2, 6);
                    orc_descramble(bits, P3_LEN_MA1);
                    emit_frame(s, bits, P3_LEN_MA1, 1);
                } else {
                    total += P3_LEN_MA3 * 12 / 5;
                    orc_viterbi(s->vit_p3, P3_LEN_MA3, 9, GEN_E1, bits);
                    s->am_errors += orc_bit_errors(s->vit_p3, bits, 9, P3_LEN_MA3, GEN_E1, PUNCT_E1, 15);
                    orc_descramble(bits, P3_LEN_MA3);
                    emit_frame(s, bits, P3_LEN_MA3, 1);
                }
            }
            float cber = (float)s->am_errors / (float)total;
            log_rec(s, ORC_REC_BER, &cber, sizeof(cber));
        }
    }
    if (bc == 7) {
        orc_am_deinterleave(s->psmi, s->sym_pl, s->sym_pu, s->sym_s, s->sym_t, s->ml_q, s->mu_q, s->eml_q, s->emu_q,
                            s->vit_p1, s->vit_p3);
        if (s->am_diversity_wait > 0) s->am_diversity_wait--;
    }
}

/* ---- sync side (sync.c:209-252, 612-767) -------------------------------------------------- */

static const signed char REF_PATTERN_AM[NSYM] = {
    0, 1, 1, 0, 0, 1, 0, -1, -1, 1, -1, -1, -1, -1, 0, -1, -1, -1, -1, -1, -1, 1, 1, -1, -1, -1, -1, -1, -1, -1, -1, -1 };

/* find_ref_am: cyclic position of the fixed part of the reference sequence (first 23 entries) */
static int locate_ref(const unsigned char *d)
{
    for (int n = 0; n < NSYM; n++) {
        int i;
        for (i = 0; i < 23; i++) {
            if (REF_PATTERN_AM[i] < 0) continue;
            if (REF_PATTERN_AM[i] != d[(n + i) % NSYM]) break;
        }
        if (i == 23) return n;
    }
    return -1;
}

/* find_block_am: validate the aligned reference sequence, return its block count, latch the system bits at bc 0 */
static int read_ref(orc_am_stream *s, const unsigned char *d)
{
    for (int n = 0; n < NSYM; n++)
        if (REF_PATTERN_AM[n] >= 0 && d[n] != REF_PATTERN_AM[n]) return -1;
    if (d[7] ^ d[8]) return -1;
    if (d[10] ^ d[11] ^ d[12] ^ d[13]) return -1;
    if (d[15] ^ d[16] ^ d[17] ^ d[18] ^ d[19] ^ d[20]) return -1;
    if (d[23] ^ d[24] ^ d[25] ^ d[26] ^ d[27] ^ d[28] ^ d[29] ^ d[30] ^ d[31]) return -1;
    int bc = (d[17] << 2) | (d[18] << 1) | d[19];
    if (bc == 0) {
        s->psmi = (d[26] << 4) | (d[27] << 3) | (d[28] << 2) | (d[29] << 1) | d[30];
        s->pli = d[7]; s->hppi = d[11]; s->aabi = d[12]; s->rdbi = d[15];
    }
    return bc;
}

static float half_turn_diff(float a, float b)   /* sync.c:284-290 */
{
    float d = a - b;
    while (d > M_PI / 2) d -= M_PI;
    while (d < -M_PI / 2) d += M_PI;
    return d;
}

static void sync_block(orc_am_stream *s)
{
    float complex (*z)[NSYM] = s->bins;
    for (int i = IDX_REF; i <= IDX_MAX; i++)
        for (int n = 0; n < NSYM; n++) z[C_A - i][n] = -conjf(z[C_A - i][n]);
    if (s->psmi != MA3)                                    /* complementary sidebands add coherently */
        for (int i = IDX_REF; i <= IDX_PIDS_OUT; i++)
            for (int n = 0; n < NSYM; n++) z[C_A + i][n] += z[C_A - i][n];

    unsigned char d[NSYM];
    for (int n = 0; n < NSYM; n++) d[n] = cimagf(z[C_A + IDX_REF][n]) <= 0 ? 0 : 1;

    if (s->sync_state == ORC_SYNC_COARSE && s->cfo_wait == 0) {
        int off = locate_ref(d);
        if (off > 0) { s->keep_extra = ((NSYM - off) % NSYM) * SYM_A; s->cfo_wait = 8; }
    } else {
        s->cfo_wait--;
    }

    if (s->sync_state == ORC_SYNC_COARSE) {
        int bc = read_ref(s, d);
        if (bc == -1) s->offset_history = 0;
        else s->offset_history = (s->offset_history << 4) | bc;
        if ((s->offset_history & 0xffff) == 0x5670) {
            s->bc = 0;
            set_sync_state(s, ORC_SYNC_FINE);
            s->am_errors = 0; s->am_diversity_wait = 4;    /* decode_reset, decode.c:563-572 */
            s->offset_history = 0;
        }
    }
    if (s->sync_state != ORC_SYNC_FINE) return;

    const int ma3 = s->psmi == MA3;
    const int pids1 = ma3 ? -IDX_PIDS_IN : IDX_PIDS_IN, pids2 = ma3 ? IDX_PIDS_IN : IDX_PIDS_OUT;
    const float complex pids1_mult = 2 * CMPLXF(1.5, -0.5) / (z[C_A + pids1][8] + z[C_A + pids1][24]);
    const float complex pids2_mult = 2 * CMPLXF(1.5, -0.5) / (z[C_A + pids2][8] + z[C_A + pids2][24]);
    uint8_t pids[2 * NSYM];
    for (int n = 0; n < NSYM; n++) {
        z[C_A + pids1][n] *= pids1_mult; pids[2 * n] = sym_qam16(z[C_A + pids1][n]);
        z[C_A + pids2][n] *= pids2_mult; pids[2 * n + 1] = sym_qam16(z[C_A + pids2][n]);
    }
    decode_pids(s, pids);

    float complex pl_mult[PW_A], pu_mult[PW_A], s_mult[PW_A], t_mult[PW_A];
    const int pri = ma3 ? IDX_INNER : IDX_OUTER, sec = IDX_MIDDLE, ter = ma3 ? IDX_MIDDLE : IDX_INNER;
    float samperr = 0;
    for (int col = 0; col < PW_A; col++) {
        const int t1 = (5 + 11 * col) % 32, t2 = (21 + 11 * col) % 32;     /* training cells of this carrier */
        pl_mult[col] = 2 * CMPLXF(2.5, -2.5) / (z[C_A - pri - col][t1] + z[C_A - pri - col][t2]);
        pu_mult[col] = 2 * CMPLXF(2.5, -2.5) / (z[C_A + pri + col][t1] + z[C_A + pri + col][t2]);
        if (!ma3) {
            s_mult[col] = 2 * CMPLXF(1.5, -0.5) / (z[C_A + sec + col][t1] + z[C_A + sec + col][t2]);
            t_mult[col] = 2 * CMPLXF(-0.5, 0.5) / (z[C_A + ter + col][t1] + z[C_A + ter + col][t2]);
        } else {
            s_mult[col] = 2 * CMPLXF(2.5, -2.5) / (z[C_A + sec + col][t1] + z[C_A + sec + col][t2]);
            t_mult[col] = 2 * CMPLXF(2.5, -2.5) / (z[C_A - ter - col][t1] + z[C_A - ter - col][t2]);
        }
        if (col > 0) {
            samperr += half_turn_diff(cargf(pl_mult[col]), cargf(pl_mult[col - 1]));
            samperr += half_turn_diff(cargf(pu_mult[col]), cargf(pu_mult[col - 1]));
        }
    }
    samperr = samperr / (2 * (PW_A - 1)) * FFT_A / (2 * M_PI);
    s->samperr = roundf(samperr);

    uint8_t pl[800], pu[800], sy[800], t[800];
    for (int n = 0; n < NSYM; n++)
        for (int col = 0; col < PW_A; col++) {
            const int tb = ma3 ? C_A - ter - col : C_A + ter + col;
            z[C_A - pri - col][n] *= pl_mult[col];
            z[C_A + pri + col][n] *= pu_mult[col];
            z[C_A + sec + col][n] *= s_mult[col];
            z[tb][n] *= t_mult[col];
            pl[n * PW_A + col] = sym_qam64(z[C_A - pri - col][n]);
            pu[n * PW_A + col] = sym_qam64(z[C_A + pri + col][n]);
            sy[n * PW_A + col] = ma3 ? sym_qam64(z[C_A + sec + col][n]) : sym_qam16(z[C_A + sec + col][n]);
            t[n * PW_A + col] = ma3 ? sym_qam64(z[tb][n]) : sym_qpsk(z[tb][n]);
        }
    decode_block(s, pl, pu, sy, t, s->bc);
    s->bc = (s->bc + 1) % 8;
}

/* sync_push, sync.c:791-797 */
static void push_symbol(orc_am_stream *s, const float complex *shifted)
{
    for (int i = C_A - IDX_MAX; i <= C_A + IDX_MAX; i++) s->bins[i][s->sym_idx] = shifted[i];
    if (++s->sym_idx == NSYM) { s->sym_idx = 0; sync_block(s); }
}

/* ---- acquire (acquire.c:98-263, AM branches) ------------------------------------------------ */

static inline float complex q15_to_cf(orc_c16 v) { return CMPLXF((float)v.r / 32767.0f, (float)v.i / 32767.0f); }   /* defines.h:106 */
static inline float norm2(float complex v) { float a = crealf(v), b = cimagf(v); return a * a + b * b; }

void orc_am_fir32(orc_c16 hist[31], const orc_c16 *in, size_t n, orc_c16 *out)
{
    build_tables();
    orc_c16 *w = malloc(sizeof(orc_c16) * (n + 31));
    memcpy(w, hist, sizeof(orc_c16) * 31);
    memcpy(w + 31, in, sizeof(orc_c16) * n);
    for (size_t t = 0; t < n; t++) {
        const orc_c16 *a = w + t;
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

/* acquire.c:129-151 with the AM geometry */
static void cp_correlate(const float complex *buf, int *samperr_out, float complex *peak)
{
    float complex sums[SYM_A], max_v = 0;
    float max_mag = -1.0f;
    int samperr = 0;
    memset(sums, 0, sizeof(sums));
    for (int i = 0; i < SYM_A; ++i)
        for (int j = 0; j < NSYM; ++j)
            sums[i] += buf[i + j * SYM_A] * conjf(buf[i + j * SYM_A + FFT_A]);
    for (int i = 0; i < SYM_A; ++i) {
        float complex v = 0;
        for (int j = 0; j < CP_A; ++j)
            v += sums[(i + j) % SYM_A] * shape_am[j] * shape_am[j + FFT_A];
        float mag = norm2(v);
        if (mag > max_mag) { max_mag = mag; max_v = v; samperr = (i + SYM_A - 15) % SYM_A; }
    }
    *samperr_out = samperr; *peak = max_v;
}

/* mix one 270-sample symbol down with the running NCO, fold the cyclic prefix (rotated by 121 samples so that
 * carrier phases are referenced to the symbol centre) and transform: acquire.c:187-195 / 239-255 */
static void demod_symbol(const float complex *sym, float complex *phase, float complex inc, float complex *shifted)
{
    float complex fftin[FFT_A], fftout[FFT_A];
    const int rot = (FFT_A - CP_A) / 2;
    for (int j = 0; j < SYM_A; ++j) {
        float complex sample = *phase * sym[j];
        if (j < CP_A) fftin[(j + rot) % FFT_A] = shape_am[j] * sample;
        else if (j < FFT_A) fftin[(j + rot) % FFT_A] = sample;
        else fftin[(j + rot) % FFT_A] += shape_am[j] * sample;
        *phase *= inc;
    }
    *phase /= cabsf(*phase);
    oracle_fft_forward(FFT_A, (const float *)fftin, (float *)fftout);
    memcpy(shifted, fftout + FFT_A / 2, sizeof(float complex) * FFT_A / 2);
    memcpy(shifted + FFT_A / 2, fftout, sizeof(float complex) * FFT_A / 2);
}

static void process_window(orc_am_stream *s)
{
    float complex buf[WIN_A], shifted[FFT_A], phase_inc;
    orc_c16 filt[WIN_A];
    float angle, angle_diff, angle_factor;
    int samperr = 0;
    const unsigned state_before = s->sync_state;

    if (s->sync_state == ORC_SYNC_FINE) {
        samperr = SYM_A / 2 + s->samperr;  s->samperr = 0;
        angle_diff = -0.0f;                                /* sync_t.angle is only ever written by the FM path */
        angle = s->prev_angle + angle_diff;
        s->prev_angle = angle;
    } else {
        float complex peak;
        orc_am_fir32(s->fir_hist, s->ring, WIN_A, filt);
        for (int i = 0; i < WIN_A; i++) buf[i] = q15_to_cf(filt[i]);
        cp_correlate(buf, &samperr, &peak);
        angle_diff = cargf(peak * cexpf(I * -s->prev_angle));
        angle_factor = (s->prev_angle) ? 0.25 : 1.0;
        angle = s->prev_angle + (angle_diff * angle_factor);
        s->prev_angle = angle;
        set_sync_state(s, ORC_SYNC_COARSE);
    }
    for (int i = 0; i < WIN_A; i++) buf[i] = q15_to_cf(s->ring[i]);

    angle -= 2 * M_PI * s->cfo;
    s->phase *= cexpf(-(SYM_A / 2 - samperr) * angle / FFT_A * I);
    phase_inc = cexpf(angle / FFT_A * I);

    {   /* acquire.c:170-235: first pass measures the analog carrier's phase per symbol; a line fit over the block gives
         * the residual frequency (slope) and phase, and while un-synchronised the strongest bin near the centre
         * gives the integer carrier offset */
        float y = 0, sum_y = 0, sum_xy = 0, sum_x2 = 0;
        float complex last_carrier = 0, temp_phase = s->phase;
        float mag_sums[FFT_A] = { 0 };
        for (int i = 0; i < NSYM; ++i) {
            demod_symbol(buf + i * SYM_A + samperr, &temp_phase, phase_inc, shifted);
            float x = SYM_A * (i - (float)(NSYM - 1) / 2);
            if (i == 0) y = cargf(shifted[C_A]);
            else y += cargf(shifted[C_A] / last_carrier);
            last_carrier = shifted[C_A];
            sum_y += y; sum_xy += x * y; sum_x2 += x * x;
            if (s->sync_state != ORC_SYNC_FINE)
                for (int j = C_A - IDX_PIDS_OUT; j <= C_A + IDX_PIDS_OUT; j++) mag_sums[j] += cabsf(shifted[j]);
        }
        if (s->sync_state != ORC_SYNC_FINE) {
            float max_mag = -1.0f;
            int max_index = -1;
            for (int j = C_A - IDX_PIDS_OUT; j <= C_A + IDX_PIDS_OUT; j++)
                if (mag_sums[j] > max_mag) { max_mag = mag_sums[j]; max_index = j; }
            s->cfo += max_index - C_A;                     /* acquire_cfo_adjust: takes effect from the next block */
        }
        phase_inc *= cexpf(-sum_xy / sum_x2 * I);
        s->phase *= cexpf((-sum_y / NSYM + (sum_xy / sum_x2) * (NSYM) * SYM_A / 2 - 0.06) * I);
    }

    for (int i = 0; i < NSYM; ++i) {
        demod_symbol(buf + i * SYM_A + samperr, &s->phase, phase_inc, shifted);
        if ((s->taps & ORC_TAP_FFT) && s->fft_syms < s->fft_limit * NSYM) { gb_put(&s->fft, shifted, sizeof(shifted)); s->fft_syms++; }
        push_symbol(s, shifted);
    }

    int keep = SYM_A + (SYM_A / 2 - samperr) + s->keep_extra;
    s->keep_extra = 0;
    memmove(s->ring, s->ring + WIN_A - keep, sizeof(orc_c16) * keep);
    s->ring_fill = keep;

    struct { int32_t state_before, state_after, samperr, cfo, keep, bc, psmi, cfo_wait, next_samperr;
             float prev_angle, phase_re, phase_im, next_angle; } r = {
        (int32_t)state_before, (int32_t)s->sync_state, samperr, s->cfo, keep, (int32_t)s->bc, s->psmi, s->cfo_wait,
        s->samperr, s->prev_angle, crealf(s->phase), cimagf(s->phase), 0.0f };
    log_rec(s, ORC_REC_BLOCK, &r, sizeof(r));
}

static void feed_q15(orc_am_stream *s, const orc_c16 *x, size_t n)
{
    if (s->taps & ORC_TAP_Q15) gb_put(&s->q15, x, sizeof(orc_c16) * n);
    while (n) {
        size_t take = WIN_A - s->ring_fill;
        if (take > n) take = n;
        memcpy(s->ring + s->ring_fill, x, sizeof(orc_c16) * take);
        s->ring_fill += take; x += take; n -= take;
        if (s->ring_fill == WIN_A) process_window(s);
    }
}

void orc_am_push_cu8(orc_am_stream *s, const uint8_t *iq, uint32_t nbytes)
{
    orc_c16 out[512];
    while (nbytes) {
        uint32_t take = nbytes > 64 * 256 ? 64 * 256 : nbytes;
        size_t n = orc_am_decimate_cu8(s->decim, iq, take, out);
        feed_q15(s, out, n);
        iq += take; nbytes -= take;
    }
}

void orc_am_push_cs16(orc_am_stream *s, const int16_t *iq, uint32_t n) { feed_q15(s, (const orc_c16 *)iq, n / 2); }   /* input.c:119-124 */

orc_am_stream *orc_am_open(void)
{
    build_tables();
    orc_am_stream *s = calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->decim = orc_am_decim_new();
    s->phase = 1; s->psmi = 1; s->pli = s->hppi = s->aabi = s->rdbi = -1;      /* acquire_reset, sync_reset */
    s->am_diversity_wait = 4; s->fft_limit = 4;
    return s;
}
void orc_am_close(orc_am_stream *s)
{
    if (!s) return;
    orc_am_decim_free(s->decim);
    free(s->log.p); free(s->q15.p); free(s->fft.p); free(s);
}
void orc_am_set_taps(orc_am_stream *s, unsigned mask, unsigned fft_limit_blocks) { s->taps = mask; s->fft_limit = fft_limit_blocks; }
void orc_am_set_p1_hook(orc_am_stream *s, orc_p1_hook hook, void *user) { s->hook = hook; s->hook_user = user; }
size_t orc_am_buf(orc_am_stream *s, int which, const uint8_t **p)
{
    gbuf *b = which == 0 ? &s->log : which == 1 ? &s->q15 : &s->fft;
    *p = b->p; return b->len;
}
