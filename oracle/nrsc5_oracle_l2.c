/* TEST INFRASTRUCTURE -- see nrsc5_oracle.h.  The one piece of L2 that feeds back into the hot path: frame_process
 * drops the receiver to SYNC_STATE_NONE when the first L2 header of a P1 frame fails its RS(255,247) check
 * (frame.c:516-540).  Restated here from frame_push's bit unpacking (frame.c:645-714), has_audio / has_fixed
 * (frame.c:138-151), fix_header (frame.c:153-179) and the Reed-Solomon decoder the reference links (third-party:
 * Phil Karn's libfec decode_rs_char as vendored in src/rs_decode.c, configured by init_rs_char(8, 0x11d, 1, 1, 8),
 * frame.c:747): syndromes, Berlekamp-Massey, Chien search, Forney -- written out from the algorithm, with the same
 * accept / reject decisions (root count == deg(lambda), nonzero derivative, zero padding after correction). */
#include <string.h>
#include "nrsc5_oracle.h"

static uint8_t gexp[512], glog[256];
static int gf_ready;

static void gf_init(void)
{
    if (gf_ready) return;
    unsigned x = 1;
    for (int i = 0; i < 255; i++) { gexp[i] = (uint8_t)x; glog[x] = (uint8_t)i; x <<= 1; if (x & 0x100) x ^= 0x11d; }
    for (int i = 255; i < 512; i++) gexp[i] = gexp[i - 255];
    gf_ready = 1;
}
static inline uint8_t gmul(uint8_t a, uint8_t b) { return (a && b) ? gexp[glog[a] + glog[b]] : 0; }
static inline uint8_t gdiv(uint8_t a, uint8_t b) { return a ? gexp[glog[a] + 255 - glog[b]] : 0; }
static inline uint8_t gpow_alpha(unsigned e) { return gexp[e % 255]; }

/* r[0] is the coefficient of x^254.  Returns the number of corrected symbols, or -1. */
int orc_rs255_247_decode(uint8_t r[255])
{
    gf_init();
    uint8_t S[8], any = 0;
    for (int i = 0; i < 8; i++) {                              /* S_i = r(alpha^(i+1)), Horner */
        uint8_t acc = r[0];
        for (int j = 1; j < 255; j++) acc = gmul(acc, gpow_alpha(i + 1)) ^ r[j];
        S[i] = acc; any |= acc;
    }
    if (!any) return 0;
    uint8_t lam[9] = { 1 }, B[9] = { 1 }, T[9];
    int L = 0;
    for (int step = 1; step <= 8; step++) {                    /* Berlekamp-Massey, errors only */
        uint8_t delta = 0;
        for (int i = 0; i < step; i++) delta ^= gmul(lam[i], S[step - 1 - i]);
        if (!delta) { memmove(B + 1, B, 8); B[0] = 0; continue; }
        T[0] = lam[0];
        for (int i = 0; i < 8; i++) T[i + 1] = lam[i + 1] ^ gmul(delta, B[i]);
        if (2 * L <= step - 1) {
            L = step - L;
            for (int i = 0; i <= 8; i++) B[i] = gdiv(lam[i], delta);
        } else { memmove(B + 1, B, 8); B[0] = 0; }
        memcpy(lam, T, 9);
    }
    int deg = 0;
    for (int i = 0; i <= 8; i++) if (lam[i]) deg = i;
    int root[8], loc[8], count = 0;
    for (int i = 1; i <= 255 && count < 8; i++) {              /* Chien search: lambda(alpha^i) == 0 */
        uint8_t q = 1;
        for (int j = 1; j <= deg; j++) q ^= gmul(lam[j], gpow_alpha((unsigned)(i * j)));
        if (q) continue;
        root[count] = i; loc[count] = i - 1;
        if (++count == deg) break;
    }
    if (count != deg) return -1;
    uint8_t om[8]; int deg_om = 0;
    for (int i = 0; i < 8; i++) {                              /* omega = S * lambda mod x^8 */
        uint8_t t = 0;
        for (int j = (deg < i ? deg : i); j >= 0; j--) t ^= gmul(S[i - j], lam[j]);
        om[i] = t; if (t) deg_om = i;
    }
    for (int k = count - 1; k >= 0; k--) {                     /* Forney (first consecutive root 1: no x^(fcr-1) factor) */
        uint8_t num = 0, den = 0;
        for (int i = deg_om; i >= 0; i--) num ^= gmul(om[i], gpow_alpha((unsigned)(i * root[k])));
        for (int i = (deg < 7 ? deg : 7) & ~1; i >= 0; i -= 2) den ^= gmul(lam[i + 1], gpow_alpha((unsigned)(i * root[k])));
        if (!den) return -1;
        if (num) r[loc[k]] ^= gdiv(num, den);
    }
    return count;
}

/* Would frame_process keep the receiver synchronised after this P1 frame?  bits = the frame as handed to frame_push
 * (one bit per byte), len = 146176 (FM) or 3750 (AM).  1 = yes / no check applies, 0 = it calls
 * input_set_sync_state(SYNC_STATE_NONE).  Frames that announce fixed-data sub-channels next to audio (PCI_AUDIO_FIXED /
 * _OPP) move audio_end (process_fixed_data, frame.c:458-514), but the header at offset 0 is still checked as long as
 * 0 < audio_end - 96 (frame.c:525), i.e. always for a P1 PDU: same decision as for plain audio frames. */
int orc_l2_first_header_ok(const uint8_t *bits, unsigned len)
{
    unsigned start, step, pci_len;
    if (len == 146176) { start = 146176 - 30000; step = 1248; pci_len = 24; }
    else if (len == 3750) { start = 120; step = 160; pci_len = 22; }
    else return 1;
    uint8_t pdu[96];
    unsigned nbytes = 0, j = 0, h = 0, val = 0, pci = 0;
    for (unsigned i = 0; i < len; i++) {
        const unsigned b0 = (i >> 3) << 3, blen = (len - b0 < 8) ? len - b0 : 8;
        const unsigned bit = bits[b0 + blen - 1 - (i & 7)];
        if (i >= start && ((i - start) % step) == 0 && h < pci_len) { pci |= bit << (23 - h); ++h; }
        else {
            val |= bit << (7 - j);
            if (++j == 8) { if (nbytes < 96) pdu[nbytes] = (uint8_t)val; nbytes++; val = 0; j = 0; }
        }
    }
    const unsigned p = pci & 0xFFFFFC;
    if (p == (0x3634CE & 0xFFFFFC)) return 1;                  /* !has_audio */
    if (nbytes <= 96) return 1;                                /* while (offset < audio_end - RS_CODEWORD_LEN) not entered */
    uint8_t r[255];
    memset(r, 0, 159);
    for (int i = 0; i < 96; i++) r[254 - i] = pdu[i];
    if (orc_rs255_247_decode(r) < 0) return 0;
    for (int i = 0; i < 159; i++) if (r[i]) return 0;
    return 1;
}

/* pids_frame_push's acceptance test (pids.c:52-86, 1032-1050): undo the per-byte bit reversal, CRC-12 over logical bits
 * 0..67 (processed from bit 67 down), compare with bits 68..79.  1 = the frame goes on to sis_decode. */
int orc_pids_crc_ok(const uint8_t bits[80])
{
    uint8_t p[80];
    for (int i = 0; i < 80; i++) p[i] = bits[((i >> 3) << 3) + 7 - (i & 7)];
    uint16_t reg = 0;
    for (int i = 67; i >= 0; i--) {
        const int low = reg & 1;
        reg >>= 1;
        reg ^= (uint16_t)(p[i] << 15);
        if (low) reg ^= 0xD010;
    }
    for (int i = 0; i < 16; i++) { const int low = reg & 1; reg >>= 1; if (low) reg ^= 0xD010; }
    reg ^= 0x955;
    unsigned expected = 0;
    for (int i = 68; i < 80; i++) expected = (expected << 1) | p[i];
    return expected == (reg & 0xfffu);
}

/* ---- audio transport index -------------------------------------------------------------------------------------- */
/* crc8, frame.c:130-136 with the table of frame.c:60-90 = MSB-first CRC, polynomial 0x31, initial value 0xFF */
uint8_t orc_crc8(const uint8_t *p, unsigned n)
{
    unsigned c = 0xFF;
    for (unsigned i = 0; i < n; i++) {
        c ^= p[i];
        for (int k = 0; k < 8; k++) c = (c & 0x80) ? ((c << 1) ^ 0x31) & 0xFF : (c << 1) & 0xFF;
    }
    return (uint8_t)c;
}

/* frame_push's switch (frame.c:651-686) */
static int l2_layout(unsigned len, unsigned *start, unsigned *step, unsigned *pci_len)
{
    switch (len) {
    case 146176: *start = 146176 - 30000; *step = 1248; *pci_len = 24; return 0;
    case 4608:   *start = 120; *step = 184;  *pci_len = 24; return 0;
    case 2304:   *start = 120; *step = 88;   *pci_len = 24; return 0;
    case 3750:   *start = 120; *step = 160;  *pci_len = 22; return 0;
    case 24000:  *start = 120; *step = 992;  *pci_len = 24; return 0;
    case 30000:  *start = 120; *step = 1240; *pci_len = 24; return 0;
    }
    return -1;
}

/* parse_hef, frame.c:198-265 */
static unsigned l2_parse_hef(const uint8_t *buf, unsigned length, orc_l2_pdu *h)
{
    const uint8_t *byte = buf, *end = buf + length;
    do {
        if (byte >= end) return length;
        switch ((*byte >> 4) & 7) {
        case 0: h->class_ind = *byte & 0xf; break;
        case 1:
            h->prog_num = (*byte >> 1) & 7;
            if (*byte & 1) {
                if (byte + 2 >= end) return length;
                byte++; h->hef_pdu_len = (uint16_t)((*byte & 0x7f) << 7);
                byte++; h->hef_pdu_len |= (*byte & 0x7f);
            }
            break;
        case 2:
            if (byte + 1 >= end) return length;
            h->access = (*byte >> 3) & 1;
            h->prog_type = (uint8_t)((*byte & 1) << 7);
            byte++; h->prog_type |= (*byte & 0x7f);
            break;
        case 3:
            if (*byte & 8) { if (byte + 4 >= end) return length; byte += 4; }
            else           { if (byte + 3 >= end) return length; byte += 3; }
            break;
        case 4:
            if (*byte & 8) {
                if (byte + 3 >= end) return length;
                h->applied_services = *byte & 7;
                byte++; h->pdu_marker  = (uint32_t)(*byte & 0x7f) << 14;
                byte++; h->pdu_marker |= (uint32_t)(*byte & 0x7f) << 7;
                byte++; h->pdu_marker |= (*byte & 0x7f);
            } else { if (byte + 1 >= end) return length; byte++; }
            break;
        default: break;
        }
    } while (*(byte++) & 0x80);
    return (unsigned)(byte - buf);
}

int orc_l2_index(const uint8_t *bits, unsigned len, orc_l2_frame *out, uint8_t *bytes_out)
{
    unsigned start0, step, pci_len;
    memset(out, 0, sizeof(*out));
    if (l2_layout(len, &start0, &step, &pci_len)) return -1;
    static uint8_t scratch[(146176 - 24) / 8 + 8];
    uint8_t *buf = bytes_out ? bytes_out : scratch;
    unsigned nbytes = 0, j = 0, h = 0, val = 0, pci = 0;
    for (unsigned i = 0; i < len; i++) {                       /* frame.c:688-709 */
        const unsigned b0 = (i >> 3) << 3, blen = (len - b0 < 8) ? len - b0 : 8;
        const unsigned bit = bits[b0 + blen - 1 - (i & 7)];
        if (i >= start0 && ((i - start0) % step) == 0 && h < pci_len) { pci |= bit << (23 - h); ++h; }
        else { val |= bit << (7 - j); if (++j == 8) { buf[nbytes++] = (uint8_t)val; val = 0; j = 0; } }
    }
    out->pci = pci; out->nbytes = nbytes;
    const unsigned p = pci & 0xFFFFFC;
    /* has_fixed next to audio: process_fixed_data (frame.c:458-514) moves audio_end by host state, never beyond length - 1;
     * the index is built with that largest value (every audio_end test below only gets stricter below it) and the consumer
     * cuts it back (orc_l2_apply_audio_end) */
    const unsigned audio_end = (p == (0xE3634C & 0xFFFFFC) || p == (0x8D8D33 & 0xFFFFFC)) ? nbytes - 1 : nbytes;
    const int is_p1 = (len == 146176 || len == 3750);          /* length == MAX_PDU_LEN || P1_PDU_LEN_AM, frame.c:537 */
    if (p == (0x3634CE & 0xFFFFFC)) { out->status = ORC_L2_NO_AUDIO; return 0; }
    unsigned offset = 0;
    out->status = ORC_L2_END;
    while (offset < audio_end - 96) {                          /* unsigned, as frame.c:525 */
        const unsigned start = offset;
        if (out->n_pdu == ORC_L2_MAX_PDUS) { out->status = ORC_L2_TOO_MANY_PDUS; break; }
        orc_l2_pdu *d = &out->pdu[out->n_pdu];
        memset(d, 0, sizeof(*d));
        uint8_t r[255];
        memset(r, 0, 159);
        for (int i = 0; i < 96; i++) r[254 - i] = buf[offset + i];
        const int corr = orc_rs255_247_decode(r);
        int ok = corr >= 0;
        for (int i = 0; ok && i < 159; i++) if (r[i]) ok = 0;
        if (!ok) { out->status = ORC_L2_HEADER_RS; out->lost_sync = (is_p1 && offset == 0); break; }
        for (int i = 0; i < 96; i++) buf[offset + i] = r[254 - i];
        const uint8_t *b = buf + offset;                       /* parse_header, frame.c:181-196 */
        d->start = start; d->rs_corrections = (uint8_t)corr;
        d->codec_mode = b[8] & 0xf; d->stream_id = (b[8] >> 4) & 3; d->pdu_seq = (uint8_t)((b[8] >> 6) | ((b[9] & 1) << 2));
        d->blend_control = (b[9] >> 1) & 3; d->per_stream_delay = b[9] >> 3; d->common_delay = b[10] & 0x3f;
        d->latency = (uint8_t)((b[10] >> 6) | ((b[11] & 1) << 2)); d->pfirst = (b[11] >> 1) & 1; d->plast = (b[11] >> 2) & 1;
        d->seq = (uint8_t)((b[11] >> 3) | ((b[12] & 1) << 5)); d->nop = (b[12] >> 1) & 0x3f; d->hef = b[12] >> 7;
        d->la_location = b[13];
        offset += 14;
        unsigned lc_bits = 16, avg = 32;                       /* calc_lc_bits / calc_avg_packets, frame.c:267-315 */
        switch (d->codec_mode) {
        case 0: break;
        case 1: case 2: case 3: if (d->stream_id == 0) { lc_bits = 12; avg = 4; } break;
        case 10: lc_bits = 12; if (d->stream_id != 0) avg = 4; break;
        case 13: lc_bits = 12; avg = 4; break;
        default: break;
        }
        const unsigned loc_bytes = (lc_bits * d->nop + 4) / 8;
        if (start + d->la_location + 1 < offset + loc_bytes || start + d->la_location >= audio_end) { out->status = ORC_L2_BAD_LOCATORS; break; }
        int bad = 0;
        for (unsigned k = 0; k < d->nop; k++) {                /* parse_location, frame.c:317-328 */
            const uint8_t *lb = buf + offset;
            unsigned loc;
            if (lc_bits == 16) loc = (unsigned)(lb[2 * k + 1] << 8) | lb[2 * k];
            else if (k % 2 == 0) loc = (unsigned)((lb[k / 2 * 3 + 1] & 0xf) << 8) | lb[k / 2 * 3];
            else loc = (unsigned)(lb[k / 2 * 3 + 2] << 4) | (lb[k / 2 * 3 + 1] >> 4);
            const unsigned prev = k ? (unsigned)(d->loc[k - 1] - start) : 0;
            if ((k == 0 && loc <= d->la_location) || (k > 0 && loc <= prev) || start + loc >= audio_end) { bad = 1; break; }
            d->loc[k] = (uint16_t)(start + loc);
        }
        if (bad) { out->status = ORC_L2_BAD_LOCATORS; break; }
        offset += loc_bytes;
        if (d->stream_id >= 2) {                               /* MAX_STREAMS, frame.c:559-564 */
            if (d->nop == 0) { out->status = ORC_L2_BAD_STREAM; break; }     /* the reference reads locations[-1] here */
            d->skipped = 1; out->n_pdu++;
            offset = d->loc[d->nop - 1] + 1u;
            continue;
        }
        if (d->hef) offset += l2_parse_hef(buf + offset, audio_end - offset, d);
        d->elastic_seq = (uint8_t)((64 + d->seq - d->pfirst) % 64);           /* frame.c:593-598 */
        unsigned oo = (64 + d->pdu_seq * avg - d->latency * 2u) % 64;
        if (((64 + d->elastic_seq - oo) % 64) >= 32) oo = (oo + 32) % 64;
        d->align_offset = (uint8_t)oo;
        d->psd_off = offset; d->psd_len = (int32_t)(start + d->la_location + 1) - (int32_t)offset;
        if (d->psd_len < 0) { out->status = ORC_L2_HEF_OVERRUN; break; }      /* the reference's inlen wraps: undefined */
        offset = start + d->la_location + 1;
        d->audio_off = offset;
        for (unsigned k = 0; k < d->nop; k++) {                /* frame.c:613-640 */
            const unsigned cnt = d->loc[k] - offset;
            if (orc_crc8(buf + offset, cnt + 1) != 0) { if (k < 32) d->crc_bad_lo |= 1u << k; else d->crc_bad_hi |= 1u << (k - 32); }
            offset += cnt + 1;
        }
        out->n_pdu++;
    }
    out->end_offset = offset;
    return 0;
}
