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
 * input_set_sync_state(SYNC_STATE_NONE).  Frames that announce fixed-data sub-channels (has_fixed) move audio_end by
 * state this restatement does not model: they are reported as 1. */
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
    if (p == (0xE3634C & 0xFFFFFC) || p == (0x8D8D33 & 0xFFFFFC)) return 1;   /* has_fixed: not modelled */
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
