// The one piece of L2 that feeds back into the hot path: frame_process drops the receiver to
// SYNC_STATE_NONE when the first L2 header of a P1 frame fails its RS(255,247) check (frame.c:516-540).  With it on
// the device the batch pipeline follows the reference through false locks without a host round trip.
//   frame_push bit unpacking        frame.c:645-714      -> l2_extract_*
//   has_audio / has_fixed           frame.c:138-151      -> l2_pci_wants_check
//   fix_header                      frame.c:153-179      -> l2_header_codeword_ok
//   decode_rs_char (libfec, vendored as src/rs_decode.c; init_rs_char(8, 0x11d, 1, 1, 8), frame.c:747)
//                                                        -> rs255_247_decode: syndromes, Berlekamp-Massey, Chien,
//                                                           Forney with the same accept / reject decisions
// Single work-item code: ~6 k GF operations per frame, once per 2.2 M input samples.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nrsc5 {

struct L2Smem { uint8_t gexp[512], glog[256], r[255], pdu[96]; int ok; };

// GF(256), x^8 + x^4 + x^3 + x^2 + 1, as compile-time tables (the block-cooperative check below copies them into LDS)
struct GfTables { uint8_t gexp[512], glog[256]; };
constexpr GfTables gf_make_tables()
{
    GfTables t{};
    unsigned x = 1;
    for (int i = 0; i < 255; i++) { t.gexp[i] = (uint8_t)x; t.glog[x] = (uint8_t)i; x <<= 1; if (x & 0x100u) x ^= 0x11du; }
    for (int i = 255; i < 512; i++) t.gexp[i] = t.gexp[i - 255];
    t.glog[0] = 0;
    return t;
}
static __device__ const GfTables GF_TABLES = gf_make_tables();

__device__ inline void l2_gf_init(L2Smem &g)                 // GF(256), x^8 + x^4 + x^3 + x^2 + 1
{
    unsigned x = 1;
    for (int i = 0; i < 255; i++) { g.gexp[i] = (uint8_t)x; g.glog[x] = (uint8_t)i; x <<= 1; if (x & 0x100u) x ^= 0x11du; }
    for (int i = 255; i < 512; i++) g.gexp[i] = g.gexp[i - 255];
    g.glog[0] = 0;
}
__device__ inline unsigned l2_mul(const L2Smem &g, unsigned a, unsigned b) { return (a && b) ? g.gexp[g.glog[a] + g.glog[b]] : 0u; }
__device__ inline unsigned l2_div(const L2Smem &g, unsigned a, unsigned b) { return a ? g.gexp[g.glog[a] + 255 - g.glog[b]] : 0u; }
__device__ inline unsigned l2_alpha(const L2Smem &g, unsigned e) { return g.gexp[e % 255u]; }

// g.r[0] = coefficient of x^254.  Returns the number of corrected symbols or -1 (g.r is corrected in place).
__device__ inline int rs255_247_decode(L2Smem &g)
{
    unsigned S[8], any = 0;
    for (int i = 0; i < 8; i++) {
        const unsigned a = l2_alpha(g, (unsigned)(i + 1));
        unsigned acc = g.r[0];
        for (int j = 1; j < 255; j++) acc = l2_mul(g, acc, a) ^ g.r[j];
        S[i] = acc; any |= acc;
    }
    if (!any) return 0;
    unsigned lam[9], B[9], T[9];
    for (int i = 0; i < 9; i++) { lam[i] = 0; B[i] = 0; }
    lam[0] = 1; B[0] = 1;
    int L = 0;
    for (int step = 1; step <= 8; step++) {
        unsigned delta = 0;
        for (int i = 0; i < step; i++) delta ^= l2_mul(g, lam[i], S[step - 1 - i]);
        if (!delta) { for (int i = 8; i > 0; i--) B[i] = B[i - 1]; B[0] = 0; continue; }
        T[0] = lam[0];
        for (int i = 0; i < 8; i++) T[i + 1] = lam[i + 1] ^ l2_mul(g, delta, B[i]);
        if (2 * L <= step - 1) {
            L = step - L;
            for (int i = 0; i <= 8; i++) B[i] = l2_div(g, lam[i], delta);
        } else { for (int i = 8; i > 0; i--) B[i] = B[i - 1]; B[0] = 0; }
        for (int i = 0; i <= 8; i++) lam[i] = T[i];
    }
    int deg = 0;
    for (int i = 0; i <= 8; i++) if (lam[i]) deg = i;
    int root[8], count = 0;
    for (int i = 1; i <= 255 && count < 8; i++) {
        unsigned q = 1;
        for (int j = 1; j <= deg; j++) q ^= l2_mul(g, lam[j], l2_alpha(g, (unsigned)(i * j)));
        if (q) continue;
        root[count] = i;
        if (++count == deg) break;
    }
    if (count != deg) return -1;
    unsigned om[8]; int deg_om = 0;
    for (int i = 0; i < 8; i++) {
        unsigned t = 0;
        for (int j = (deg < i ? deg : i); j >= 0; j--) t ^= l2_mul(g, S[i - j], lam[j]);
        om[i] = t; if (t) deg_om = i;
    }
    for (int k = count - 1; k >= 0; k--) {
        unsigned num = 0, den = 0;
        for (int i = deg_om; i >= 0; i--) num ^= l2_mul(g, om[i], l2_alpha(g, (unsigned)(i * root[k])));
        for (int i = (deg < 7 ? deg : 7) & ~1; i >= 0; i -= 2) den ^= l2_mul(g, lam[i + 1], l2_alpha(g, (unsigned)(i * root[k])));
        if (!den) return -1;
        if (num) g.r[root[k] - 1] ^= (uint8_t)l2_div(g, num, den);
    }
    return count;
}

__device__ inline bool l2_pci_wants_check(unsigned pci)
{
    const unsigned p = pci & 0xFFFFFCu;
    if (p == (0x3634CEu & 0xFFFFFCu)) return false;                                          // !has_audio (PCI_FIXED)
    // PCI_AUDIO_FIXED / _OPP: frame_process still runs fix_header at offset 0 -- the loop condition 0 < audio_end - 96
    // (frame.c:525) holds unless the fixed-data sub-channels fill all but 96 bytes of the 18 269-byte PDU, which no
    // service mode allows -- so the check applies to them as to plain audio frames.
    return true;
}

__device__ inline bool l2_header_codeword_ok(L2Smem &g)       // fix_header on g.pdu[0..95]
{
    for (int i = 0; i < 159; i++) g.r[i] = 0;
    for (int i = 0; i < 96; i++) g.r[254 - i] = g.pdu[i];
    if (rs255_247_decode(g) < 0) return false;
    for (int i = 0; i < 159; i++) if (g.r[i]) return false;
    return true;
}

__device__ inline unsigned l2_bit(const uint32_t *w, unsigned i) { return (w[i >> 5] >> (i & 31)) & 1u; }

// FM P1 frame (146176 descrambled bits, packed LSB-first): the 24 PCI bits sit at 116176 + 1248 h, far behind the first 96 PDU
// bytes, and frame_push's per-byte bit reversal turns PDU byte q into byte q of the packed frame.
// AM P1 frame (3750 bits): 22 PCI bits at 120 + 160 h are interleaved with the first PDU bytes.

// ---- the check by a whole workgroup ---------------------------------------------------------------------------
// As single work-item code the check was ~0.2 ms at the end of every P1 decode (table build, bit unpacking, 8 x 254 dependent
// GF multiplies for the syndromes) with the rest of the workgroup waiting.  Here EVERY work-item of the block calls; the first
// wave computes the eight syndromes S_i = sum_q pdu[q] alpha^((i+1) q) one or two PDU bytes per lane.  All zero (the usual case):
// the code word is accepted as rs255_247_decode accepts it (no correction, the zero padding untouched).  Otherwise work-item 0
// runs the serial decoder above (l2_header_codeword_ok) -- same accept / reject decisions by construction.
__device__ inline void l2_gf_init_block(L2Smem &g)
{
    for (int k = threadIdx.x; k < 512; k += blockDim.x) g.gexp[k] = GF_TABLES.gexp[k];
    for (int k = threadIdx.x; k < 256; k += blockDim.x) g.glog[k] = GF_TABLES.glog[k];
}

// g.pdu, the tables and `pci` (wave 0) are in place and the block has synchronised; returns the verdict in every work-item
__device__ inline bool l2_header_verdict_block(L2Smem &g, unsigned pci)
{
    if (threadIdx.x < 64) {
        unsigned lo = 0, hi = 0;                               // S_0..S_3 / S_4..S_7, one byte each
        for (unsigned q = threadIdx.x; q < 96; q += 64) {
            const unsigned v = g.pdu[q];
            if (!v) continue;
            const unsigned lg = g.glog[v];
            for (unsigned i = 0; i < 4; i++) {
                lo ^= (unsigned)g.gexp[(lg + (i + 1) * q) % 255u] << (8 * i);
                hi ^= (unsigned)g.gexp[(lg + (i + 5) * q) % 255u] << (8 * i);
            }
        }
        for (int m = 32; m >= 1; m >>= 1) { lo ^= (unsigned)__shfl_xor((int)lo, m); hi ^= (unsigned)__shfl_xor((int)hi, m); }
        if (threadIdx.x == 0) g.ok = !l2_pci_wants_check(pci) ? 1 : !(lo | hi) ? 1 : l2_header_codeword_ok(g) ? 1 : 0;
    }
    __syncthreads();
    return g.ok != 0;
}

__device__ inline bool l2_first_header_ok_fm_block(const uint32_t *w, L2Smem &g)
{
    l2_gf_init_block(g);
    unsigned pci = 0;
    if (threadIdx.x < 64) {
        unsigned bit = 0;
        if (threadIdx.x < 24) { const unsigned i = 116176u + 1248u * threadIdx.x; bit = l2_bit(w, (i & ~7u) + 7u - (i & 7u)); }
        pci = __brev((unsigned)__ballot((int)bit)) >> 8;       // lane h -> bit 23 - h
    }
    for (int q = threadIdx.x; q < 96; q += blockDim.x) g.pdu[q] = (uint8_t)(w[q >> 2] >> (8 * (q & 3)));
    __syncthreads();
    return l2_header_verdict_block(g, pci);
}

__device__ inline unsigned l2_am_bit(const uint32_t *w, unsigned i)      // frame_push's per-byte bit reversal (the last byte of the 3750 bits is short)
{
    const unsigned len = 3750, b0 = i & ~7u, blen = (len - b0 < 8) ? len - b0 : 8;
    return l2_bit(w, b0 + blen - 1 - (i & 7));
}

__device__ inline bool l2_first_header_ok_am_block(const uint32_t *w, L2Smem &g)
{
    l2_gf_init_block(g);
    unsigned pci = 0;
    if (threadIdx.x < 64) {
        const unsigned bit = threadIdx.x < 22 ? l2_am_bit(w, 120u + 160u * threadIdx.x) : 0u;
        pci = __brev((unsigned)__ballot((int)bit)) >> 8;
    }
    // PDU byte q = data bits 8q .. 8q+7; data bit k sits at frame bit k + (PCI bits before it): 120 + 160 h are PCI positions
    for (unsigned q = threadIdx.x; q < 96; q += blockDim.x) {
        unsigned val = 0;
        for (unsigned j = 0; j < 8; j++) { const unsigned k = 8 * q + j, i = k < 120 ? k : k + (k - 120) / 159 + 1; val |= l2_am_bit(w, i) << (7 - j); }
        g.pdu[q] = (uint8_t)val;
    }
    __syncthreads();
    return l2_header_verdict_block(g, pci);
}

// pids_frame_push's acceptance test (pids.c:52-86, 1032-1050) on the 80 descrambled bits packed LSB-first in w[0..2]:
// undo the per-byte bit reversal, CRC-12 over logical bits 0..67 (from bit 67 down), compare with bits 68..79
__device__ inline bool pids_crc_ok(const uint32_t *w)
{
    auto logical = [&](int i) -> unsigned { const int k = ((i >> 3) << 3) + 7 - (i & 7); return (w[k >> 5] >> (k & 31)) & 1u; };
    unsigned reg = 0;
    for (int i = 67; i >= 0; i--) {
        const unsigned low = reg & 1u;
        reg >>= 1;
        reg ^= logical(i) << 15;
        if (low) reg ^= 0xD010u;
    }
    for (int i = 0; i < 16; i++) { const unsigned low = reg & 1u; reg >>= 1; if (low) reg ^= 0xD010u; }
    reg ^= 0x955u;
    unsigned expected = 0;
    for (int i = 68; i < 80; i++) expected = (expected << 1) | logical(i);
    return expected == (reg & 0xfffu);
}

// The same test by the 64 lanes of ONE wave (the inline PIDS decode of the streaming seam's sync kernel: on one lane the 84-trip bit loop above was ~5 000 shader cycles of a chain
// the host waits for).  The CRC register is linear in the message bits (it starts at zero): its final value is the XOR of one 16-bit term per set bit -- pids_crc_term(i), what a
// lone 1 at logical bit i leaves after the remaining shifts -- so lane i contributes its term, and every bit of the XOR is the parity of a ballot.
constexpr unsigned pids_crc_term(int i)
{
    unsigned reg = 0;
    for (int j = 67; j >= 0; j--) { const unsigned low = reg & 1u; reg >>= 1; reg ^= (j == i ? 1u : 0u) << 15; if (low) reg ^= 0xD010u; }
    for (int j = 0; j < 16; j++) { const unsigned low = reg & 1u; reg >>= 1; if (low) reg ^= 0xD010u; }
    return reg;
}
struct PidsCrcTerms { uint16_t t[68]; };
constexpr PidsCrcTerms pids_crc_terms()
{
    PidsCrcTerms r{};
    for (int i = 0; i < 68; i++) r.t[i] = (uint16_t)pids_crc_term(i);
    return r;
}
__device__ inline bool pids_crc_ok_wave(const uint32_t *w)      // w[0..2]: wave-uniform; call with all 64 lanes
{
    constexpr PidsCrcTerms T = pids_crc_terms();
    auto logical = [&](int i) -> unsigned { const int k = ((i >> 3) << 3) + 7 - (i & 7); return (w[k >> 5] >> (k & 31)) & 1u; };
    const int lane = threadIdx.x & 63;
    unsigned v = logical(lane) ? (unsigned)T.t[lane] : 0u;
    if (lane < 4 && logical(64 + lane)) v ^= (unsigned)T.t[64 + lane];
    unsigned reg = 0;
#pragma unroll
    for (int b = 0; b < 12; b++) reg |= (unsigned)(__popcll(__ballot((int)((v >> b) & 1u))) & 1) << b;
    reg ^= 0x955u;
    unsigned expected = 0;
    for (int i = 68; i < 80; i++) expected = (expected << 1) | logical(i);
    return expected == (reg & 0xfffu);
}

}  // namespace nrsc5
