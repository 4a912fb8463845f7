// Batch consumer of the L2 audio-transport index: turns nrsc5hip_l2_frame + PDU bytes into the reference's
// NRSC5_EVENT_HDC stream for thousands of streams without one nrsc5_t (22.9 MB: output_t alone holds 8 x 2 elastic buffers of
// 64 x 18 269 bytes, output.h:104-122) per stream.  Host code only -- no device work, no HIP calls.
//
//   frame_process's hand-off   output_align / output_push            frame.c:590-640, output.c:31-92   -> nrsc5hip_hdc_push_frame
//   output_advance             pops 2 (FM) / 4 (AM) packets per program and block, NRSC5_EVENT_HDC for every complete one
//                                                                   output.c:100-168, nrsc5.c:709-728 -> nrsc5hip_hdc_advance
//   dump_hdc / write_adts_header (--dump-hdc)                        main.c:182-212                    -> nrsc5hip_hdc_adts
//
// State per stream: for every program that ever carried audio, 64 packet slots (size, flags, shape) whose payload buffers
// only ever grow to the largest packet that passed through them -- ~40 KB per program at the usual 300-600 byte packets.
#include <stdint.h>
#include <string.h>
#include <memory>
#include <new>
#include <vector>
#include "nrsc5hip.h"

namespace {

constexpr int ELASTIC_LEN = 64;                                // ELASTIC_BUFFER_LEN, defines.h:71
constexpr int MAX_PROGRAMS = 8, MAX_STREAMS = 2;               // defines.h:67-69
enum { SHAPE_NONE = 0, SHAPE_FULL = 1, SHAPE_HALF_FRONT = 2, SHAPE_HALF_BACK = 3 };   // packet shapes, output.h
constexpr unsigned FLAG_CRC_ERROR = 1;                         // PACKET_FLAG_CRC_ERROR

// process_fixed_data's state (ccc_data_t, frame.h:30-38) as far as audio_end depends on it
struct Ccc {
    unsigned sync_width = 0, sync_count = 0;
    uint8_t buf[32];
    int idx = -1, fixed_ready = 0;
    unsigned length[4] = {0, 0, 0, 0};
    void reset() { sync_width = 0; sync_count = 0; idx = -1; fixed_ready = 0; for (unsigned &l : length) l = 0; }
};

struct Packet { std::vector<uint8_t> data; unsigned size = 0, flags = 0, shape = SHAPE_NONE; };
struct Elastic { Packet packets[ELASTIC_LEN]; };
struct Stream {
    int audio_offset[MAX_PROGRAMS][MAX_STREAMS];
    std::unique_ptr<Elastic> elastic[MAX_PROGRAMS];            // stream_id 0 only: output_push ignores the enhanced stream (output.c:52-53)
    Ccc ccc[3];                                                // per logical channel P1 / P3 / P4
    Stream() { reset(); }
    void reset()
    {
        for (auto &a : audio_offset) for (int &v : a) v = -1;  // output_reset, output.c:204-218
        for (auto &e : elastic) if (e) for (Packet &p : e->packets) { p.size = 0; p.flags = 0; p.shape = SHAPE_NONE; }
        for (Ccc &c : ccc) c.reset();
    }
};

}  // namespace

struct nrsc5hip_hdc { std::vector<Stream> streams; };

extern "C" int nrsc5hip_hdc_create(int nstreams, nrsc5hip_hdc **out)
{
    if (!out || nstreams < 1) return NRSC5HIP_EINVAL;
    nrsc5hip_hdc *h = new (std::nothrow) nrsc5hip_hdc();
    if (!h) return NRSC5HIP_ENOMEM;
    try { h->streams.resize((size_t)nstreams); } catch (...) { delete h; return NRSC5HIP_ENOMEM; }
    *out = h;
    return NRSC5HIP_OK;
}

extern "C" void nrsc5hip_hdc_destroy(nrsc5hip_hdc *h) { delete h; }

extern "C" int nrsc5hip_hdc_reset(nrsc5hip_hdc *h, int stream)
{
    if (!h || stream < 0 || stream >= (int)h->streams.size()) return NRSC5HIP_EINVAL;
    h->streams[stream].reset();
    return NRSC5HIP_OK;
}

// fcs16 (frame.c:138-144): PPP FCS-16, reflected polynomial 0x8408, good residue 0xf0b8
static unsigned fcs16(const uint8_t *p, unsigned n)
{
    unsigned crc = 0xffff;
    for (unsigned i = 0; i < n; i++) {
        crc ^= p[i];
        for (int k = 0; k < 8; k++) crc = (crc & 1u) ? (crc >> 1) ^ 0x8408u : crc >> 1;
    }
    return crc & 0xffffu;
}

static void ccc_message(Ccc &c, uint8_t *buf, unsigned len)     // process_fixed_ccc, frame.c:393-431
{
    unsigned n = 0;                                            // unescape_hdlc, frame.c:328-340
    for (unsigned i = 0; i < len; i++) { if (buf[i] == 0x7D) { ++i; buf[n++] = (uint8_t)(buf[i] | 0x20); } else buf[n++] = buf[i]; }
    if (n == 0 || c.fixed_ready) return;
    if (fcs16(buf, n) != 0xf0b8u) return;
    for (unsigned i = 0; i < 4; i++) {
        c.length[i] = 0;
        if (5 + i * 4 <= n) {
            const unsigned mode = buf[1 + i * 4] | (buf[2 + i * 4] << 8), length = buf[3 + i * 4] | (buf[4 + i * 4] << 8);
            if (mode == 0) c.length[i] = length;
        }
    }
    c.fixed_ready = 1;
}

extern "C" unsigned nrsc5hip_hdc_fixed_audio_end(nrsc5hip_hdc *h, int stream, int lc, const uint8_t *b, unsigned nbytes)
{
    if (!h || !b || nbytes == 0 || stream < 0 || stream >= (int)h->streams.size() || lc < 0 || lc > 2) return nbytes;
    Ccc &c = h->streams[stream].ccc[lc];
    unsigned p = nbytes - 1;                                   // process_fixed_data, frame.c:458-514
    if (c.sync_count < 2) {
        const uint8_t byte = b[p];
        const unsigned width = byte == 0 ? 1u : ((byte >> 4) == (byte & 0xf)) ? (byte & 0xfu) * 2u : 0u;      // sync_width, frame.c:448-456
        if (width > 0 && c.sync_width == width) c.sync_count++; else c.sync_count = 0;
        c.sync_width = width;
        if (c.sync_count < 2) return p;
    }
    p -= c.sync_width;
    for (unsigned i = 0; i < c.sync_width; i++) {              // parse_hdlc into ccc_buf, frame.c:369-391
        const uint8_t byte = b[p + i];
        if (byte == 0x7E) { if (c.idx >= 0) ccc_message(c, c.buf, (unsigned)c.idx); c.idx = 0; }
        else if (c.idx >= 0) { if (c.idx == (int)sizeof(c.buf)) { c.idx = -1; continue; } c.buf[c.idx++] = byte; }
    }
    if (!c.fixed_ready) return p;
    for (int i = 3; i >= 0; i--) p -= c.length[i];
    return p;
}

extern "C" int nrsc5hip_l2_apply_audio_end(nrsc5hip_l2_frame *ix, unsigned audio_end)
{
    if (!ix) return NRSC5HIP_EINVAL;
    for (unsigned k = 0; k < ix->n_pdu && k < NRSC5HIP_L2_MAX_PDUS; k++) {
        const nrsc5hip_l2_pdu &p = ix->pdu[k];
        bool cut = !(p.start < audio_end - 96u);                                  // while (offset < audio_end - RS_CODEWORD_LEN), unsigned
        if (!cut && p.start + p.la_location >= audio_end) cut = true;             // frame.c:548
        for (unsigned j = 0; !cut && j < p.nop && j < NRSC5HIP_L2_MAX_PACKETS; j++) if (p.loc[j] >= audio_end) cut = true;   // frame.c:554
        if (cut) { ix->n_pdu = k; ix->status = NRSC5HIP_L2_AUDIO_END; ix->end_offset = p.start; if (k == 0 && p.start == 0) ix->lost_sync = 0; return (int)k; }
        if (p.hef && !p.skipped && p.psd_off > audio_end) return -1;              // parse_hef(.., audio_end - offset) would have been cut short
    }
    // a header failure recorded beyond the audio region never happened for the reference
    if (ix->status == NRSC5HIP_L2_HEADER_RS && !(ix->end_offset < audio_end - 96u)) { ix->status = NRSC5HIP_L2_AUDIO_END; ix->lost_sync = 0; }
    return (int)ix->n_pdu;
}

extern "C" int nrsc5hip_hdc_frame_reset(nrsc5hip_hdc *h, int stream)
{
    if (!h || stream < 0 || stream >= (int)h->streams.size()) return NRSC5HIP_EINVAL;
    for (Ccc &c : h->streams[stream].ccc) c.reset();
    return NRSC5HIP_OK;
}

extern "C" int nrsc5hip_hdc_push_frame(nrsc5hip_hdc *h, int stream, int lc, const nrsc5hip_l2_frame *ix_in, const uint8_t *pdu_bytes)
{
    if (!h || !ix_in || !pdu_bytes || stream < 0 || stream >= (int)h->streams.size() || lc < 0 || lc > 2) return NRSC5HIP_EINVAL;
    Stream &st = h->streams[stream];
    const nrsc5hip_l2_frame *ix = ix_in;
    std::unique_ptr<nrsc5hip_l2_frame> cut;
    if ((ix_in->pci & 0xFFFFFCu) == (0x3634CEu & 0xFFFFFCu)) {
        // PCI_FIXED: has_fixed() but !has_audio() (frame.c:138-151) -- frame_process still runs process_fixed_data on it (sync width
        // tracking, CCC messages, fixed_ready) before it returns, and later audio + fixed frames of the channel depend on that state
        (void)nrsc5hip_hdc_fixed_audio_end(h, stream, lc, pdu_bytes, ix_in->nbytes);
        return NRSC5HIP_OK;
    }
    if (NRSC5HIP_L2_PCI_HAS_FIXED(ix_in->pci)) {
        const unsigned audio_end = nrsc5hip_hdc_fixed_audio_end(h, stream, lc, pdu_bytes, ix_in->nbytes);
        cut.reset(new nrsc5hip_l2_frame(*ix_in));
        if (nrsc5hip_l2_apply_audio_end(cut.get(), audio_end) < 0) return NRSC5HIP_EINVAL;
        ix = cut.get();
    }
    for (unsigned k = 0; k < ix->n_pdu && k < NRSC5HIP_L2_MAX_PDUS; k++) {
        const nrsc5hip_l2_pdu &p = ix->pdu[k];
        if (p.skipped || p.prog_num >= MAX_PROGRAMS || p.stream_id >= MAX_STREAMS) continue;     // frame.c:559-564
        st.audio_offset[p.prog_num][p.stream_id] = p.align_offset;                                // output_align
        if (p.stream_id != 0) continue;                                                             // output.c:52-53
        if (!st.elastic[p.prog_num]) st.elastic[p.prog_num].reset(new Elastic());
        Elastic &el = *st.elastic[p.prog_num];
        unsigned off = p.audio_off;
        for (unsigned j = 0; j < p.nop && j < NRSC5HIP_L2_MAX_PACKETS; j++) {                      // frame.c:613-640 -> output_push
            const unsigned size = (unsigned)p.loc[j] - off;
            const uint8_t *data = pdu_bytes + off;
            const unsigned flags = ((j < 32 ? p.crc_bad_lo >> j : p.crc_bad_hi >> (j - 32)) & 1u) ? FLAG_CRC_ERROR : 0u;
            const unsigned shape = (j == 0 && p.pfirst) ? SHAPE_HALF_BACK : (j == p.nop - 1u && p.plast) ? SHAPE_HALF_FRONT : SHAPE_FULL;
            Packet &pkt = el.packets[(p.elastic_seq + j) % ELASTIC_LEN];
            if (shape == SHAPE_HALF_BACK && pkt.shape == SHAPE_HALF_FRONT) {
                pkt.flags |= flags; pkt.shape = SHAPE_FULL;
                if (!(pkt.flags & FLAG_CRC_ERROR)) {
                    if (pkt.data.size() < pkt.size + size) pkt.data.resize(pkt.size + size);
                    memcpy(pkt.data.data() + pkt.size, data, size);
                    pkt.size += size;
                } else pkt.size = 0;
            } else if (shape != SHAPE_HALF_BACK) {
                pkt.flags = flags; pkt.shape = shape;
                if (!(pkt.flags & FLAG_CRC_ERROR)) {
                    if (pkt.data.size() < size) pkt.data.resize(size);
                    memcpy(pkt.data.data(), data, size);
                    pkt.size = size;
                } else pkt.size = 0;
            }
            off = (unsigned)p.loc[j] + 1u;
        }
    }
    return NRSC5HIP_OK;
}

extern "C" int nrsc5hip_hdc_advance(nrsc5hip_hdc *h, int stream, int mode, nrsc5hip_hdc_cb cb, void *opaque)
{
    if (!h || stream < 0 || stream >= (int)h->streams.size()) return NRSC5HIP_EINVAL;
    Stream &st = h->streams[stream];
    const int audio_frames = mode == NRSC5HIP_MODE_FM ? 2 : 4;                                       // output.c:103
    int delivered = 0;
    for (unsigned program = 0; program < (unsigned)MAX_PROGRAMS; program++) {
        int &ao = st.audio_offset[program][0];
        if (ao == -1) continue;
        for (int f = 0; f < audio_frames; f++) {
            if (st.elastic[program]) {
                Packet &pkt = st.elastic[program]->packets[ao];
                if (pkt.shape == SHAPE_FULL) {                                                         // nrsc5_report_hdc, nrsc5.c:709-728
                    if (cb) cb(opaque, stream, program, pkt.size ? pkt.data.data() : nullptr, pkt.size, (pkt.flags & FLAG_CRC_ERROR) ? 1u : 0u);
                    delivered++;
                }
                pkt.size = 0; pkt.flags = 0; pkt.shape = SHAPE_NONE;                                    // pkt_reset
            }
            ao = (ao + 1) % ELASTIC_LEN;
        }
    }
    return delivered;
}

extern "C" size_t nrsc5hip_hdc_adts(const uint8_t *data, unsigned count, uint8_t *out)
{
    // write_adts_header (main.c:182-204): sync 0xFFF, MPEG-4, layer 0, no CRC, AAC-LC (profile field 1), 22050 Hz (index 7),
    // 2 channels, frame length = count + 7, buffer fullness 0x7FF, 1 AAC frame
    if (!out) return 0;
    const unsigned len = count + 7;
    uint64_t bits = 0;
    auto add = [&](uint64_t v, int n) { bits = (bits << n) | (v & ((1ull << n) - 1)); };
    add(0xFFF, 12); add(0, 1); add(0, 2); add(1, 1); add(1, 2); add(7, 4); add(0, 1); add(2, 3); add(0, 1); add(0, 1); add(0, 1); add(0, 1);
    add(len, 13); add(0x7FF, 11); add(0, 2);                                                            // 56 bits
    for (int k = 0; k < 7; k++) out[k] = (uint8_t)(bits >> (8 * (6 - k)));
    if (count && data) memcpy(out + 7, data, count);
    return len;
}

extern "C" size_t nrsc5hip_hdc_host_bytes(const nrsc5hip_hdc *h)
{
    if (!h) return 0;
    size_t n = sizeof(*h) + h->streams.capacity() * sizeof(Stream);
    for (const Stream &st : h->streams)
        for (const auto &e : st.elastic)
            if (e) { n += sizeof(Elastic); for (const Packet &p : e->packets) n += p.data.capacity(); }
    return n;
}
