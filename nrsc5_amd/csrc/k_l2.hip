// L2 audio transport index: frame_push + the walk of frame_process on the device, as a post-pass over decoded
// logical frames that are already in HBM.  One workgroup per frame.
//   frame_push: PCI extraction + per-byte bit reversal          frame.c:645-714   -> l2_unpack (closed-form bit map)
//   has_audio / has_fixed                                       frame.c:138-151
//   fix_header, decode_rs_char                                  frame.c:153-179   -> rs255_247_decode (l2_header.h)
//   parse_header, calc_lc_bits, calc_avg_packets, parse_location frame.c:181-196,267-328
//   parse_hef                                                   frame.c:198-265
//   elastic-buffer sequence / output_align offset               frame.c:593-600
//   crc8 of every packet                                        frame.c:130-136,613-640 -> one work-item per packet
// Output: nrsc5hip_l2_frame (what frame_process hands to output_align / parse_hdlc / output_push, as offsets) and,
// optionally, the PDU bytes with the RS-corrected headers those offsets refer to.
// Frames whose PCI announces fixed-data sub-channels next to audio (PCI_AUDIO_FIXED / _OPP): audio_end then depends on the
// CCC state machine of process_fixed_data (frame.c:458-514), which is host state.  Every condition of the walk that
// involves audio_end only gets stricter as audio_end shrinks, and process_fixed_data never returns more than length - 1, so
// the device walks with audio_end = nbytes - 1 (a superset of what the reference will accept) and the host cuts the index
// back with the true value (nrsc5hip_l2_apply_audio_end; nrsc5hip_hdc_push_frame does both).
#include <hip/hip_runtime.h>
#include "nrsc5hip.h"
#include "kernels.h"
#include "l2_header.h"

namespace nrsc5 {

__device__ inline int stream_of(const int *ids, int idx) { return ids ? ids[idx] : idx; }

struct L2IndexSmem {
    L2Smem rs;
    uint8_t crc_tab[256];
    uint8_t bytes[L2_MAX_BYTES + 3];
    int go;                         // walk state broadcast: 1 = a PDU's packets are ready for the CRC pass, 0 = done
    unsigned nop, audio_off, crc_bad[2];
    uint16_t loc[64];
    nrsc5hip_l2_pdu hdr;            // the PDU being parsed (work-item 0)
};

__device__ inline bool l2_layout(int nbits, unsigned &start, unsigned &step, unsigned &pci_len)      // frame.c:651-686
{
    switch (nbits) {
    case 146176: start = 146176 - 30000; step = 1248; pci_len = 24; return true;
    case 4608:   start = 120; step = 184;  pci_len = 24; return true;
    case 2304:   start = 120; step = 88;   pci_len = 24; return true;
    case 3750:   start = 120; step = 160;  pci_len = 22; return true;
    case 24000:  start = 120; step = 992;  pci_len = 24; return true;
    case 30000:  start = 120; step = 1240; pci_len = 24; return true;
    }
    return false;
}

// logical bit i of the frame (after frame_push's swap of the bit order inside each group of 8, frame.c:690-693)
__device__ inline unsigned l2_logical_bit(const uint32_t *w, unsigned i, unsigned len)
{
    const unsigned b0 = i & ~7u, blen = (len - b0 < 8u) ? len - b0 : 8u;
    return l2_bit(w, b0 + blen - 1u - (i & 7u));
}

// parse_hef (frame.c:198-265) on buf[0 .. length): returns the bytes consumed (length when a field runs off the end)
__device__ inline unsigned l2_parse_hef(const uint8_t *buf, unsigned length, nrsc5hip_l2_pdu &h)
{
    unsigned at = 0;
    for (;;) {
        if (at >= length) return length;
        const unsigned b = buf[at];
        switch ((b >> 4) & 7u) {
        case 0: h.class_ind = (uint8_t)(b & 0xfu); break;
        case 1:
            h.prog_num = (uint8_t)((b >> 1) & 7u);
            if (b & 1u) {
                if (at + 2 >= length) return length;
                h.hef_pdu_len = (uint16_t)(((buf[at + 1] & 0x7fu) << 7) | (buf[at + 2] & 0x7fu));
                at += 2;
            }
            break;
        case 2:
            if (at + 1 >= length) return length;
            h.access = (uint8_t)((b >> 3) & 1u);
            h.prog_type = (uint8_t)(((b & 1u) << 7) | (buf[at + 1] & 0x7fu));
            at += 1;
            break;
        case 3:
            if (b & 8u) { if (at + 4 >= length) return length; at += 4; }
            else        { if (at + 3 >= length) return length; at += 3; }
            break;
        case 4:
            if (b & 8u) {
                if (at + 3 >= length) return length;
                h.applied_services = (uint8_t)(b & 7u);
                h.pdu_marker = ((uint32_t)(buf[at + 1] & 0x7fu) << 14) | ((uint32_t)(buf[at + 2] & 0x7fu) << 7) | (buf[at + 3] & 0x7fu);
                at += 3;
            } else { if (at + 1 >= length) return length; at += 1; }
            break;
        default: break;
        }
        if (!(buf[at++] & 0x80u)) break;
    }
    return at;
}

// One workgroup of 256 work-items indexes one frame: w = packed frame bits, dst (may be null) receives the PDU bytes.
__device__ inline void l2_index_frame(L2IndexSmem &sm, const uint32_t *w, unsigned len, nrsc5hip_l2_frame &out, uint8_t *dst)
{
    const int tid = (int)threadIdx.x;

    for (unsigned k = (unsigned)tid; k < sizeof(nrsc5hip_l2_frame) / 4u; k += 256u) ((uint32_t *)&out)[k] = 0u;
    {   // CRC-8 table: MSB-first, polynomial 0x31 (the table of frame.c:60-90)
        unsigned c = (unsigned)tid;
        for (int k = 0; k < 8; k++) c = (c & 0x80u) ? ((c << 1) ^ 0x31u) & 0xffu : (c << 1) & 0xffu;
        sm.crc_tab[tid] = (uint8_t)c;
    }
    if (tid == 0) l2_gf_init(sm.rs);
    unsigned start0 = 0, step = 1, pci_len = 0;
    const bool known = l2_layout((int)len, start0, step, pci_len);
    const unsigned nbytes = known ? (len - pci_len) / 8u : 0u;

    // PDU bit p sits at logical index p + c, c = number of PCI positions at or below it: the h-th PCI bit is logical index
    // start0 + step h, i.e. it precedes PDU bit start0 + (step - 1) h
    for (unsigned q = (unsigned)tid; q < nbytes; q += 256u) {
        unsigned val = 0;
        for (unsigned r = 0; r < 8u; r++) {
            const unsigned p = 8u * q + r;
            unsigned c = 0;
            if (p >= start0) { c = (p - start0) / (step - 1u) + 1u; if (c > pci_len) c = pci_len; }
            val |= l2_logical_bit(w, p + c, len) << (7u - r);
        }
        sm.bytes[q] = (uint8_t)val;
    }
    __syncthreads();                                        // also orders the zero fill of `out` before thread 0's stores
    __threadfence_block();

    unsigned offset = 0, status = NRSC5HIP_L2_END, n_pdu = 0, lost = 0, audio_end = nbytes;   // audio_end: work-item 0's
    bool walking = known;
    if (tid == 0 && known) {
        unsigned pci = 0;
        for (unsigned h = 0; h < pci_len; h++) pci |= l2_logical_bit(w, start0 + step * h, len) << (23u - h);
        out.pci = pci; out.nbytes = nbytes;
        const unsigned p = pci & 0xFFFFFCu;
        if (p == (0x3634CEu & 0xFFFFFCu)) { status = NRSC5HIP_L2_NO_AUDIO; walking = false; }
        else if (p == (0xE3634Cu & 0xFFFFFCu) || p == (0x8D8D33u & 0xFFFFFCu)) audio_end = nbytes - 1u;   // has_fixed: see the header
    }
    const bool is_p1 = (len == 146176u || len == 3750u);   // length == MAX_PDU_LEN || P1_PDU_LEN_AM, frame.c:537

    for (;;) {
        if (tid == 0) {
            sm.go = 0;
            while (walking && offset < audio_end - 96u) {                 // unsigned, as frame.c:525
                const unsigned start = offset;
                if (n_pdu == NRSC5HIP_L2_MAX_PDUS) { status = NRSC5HIP_L2_TOO_MANY_PDUS; walking = false; break; }
                nrsc5hip_l2_pdu &d = out.pdu[n_pdu];
                for (int i = 0; i < 96; i++) sm.rs.pdu[i] = sm.bytes[offset + i];
                for (int i = 0; i < 159; i++) sm.rs.r[i] = 0;
                for (int i = 0; i < 96; i++) sm.rs.r[254 - i] = sm.rs.pdu[i];
                const int corr = rs255_247_decode(sm.rs);
                bool ok = corr >= 0;
                for (int i = 0; ok && i < 159; i++) if (sm.rs.r[i]) ok = false;
                if (!ok) { status = NRSC5HIP_L2_HEADER_RS; lost = (is_p1 && offset == 0) ? 1u : 0u; walking = false; break; }
                for (int i = 0; i < 96; i++) sm.bytes[offset + i] = sm.rs.r[254 - i];
                const uint8_t *b = sm.bytes + offset;                     // parse_header, frame.c:181-196
                nrsc5hip_l2_pdu &h = sm.hdr;
                for (unsigned k = 0; k < sizeof(h) / 4u; k++) ((uint32_t *)&h)[k] = 0u;
                h.start = start; h.rs_corrections = (uint8_t)corr;
                h.codec_mode = (uint8_t)(b[8] & 0xfu); h.stream_id = (uint8_t)((b[8] >> 4) & 3u);
                h.pdu_seq = (uint8_t)((b[8] >> 6) | ((b[9] & 1u) << 2));
                h.blend_control = (uint8_t)((b[9] >> 1) & 3u); h.per_stream_delay = (uint8_t)(b[9] >> 3);
                h.common_delay = (uint8_t)(b[10] & 0x3fu); h.latency = (uint8_t)((b[10] >> 6) | ((b[11] & 1u) << 2));
                h.pfirst = (uint8_t)((b[11] >> 1) & 1u); h.plast = (uint8_t)((b[11] >> 2) & 1u);
                h.seq = (uint8_t)((b[11] >> 3) | ((b[12] & 1u) << 5)); h.nop = (uint8_t)((b[12] >> 1) & 0x3fu);
                h.hef = (uint8_t)(b[12] >> 7); h.la_location = b[13];
                offset += 14u;
                unsigned lc_bits = 16u, avg = 32u;                        // calc_lc_bits / calc_avg_packets
                switch (h.codec_mode) {
                case 1: case 2: case 3: if (h.stream_id == 0) { lc_bits = 12u; avg = 4u; } break;
                case 10: lc_bits = 12u; if (h.stream_id != 0) avg = 4u; break;
                case 13: lc_bits = 12u; avg = 4u; break;
                default: break;
                }
                const unsigned loc_bytes = (lc_bits * h.nop + 4u) / 8u;
                if (start + h.la_location + 1u < offset + loc_bytes || start + h.la_location >= audio_end) { status = NRSC5HIP_L2_BAD_LOCATORS; walking = false; break; }
                bool bad = false;
                const uint8_t *lb = sm.bytes + offset;
                for (unsigned k = 0; k < h.nop; k++) {                    // parse_location, frame.c:317-328
                    unsigned loc;
                    if (lc_bits == 16u) loc = ((unsigned)lb[2 * k + 1] << 8) | lb[2 * k];
                    else if ((k & 1u) == 0) loc = ((unsigned)(lb[k / 2 * 3 + 1] & 0xfu) << 8) | lb[k / 2 * 3];
                    else loc = ((unsigned)lb[k / 2 * 3 + 2] << 4) | (lb[k / 2 * 3 + 1] >> 4);
                    const unsigned prev = k ? (unsigned)h.loc[k - 1] - start : 0u;
                    if ((k == 0 && loc <= h.la_location) || (k > 0 && loc <= prev) || start + loc >= audio_end) { bad = true; break; }
                    h.loc[k] = (uint16_t)(start + loc);
                }
                if (bad) { status = NRSC5HIP_L2_BAD_LOCATORS; walking = false; break; }
                offset += loc_bytes;
                if (h.stream_id >= 2u) {                                  // MAX_STREAMS, frame.c:559-564
                    if (h.nop == 0) { status = NRSC5HIP_L2_BAD_STREAM; walking = false; break; }
                    h.skipped = 1; d = h; n_pdu++;
                    offset = (unsigned)h.loc[h.nop - 1] + 1u;
                    continue;
                }
                if (h.hef) offset += l2_parse_hef(sm.bytes + offset, audio_end - offset, h);
                h.elastic_seq = (uint8_t)((64u + h.seq - h.pfirst) % 64u);                  // frame.c:593-598
                unsigned oo = (64u + h.pdu_seq * avg - h.latency * 2u) % 64u;
                if (((64u + h.elastic_seq - oo) % 64u) >= 32u) oo = (oo + 32u) % 64u;
                h.align_offset = (uint8_t)oo;
                h.psd_off = offset; h.psd_len = (int32_t)(start + h.la_location + 1u) - (int32_t)offset;
                if (h.psd_len < 0) { status = NRSC5HIP_L2_HEF_OVERRUN; walking = false; break; }
                offset = start + h.la_location + 1u;
                h.audio_off = offset;
                d = h;
                sm.nop = h.nop; sm.audio_off = offset; sm.crc_bad[0] = sm.crc_bad[1] = 0u;
                for (unsigned k = 0; k < h.nop; k++) sm.loc[k] = h.loc[k];
                if (h.nop) offset = (unsigned)h.loc[h.nop - 1] + 1u;
                sm.go = 1;
                break;
            }
        }
        __syncthreads();
        if (!sm.go) break;
        if ((unsigned)tid < sm.nop) {                                     // crc8 over payload + CRC byte == 0, frame.c:615-616
            const unsigned from = tid ? (unsigned)sm.loc[tid - 1] + 1u : sm.audio_off, to = sm.loc[tid];
            unsigned crc = 0xffu;
            for (unsigned k = from; k <= to; k++) crc = sm.crc_tab[crc ^ sm.bytes[k]];
            if (crc) atomicOr(&sm.crc_bad[tid >> 5], 1u << (tid & 31));
        }
        __syncthreads();
        if (tid == 0) { out.pdu[n_pdu].crc_bad_lo = sm.crc_bad[0]; out.pdu[n_pdu].crc_bad_hi = sm.crc_bad[1]; n_pdu++; }
    }
    if (tid == 0) {
        out.n_pdu = n_pdu; out.status = known ? status : (unsigned)NRSC5HIP_L2_BAD_LENGTH;
        out.end_offset = offset; out.lost_sync = lost;
    }
    if (dst) for (unsigned q = (unsigned)tid; q < nbytes; q += 256u) dst[q] = sm.bytes[q];
}

__global__ __launch_bounds__(256) void k_l2_index(const L2Job *jobs, nrsc5hip_l2_frame *frames, uint8_t *bytes_out, long long stride)
{
    __shared__ L2IndexSmem sm;
    const L2Job job = jobs[blockIdx.x];
    l2_index_frame(sm, job.words, (unsigned)job.nbits, frames[blockIdx.x], bytes_out ? bytes_out + (long long)blockIdx.x * stride : nullptr);
}

// Fused variant (engine option l2_index): runs on the decode stream right behind k_p1_traceback and indexes the P1 frame
// that kernel just finished for this stream, if any, into the slot's entry of the index ring.
__global__ __launch_bounds__(256) void k_l2_index_window(DevBuffers db, const int *ids, int parity)
{
    __shared__ L2IndexSmem sm;
    const int s = stream_of(ids, blockIdx.x);
    StreamState &st = db.state[s];
    const int slot = st.p1_l2slot[parity] - 1;                 // block-uniform; written by k_p1_traceback
    if (slot < 0) return;
    l2_index_frame(sm, db.p1_ring + ((size_t)s * db.p1_slots + slot) * P1_WORDS, (unsigned)P1_LEN, db.l2_ring[(size_t)s * db.p1_slots + slot], nullptr);
    if (threadIdx.x == 0) st.p1_l2slot[parity] = 0;
}

void launch_l2_index_window(const DevBuffers &db, int nstreams, const int *stream_ids, int parity, hipStream_t st)
{
    hipLaunchKernelGGL(k_l2_index_window, dim3(nstreams), dim3(256), 0, st, db, stream_ids, parity);
}

// P3 / P4 frames of the extended sidebands: k_px_decode leaves PxJob::pad = 1 on every frame it decoded in this window
__global__ __launch_bounds__(256) void k_l2_index_px_window(DevBuffers db, const int *ids, int parity)
{
    __shared__ L2IndexSmem sm;
    const int s = stream_of(ids, blockIdx.y), j = blockIdx.x, ch = j & 1;   // j = pair slot * 2 + channel
    PxJob &job = db.px_job[((size_t)s * NWIN + parity) * 16 + j];
    if (job.pad != 1) return;                                  // block-uniform (fresh entries are 0xff-filled)
    const size_t fr = ((size_t)s * db.px_slots + job.slot) * 2 + ch;
    l2_index_frame(sm, db.px_ring + fr * PX_WORDS, (unsigned)job.len, db.l2_px_ring[fr], nullptr);
    if (threadIdx.x == 0) job.pad = 0;
}

void launch_l2_index_px_window(const DevBuffers &db, int nstreams, const int *stream_ids, int parity, hipStream_t st)
{
    hipLaunchKernelGGL(k_l2_index_px_window, dim3(16, nstreams), dim3(256), 0, st, db, stream_ids, parity);
}

// AM: frame r = 0..7 is the P1 frame of block r (3750 bits), 8 the P3 frame (24000 / 30000 bits) of the L1 frame in `slot`
__device__ inline void l2_index_am_frame(L2IndexSmem &sm, const DevBuffers &db, int s, int slot, int r, bool ma3)
{
    const uint32_t *words = db.p1_ring + ((size_t)s * db.p1_slots + slot) * P1_WORDS + (r < 8 ? r * AM_P1_WORDS : AM_P3_WORD0);
    const unsigned nbits = r < 8 ? (unsigned)AM_P1_LEN : (unsigned)(ma3 ? AM_P3_LEN_MA3 : AM_P3_LEN_MA1);
    l2_index_frame(sm, words, nbits, db.l2_am_ring[((size_t)s * db.p1_slots + slot) * 9 + r], nullptr);
}

// window pipeline: the last k_am_decode of an L1 frame leaves AmJob::pad = mask of the frames it produced
__global__ __launch_bounds__(256) void k_l2_index_am_window(DevBuffers db, const int *ids, int parity)
{
    __shared__ L2IndexSmem sm;
    const int s = stream_of(ids, blockIdx.y), r = blockIdx.x;
    AmJob &job = db.am_job[(size_t)s * NWIN + parity];
    if (!((job.pad >> r) & 1)) return;                         // block-uniform
    l2_index_am_frame(sm, db, s, job.slot, r, job.psmi == AM_MA3);
    if (threadIdx.x == 0) atomicAnd(&job.pad, ~(1 << r));
}

void launch_l2_index_am_window(const DevBuffers &db, int nstreams, const int *stream_ids, int parity, hipStream_t st)
{
    hipLaunchKernelGGL(k_l2_index_am_window, dim3(9, nstreams), dim3(256), 0, st, db, stream_ids, parity);
}

// in order: what k_am_viterbi delivered in this step (same conditions)
__global__ __launch_bounds__(256) void k_l2_index_am_step(DevBuffers db, const int *ids)
{
    __shared__ L2IndexSmem sm;
    const int s = stream_of(ids, blockIdx.y);
    const StreamState &st = db.state[s];
    const AmStream &am = db.am[s];
    if (!st.active || am.dec_bc < 0 || am.am_diversity_wait != 0) return;      // block-uniform
    const int bc = am.dec_bc;
    if (blockIdx.x == 1 && (bc != 7 || am.dec_rdbi)) return;
    l2_index_am_frame(sm, db, s, am.frame_slot, blockIdx.x == 0 ? bc : 8, am.dec_psmi == AM_MA3);
}

void launch_l2_index_am_step(const DevBuffers &db, int nstreams, const int *stream_ids, hipStream_t st)
{
    hipLaunchKernelGGL(k_l2_index_am_step, dim3(2, nstreams), dim3(256), 0, st, db, stream_ids);
}

void launch_l2_index(const L2Job *jobs, int njobs, nrsc5hip_l2_frame *frames, uint8_t *bytes_out, long long stride, hipStream_t st)
{
    if (njobs < 1) return;
    hipLaunchKernelGGL(k_l2_index, dim3(njobs), dim3(256), 0, st, jobs, frames, bytes_out, stride);
}

}  // namespace nrsc5
