/* TEST INFRASTRUCTURE -- CPU restatement of the reference's IQ -> P1/PIDS hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this
 * library, and only as the checker.  The product path (nrsc5_amd/csrc, libnrsc5hip.so)
 * never links, loads or calls anything in oracle/.
 *
 * Pinning: the reference ships no unit tests or stage vectors; this
 * restatement is pinned against the UNMODIFIED reference compiled into oracle/_ref/
 * (tests/test_oracle_vs_reference.py: exact Q15 stream, soft bits, PIDS/P1 frames,
 * per-block timing/CFO trace) and against the fixtures under tests/golden/ that the
 * same reference build produced.  The FFT (third-party fftw3f in the reference,
 * absent here) is oracle/cpu_fft.c in both builds: parity unpinned at that boundary,
 * tolerance-checked.
 *
 * Each function cites the reference lines it restates.  FM: primary-main processing, service mode
 * MP1 end to end (nrsc5_oracle.c).  AM: MA1 and MA3 (nrsc5_oracle_am.c).
 */
#pragma once
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { int16_t r, i; } orc_c16;

enum { ORC_SYNC_NONE = 0, ORC_SYNC_COARSE = 1, ORC_SYNC_FINE = 2 };

/* ordered log record kinds: identical numbering/payloads to oracle/ref_shim/ref_harness.c */
enum {
    ORC_REC_BLOCK = 1, ORC_REC_STATE, ORC_REC_SOFT, ORC_REC_PIDS, ORC_REC_FRAME, ORC_REC_SYNC,
    ORC_REC_LOST_SYNC, ORC_REC_MER, ORC_REC_BER, ORC_REC_HDC, ORC_REC_VIT,
    ORC_REC_AMSYM,                /* AM: u32 bc + pl[800] + pu[800] + s[800] + t[800] hard symbols of one block */
    ORC_REC_PXSOFT                /* FM: u32 channel (0 PX1, 1 PX2), u32 bc, u32 len + soft bits handed to decode_push_px1/px2 */
};
enum { ORC_TAP_Q15 = 1, ORC_TAP_FFT = 2, ORC_TAP_SOFT = 4 };

/* ---- stage functions: each is the CPU twin of one HIP kernel ------------------------- */

/* K1  input.c:52-69 + firdecim_q15.c:137-165.  cu8 -> Q15 -> 15-tap half-band, 2:1.
 * hist = last 14 Q15 samples pushed (zeros for a fresh session); nbytes % 4 == 0.
 * Returns the number of output samples (nbytes / 4). */
size_t orc_halfband_fm_cu8(orc_c16 hist[14], const uint8_t *iq, size_t nbytes, orc_c16 *out);

/* K2a firdecim_q15.c:95-109,154-158 with acquire.c:28-61 taps.  1-in/1-out 32-tap band-select
 * FIR; hist = last 31 samples pushed. */
void orc_fir32_fm(orc_c16 hist[31], const orc_c16 *in, size_t n, orc_c16 *out);

/* K2b acquire.c:122-157.  Cyclic-prefix correlation over a 33-symbol Q15 window already
 * FIR-filtered; returns samperr and the correlation value at the peak. */
void orc_cp_correlate_fm(const orc_c16 *filtered /*71280*/, int *samperr, float peak[2]);

/* K6  decode.c:296-342: gather + depuncture of one L1 frame / one block. */
void orc_deinterleave_p1(const int8_t *pm /*16*23040*/, int8_t *out /*438528*/);
void orc_deinterleave_pids(const int8_t *pm /*16*23040*/, unsigned bc, int8_t *out /*240*/);

/* K6x decode.c:344-376: interleaver IV of the extended sidebands (PX1 -> P3, PX2 -> P4).  frame_len = 4608 (MP3/MP11)
 * or 2304 (MP2); pair = the 2 * frame_len soft bits of a block pair; mem[32 * frame_len], pos, taken[4], ready persist. */
void orc_interleave_px(int8_t *mem, unsigned *pos, unsigned taken[4], int *ready, const int8_t *pair, unsigned frame_len, int8_t *out /*3*frame_len*/);

/* K7  conv_dec.c:402-453 + conv_gen.h:32-101: tail-biting soft Viterbi, rate 1/3.
 * k = 7 (gens 0133,0171,0165) or k = 9 with the three generators given. */
int orc_viterbi(const int8_t *in /*3*len*/, int len, int k, const unsigned gens[3], uint8_t *out /*len*/);
int orc_viterbi_k7(const int8_t *in, int len, uint8_t *out);

/* K8  decode.c:279-294 / 234-265 */
void orc_descramble(uint8_t *bits, unsigned len);
int orc_bit_errors_k7(const int8_t *coded, const uint8_t *decoded, int len);

/* forward FFT used by the mixer stage (oracle/cpu_fft.c) is declared in cpu_fft.h */

/* ---- whole-path stream object: mirrors input_push_cu8/cs16 (input.c:96-124) ----------- */
typedef struct orc_stream orc_stream;

/* L2 feedback hook (frame.c:535-540): called with each decoded P1 frame (146176 bits, one per
 * byte, already descrambled); return nonzero to drop the stream to SYNC_NONE exactly where
 * the reference's frame_process would. */
typedef int (*orc_p1_hook)(void *user, const uint8_t *bits, unsigned len);

orc_stream *orc_open(void);
void orc_close(orc_stream *s);
void orc_reset(orc_stream *s);                       /* input_reset, fresh-session semantics */
void orc_set_taps(orc_stream *s, unsigned mask, unsigned fft_limit_blocks);
void orc_set_p1_hook(orc_stream *s, orc_p1_hook hook, void *user);
void orc_push_cu8(orc_stream *s, const uint8_t *iq, uint32_t nbytes);   /* nbytes % 4 == 0 */
void orc_push_cs16(orc_stream *s, const int16_t *iq, uint32_t n);       /* n % 2 == 0 */
void orc_force_resync(orc_stream *s);                /* input_set_sync_state(NONE) from L2 */
size_t orc_buf(orc_stream *s, int which, const uint8_t **p);   /* 0 log, 1 q15, 2 fft */
void orc_clear_bufs(orc_stream *s);

/* per-block internal state, for stage-level parity of the HIP sync kernel */
typedef struct {
    int32_t sync_state, bc, psmi, cfo_wait, samperr_next, mer_cnt, acq_cfo, keep_extra;
    float angle_next, prev_angle, phase_re, phase_im, error_lb, error_ub;
    float costas_freq[30], costas_phase[30];   /* refs: lower i=0..14 at [2i], upper at [2i+1] */
} orc_sync_snapshot;
void orc_snapshot(const orc_stream *s, orc_sync_snapshot *out);

/* ---- the L2 -> L1 feedback (nrsc5_oracle_l2.c) ------------------------------------------------ */
/* frame.c:516-540 + fix_header + RS(255,247): 1 = frame_process keeps sync after this P1 frame, 0 = it drops to NONE */
int orc_rs255_247_decode(uint8_t r[255]);
int orc_l2_first_header_ok(const uint8_t *bits, unsigned len);
/* pids.c:52-86,1032-1050: does pids_frame_push accept this PIDS frame (CRC-12)? */
int orc_pids_crc_ok(const uint8_t bits[80]);

/* The audio-transport walk of frame_push / frame_process (frame.c:516-714) restated as an index: PDU bytes (PCI removed,
 * bit order restored, RS-corrected headers) plus, per audio PDU, the header fields, the PSD span and every packet with
 * its CRC-8 verdict -- i.e. exactly what frame_process hands to output_align / parse_hdlc / output_push. */
#define ORC_L2_MAX_PDUS 16
enum { ORC_L2_END = 0, ORC_L2_NO_AUDIO, ORC_L2_FIXED_DATA, ORC_L2_HEADER_RS, ORC_L2_BAD_LOCATORS, ORC_L2_TOO_MANY_PDUS,
       ORC_L2_HEF_OVERRUN, ORC_L2_BAD_STREAM };
typedef struct orc_l2_pdu {
    uint32_t start, psd_off; int32_t psd_len; uint32_t audio_off, crc_bad_lo, crc_bad_hi, pdu_marker;
    uint16_t hef_pdu_len, loc[64];
    uint8_t codec_mode, stream_id, pdu_seq, blend_control, per_stream_delay, common_delay, latency, pfirst, plast, seq, nop,
            hef, la_location, rs_corrections, class_ind, prog_num, access, prog_type, applied_services, elastic_seq,
            align_offset, skipped;
} orc_l2_pdu;
typedef struct orc_l2_frame {
    uint32_t pci, nbytes, n_pdu, status, end_offset, lost_sync;
    orc_l2_pdu pdu[ORC_L2_MAX_PDUS];
} orc_l2_frame;
/* bits = the frame as handed to frame_push (one bit per byte); len one of 146176, 4608, 2304, 3750, 24000, 30000.
 * bytes (may be NULL): room for (len - pci_len) / 8 PDU bytes.  Returns 0, or -1 for an unknown length. */
int orc_l2_index(const uint8_t *bits, unsigned len, orc_l2_frame *out, uint8_t *bytes);
uint8_t orc_crc8(const uint8_t *p, unsigned n);

/* ---- AM (nrsc5_oracle_am.c) ------------------------------------------------------------ */

/* K1-AM  input.c:52-94: cu8 -> (Q15 >> 4) -> five cascaded 15-tap half-bands, 32:1.  Any nbytes % 4 == 0;
 * stage phases carry over between calls.  Returns the number of output samples. */
typedef struct orc_am_decim orc_am_decim;
orc_am_decim *orc_am_decim_new(void);
void orc_am_decim_free(orc_am_decim *d);
size_t orc_am_decimate_cu8(orc_am_decim *d, const uint8_t *iq, size_t nbytes, orc_c16 *out);

/* K2a-AM firdecim_q15.c:95-109 with the acquire.c:63-96 taps */
void orc_am_fir32(orc_c16 hist[31], const orc_c16 *in, size_t n, orc_c16 *out);

/* K6-AM  interleaver_ma1 (decode.c:74-231) incl. the 3-frame diversity delay lines (54000 bits each, updated in
 * place), and the PIDS bit gather of decode_process_pids_am (decode.c:476-501) */
void orc_am_deinterleave(int psmi, const uint8_t *pl, const uint8_t *pu, const uint8_t *s, const uint8_t *t,
                         uint8_t *ml_q, uint8_t *mu_q, uint8_t *eml_q, uint8_t *emu_q, int8_t *vit_p1, int8_t *vit_p3);
void orc_am_deinterleave_pids(const uint8_t sym[64], int pids1_disabled, int8_t out[240]);

/* K8  decode.c:234-261: re-encode and count sign disagreements at unpunctured positions */
int orc_bit_errors(const int8_t *coded, const uint8_t *decoded, int k, int len, const unsigned gens[3],
                   const uint8_t *puncture, int plen);

/* whole-path AM stream: mirrors input_push_cu8/cs16 in NRSC5_MODE_AM; the hook sees every 3750-bit P1 frame */
typedef struct orc_am_stream orc_am_stream;
orc_am_stream *orc_am_open(void);
void orc_am_close(orc_am_stream *s);
void orc_am_set_taps(orc_am_stream *s, unsigned mask, unsigned fft_limit_blocks);
void orc_am_set_p1_hook(orc_am_stream *s, orc_p1_hook hook, void *user);
void orc_am_push_cu8(orc_am_stream *s, const uint8_t *iq, uint32_t nbytes);
void orc_am_push_cs16(orc_am_stream *s, const int16_t *iq, uint32_t n);
void orc_am_force_resync(orc_am_stream *s);
size_t orc_am_buf(orc_am_stream *s, int which, const uint8_t **p);

#ifdef __cplusplus
}
#endif
