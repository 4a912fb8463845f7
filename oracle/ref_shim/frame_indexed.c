/* TEST INFRASTRUCTURE.  The reference's src/frame.c, included verbatim from where it lies (so that its static helpers --
 * parse_hdlc, aas_push, ... -- are visible; nothing is copied into this repo), followed by the one function a maintainer
 * of the reference would add to take the device's L2 audio-transport index instead of re-walking the frame
 * ("Taking the L2 audio index").  oracle/Makefile compiles this translation unit in place of frame.o, so
 * the checker library can run both entry points on the same frames and compare every call they make into output.c. */
#include "frame.c"                        /* -I$(REF)/src */
#include "../../include/nrsc5hip.h"

/* the cut libnrsc5hip exports as nrsc5hip_l2_apply_audio_end (nrsc5_amd/csrc/hdc_consumer.hip), restated here because this
 * translation unit is linked into the reference-only checker: an index built with audio_end = nbytes - 1 taken back to the
 * audio_end process_fixed_data returned (the loop / locator conditions of frame.c:525,547-556) */
static int apply_audio_end(nrsc5hip_l2_frame *ix, unsigned int audio_end)
{
    for (unsigned int k = 0; k < ix->n_pdu; k++)
    {
        const nrsc5hip_l2_pdu *p = &ix->pdu[k];
        int cut = !(p->start < audio_end - RS_CODEWORD_LEN) || p->start + p->la_location >= audio_end;
        for (unsigned int j = 0; !cut && j < p->nop; j++) if (p->loc[j] >= audio_end) cut = 1;
        if (cut) { ix->n_pdu = k; if (k == 0 && p->start == 0) ix->lost_sync = 0; return (int)k; }
        if (p->hef && !p->skipped && p->psd_off > audio_end) return -1;
    }
    if (ix->status == NRSC5HIP_L2_HEADER_RS && !(ix->end_offset < audio_end - RS_CODEWORD_LEN)) ix->lost_sync = 0;
    return (int)ix->n_pdu;
}

/* same effects as frame_push + frame_process (frame.c:516-714) for a frame whose walk the device already did */
void frame_push_indexed(frame_t *st, const nrsc5hip_l2_frame *ix_in, const uint8_t *pdu_bytes, logical_channel_t lc)
{
    nrsc5hip_l2_frame cut;
    const nrsc5hip_l2_frame *ix = ix_in;
    memcpy(st->buffer, pdu_bytes, ix->nbytes);                 /* PCI removed, bit order restored, headers RS-corrected */
    st->pci = ix->pci;
    if (has_fixed(st))                                         /* frame.c:521-522: the fixed-data state machine stays host code */
    {
        const unsigned int audio_end = (unsigned int)process_fixed_data(st, ix->nbytes, lc);
        if (!has_audio(st)) return;
        cut = *ix_in;
        if (apply_audio_end(&cut, audio_end) < 0) { frame_process(st, ix->nbytes, lc); return; }   /* (never seen: HEF into the fixed region) */
        ix = &cut;
    }
    if (ix->lost_sync) input_set_sync_state(st->input, SYNC_STATE_NONE);          /* frame.c:537-538 */
    for (unsigned int k = 0; k < ix->n_pdu; k++)
    {
        const nrsc5hip_l2_pdu *p = &ix->pdu[k];
        if (p->skipped) continue;                              /* stream_id >= MAX_STREAMS, frame.c:559-564 */
        audio_service_t *service = &st->services[p->prog_num];
        if (p->stream_id == 0 && (service->access != p->access || service->type != p->prog_type
            || service->codec_mode != p->codec_mode || service->blend_control != p->blend_control
            || service->digital_audio_gain != p->per_stream_delay || service->common_delay != p->common_delay
            || service->latency != p->latency))
        {
            service->access = p->access; service->type = p->prog_type; service->codec_mode = p->codec_mode;
            service->blend_control = p->blend_control; service->digital_audio_gain = p->per_stream_delay;
            service->common_delay = p->common_delay; service->latency = p->latency;
            nrsc5_report_audio_service(st->input->radio, p->prog_num, service->access, service->type, service->codec_mode,
                                       service->blend_control,
                                       (service->digital_audio_gain < 16) ? service->digital_audio_gain : (service->digital_audio_gain - 32),
                                       service->common_delay * 4, service->latency * 2);
        }
        output_align(st->input->output, p->prog_num, p->stream_id, p->align_offset);
        parse_hdlc(st, aas_push, st->psd_buf[p->prog_num], &st->psd_idx[p->prog_num], MAX_AAS_LEN,
                   st->buffer + p->psd_off, (size_t)p->psd_len, lc);
        unsigned int off = p->audio_off;
        for (unsigned int j = 0; j < p->nop; j++)
        {
            packet_ref_t ref;
            ref.program = p->prog_num;
            ref.stream_id = p->stream_id;
            ref.data = st->buffer + off;
            ref.size = p->loc[j] - off;
            ref.seq = (p->elastic_seq + j) % ELASTIC_BUFFER_LEN;
            ref.flags = ((j < 32 ? p->crc_bad_lo >> j : p->crc_bad_hi >> (j - 32)) & 1) ? PACKET_FLAG_CRC_ERROR : PACKET_FLAG_NONE;
            ref.shape = (j == 0 && p->pfirst) ? PACKET_HALF_BACK : (j == p->nop - 1u && p->plast) ? PACKET_HALF_FRONT : PACKET_FULL;
            output_push(st->input->output, &ref);
            off = p->loc[j] + 1u;
        }
    }
}
