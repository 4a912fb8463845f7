"""`-m "not gpu"`: pins the oracle's restatement of frame_push / frame_process (oracle/nrsc5_oracle_l2.c, orc_l2_index)
against the UNMODIFIED reference: every logical frame is handed to the reference's own frame_push and the calls it
makes to output_align / output_push (tapped with --wrap in oracle/ref_shim/ref_harness.c) must be exactly the ones the
index describes -- program, stream, elastic-buffer sequence, size, CRC flag, half-packet shape and the bytes."""
import numpy as np
import pytest

from oracle import port
from nrsc5_amd import synth_l2


from tests.common import l2_expected_taps as expected_taps, l2_reference_taps as reference_taps  # noqa: E402


@pytest.mark.parametrize("nbits", sorted(synth_l2.LAYOUT))
def test_l2_index_matches_reference_frame_process(oracle, reflib, nbits):
    cases = synth_l2.test_frames(nbits)
    statuses = set()
    for name, bits, safe in cases:
        idx, by = oracle.l2_index(bits)
        statuses.add(port.L2_STATUS[idx["status"]])
        if not safe:
            continue                      # reference undefined (its parse_hdlc length wraps)
        log = reflib.l2_frames([bits])[0]
        assert expected_taps(idx, by) == reference_taps(log), (nbits, name)
        if idx["n_pdu"]:
            assert any(k == "l2pkt" for k, _ in log) or all(d["nop"] == 0 or d["skipped"] for d in idx["pdus"])
    assert {"end", "no_audio", "header_rs", "bad_locators", "hef_overrun", "bad_stream"} <= statuses
    assert "fixed_data" not in statuses   # frames with fixed-data sub-channels are indexed like the others since round 2


def test_l2_index_sequence_through_one_session(oracle, reflib):
    """Frames pushed back to back through ONE reference session (services / elastic buffer state evolving) still yield,
    frame by frame, exactly the calls the stateless index describes."""
    frames = [b for _, b, safe in synth_l2.test_frames(146176, seed=3) if safe]
    logs = reflib.l2_frames(frames)
    for bits, log in zip(frames, logs):
        idx, by = oracle.l2_index(bits)
        assert expected_taps(idx, by) == reference_taps(log)


def test_l2_index_agrees_with_first_header_check(oracle):
    """lost_sync of the index == the L2 -> L1 feedback decision used by the engine (orc_l2_first_header_ok)."""
    for nbits in (146176, 3750):
        for name, bits, _ in synth_l2.test_frames(nbits, seed=5):
            idx, _ = oracle.l2_index(bits)
            assert bool(idx["lost_sync"]) == (not oracle.l2_first_header_ok(bits)), (nbits, name)


def test_l2_index_on_synth_capture_frames(oracle):
    """The PDUs of the end-to-end signal source: 32 packets, all CRCs good, header fields as generated."""
    from nrsc5_amd import synth
    rng = np.random.default_rng(4)
    pdu, packets = synth.make_audio_pdu(3, rng)
    idx, by = oracle.l2_index(synth.p1_frame_bits(pdu))
    assert idx["n_pdu"] == 1 and idx["pdus"][0]["nop"] == 32 and idx["pdus"][0]["crc_bad_lo"] == 0
    d = idx["pdus"][0]
    off = d["audio_off"]
    for j, loc in enumerate(d["loc"]):
        assert bytes(by[off:loc]) == packets[j]
        off = loc + 1


def test_l2_index_random_structures_match_reference(oracle, reflib):
    """Randomised frames: 1..6 PDUs with random codec modes (12- / 16-bit locators, unknown modes), packet counts 0..63,
    enhanced streams, HEF combinations, half packets, CRC failures, then 0..6 corrupted bytes in a random header."""
    rng = np.random.default_rng(99)
    statuses = {}
    for trial in range(150):
        nbits, bits = synth_l2.random_frame(rng)
        idx, by = oracle.l2_index(bits)
        st = port.L2_STATUS[idx["status"]]
        statuses[st] = statuses.get(st, 0) + 1
        if st in ("hef_overrun", "bad_stream", "too_many_pdus"):
            continue
        log = reflib.l2_frames([bits])[0]
        assert expected_taps(idx, by) == reference_taps(log), (trial, nbits, st)
    assert statuses.get("end", 0) + statuses.get("bad_locators", 0) >= 40 and statuses.get("header_rs", 0) >= 10, statuses


def test_l2_golden_is_current(oracle, reflib):
    """The committed golden equals what the reference produces now, and the oracle's index implies the same calls."""
    import hashlib
    import json
    import os
    from tests import common
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "l2_reference_taps.json")))
    for nbits in sorted(synth_l2.LAYOUT):
        frames = {name: bits for name, bits, _ in synth_l2.test_frames(nbits)}
        for c in gold[str(nbits)]:
            bits = frames[c["name"]]
            assert hashlib.sha1(bits.tobytes()).hexdigest() == c["bits_sha1"]
            now = json.loads(json.dumps(common.l2_taps_digest(reference_taps(reflib.l2_frames([bits])[0]))))
            idx, by = oracle.l2_index(bits)
            mine = json.loads(json.dumps(common.l2_taps_digest(expected_taps(idx, by))))
            assert now == c["taps"] == mine, (nbits, c["name"])


def _all_l2_taps(log):
    out = []
    for k, v in log:
        if k in ("l2align", "l2pkt", "state"):
            out += reference_taps([(k, v)])
        elif k == "l2aas":
            out.append((k, v["data"]))
        elif k == "l2svc":
            out.append((k,) + tuple(v.values()))
        elif k == "hdc":
            out.append((k, v["program"], v["count"], v["flags"], bytes(v["data"])))
    return out


def test_frame_push_indexed_equals_frame_push_inside_the_reference(oracle, reflib):
    """The binding INTEGRATION.md proposes, executed: oracle/ref_shim/frame_indexed.c includes the reference's frame.c
    verbatim and adds frame_push_indexed(); fed with the index (C-ABI struct layout) + PDU bytes it must make the reference
    do exactly what its own frame_push does with the raw bits -- output_align, output_push, PSD / AAS packets through
    parse_hdlc, audio-service reports, HDC events out of the elastic buffer, sync loss -- over whole sessions."""
    sessions = {
        "psd": synth_l2.psd_sequence(seed=1),
        "psd_am": synth_l2.psd_sequence(seed=2, nbits=24000, n_frames=4),
        "branches": [b for _, b, safe in synth_l2.test_frames(146176, seed=7) if safe],
        "random": [synth_l2.random_frame(np.random.default_rng(500 + k))[1] for k in range(40)],
    }
    n_aas = n_svc = n_pkt = 0
    for name, frames in sessions.items():
        keep, items = [], []
        for bits in frames:
            fr, by = oracle.l2_index_struct(bits)
            if port.L2_STATUS[fr.status] in ("hef_overrun", "bad_stream", "too_many_pdus"):
                continue                  # the host keeps walking those itself
            keep.append(bits)
            items.append((fr, by))
        direct = reflib.l2_frames(keep)
        indexed = reflib.l2_frames_indexed(items)
        for k, (a, b) in enumerate(zip(direct, indexed)):
            ta, tb = _all_l2_taps(a), _all_l2_taps(b)
            assert ta == tb, (name, k, [x[:6] for x in ta[:4]], [x[:6] for x in tb[:4]])
            n_aas += sum(1 for t in ta if t[0] == "l2aas"); n_svc += sum(1 for t in ta if t[0] == "l2svc"); n_pkt += sum(1 for t in ta if t[0] == "l2pkt")
    assert n_aas >= 8 and n_svc >= 6 and n_pkt >= 500, (n_aas, n_svc, n_pkt)


def test_fixed_data_frames_index_cut_back_equals_reference(oracle, reflib, emu_lib):
    """Frames with fixed-data sub-channels (has_fixed, frame.c:458-514): the index is built with audio_end = nbytes - 1 and the
    consumer cuts it back with process_fixed_data's value.  One reference session over frames whose audio_end moves from
    length - 1 to length - 17 to length - 4017: (1) the host restatement of the CCC state machine (nrsc5hip_hdc_fixed_audio_end)
    + nrsc5hip_l2_apply_audio_end imply exactly the reference's output_align / output_push calls, frame by frame;
    (2) frame_push_indexed inside the reference (which runs the reference's own process_fixed_data) does what frame_push does."""
    import ctypes
    from nrsc5_amd import engine as eng
    frames = synth_l2.fixed_data_session(seed=1)
    direct = reflib.l2_frames(frames)
    lib = eng.load_library(emu_lib)                              # host-only functions of the product library (no device needed)
    H = eng.HdcConsumer(1, lib=lib)
    ends, kept = [], []
    for bits, log in zip(frames, direct):
        fr, by = oracle.l2_index_struct(bits)
        assert fr.status != port.L2_STATUS.index("fixed_data") and fr.n_pdu == 5          # walked with the largest audio_end
        audio_end = H.fixed_audio_end(0, 0, np.frombuffer(by, dtype=np.uint8)[:fr.nbytes])
        cut = eng.L2Frame.from_buffer_copy(fr)
        n = lib.nrsc5hip_l2_apply_audio_end(ctypes.byref(cut), audio_end)
        assert n >= 0
        ends.append(audio_end); kept.append(n)
        assert expected_taps(eng.l2_frame_to_dict(cut), np.frombuffer(by, dtype=np.uint8)) == reference_taps(log)
    nb = (146176 - 24) // 8
    assert ends[0] == ends[1] == nb - 1 and nb - 17 in ends and ends[-1] == nb - 17 - 4000, ends
    assert kept[0] == 5 and kept[-1] < 5, kept                                              # the cut really removes PDUs
    H.close()
    items = [oracle.l2_index_struct(b) for b in frames]
    indexed = reflib.l2_frames_indexed(items)
    for a, b in zip(direct, indexed):
        assert _all_l2_taps(a) == _all_l2_taps(b)


def test_fixed_only_frames_advance_the_ccc_state(oracle, reflib, emu_lib):
    """PCI_FIXED frames (has_fixed && !has_audio, frame.c:138-151): frame_process runs process_fixed_data on them before it
    returns, so the sync-width count, the CCC message and fixed_ready advance on frames that deliver no audio.  A channel that
    mixes fixed-only and audio + fixed frames must be cut at the reference's audio_end afterwards: nrsc5hip_hdc_push_frame does
    the state update for the fixed-only frames."""
    import ctypes
    from nrsc5_amd import engine as eng
    frames = synth_l2.fixed_data_session(seed=2, fixed_only=(1, 3))
    direct = reflib.l2_frames(frames)
    lib = eng.load_library(emu_lib)
    H = eng.HdcConsumer(1, lib=lib)
    ends = []
    for fi, (bits, log) in enumerate(zip(frames, direct)):
        fr, by = oracle.l2_index_struct(bits)
        b = np.frombuffer(by, dtype=np.uint8)
        if fi in (1, 3):
            assert fr.status == port.L2_STATUS.index("no_audio") and reference_taps(log) == []
            H.push_frame(0, fr, b, 0)                            # state update only
            continue
        audio_end = H.fixed_audio_end(0, 0, b[:fr.nbytes])
        cut = eng.L2Frame.from_buffer_copy(fr)
        assert lib.nrsc5hip_l2_apply_audio_end(ctypes.byref(cut), audio_end) >= 0
        ends.append(audio_end)
        assert expected_taps(eng.l2_frame_to_dict(cut), b) == reference_taps(log), fi
    nb = (146176 - 24) // 8
    # frame 0 and the fixed-only frame 1 confirm the sync byte: frame 2 already sees the CCC region, frame 4 the sub-channel
    assert ends[0] == nb - 1 and ends[1] == nb - 17 and ends[-1] == nb - 17 - 4000, ends
    H.close()
