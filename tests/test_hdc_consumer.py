"""`-m "not gpu"`: the batch HDC consumer (nrsc5hip_hdc_*) fed with the L2 index reproduces the NRSC5_EVENT_HDC
sequence of the UNMODIFIED reference (program, byte count, flags, payload -- in order), at a few dozen KB of host state per
stream instead of a 22.9 MB nrsc5_t.  The index comes from the CPU-emulated build here (kernel logic); the -m gpu twin in
test_gpu_parity.py takes it from the device."""
import numpy as np
import pytest

from nrsc5_amd import engine as eng, synth
from oracle import ref
from tests import common, engine_checks as ec


def _reference_hdc(reflib, iq, mode=ref.MODE_FM):
    log, _, _ = reflib.run(iq, mode=mode, taps=ref.TAP_HDC)
    return [(v["program"], v["count"], v["flags"], bytes(v["data"])) for k, v in log if k == "hdc"]


@pytest.mark.parametrize("kw", [dict(n_frames=4, seed=91, cfo_hz=35.0, offset=700, snr_db=22),
                                dict(n_frames=0, n_blocks=50, seed=23, cfo_hz=0.0, offset=1234, snr_db=20)])       # second: false lock, LOST_SYNC, re-acquisition
def test_hdc_events_equal_reference(emu_lib, reflib, kw):
    ec.check_hdc_consumer(emu_lib, reflib, [synth.fm_mp1_capture(**kw)])


def test_adts_framing_matches_dump_hdc():
    """write_adts_header (main.c:182-204) bit for bit: fixed fields + 13-bit frame length"""
    lib = eng.load_library(ec_lib())
    H = eng.HdcConsumer(1, lib=lib)
    payload = bytes(range(200)) * 3
    out = H.adts(payload)
    n = len(payload) + 7
    assert out[7:] == payload and len(out) == n
    bits = int.from_bytes(out[:7], "big")
    assert bits >> 44 == 0xFFF and (bits >> 40) & 0xF == 0b0001 and (bits >> 38) & 3 == 1 and (bits >> 34) & 0xF == 7
    assert (bits >> 30) & 7 == 2 and (bits >> 13) & 0x1FFF == n and (bits >> 2) & 0x7FF == 0x7FF and bits & 3 == 0
    H.close()


def ec_lib():
    from nrsc5_amd import build
    return build.build_emu()


def test_2048_streams_bounded_host_state(emu_lib, oracle):
    """2048 streams through one consumer: host state stays ~tens of KB per stream (the reference: 22.9 MB per session)."""
    import resource
    rng = np.random.default_rng(5)
    frames = []
    for f in range(3):
        pdu, _ = synth.make_audio_pdu(f, rng)
        bits = synth.p1_frame_bits(pdu)
        frames.append(oracle.l2_index_struct(bits))                 # (C-ABI struct, PDU bytes): what the device index returns
    lib = eng.load_library(emu_lib)
    H = eng.HdcConsumer(2048, lib=lib)
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    delivered = 0
    for rnd in range(3):
        fr, by = frames[rnd]
        for s in range(2048):
            for _ in range(16):
                delivered += H.lib.nrsc5hip_hdc_advance(H._h, s, eng.MODE_FM, eng.HDC_CB(0), None)   # null callback: count only
            H.push_frame(s, fr, np.frombuffer(by, dtype=np.uint8), 0)
    per_stream = H.host_bytes() / 2048
    rss_growth_mb = (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss - rss0) / 1024
    assert delivered >= 2048 * 32                                    # packets flowed
    assert per_stream < 128 * 1024, per_stream                        # vs 22.9 MB
    assert rss_growth_mb < 2048 * 0.2, rss_growth_mb
    H.close()
