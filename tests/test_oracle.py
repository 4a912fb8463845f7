"""CPU-only: pins the oracle restatement against (a) the unmodified reference build and (b) the
golden fixtures that build produced; plus synthesizer sanity.  No GPU, no HIP library calls."""
import os

import numpy as np
import pytest

from tests import common
from nrsc5_amd import synth

GOLDEN_DIR = os.path.join(os.path.dirname(__file__), "golden")


def _golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


@pytest.mark.parametrize("name", list(common.GOLDEN_CASES))
def test_capture_regenerates_bit_identically(name, captures):
    g = _golden(name)
    assert common.sha256(captures(name).iq) == str(g["iq_sha"]), "synthetic capture bytes differ on this host"


@pytest.mark.parametrize("name", list(common.GOLDEN_CASES))
def test_oracle_matches_golden_reference_trace(name, captures, oracle):
    """Exact: Q15 stream, soft bits, PIDS/P1 frames, integer block trace.  Floats: rtol 1e-4."""
    from oracle import port
    g = _golden(name)
    log, q15, _ = oracle.run(captures(name).iq, taps=port.TAP_Q15 | port.TAP_SOFT)
    assert common.sha256(q15) == str(g["q15_sha"])
    soft = [v for k, v in log if k == "soft"]
    assert [v["bc"] for v in soft] == g["soft_bc"].tolist()
    assert [common.sha256(v["bits"]) for v in soft] == g["soft_sha"].tolist()
    diffs = common.compare_logs(common.arrays_to_log(g), common.strip_states(log))
    assert not diffs, diffs[:10]


@pytest.mark.parametrize("name", ["fm_cu8_cfo137", "fm_cs16_cfo60"])
def test_golden_frames_equal_transmitted_truth(name):
    g = _golden(name)
    assert g["p1"].shape[0] >= 1
    # frame k of the reference output is L1 frame k of the transmission
    assert np.array_equal(g["p1"], g["truth_p1"][:g["p1"].shape[0]])


@pytest.mark.parametrize("sse", [False, True])
def test_oracle_bit_identical_to_reference(sse, oracle):
    """Full ordered event log incl. floats and soft bits, 0 tolerance, against both reference builds."""
    from oracle import ref, port
    if not ref.available(sse):
        pytest.skip("reference build absent")
    R = ref.RefLib(sse=sse)
    for kw in (dict(n_frames=0, n_blocks=24, seed=3, cfo_hz=2000.0, offset=1234, snr_db=15),
               dict(n_frames=0, n_blocks=20, seed=5, cfo_hz=0.0, offset=0, snr_db=30),
               dict(n_frames=0, n_blocks=36, seed=6, cfo_hz=137.0, offset=777, snr_db=12),
               dict(n_frames=0, n_blocks=20, seed=8, cfo_hz=-77.0, offset=500, snr_db=22, fmt="cs16")):
        cap = synth.fm_mp1_capture(**kw)
        rl, rq, rf = R.run(cap.iq, taps=ref.TAP_Q15 | ref.TAP_SOFT | ref.TAP_FFT, fft_blocks=2)
        ol, oq, of = oracle.run(cap.iq, taps=port.TAP_Q15 | port.TAP_SOFT | port.TAP_FFT, fft_blocks=2)
        assert np.array_equal(rq, oq)
        assert np.array_equal(rf, of)
        assert not common.compare_logs(rl, ol, rtol=0.0, skip_kinds=("hdc",))


@pytest.mark.parametrize("name", list(common.IMPAIRED_FM_CASES))
def test_oracle_impaired_channel_bit_identical_to_reference(name, oracle):
    """Sample-clock error, echoes, analog host, clipping, fading (nrsc5_amd/channel.py): the FINE-state timing feedback
    (sync.c:426-463 -> acquire.c:110-119,259 -> sync_adjust sync.c:769-777) works on every block of these captures.  Restatement
    == both builds of the UNMODIFIED reference incl. its L2 decision, 0 tolerance: Q15 stream, soft bits, every float."""
    from oracle import ref, port
    if not ref.available(False):
        pytest.skip("reference build absent")
    cap = synth.fm_mp1_capture(**common.IMPAIRED_FM_CASES[name])
    ol, oq, _ = oracle.run(cap.iq, taps=port.TAP_Q15 | port.TAP_SOFT, p1_hook=oracle.l2_hook())
    for sse in (False, True):
        rl, rq, _ = ref.RefLib(sse=sse).run(cap.iq, taps=ref.TAP_Q15 | ref.TAP_SOFT)
        assert np.array_equal(rq, oq)
        assert not common.compare_logs(rl, ol, rtol=0.0, skip_kinds=("hdc",))
    blocks = [v for k, v in ol if k == "block"]
    fine = [b for b in blocks if b["state_before"] == 2]
    if cap_has_drift(common.IMPAIRED_FM_CASES[name]):
        assert sum(1 for b in fine if b["samperr"] != 1080) >= len(fine) // 2, "the capture does not exercise the timing feedback"
    assert any(k == "frame" for k, _ in ol)


def cap_has_drift(kw) -> bool:
    return abs(kw["chan"].ppm) >= 25.0


def test_oracle_sync_loss_under_drift_bit_identical_to_reference(oracle, reflib):
    """A burst breaks a P1 frame's first header 30-50 blocks into a drifting, tracked stream: LOST_SYNC, the re-acquisition from
    prev_angle != 0 and everything after it (frame.c:535-540, acquire.c:110-158)."""
    from tests import engine_checks as ec
    for cap in ec.drift_replay_captures():
        rl, _, _ = reflib.run(cap.iq)
        ol, _, _ = oracle.run(cap.iq, p1_hook=oracle.l2_hook())
        assert not common.compare_logs(rl, ol, rtol=0.0)
        kinds = [k for k, _ in ol]
        assert "lost_sync" in kinds and max(i for i, k in enumerate(kinds) if k == "lost_sync") > 60    # deep into the tracked stretch


def test_oracle_noise_only_stays_unsynchronised(oracle, reflib):
    rng = np.random.default_rng(1)
    iq = rng.integers(100, 156, size=2 * 1488375, dtype=np.uint8)   # 1 s of noise
    rl, _, _ = reflib.run(iq)
    ol, _, _ = oracle.run(iq)
    assert not common.compare_logs(rl, ol, rtol=0.0)
    assert all(v["state_after"] != 2 for k, v in ol if k == "block")


def test_oracle_push_size_invariance(oracle, captures):
    cap = captures("fm_cu8_cfo-2400")
    a, _, _ = oracle.run(cap.iq, chunk=32768)
    b, _, _ = oracle.run(cap.iq, chunk=4100)
    c, _, _ = oracle.run(cap.iq, chunk=1000004)
    assert not common.compare_logs(a, b, rtol=0.0) and not common.compare_logs(a, c, rtol=0.0)


def test_oracle_viterbi_matches_reference_decoder(oracle, reflib):
    rng = np.random.default_rng(2)
    for kind, n in (("pids", 80), ("p3", 2304), ("p3", 4608), ("p1", 146176)):
        soft = rng.integers(-127, 128, size=3 * n, dtype=np.int8)
        assert np.array_equal(oracle.viterbi_k7(soft), reflib.conv_decode(soft, kind))
    # K=9 codes of the AM path (decode.c:47-61)
    soft = rng.integers(-1, 2, size=3 * 3750, dtype=np.int8)
    assert np.array_equal(oracle.viterbi(soft, 9, (0o561, 0o657, 0o711)), reflib.conv_decode(soft, "e1"))
    assert np.array_equal(oracle.viterbi(soft, 9, (0o561, 0o753, 0o711)), reflib.conv_decode(soft, "e2"))


def test_l2_feedback_hook_drops_to_none(oracle, captures):
    cap = captures("fm_cu8_cfo137")
    log, _, _ = oracle.run(cap.iq, p1_hook=lambda bits: 1)
    kinds = [k for k, _ in log]
    assert "lost_sync" in kinds
    i = kinds.index("lost_sync")
    assert kinds[i - 1] == "state" and kinds[i - 2] == "frame"
    nxt = next(v for k, v in log[i:] if k == "block")   # the frame's own block record follows
    assert nxt["state_after"] == 0


def test_synth_interleaver_tiles_matrix():
    assert len(np.unique(np.concatenate([synth.P1_IDX, synth.PIDS_IDX]))) == 16 * synth.PM_BLOCK


@pytest.mark.parametrize("mode,kw", [
    ("MP2", dict(n_blocks=52, seed=31, cfo_hz=20.0, offset=300, snr_db=25)),
    ("MP3", dict(n_blocks=54, seed=32, cfo_hz=-150.0, offset=500, snr_db=18, fmt="cs16")),
    ("MP11", dict(n_blocks=52, seed=33, cfo_hz=0.0, offset=64, snr_db=14)),
])
def test_oracle_extended_sidebands_bit_identical_to_reference(mode, kw, oracle, reflib):
    """PX1 / PX2 soft bits -> interleaver IV -> P3 / P4 frames (decode.c:344-437): full log, 0 tolerance."""
    from oracle import ref, port
    cap = synth.fm_mp1_capture(0, mode=mode, **kw)
    rl, _, _ = reflib.run(cap.iq, taps=ref.TAP_SOFT)
    ol, _, _ = oracle.run(cap.iq, taps=port.TAP_SOFT)
    assert not common.compare_logs(rl, ol, rtol=0.0, skip_kinds=("hdc",))
    px = [v for k, v in rl if k == "frame" and v["lc"] != 0]
    assert len(px) >= (4 if mode == "MP11" else 2)


@pytest.mark.parametrize("mode", ["MP5", "MP6", "PSMI7"])
def test_oracle_compatibility_modes_5_6_bit_identical_to_reference(mode, oracle, reflib):
    """PSMI values whose compatibility mode is 5 or 6 (Table 6-4, sync.c:29-35): the reference tracks and equalises 14
    partitions per sideband (MER over all of them) but routes none of the extended ones to the decoder -- no P3 / P4 frames."""
    from oracle import ref, port
    cap = synth.fm_mp1_capture(0, mode=mode, n_blocks=50, seed=41, cfo_hz=35.0, offset=420, snr_db=22)
    rl, _, _ = reflib.run(cap.iq, taps=ref.TAP_SOFT)
    ol, _, _ = oracle.run(cap.iq, taps=port.TAP_SOFT)
    assert not common.compare_logs(rl, ol, rtol=0.0, skip_kinds=("hdc",))
    assert [v["psmi"] for k, v in rl if k == "sync"] == [synth.MODES[mode][0]]
    assert not [v for k, v in rl if k == "frame" and v["lc"] != 0]
    assert sum(1 for k, v in rl if k == "frame" and v["lc"] == 0) >= 1 and any(k == "mer" for k, _ in rl)


@pytest.mark.parametrize("name", ["fm_mp11_cs16", "fm_mp2_cu8"])
def test_golden_px_frames_equal_transmitted_truth(name):
    g = _golden(name)
    lcs = g["frame_lc"][g["frame_lc"] != 0]
    assert lcs.size >= 2 and g["sync"].shape[0] == 1
    i3 = i4 = None
    for lc, row in zip(lcs, g["px"]):
        truth = g["truth_p3"] if lc == 1 else g["truth_p4"]
        hit = [k for k in range(truth.shape[0]) if np.array_equal(truth[k], row)]
        assert len(hit) == 1
        if lc == 1:
            assert i3 is None or hit[0] == i3 + 1
            i3 = hit[0]
        else:
            assert i4 is None or hit[0] == i4 + 1
            i4 = hit[0]


def test_interleaver_iv_is_convolutional():
    """The synthesizer relies on interleaver_iv being shift-invariant by one block pair (delays 1..N)."""
    for L in (2304, 4608):
        d = synth.interleaver_iv_delays(L)
        assert d.min() >= 1 and d.max() <= 32 * L and len(np.unique((np.arange(2 * L) - d) % (32 * L))) == 2 * L


def test_rs_decoder_matches_reference_decoder(oracle):
    """Restated RS(255,247) (oracle/nrsc5_oracle_l2.c) == the decoder the reference links (rs_decode.c via oracle/_ref):
    same return value and same corrected word for 0..7 symbol errors, incl. its mis-corrections beyond 4 errors."""
    import ctypes
    from oracle import ref
    if not ref.available():
        pytest.skip("reference build absent")
    lib = ctypes.CDLL(ref.lib_path())
    lib.init_rs_char.restype = ctypes.c_void_p
    lib.init_rs_char.argtypes = [ctypes.c_uint] * 5
    lib.decode_rs_char.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    rs = lib.init_rs_char(8, 0x11d, 1, 1, 8)
    rng = np.random.default_rng(0)
    seen = set()
    for _ in range(3000):
        data = [int(x) for x in rng.integers(0, 256, size=247)]
        w = np.array(data + synth._GF.rs_parity(data), dtype=np.uint8)
        ne = int(rng.integers(0, 8))
        pos = rng.choice(255, size=ne, replace=False)
        w[pos] ^= rng.integers(1, 256, size=ne).astype(np.uint8)
        a = w.copy()
        rc_ref = lib.decode_rs_char(rs, a.ctypes.data, None, 0)
        rc, b = oracle.rs_decode(w)
        assert rc == rc_ref and (rc < 0 or np.array_equal(a, b))
        seen.add((ne > 4, rc_ref < 0))
    assert len(seen) >= 3


@pytest.mark.parametrize("kw", [
    dict(n_frames=0, n_blocks=40, seed=23, cfo_hz=0.0, offset=1234, snr_db=20.0),          # false lock -> header fails -> resync
    dict(n_frames=0, n_blocks=52, seed=15, cfo_hz=80.0, offset=2100, snr_db=22.0, mode="MP2"),
    dict(n_frames=0, n_blocks=36, seed=25, cfo_hz=10.0, offset=777, snr_db=20.0),          # good lock: no feedback
])
def test_l2_feedback_restatement_matches_reference_l2(kw, oracle, reflib):
    """The unmodified reference (with its real L2) vs the oracle driven by orc_l2_first_header_ok: identical logs,
    incl. LOST_SYNC and the re-acquisition after it."""
    cap = synth.fm_mp1_capture(**kw)
    rl, _, _ = reflib.run(cap.iq)
    ol, _, _ = oracle.run(cap.iq, p1_hook=oracle.l2_hook())
    assert not common.compare_logs(rl, ol, rtol=0.0)
    assert (kw["offset"] == 777) != any(k == "lost_sync" for k, _ in rl)


def test_pids_crc_restatement_matches_reference_sis_events(oracle, reflib):
    """Every block carries a new station id, every fifth PIDS frame a broken CRC: the unmodified reference fires
    NRSC5_EVENT_STATION_ID exactly for the frames the restated CRC-12 accepts, with their ids in order."""
    cap = synth.fm_mp1_capture(0, n_blocks=40, seed=77, cfo_hz=30.0, offset=777, snr_db=25, station_ids=True)
    log, _, _ = reflib.run(cap.iq)
    pids = [v["bits"] for k, v in log if k == "pids"]
    ok = [oracle.pids_crc_ok(b) for b in pids]

    def station_id(b):
        p = np.array([b[((i >> 3) << 3) + 7 - (i & 7)] for i in range(80)])
        return int("".join(str(int(x)) for x in p[19:38]), 2)
    assert 0 < sum(ok) < len(ok)
    assert [station_id(b) for b, o in zip(pids, ok) if o] == [v["fcc"] for k, v in log if k == "station"]


def test_reset_of_a_used_session_stale_decimator_window_is_pinned(oracle, reflib):
    """input_reset on a USED session (nrsc5_set_mode on a live pipe session): firdecim_q15_reset only rewinds the window index
    (firdecim_q15.c:53-56), so the first outputs of the half-band see 14 stale samples of the previous capture where a fresh
    session has zeros.  Pinned here for the ORACLE, which restates fresh sessions only: the deviation is confined to the first 7
    decimated samples after the reset; for this pair of captures every later sample and the complete event log of the second one
    equal a fresh session's.  (The ENGINE reproduces the used session: nrsc5hip_stream_reset keeps the windows, nrsc5hip_stream_fresh
    is the new session -- engine_checks.check_reset_keeps_fir_windows, where the captures are chosen so that the logs differ.)"""
    from oracle import ref
    from nrsc5_amd import synth
    a = synth.fm_mp1_capture(0, seed=81, cfo_hz=-55.0, offset=2222, snr_db=20, n_blocks=6)
    b = synth.fm_mp1_capture(0, seed=82, cfo_hz=120.0, offset=901, snr_db=20, n_blocks=20)
    _, log_b, q15_b = reflib.run_with_mode_switch(a.iq, b.iq, taps=ref.TAP_Q15)
    fresh_log, fresh_q15, _ = reflib.run(b.iq, taps=ref.TAP_Q15)
    n = min(len(q15_b), len(fresh_q15))
    assert n > 100000
    differ = np.nonzero((q15_b[:n] != fresh_q15[:n]).any(axis=1))[0]
    assert differ.size > 0 and differ.max() < 7, differ[:10]           # the quirk exists, and this is all of it
    diffs = common.compare_logs(common.strip_states(fresh_log), common.strip_states(log_b))
    assert not diffs, diffs[:5]
    assert sum(1 for k, _ in log_b if k == "frame") >= 1
    # and the restatement (fresh-session semantics, like the engine) equals both from sample 7 on
    o_log, o_q15, _ = oracle.run(b.iq, taps=1)
    assert np.array_equal(o_q15[7:n], q15_b[7:n])
