"""CPU-only: pins the AM restatement (oracle/nrsc5_oracle_am.c) against the unmodified reference build and
the golden AM fixtures it produced; AM synthesizer sanity.  No GPU, no HIP library calls."""
import os

import numpy as np
import pytest

from tests import common
from nrsc5_amd import synth_am

GOLDEN_DIR = os.path.join(os.path.dirname(__file__), "golden")
E1, E2 = (0o561, 0o657, 0o711), (0o561, 0o753, 0o711)


def _golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


@pytest.mark.parametrize("name", list(common.GOLDEN_AM_CASES))
def test_am_capture_regenerates_bit_identically(name, captures):
    assert common.sha256(captures(name).iq) == str(_golden(name)["iq_sha"])


@pytest.mark.parametrize("name", list(common.GOLDEN_AM_CASES))
def test_am_oracle_matches_golden_reference_trace(name, captures, oracle):
    """Exact: Q15 stream, hard symbols, PIDS/P1/P3 frames, integer block trace.  Floats: rtol 1e-4."""
    from oracle import port
    g = _golden(name)
    log, q15, _ = oracle.run(captures(name).iq, taps=port.TAP_Q15 | port.TAP_SOFT, mode=1)
    assert common.sha256(q15) == str(g["q15_sha"])
    sym = [v for k, v in log if k == "amsym"]
    assert [v["bc"] for v in sym] == g["sym_bc"].tolist()
    assert [common.sha256(np.concatenate([v["pl"], v["pu"], v["s"], v["t"]])) for v in sym] == g["sym_sha"].tolist()
    diffs = common.compare_logs(common.am_arrays_to_log(g), common.strip_states(log))
    assert not diffs, diffs[:10]


@pytest.mark.parametrize("name", list(common.GOLDEN_AM_CASES))
def test_am_golden_frames_equal_transmitted_truth(name):
    """The reference decodes the synthetic MA1 signal to the transmitted bits: P1 frames are content frames
    4.. (3-frame diversity delay + am_diversity_wait), P3 frames are not delayed."""
    g = _golden(name)
    n1, n3 = g["p1"].shape[0], g["p3"].shape[0]
    # a drifting sample clock costs the reference coded-bit errors even without noise (whole-sample timing corrections only); the
    # decoded frames are still the transmitted ones
    assert n1 >= 16 and n3 >= 2 and np.all(g["ber"] <= (0.01 if "ppm" in name else 0.0))
    first = next(i for i in range(g["truth_p1"].shape[0]) if np.array_equal(g["truth_p1"][i], g["p1"][0]))
    assert first % 8 == 0 and np.array_equal(g["p1"], g["truth_p1"][first:first + n1])
    # MA1: the P3 code word is not diversity-delayed (decoded frame = the one just received); MA3: it is, like P1
    f3 = first // 8 + (0 if name == "am_ma3_cs16" else 3)
    assert np.array_equal(g["p3"], g["truth_p3"][f3:f3 + n3])


@pytest.mark.parametrize("name", list(common.IMPAIRED_AM_CASES))
def test_am_oracle_impaired_channel_bit_identical_to_reference(name, oracle):
    """AM twin of test_oracle_impaired_channel_bit_identical_to_reference: sample-clock error, an echo inside the cyclic prefix,
    fading.  Restatement == both builds of the unmodified reference incl. its L2 decision, 0 tolerance."""
    from oracle import ref, port
    if not ref.available(False):
        pytest.skip("reference build absent")
    cap = synth_am.am_ma1_capture(**common.IMPAIRED_AM_CASES[name])
    ol, oq, _ = oracle.run(cap.iq, taps=port.TAP_Q15 | port.TAP_SOFT, mode=1, p1_hook=oracle.l2_hook())
    for sse in (False, True):
        rl, rq, _ = ref.RefLib(sse=sse).run(cap.iq, mode=ref.MODE_AM, taps=ref.TAP_Q15 | ref.TAP_SOFT)
        assert np.array_equal(rq, oq)
        assert not common.compare_logs(rl, ol, rtol=0.0, skip_kinds=("hdc",))


@pytest.mark.parametrize("sse", [False, True])
def test_am_oracle_bit_identical_to_reference(sse, oracle):
    """Full ordered event log incl. floats and hard symbols, 0 tolerance, against both reference builds."""
    from oracle import ref, port
    if not ref.available(sse):
        pytest.skip("reference build absent")
    R = ref.RefLib(sse=sse)
    for kw in (dict(n_frames=10, seed=3, cfo_hz=3.0, offset=1234),
               dict(n_frames=7, seed=4, cfo_hz=200.0, offset=0),
               dict(n_frames=7, seed=5, cfo_hz=-40.0, offset=9000, noise=2.0),
               dict(n_frames=2, seed=6, cfo_hz=10.0, offset=64 * 300 + 12, fmt="cu8"),
               dict(n_frames=9, seed=8, cfo_hz=-6.0, offset=2000, mode="MA3"),
               dict(n_frames=10, seed=21, cfo_hz=1.5, offset=700, rdbi=1)):          # reduced digital bandwidth: no P3 frame (decode.c:524)
        cap = synth_am.am_ma1_capture(**kw)
        rl, rq, rf = R.run(cap.iq, mode=ref.MODE_AM, taps=ref.TAP_Q15 | ref.TAP_SOFT | ref.TAP_FFT, fft_blocks=2)
        ol, oq, of = oracle.run(cap.iq, mode=1, taps=port.TAP_Q15 | port.TAP_SOFT | port.TAP_FFT, fft_blocks=2)
        assert np.array_equal(rq, oq)
        assert np.array_equal(rf, of)
        assert not common.compare_logs(rl, ol, rtol=0.0, skip_kinds=("hdc",))


def test_am_oracle_noise_only_and_push_sizes(oracle, reflib):
    from oracle import ref
    rng = np.random.default_rng(1)
    iq = rng.integers(-3000, 3000, size=2 * 46512 * 3, dtype=np.int16)      # 3 s of noise
    rl, _, _ = reflib.run(iq, mode=ref.MODE_AM)
    ol, _, _ = oracle.run(iq, mode=1)
    assert not common.compare_logs(rl, ol, rtol=0.0)
    assert all(v["state_after"] != 2 for k, v in ol if k == "block")
    cap = synth_am.am_ma1_capture(3, seed=9, fmt="cu8")
    a, _, _ = oracle.run(cap.iq, mode=1, chunk=32768)
    b, _, _ = oracle.run(cap.iq, mode=1, chunk=4100)        # not a multiple of 64: stage phases carry over
    r, _, _ = reflib.run(cap.iq, mode=ref.MODE_AM, chunk=4100)
    assert not common.compare_logs(a, b, rtol=0.0) and not common.compare_logs(r, b, rtol=0.0)


def test_am_decimator_chunking(oracle):
    rng = np.random.default_rng(5)
    iq = rng.integers(0, 256, size=64 * 500 + 8, dtype=np.uint8)
    whole = oracle.am_decimate_cu8([iq])
    parts = oracle.am_decimate_cu8([iq[:4], iq[4:1000], iq[1000:1004], iq[1004:20000], iq[20000:]])
    assert whole.shape[0] == 500 and np.array_equal(whole, parts)


def test_am_l2_feedback_hook_drops_to_none(oracle, captures):
    cap = captures("am_cs16_cfo3")
    log, _, _ = oracle.run(cap.iq, mode=1, p1_hook=lambda bits: 1)
    kinds = [k for k, _ in log]
    assert "lost_sync" in kinds
    i = kinds.index("lost_sync")
    assert kinds[i - 1] == "state" and kinds[i - 2] == "frame"


def test_am_synth_cells_tile_matrices():
    """Data + training cells exactly tile the 8 x 32 x 25 symbol matrices, bit by bit."""
    t = synth_am._IDX
    for keys, nbits in ((("bl", "ml"), 6), (("bu", "mu"), 6), (("el",), 2), (("eu",), 4)):
        used = np.zeros((6400, 8), dtype=np.int32)
        for k in keys:
            cell, p = t[k]
            np.add.at(used, (cell, p), 1)
        assert used[:, nbits:].sum() == 0 and used.max() == 1
        per_cell = used[:, :nbits].sum(axis=1)
        assert set(np.unique(per_cell)) == {0, nbits} and (per_cell == 0).sum() == 8 * 50     # 2 training cells / carrier / block


def test_am_l2_feedback_restatement_matches_reference_l2(oracle, reflib):
    """An interference burst wipes out a P1 frame: the reference's frame_process drops sync on the 466-byte PDU's header."""
    from oracle import ref
    cap = synth_am.am_ma1_capture(n_frames=16, seed=9, cfo_hz=2.0, offset=500, burst=(8.3, 0.5, 40.0))
    rl, _, _ = reflib.run(cap.iq, mode=ref.MODE_AM)
    ol, _, _ = oracle.run(cap.iq, mode=1, p1_hook=oracle.l2_hook())
    assert any(k == "lost_sync" for k, _ in rl)
    assert not common.compare_logs(rl, ol, rtol=0.0)
