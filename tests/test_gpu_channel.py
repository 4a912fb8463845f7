"""`-m gpu`: parity of the real gfx950 library on IMPAIRED channels (nrsc5_amd/channel.py) -- sample-clock error of +-25 ... +-100 ppm
(sync.samperr != 0 on every FINE block: sync.c:455 -> acquire.c:112,259 -> sync_adjust, sync.c:769-777), echoes inside the cyclic
prefix, an analog FM host 20 dB above the digital sidebands, ADC clipping, block-scale fading; streaming seam, zero-copy batch with
replay, AM.  The oracle is pinned on exactly these captures against the unmodified reference (tests/test_oracle.py,
tests/test_oracle_am.py: 0 tolerance) and three of them are golden fixtures produced by it."""
import numpy as np
import pytest

from tests import common, engine_checks as ec
from nrsc5_amd import engine as eng, synth, synth_am

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(common.IMPAIRED_FM_CASES))
@pytest.mark.parametrize("p1_async", [False, True])
def test_gpu_impaired_channel_streaming(hip_lib, oracle, name, p1_async):
    ec.check_oracle_end_to_end(hip_lib, oracle, common.IMPAIRED_FM_CASES[name], p1_async=p1_async)


@pytest.mark.parametrize("p1_async,l2_feedback", [(True, True), (False, False)])
def test_gpu_impaired_channel_zero_copy_batch(hip_lib, p1_async, l2_feedback):
    """Every cu8 case in ONE zero-copy batch == each through the streaming seam, record for record, 0 tolerance: `keep` != 2160
    moves the position the fused half-band reads the capture from on every block."""
    caps = [synth.fm_mp1_capture(**kw) for kw in common.IMPAIRED_FM_CASES.values() if kw.get("fmt", "cu8") == "cu8"]
    ec.check_zero_copy_batch(hip_lib, caps, p1_async=p1_async, l2_feedback=l2_feedback)


@pytest.mark.parametrize("lag", [0, 3])
def test_gpu_replay_under_drift(hip_lib, oracle, lag):
    ec.check_deferred_feedback_equals_reference(hip_lib, oracle, n_blocks=96, verdict_lag=lag, caps=ec.drift_replay_captures())


@pytest.mark.parametrize("name", list(common.IMPAIRED_AM_CASES))
def test_gpu_am_impaired_channel(hip_lib, oracle, name):
    ec.check_am_oracle_end_to_end(hip_lib, oracle, common.IMPAIRED_AM_CASES[name])


@pytest.mark.parametrize("p1_async", [False, True])
def test_gpu_am_impaired_batch_equals_streaming(hip_lib, p1_async):
    ec.check_am_batch_equals_streaming(hip_lib, [kw for kw in common.IMPAIRED_AM_CASES.values() if kw.get("fmt", "cs16") == "cs16"], p1_async=p1_async)


def test_gpu_am_replay_under_drift(hip_lib, oracle):
    """AM window pipeline + on-device L2 feedback on drifting captures with interference bursts."""
    from nrsc5_amd import channel
    kws = [dict(n_frames=16, seed=9, cfo_hz=2.0, offset=500, burst=(8.3, 0.5, 40.0), chan=channel.Impairments(ppm=20.0)),
           dict(n_frames=16, seed=10, cfo_hz=-3.0, offset=900, burst=(9.6, 0.3, 40.0), chan=channel.Impairments(ppm=-35.0)),
           dict(n_frames=14, seed=42, cfo_hz=-8.0, offset=700, chan=channel.Impairments(ppm=-50.0))]
    ec.check_am_deferred_feedback_equals_reference(hip_lib, oracle, verdict_lag=2, kws=kws, min_lost=1)
