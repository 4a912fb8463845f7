"""CPU (`-m "not gpu"`): the drop-in translation unit (integration/input_hip.c) under the reference's untouched L4 / L2 code, over
the CPU-EMULATED twin of the library (tests/simt: same .hip sources compiled with g++) -- the shim's control flow (block-exact
pieces, deliver-before-step order, polling, the zero-length flush) against the plain reference build through the public pipe API.
Needs /root/reference (build container); the `-m gpu` twin with the real library is tests/test_gpu_dropin.py."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from tests import common
from oracle import ref

INTEG = os.path.join(common.ROOT, "integration")


@pytest.fixture(scope="module")
def emu_dropin(emu_lib):
    if not os.path.isdir("/root/reference/src"):
        pytest.skip("needs /root/reference")
    if not os.path.exists(os.path.join(common.ROOT, "oracle", "_ref", "libnrsc5_plain.so")):
        pytest.skip("oracle/_ref/libnrsc5_plain.so not built")
    subprocess.check_call(["make", "-C", INTEG, "emu"], stdout=subprocess.DEVNULL)
    return os.path.join(INTEG, "_build", "libnrsc5_emudropin.so")


def _run(path, iq, chunk=32768, mode=0):
    lib = ctypes.CDLL(path)
    lib.pipe_run.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    lib.pipe_run.restype = ctypes.c_size_t
    p = ctypes.c_void_p()
    n = lib.pipe_run(iq.ctypes.data, iq.size, chunk, mode, int(iq.dtype == np.int16), ctypes.byref(p))
    return ref.parse_log(ctypes.string_at(p, n))


def _compare_events(exp, got):
    assert [k for k, _ in exp] == [k for k, _ in got]
    for (k, a), (_, b) in zip(exp, got):
        if k == "hdc":
            assert a["program"] == b["program"] and a["flags"] == b["flags"] and a["data"] == b["data"]
        elif k in ("sync", "mer", "ber"):
            for f in a:
                assert common.float_close(f, float(a[f]), float(b[f])), (k, f, a[f], b[f])


@pytest.mark.parametrize("name,chunk", [("fm_cu8_cfo137", 32768), ("fm_cu8_cfo-2400", 32768), ("fm_cu8_cfo137", 1 << 20)])
def test_emu_dropin_events_match_reference(emu_dropin, captures, name, chunk):
    """FM: a capture whose first lock is false (LOST_SYNC through frame.c -> input_set_sync_state -> nrsc5hip_force_resync while the
    engine runs with deferred waits and manual steps), one with a CFO search, and one fed in calls that span many blocks."""
    iq = np.ascontiguousarray(captures(name).iq)
    exp = _run(os.path.join(common.ROOT, "oracle", "_ref", "libnrsc5_plain.so"), iq, chunk)
    got = _run(emu_dropin, iq, chunk)
    assert len(exp) >= 2
    _compare_events(exp, got)


def test_emu_dropin_fifo_seam_events_match_reference(emu_dropin, captures, monkeypatch):
    """The same through the FIFO seam of rounds 3 - 5 (NRSC5HIP_HOST_CAPTURE=0: pinned staging + decimator kernel): what cs16 / AM sessions and the fall-back of a capture
    use.  The default since round 6 keeps an FM cu8 session's bytes in a pinned capture the stream reads in place (every other test of this file)."""
    monkeypatch.setenv("NRSC5HIP_HOST_CAPTURE", "0")
    iq = np.ascontiguousarray(captures("fm_cu8_cfo137").iq)
    exp = _run(os.path.join(common.ROOT, "oracle", "_ref", "libnrsc5_plain.so"), iq, 32768)
    got = _run(emu_dropin, iq, 32768)
    assert len(exp) >= 2
    _compare_events(exp, got)


def test_emu_dropin_am(emu_dropin, captures):
    name = next(iter(common.GOLDEN_AM_CASES))
    iq = np.ascontiguousarray(captures(name).iq)
    exp = _run(os.path.join(common.ROOT, "oracle", "_ref", "libnrsc5_plain.so"), iq, mode=1)
    got = _run(emu_dropin, iq, mode=1)
    assert any(k == "ber" for k, _ in exp)
    _compare_events(exp, got)


def _run_opts(path, iq, flags, chunk=32768, mode=0):
    lib = ctypes.CDLL(path)
    lib.pipe_run_opts.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_void_p)]
    lib.pipe_run_opts.restype = ctypes.c_size_t
    p, at_end = ctypes.c_void_p(), ctypes.c_size_t()
    n = lib.pipe_run_opts(iq.ctypes.data, iq.size, chunk, mode, int(iq.dtype == np.int16), flags, ctypes.byref(at_end), ctypes.byref(p))
    return ref.parse_log(ctypes.string_at(p, n)), int(at_end.value), int(n)


@pytest.mark.parametrize("strict", [0, 1])
def test_emu_dropin_close_without_flush_delivers_the_last_block(emu_dropin, captures, strict, monkeypatch):
    """src/main.c:1095-1121 ends a file by calling nrsc5_close -- no zero-length nrsc5_pipe_samples_* call.  With the drop-in's overlapped
    delivery (opt-in since round 6) the events of the block that was on the device when the loop ended must then come out of nrsc5_close
    (input_free -> deliver) in the overlapped mode (NRSC5HIP_SYNC_DELIVERY=0 / NRSC5HIP_OVERLAP_DELIVERY=1); in the default mode -- the reference's contract, also NRSC5HIP_SYNC_DELIVERY=1 --
    every event has been delivered inside the call that completed its block.
    Either way the complete log equals the plain reference's."""
    monkeypatch.setenv("NRSC5HIP_SYNC_DELIVERY", str(strict))
    iq = np.ascontiguousarray(captures("fm_cu8_cfo137").iq)
    exp = _run(os.path.join(common.ROOT, "oracle", "_ref", "libnrsc5_plain.so"), iq)
    got, at_end, total = _run_opts(emu_dropin, iq, flags=1)
    _compare_events(exp, got)
    if strict:
        assert at_end == total, "strict delivery: nothing may be left for nrsc5_close"


def run_two(path, a, b, mode_a=0, mode_b=0, chunk=32768):
    lib = ctypes.CDLL(path)
    lib.pipe_run_two.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_uint,
                                 ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_void_p)]
    lib.pipe_run_two.restype = ctypes.c_size_t
    p, split = ctypes.c_void_p(), ctypes.c_size_t()
    n = lib.pipe_run_two(a.ctypes.data, a.size, mode_a, int(a.dtype == np.int16), b.ctypes.data, b.size, mode_b, int(b.dtype == np.int16), chunk, ctypes.byref(split), ctypes.byref(p))
    raw = ctypes.string_at(p, n)
    return ref.parse_log(raw[:split.value]), ref.parse_log(raw[split.value:])


def check_dropin_set_mode_on_a_live_session(dropin):
    """nrsc5_set_mode on a live pipe session, through the public API only: capture A (a real signal: SYNC, audio packets), the reset, then a
    capture B whose symbol boundary sits in the first samples of the acquisition window -- the reference's acquisition filter and half-band
    still hold samples of A there (firdecim_q15_reset, firdecim_q15.c:53-56), so B's events differ from a fresh session's (asserted on the
    plain reference) and the drop-in must follow the USED session: nrsc5hip_stream_reset keeps the windows.  Also: every event of A is
    delivered before nrsc5_set_mode returns (the block in flight on the device belongs to the session that ends there)."""
    from nrsc5_amd import synth
    plain = os.path.join(common.ROOT, "oracle", "_ref", "libnrsc5_plain.so")
    rng = np.random.default_rng(5)
    sig = synth.fm_mp1_capture(0, seed=81, cfo_hz=-55.0, offset=2222, snr_db=20, n_blocks=20).iq
    a = np.concatenate([sig[:sig.size - sig.size % 4], rng.integers(0, 256, size=4 * 71280 * 2 + 4 * 5000, dtype=np.uint8)])    # ends un-synchronised, at full scale
    b = synth.fm_mp1_capture(0, seed=82, cfo_hz=120.0, offset=0, snr_db=20, n_blocks=20).iq
    b = b[:b.size - b.size % 4]
    exp_a, exp_b = run_two(plain, a, b)
    fresh_b = _run(plain, b)
    got_a, got_b = run_two(dropin, a, b)
    assert any(k == "sync" for k, _ in exp_a) and any(k == "sync" for k, _ in exp_b)
    _compare_events(exp_a, got_a)
    _compare_events(exp_b, got_b)
    with pytest.raises(AssertionError):                            # the point of the captures: a fresh session is NOT what the reference does here
        _compare_events(fresh_b, exp_b)


def test_emu_dropin_set_mode_on_a_live_session(emu_dropin):
    check_dropin_set_mode_on_a_live_session(emu_dropin)


def test_emu_dropin_fm_then_am_on_one_session(emu_dropin):
    """FM capture that ends synchronised, nrsc5_set_mode(AM), AM capture -- through the public API on the emulated twin: sync_reset leaves sync_t.angle alone and the AM
    path never writes it, so the reference's first synchronised AM block turns by the FM session's last angle; the events of both captures (incl. the LOST_SYNC the reset
    itself fires, delivered by the drop-in's input_set_sync_state) equal the plain reference's."""
    from nrsc5_amd import synth, synth_am
    plain = os.path.join(common.ROOT, "oracle", "_ref", "libnrsc5_plain.so")
    a = synth.fm_mp1_capture(0, seed=71, cfo_hz=33.0, offset=640, snr_db=20, n_blocks=20).iq
    a = a[:a.size - a.size % 4]
    b = synth_am.am_ma1_capture(9, seed=72, cfo_hz=1.0, offset=900).iq
    exp_a, exp_b = run_two(plain, a, b, mode_a=0, mode_b=1)
    got_a, got_b = run_two(emu_dropin, a, b, mode_a=0, mode_b=1)
    assert any(k == "sync" for k, _ in exp_a) and exp_a[-1][0] == "lost_sync" and any(k == "sync" for k, _ in exp_b)
    _compare_events(exp_a, got_a)
    _compare_events(exp_b, got_b)
