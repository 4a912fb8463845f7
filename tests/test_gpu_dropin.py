"""`-m gpu`: the DROP-IN seen from the reference's own public API (nrsc5.h): the reference's L4/L2/L1' code,
unmodified, linked with integration/input_hip.c + libnrsc5hip.so, against the plain reference library.
Both are driven only through nrsc5_open_pipe / nrsc5_set_mode / nrsc5_set_callback / nrsc5_pipe_samples_cu8.
The libraries are prebuilt in the build container (integration/Makefile / oracle/Makefile need /root/reference)."""
import ctypes
import os

import numpy as np
import pytest

from tests import common
from oracle import ref

pytestmark = pytest.mark.gpu
BUILD = {"libnrsc5_hipdropin.so": os.path.join(common.ROOT, "integration", "_build"), "libnrsc5_plain.so": os.path.join(common.ROOT, "oracle", "_ref")}


def _run(libname, iq, chunk=32768, mode=0):
    path = os.path.join(BUILD[libname], libname)
    if not os.path.exists(path):
        pytest.skip(f"{libname} not prebuilt (integration/Makefile / oracle/Makefile need /root/reference)")
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
                va, vb = a[f], b[f]
                assert common.float_close(f, float(va), float(vb)), (k, f, va, vb)


@pytest.mark.parametrize("name", list(common.GOLDEN_AM_CASES))
def test_dropin_public_api_am(name, captures):
    """nrsc5_set_mode(NRSC5_MODE_AM) + nrsc5_pipe_samples_cs16 / _cu8: same HDC packets, SYNC (incl. the system
    control bits) and BER events as the unmodified reference."""
    iq = np.ascontiguousarray(captures(name).iq)
    exp = _run("libnrsc5_plain.so", iq, mode=1)
    got = _run("libnrsc5_hipdropin.so", iq, mode=1)
    assert sum(k == "hdc" for k, _ in exp) >= 4 and any(k == "ber" for k, _ in exp)
    _compare_events(exp, got)


@pytest.mark.parametrize("name", ["fm_cu8_cfo137", "fm_cu8_cfo-2400", "fm_mp2_cu8"])
def test_dropin_public_api_events_match_reference(name, captures):
    cap = captures(name)
    iq = np.ascontiguousarray(cap.iq)
    exp = _run("libnrsc5_plain.so", iq)
    got = _run("libnrsc5_hipdropin.so", iq)
    assert [k for k, _ in exp] == [k for k, _ in got]
    assert any(k == "hdc" for k, _ in exp) or name != "fm_cu8_cfo137"
    for (k, a), (_, b) in zip(exp, got):
        if k == "hdc":
            assert a["program"] == b["program"] and a["flags"] == b["flags"] and a["data"] == b["data"]
        elif k in ("sync", "mer", "ber"):
            for f in a:
                va, vb = a[f], b[f]
                assert common.float_close(f, float(va), float(vb)), (k, f, va, vb)


@pytest.mark.parametrize("name", ["fm_cu8_cfo137", "fm_cu8_cfo-2400"])
def test_dropin_public_api_fifo_seam_events_match_reference(name, captures, monkeypatch):
    """The FIFO seam of rounds 3 - 5 (NRSC5HIP_HOST_CAPTURE=0: pinned staging, decimator kernel on the ingest stream) through the public API -- what cs16 / AM sessions and a
    capture's fall-back use; every other FM cu8 test of this file runs the round-6 default (the session's bytes in a pinned capture the stream reads in place)."""
    monkeypatch.setenv("NRSC5HIP_HOST_CAPTURE", "0")
    iq = np.ascontiguousarray(captures(name).iq)
    exp = _run("libnrsc5_plain.so", iq)
    got = _run("libnrsc5_hipdropin.so", iq)
    assert len(exp) >= 2
    _compare_events(exp, got)


def test_two_dropin_sessions_in_one_process(captures):
    """Two nrsc5_open_pipe sessions of one process (one engine each: own HIP streams, staging and state) fed alternately in
    32768-byte calls: each delivers exactly the events it delivers alone (= the plain reference's)."""
    a = np.ascontiguousarray(captures("fm_cu8_cfo137").iq)
    b = np.ascontiguousarray(captures("fm_cu8_cfo-2400").iq)
    lib = ctypes.CDLL(os.path.join(BUILD["libnrsc5_hipdropin.so"], "libnrsc5_hipdropin.so"))
    if not hasattr(lib, "pipe_run_pair"):
        pytest.skip("drop-in not prebuilt with pipe_run_pair")
    vp, sz = ctypes.c_void_p, ctypes.c_size_t
    lib.pipe_run_pair.argtypes = [vp, sz, vp, sz, ctypes.c_uint, ctypes.POINTER(vp), ctypes.POINTER(sz), ctypes.POINTER(vp), ctypes.POINTER(sz)]
    p0, p1, n0, n1 = vp(), vp(), sz(), sz()
    assert lib.pipe_run_pair(a.ctypes.data, a.size, b.ctypes.data, b.size, 32768, ctypes.byref(p0), ctypes.byref(n0), ctypes.byref(p1), ctypes.byref(n1)) == 0
    got = [ref.parse_log(ctypes.string_at(p0, n0.value)), ref.parse_log(ctypes.string_at(p1, n1.value))]
    for iq, g in zip((a, b), got):
        exp = _run("libnrsc5_plain.so", iq)
        assert len(exp) >= 2
        _compare_events(exp, g)


def test_batch_shard_c_host_equals_python_path(tmp_path):
    """integration/batch_shard.c -- plain C, one host thread and one engine per visible GPU over the C ABI, stream k -> GPU k mod N --
    prints per stream the summary bench.py's ranks gather (blocks, P1 / PIDS counts, crc32 over the packed P1 frames in record
    order); here with N = 1 against the same captures through the Python binding."""
    import subprocess
    import zlib
    from nrsc5_amd import engine as eng, synth, channel
    from tests import engine_checks as ec
    exe = os.path.join(common.ROOT, "integration", "_build", "batch_shard")
    if not os.path.exists(exe):
        pytest.skip("integration/_build/batch_shard not prebuilt")
    caps = [synth.fm_mp1_capture(0, seed=80 + k, cfo_hz=c, offset=o, snr_db=20, n_blocks=40, chan=ch)
            for k, (c, o, ch) in enumerate([(40.0, 123, None), (-130.0, 3001, channel.Impairments(ppm=70.0)), (10.0, 1700, None), (250.0, 9, channel.Impairments(ppm=-45.0))])]
    per = min(c.iq.size for c in caps); per -= per % 4
    path = tmp_path / "caps.cu8"
    with open(path, "wb") as f:
        for c in caps:
            f.write(c.iq[:per].tobytes())
    out = subprocess.run([exe, str(path), str(len(caps)), str(per)], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, GPU_MAX_HW_QUEUES="8"))
    assert out.returncode == 0, out.stderr
    lines = [ln.split() for ln in out.stdout.strip().splitlines()]
    assert len(lines) == len(caps)
    # the Python path: same engine configuration, same entry points
    n = len(caps)
    stride = per + (-per) % 256
    buf = np.zeros((n, stride), dtype=np.uint8)
    for k, c in enumerate(caps):
        buf[k, :per] = c.iq[:per]
    nframes = per // (16 * 276480) + 1
    E = eng.Engine(max_streams=n, q15_capacity=2 * 71280, record_capacity=max(512, 2 * 16 * nframes + 64), p1_slots=nframes + 12, p1_async=True, l2_feedback=True, batch_zero_copy=True)
    dev = ec._to_device(E, buf)
    E.batch_append_cu8(dev, stride, [per] * n)
    E.batch_process(n)
    recs, counts, frames = E.batch_fetch_view(n)
    for k in range(n):
        r = recs[k, :counts[k]]
        h, p1 = 0, 0
        for x in r:
            if int(x["flags"]) & eng.REC_P1:
                h = zlib.crc32(frames[k, int(x["p1_slot"])].tobytes(), h); p1 += 1
        want = ["stream", str(k), "gpu", "0", "blocks", str(len(r)), "p1", str(p1), "pids", str(int(((r["flags"] & eng.REC_PIDS) != 0).sum())),
                "fine", str(int((r["state_after"] == eng.SYNC_FINE).sum())), "crc32", "%08x" % h]
        assert lines[k] == want, (lines[k], want)
        assert p1 >= 1
    ec._free_device(E, dev)
    E.close()


@pytest.mark.parametrize("strict", [0, 1])
def test_dropin_close_without_flush_delivers_the_last_block(captures, strict, monkeypatch):
    """src/main.c:1095-1121 ends a file with nrsc5_close, no zero-length nrsc5_pipe_samples_* call: the events of a block that is still on
    the device when the feeding loop ends must come out of nrsc5_close (overlapped delivery, opt-in since round 6); in the default mode (the reference's contract;
    also NRSC5HIP_SYNC_DELIVERY=1) nothing may be left for it.  The complete log equals the plain reference's in both modes."""
    monkeypatch.setenv("NRSC5HIP_SYNC_DELIVERY", str(strict))
    path = os.path.join(BUILD["libnrsc5_hipdropin.so"], "libnrsc5_hipdropin.so")
    if not os.path.exists(path):
        pytest.skip("libnrsc5_hipdropin.so not prebuilt")
    iq = np.ascontiguousarray(captures("fm_cu8_cfo137").iq)
    exp = _run("libnrsc5_plain.so", iq)
    lib = ctypes.CDLL(path)
    lib.pipe_run_opts.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_void_p)]
    lib.pipe_run_opts.restype = ctypes.c_size_t
    p, at_end = ctypes.c_void_p(), ctypes.c_size_t()
    n = lib.pipe_run_opts(iq.ctypes.data, iq.size, 32768, 0, 0, 1, ctypes.byref(at_end), ctypes.byref(p))
    got = ref.parse_log(ctypes.string_at(p, n))
    _compare_events(exp, got)
    if strict:
        assert at_end.value == n, "strict delivery: nothing may be left for nrsc5_close"


def test_dropin_set_mode_on_a_live_session():
    """the reference's reset of a USED session (FIR windows rewound, not cleared) through the public API, on the real library"""
    from tests import test_emu_dropin as emu
    path = os.path.join(BUILD["libnrsc5_hipdropin.so"], "libnrsc5_hipdropin.so")
    if not os.path.exists(path) or not os.path.exists(os.path.join(BUILD["libnrsc5_plain.so"], "libnrsc5_plain.so")):
        pytest.skip("drop-in / plain reference libraries not prebuilt (need /root/reference)")
    emu.check_dropin_set_mode_on_a_live_session(path)
