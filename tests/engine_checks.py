"""Parity checks of the HIP engine against the oracle / golden fixtures, written once and run twice:
  * tests/test_emu_logic.py  -- on the CPU SIMT-emulated build (kernel LOGIC only, `-m "not gpu"`),
  * tests/test_gpu_parity.py -- on the real gfx950 library through the C ABI (`-m gpu`)."""
from __future__ import annotations

import os

import numpy as np

from tests import common
from nrsc5_amd import engine as eng, synth

GOLDEN_DIR = os.path.join(os.path.dirname(__file__), "golden")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def check_halfband(lib, oracle, n=20011, seed=0):
    rng = np.random.default_rng(seed)
    E = eng.Engine(max_streams=1, q15_capacity=2 * 71280 + n, lib_path=lib)
    for size in (4, 8, 60, 4 * 1021, 4 * n):               # ragged sizes incl. shorter than the filter
        iq = rng.integers(0, 256, size=size, dtype=np.uint8)
        got = E.stage_halfband_fm_cu8(iq)
        exp, _ = oracle.halfband_fm_cu8(iq)
        assert np.array_equal(got, exp), f"half-band mismatch at size {size}"
    # extremes: full-scale input exercises the int16 accumulator range
    iq = np.tile(np.array([255, 0, 0, 255], dtype=np.uint8), 512)
    assert np.array_equal(E.stage_halfband_fm_cu8(iq), oracle.halfband_fm_cu8(iq)[0])
    E.close()


def check_halfband_streaming_history(lib, oracle):
    """Chunked pushes carry the 14-sample history exactly like one big push."""
    rng = np.random.default_rng(3)
    iq = rng.integers(0, 256, size=4 * 30000, dtype=np.uint8)
    exp, _ = oracle.halfband_fm_cu8(iq)
    E = eng.Engine(max_streams=1, q15_capacity=200000, lib_path=lib)
    cuts = [0, 4, 12, 4 * 7, 4 * 1000, 4 * 1003, 4 * 20000, iq.size]
    for a, b in zip(cuts[:-1], cuts[1:]):
        E.push_cu8(0, iq[a:b])
    import ctypes
    out = np.zeros((30000, 2), dtype=np.int16)
    # the FIFO slab of stream 0 starts at q15[0]; nothing was consumed (30000 < 71280)
    E.lib.nrsc5hip_debug_fetch  # keep symbol referenced
    from nrsc5_amd.engine import _Config  # noqa: F401
    got = _fetch_q15(E, 30000)
    assert np.array_equal(got, exp)
    E.close()


def _fetch_q15(E, n):
    """Reads the head of stream 0's FIFO through the stage API contract (test helper)."""
    # re-decimating is not possible without the input; use the debug path: FIFO is exposed via
    # a dedicated symbol only in the test helper below
    import ctypes
    fn = E.lib.nrsc5hip_debug_fetch_q15
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
    out = np.zeros((n, 2), dtype=np.int16)
    rc = fn(E._h, 0, n, out.ctypes.data)
    assert rc == 0
    return out


def check_fft(lib, oracle, n=4, seed=1, form=0):
    rng = np.random.default_rng(seed)
    E = eng.Engine(max_streams=1, q15_capacity=2 * 71280, lib_path=lib)
    if form:
        E.tune(eng.TUNE_MIXFFT_SYMS, form)                      # 32: the 256-lane FFT (8 x 8 x 8 x 4)
    x = (rng.standard_normal((n, 2048)) + 1j * rng.standard_normal((n, 2048))).astype(np.complex64)
    x[0] = 0; x[0, 5] = 1.0                                   # impulse: every twiddle path
    got = E.stage_fft2048(x)
    ref = np.fft.fft(x.astype(np.complex128), axis=1)
    err = np.linalg.norm(got - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-30)
    assert err.max() < 1e-5, f"FFT relative L2 error {err.max()}"     # <= 1e-5 vs float64 DFT
    orc = np.stack([oracle.fft(r) for r in x])
    assert np.abs(got - orc).max() <= 1e-4 * np.abs(orc).max()
    E.close()


def check_viterbi(lib, oracle, lens=(80, 2304), frames=3, seed=2, structured=True):
    rng = np.random.default_rng(seed)
    E = eng.Engine(max_streams=1, q15_capacity=2 * 71280, lib_path=lib)
    for L in lens:
        soft = rng.integers(-127, 128, size=(frames, 3 * L), dtype=np.int8)
        soft[:, 5::6] = 0                                      # punctured positions
        if structured:
            soft[0] = 0                                        # all-erasure frame: every ACS is a tie
            soft[1, :] = 127                                   # saturated
        got = E.stage_viterbi_k7(soft, L)
        exp = np.stack([oracle.viterbi_k7(s) for s in soft])
        assert np.array_equal(got, exp), f"Viterbi mismatch at len {L}"
    E.close()


def check_viterbi_roundtrip(lib, L=4608, frames=4, seed=9, flip=0.004):
    """encode -> noisy channel -> decode returns the message (domain property, any size)."""
    rng = np.random.default_rng(seed)
    E = eng.Engine(max_streams=1, q15_capacity=2 * 71280, lib_path=lib)
    msg = rng.integers(0, 2, size=(frames, L), dtype=np.uint8)
    coded = synth.conv_encode_k7(msg).reshape(frames, 3 * L).astype(np.int16) * 2 - 1
    soft = coded * 48 + rng.normal(0, 10, size=coded.shape)      # ~13.6 dB Es/N0 + sparse sign flips: error-free by a wide margin
    soft[rng.random(coded.shape) < flip] *= -1
    soft = np.clip(np.rint(soft), -127, 127).astype(np.int8)
    soft[:, 5::6] = 0
    got = E.stage_viterbi_k7(soft, L)
    assert np.array_equal(got, msg)
    E.close()


def run_capture(lib, cap, p1_async=False, chunk=32768 * 8, l2_feedback=False, tune=()):
    E = eng.Engine(max_streams=1, q15_capacity=max(2 * 71280 + chunk, 400000), lib_path=lib, p1_async=p1_async, l2_feedback=l2_feedback)
    for knob, value in tune:
        E.tune(knob, value)
    common.run_engine_streaming(E, 0, cap.iq, chunk=chunk)
    recs = E.drain(0)
    log = eng.records_to_log(E, 0, recs)
    return E, recs, log


def check_golden_end_to_end(lib, name, captures, tune=()):
    """Streaming seam vs the golden trace of the unmodified reference: frames exact, floats 1e-4."""
    g = golden(name)
    cap = captures(name)
    assert common.sha256(cap.iq) == str(g["iq_sha"])
    E, recs, log = run_capture(lib, cap, tune=tune)
    diffs = common.compare_logs(common.arrays_to_log(g), common.strip_states(log))
    assert not diffs, diffs[:10]
    E.close()
    return log


def check_oracle_end_to_end(lib, oracle, kw, soft_tol=1, garbage_frames_ok=False, p1_async=False):
    from oracle import port
    cap = synth.fm_mp1_capture(**kw)
    ol, _, _ = oracle.run(cap.iq, taps=port.TAP_SOFT)
    E, recs, log = run_capture(lib, cap, p1_async=p1_async)
    diffs = common.compare_logs(common.strip_states(ol), common.strip_states(log))
    if kw.get("mode", "MP1") != "MP1":
        # MP2/3/11: the Costas loops of the reference carriers that the new service mode adds start pulling in at lock;
        # until they settle the extended partitions equalise to garbage whose error power is chaotic in the last ulp of
        # libm.  The first EVENT_MER after lock (which averages those blocks) is therefore compared to 0.05 dB only.
        a = [v for k, v in common.strip_states(ol) if k == "mer"][:1]
        b = [v for k, v in common.strip_states(log) if k == "mer"][:1]
        kept = [r for r in common.strip_states(ol) if r[0] not in ("hdc", "soft", "vit", "amsym", "pxsoft")]   # compare_logs' view
        first = next((i for i, (k, _) in enumerate(kept) if k == "mer"), -1)
        if a and b and all(abs(a[0][f] - b[0][f]) < 0.05 for f in ("lower", "upper")):
            diffs = [d for d in diffs if not d.startswith(f"#{first} mer.")]
    if garbage_frames_ok:
        bad_ber = [v["cber"] for k, v in ol if k == "ber"]
        assert bad_ber and min(bad_ber) > 0.02, "expected an undecodable (false-lock) capture"
        diffs = [d for d in diffs if "frame.bits" not in d and "ber.cber" not in d]
        for (ka, va), (kb, vb) in zip(common.strip_states(ol), common.strip_states(log)):
            if ka == "frame":
                assert (va["bits"] != vb["bits"]).mean() < 0.01
    assert not diffs, diffs[:10]
    # soft bits are diagnostic: at most +-1 LSB on a small fraction of cells
    softs = [v for k, v in ol if k == "soft"]
    if softs:
        pm, _ = E.debug_fetch(0)
        last = softs[-1]
        d = pm.reshape(16, 23040)[last["bc"]].astype(int) - last["bits"].astype(int)
        assert np.abs(d).max() <= soft_tol and (d != 0).mean() < 0.01
    pxs = [v for k, v in ol if k == "pxsoft"]
    if pxs and pxs[-1]["bc"] % 2 == 1:                        # extended sidebands: soft bits of the last block pair
        px = E.debug_fetch_px(0)
        nch = 1 + max(x["ch"] for x in pxs[-4:])
        for v in pxs[-2 * nch:]:                              # the two blocks of the last pair, each channel
            n = len(v["bits"])
            got = px[v["ch"]].reshape(-1)[(v["bc"] % 2) * n:(v["bc"] % 2 + 1) * n]
            d = got.astype(int) - v["bits"].astype(int)
            assert np.abs(d).max() <= soft_tol and (d != 0).mean() < 0.01
    E.close()
    return log


def check_batch_equals_streaming(lib, caps, p1_async):
    """Batch path (device-resident captures, many streams per step) == per-stream streaming results."""
    n = len(caps)
    singles = []
    for cap in caps:
        E, recs, log = run_capture(lib, cap)
        singles.append(log)
        E.close()
    longest = max(c.iq.size for c in caps)
    stride = (longest + 15) // 16 * 16
    host = np.zeros((n, stride), dtype=np.uint8)
    for k, c in enumerate(caps):
        host[k, :c.iq.size] = c.iq
    E = eng.Engine(max_streams=n, q15_capacity=stride // 4 + 1024, record_capacity=256, p1_slots=4, p1_async=p1_async, lib_path=lib)
    dev = _to_device(E, host)
    E.batch_append_cu8(dev, stride, [c.iq.size - c.iq.size % 4 for c in caps])
    steps = E.batch_process(n)
    if p1_async:                                              # exercise the zero-copy view path too
        recs, counts, frames = E.batch_fetch_view(n)
    else:
        recs, counts, frames = E.batch_fetch(n)
    for k in range(n):
        log = eng.records_to_log(E, k, recs[k, :counts[k]], frames[k])
        diffs = common.compare_logs(singles[k], log, rtol=0.0)
        assert not diffs, (k, diffs[:10])
    if p1_async and n > 1:
        # second pass on the same engine with the captures rotated by one stream: the view's frames now come from the pinned
        # mirror the traceback kernel writes (set up by the first view), not from a copy of the ring
        E.reset_all()
        rot = [(k + 1) % n for k in range(n)]
        host2 = host[rot]
        dev2 = _to_device(E, host2)
        E.batch_append_cu8(dev2, stride, [caps[r].iq.size - caps[r].iq.size % 4 for r in rot])
        E.batch_process(n)
        recs, counts, frames = E.batch_fetch_view(n)
        for k in range(n):
            log = eng.records_to_log(E, k, recs[k, :counts[k]], frames[k])
            diffs = common.compare_logs(singles[rot[k]], log, rtol=0.0)
            assert not diffs, ("mirror pass", k, diffs[:10])
        _free_device(E, dev2)
    _free_device(E, dev)
    E.close()
    return steps


def check_zero_copy_batch(lib, caps, p1_async=True, l2_feedback=False, mixfft_syms=0, rtol=0.0, singles_tuned=False):
    """Engine option batch_zero_copy (K1 fused into the symbol kernel, captures read in place) == the copying batch path
    == the streaming seam, record for record; plus the error behaviour of the attached state."""
    import pytest
    n = len(caps)
    singles = []
    for cap in caps:
        E, recs, log = run_capture(lib, cap, l2_feedback=l2_feedback, tune=((eng.TUNE_MIXFFT_SYMS, mixfft_syms),) if singles_tuned else ())
        singles.append(log)
        E.close()
    stride = max(c.iq.size for c in caps); stride += (-stride) % 16
    host = np.zeros((n, stride), dtype=np.uint8)
    for k, c in enumerate(caps):
        host[k, :c.iq.size] = c.iq
    E = eng.Engine(max_streams=n, q15_capacity=2 * 71280, record_capacity=512, p1_slots=8, p1_async=p1_async, l2_feedback=l2_feedback,
                   batch_zero_copy=True, lib_path=lib)                           # FIFO at its minimum: nothing is copied into it
    if mixfft_syms:
        E.tune(eng.TUNE_MIXFFT_SYMS, mixfft_syms)                                   # symbol-kernel variants: identical bins
    dev = _to_device(E, host)
    sizes = [c.iq.size - c.iq.size % 4 for c in caps]
    E.batch_append_cu8(dev, stride, sizes)
    with pytest.raises(eng.Nrsc5HipError):
        E.batch_append_cu8(dev, stride, sizes)                                   # one append per reset
    with pytest.raises(eng.Nrsc5HipError):
        E.push_cu8(0, np.zeros(64, dtype=np.uint8))                              # attached streams take no pushes
    E.batch_process(n)
    recs, counts, frames = E.batch_fetch_view(n) if p1_async else E.batch_fetch(n)
    for k in range(n):
        log = eng.records_to_log(E, k, recs[k, :counts[k]], frames[k])
        diffs = common.compare_logs(singles[k], log, rtol=rtol)
        assert not diffs, (k, diffs[:10])
    # reset detaches: the same engine then takes the streaming seam again
    E.reset_all()
    E2log = []
    common.run_engine_streaming(E, 0, caps[0].iq[:4 * 70000], chunk=4 * 35000)
    assert len(E.drain(0)) == 0                                                  # (less than one window; the point is that the push is accepted)
    _free_device(E, dev)
    E.close()


def _to_device(E, host: np.ndarray) -> int:
    import ctypes
    fn = E.lib.nrsc5hip_debug_alloc_copy
    fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    fn.restype = ctypes.c_void_p
    p = fn(host.ctypes.data, host.nbytes)
    assert p
    return p


def _free_device(E, p: int):
    import ctypes
    fn = E.lib.nrsc5hip_debug_free
    fn.argtypes = [ctypes.c_void_p]
    fn(p)


def check_interleaved_streams(lib, p1_async=False):
    """Streams of one engine advanced through alternating calls: the scheduler's launch flags (acquisition / PX kernels)
    measured on one stream set must not leak into a call that lists another set.  (a) two streams fed by block-sized
    interleaved pushes, the second starting late while the first is FINE; (b) batch_process over alternating subsets.
    Each stream's log must equal its single-stream run."""
    caps = [synth.fm_mp1_capture(0, seed=90 + k, cfo_hz=c, offset=o, snr_db=20, n_blocks=nb, mode=m)
            for k, (c, o, nb, m) in enumerate([(20.0, 500, 12, "MP1"), (-700.0, 2900, 10, "MP1"), (5.0, 64, 44, "MP3")])]
    singles = []
    for cap in caps:
        E, recs, log = run_capture(lib, cap, p1_async=p1_async)
        singles.append(log)
        E.close()
    # (a) interleaved pushes of one block's worth of input; stream 1 starts after stream 0 locked
    E = eng.Engine(max_streams=3, q15_capacity=400000, record_capacity=256, p1_slots=4, p1_async=p1_async, lib_path=lib)
    chunk = 4 * 70200
    pos = [0, -3 * chunk, 0]
    while any(p < c.iq.size for p, c in zip(pos, caps)):
        for k, c in enumerate(caps):
            if 0 <= pos[k] < c.iq.size:
                part = c.iq[pos[k]:pos[k] + chunk]
                E.push_cu8(k, part[:part.size - part.size % 4])
            pos[k] += chunk
    for k in range(3):
        log = eng.records_to_log(E, k, E.drain(k))
        diffs = common.compare_logs(singles[k], log, rtol=0.0)
        assert not diffs, ("interleaved pushes", k, diffs[:6])
    E.close()
    # (b) batch over alternating subsets, a few steps at a time
    n = len(caps)
    stride = max(c.iq.size for c in caps); stride += (-stride) % 16
    host = np.zeros((n, stride), dtype=np.uint8)
    for k, c in enumerate(caps):
        host[k, :c.iq.size] = c.iq
    E = eng.Engine(max_streams=n, q15_capacity=stride // 4 + 1024, record_capacity=256, p1_slots=4, p1_async=p1_async, lib_path=lib)
    dev = _to_device(E, host)
    E.batch_append_cu8(dev, stride, [c.iq.size - c.iq.size % 4 for c in caps])
    for rnd in range(40):
        did = 0
        for subset in ([0], [1, 2], [0, 2], [1]):
            did += E.batch_process(len(subset), stream_ids=subset, max_steps=3)
        if did == 0:
            break
    recs, counts, frames = E.batch_fetch(n)
    for k in range(n):
        log = eng.records_to_log(E, k, recs[k, :counts[k]], frames[k])
        diffs = common.compare_logs(singles[k], log, rtol=0.0)
        assert not diffs, ("subset batches", k, diffs[:6])
    _free_device(E, dev)
    E.close()


def check_small_fifo_compaction(lib, name, captures):
    """Streaming seam with the minimum FIFO: the unread tail is compacted many times; results unchanged."""
    g = golden(name)
    cap = captures(name)
    chunk = 4 * 9000
    E = eng.Engine(max_streams=1, q15_capacity=2 * 71280 + chunk // 4, lib_path=lib)
    common.run_engine_streaming(E, 0, cap.iq, chunk=chunk)
    log = eng.records_to_log(E, 0, E.drain(0))
    diffs = common.compare_logs(common.arrays_to_log(g), common.strip_states(log))
    assert not diffs, diffs[:10]
    E.close()


def check_api_edges(lib):
    """Error behaviour of the C ABI: bad arguments are rejected with codes, never crashes; empty pushes are no-ops."""
    import pytest
    with pytest.raises(eng.Nrsc5HipError):
        eng.Engine(max_streams=0, lib_path=lib)
    with pytest.raises(eng.Nrsc5HipError):
        eng.Engine(max_streams=1, q15_capacity=1000, lib_path=lib)          # below 2 * 71280
    E = eng.Engine(max_streams=2, q15_capacity=2 * 71280 + 64, lib_path=lib)
    E.push_cu8(0, np.zeros(0, dtype=np.uint8))                              # empty push
    assert len(E.drain(0)) == 0
    with pytest.raises(eng.Nrsc5HipError):
        E.push_cu8(5, np.zeros(8, dtype=np.uint8))                          # stream out of range
    with pytest.raises(eng.Nrsc5HipError):
        E._check(E.lib.nrsc5hip_push_cu8(E._h, 0, np.zeros(8, dtype=np.uint8).ctypes.data, 6))   # length not a multiple of 4
    with pytest.raises(eng.Nrsc5HipError):
        E.batch_append_cu8(0, 16, [4 * (2 * 71280 + 1000)])                 # exceeds q15_capacity -> EOVERFLOW before touching memory
    # a short push below one window produces no records, and a reset brings the stream back to a fresh session
    E.push_cu8(1, np.full(4 * 1000, 127, dtype=np.uint8))
    assert len(E.drain(1)) == 0
    E.reset(1)
    E.close()


def check_large_push_minimum_fifo(lib, oracle):
    """ONE push of a whole capture into an engine whose FIFO is at the legal minimum (2 windows): the fast seam stages at most up to
    the sample that completes the next block, so the push is processed block by block, never overflows and leaves the host mirror
    exact (round 3: EOVERFLOW with the samples dropped but counted).  Log == oracle; bytes_to_next_block stays consistent."""
    cap = synth.fm_mp1_capture(0, seed=61, cfo_hz=33.0, offset=1999, snr_db=20, n_blocks=10)
    ol, _, _ = oracle.run(cap.iq)
    E = eng.Engine(max_streams=1, q15_capacity=2 * 71280, lib_path=lib)
    E.push_cu8(0, cap.iq[:cap.iq.size - cap.iq.size % 4])                   # ~2.8 MB in one call
    log = eng.records_to_log(E, 0, E.drain(0))
    diffs = common.compare_logs(common.strip_states(ol), common.strip_states(log))
    assert not diffs, diffs[:5]
    nb = E.lib.nrsc5hip_bytes_to_next_block(E._h, 0, 1)
    assert 0 < nb <= 4 * 71280
    E.close()


def check_cs16_batch(lib, captures):
    """cs16 captures through the batch path (bypasses K1) == golden."""
    g = golden("fm_cs16_cfo60")
    cap = captures("fm_cs16_cfo60")
    E = eng.Engine(max_streams=1, q15_capacity=cap.iq.size // 2 + 1024, record_capacity=256, p1_slots=4, p1_async=True, lib_path=lib)
    dev = _to_device(E, cap.iq)
    E.batch_append_cs16(dev, 0, [cap.iq.size])
    E.batch_process(1)
    recs, counts, frames = E.batch_fetch(1)
    log = eng.records_to_log(E, 0, recs[0, :counts[0]], frames[0])
    diffs = common.compare_logs(common.arrays_to_log(g), common.strip_states(log))
    assert not diffs, diffs[:10]
    _free_device(E, dev)
    E.close()


# ---- AM -----------------------------------------------------------------------------------------------------------
E1_GENS, E2_GENS = (0o561, 0o657, 0o711), (0o561, 0o753, 0o711)


def check_viterbi_k9(lib, oracle, lens=(80, 3750), frames=3, seed=4):
    rng = np.random.default_rng(seed)
    E = eng.Engine(max_streams=1, q15_capacity=2 * 71280, lib_path=lib)
    for L in lens:
        for gens in (E1_GENS, E2_GENS):
            soft = rng.integers(-1, 2, size=(frames, 3 * L), dtype=np.int8)       # AM inputs are hard +-1 / 0 (punctured)
            soft[0] = 0                                                            # all-erasure: every ACS is a tie
            soft[1] = 1
            got = E.stage_viterbi_k9(soft, L, gens)
            exp = np.stack([oracle.viterbi(s, 9, gens) for s in soft])
            assert np.array_equal(got, exp), f"K=9 Viterbi mismatch at len {L} gens {gens}"
    E.close()


def check_first_header(lib, oracle, seeds=(5, 6)):
    """The workgroup form of frame_process's first-header check (l2_header.h: parallel syndromes, serial RS decoder only when they
    are non-zero) against the oracle's restatement of frame.c:516-540 on every case of synth_l2.test_frames -- clean headers, 1 / 4
    (correctable) and 5 / 9 (not) byte errors, every PCI, garbage -- plus a sweep of 0..6 errors; FM and AM frames; one wave and
    several per workgroup."""
    from nrsc5_amd import synth_l2
    E = eng.Engine(max_streams=1, q15_capacity=2 * 71280, lib_path=lib)
    for nbits in (146176, 3750):
        frames = []
        for seed in seeds:
            frames += [bits for _, bits, _ in synth_l2.test_frames(nbits, seed=seed)]
        rng = np.random.default_rng(nbits)
        body = synth_l2.make_pdu(rng, synth_l2.pdu_bytes_of(nbits), nop=2)
        for k in range(7):
            for _ in range(3):
                frames.append(synth_l2.frame_from_bytes(synth_l2.corrupt(body, rng.choice(96, size=k, replace=False), rng), nbits))
        bits = np.stack(frames).astype(np.uint8)
        exp = np.array([1 if oracle.l2_first_header_ok(b) else 0 for b in bits], dtype=np.int32)
        assert 0 < exp.sum() < len(exp)
        for threads in (64, 256):
            got = E.stage_first_header(bits, threads)
            assert np.array_equal(got, exp), (nbits, threads, np.nonzero(got != exp)[0][:8])
    E.close()


def conv_encode_k9(bits: np.ndarray, gens) -> np.ndarray:
    """Tail-biting rate-1/3 K=9 code words as decode.c's bit_errors() re-encodes them: register bit 8 - k = bits[i - k]."""
    L = len(bits)
    reg = np.zeros(L, dtype=np.uint32)
    for k in range(9):
        reg |= np.roll(bits, k).astype(np.uint32) << (8 - k)
    par = lambda v: np.array([bin(int(x)).count("1") & 1 for x in v], dtype=np.int8)
    return np.stack([par(reg & g) for g in gens], axis=1).reshape(3 * L)


def check_viterbi_k9_segmented(lib, oracle, lens=(3750, 24000), segments=(2, 3, 8), seed=14):
    """K=9 decode in segment waves (k_am.hip: forward pass AND traceback speculate across segment boundaries, a checking wave accepts
    or re-runs each segment): the sequential decoder's bits for any segment count -- on noise, on the all-erasure frame (every ACS a
    tie), on constant input and on noisy code words; and again with warm-up and run-in switched off (test hook), when the
    speculation is wrong at nearly every boundary of an informative frame and the result rests on the repairs (counted)."""
    rng = np.random.default_rng(seed)
    E = eng.Engine(max_streams=1, q15_capacity=2 * 71280, lib_path=lib)
    for L in lens:
        for gens in (E1_GENS, E2_GENS):
            soft = rng.integers(-1, 2, size=(5, 3 * L), dtype=np.int8)
            soft[0] = 0
            soft[1] = 1
            for f in (2, 3):
                msg = rng.integers(0, 2, size=L, dtype=np.uint8)
                cw = conv_encode_k9(msg, gens).astype(np.int8) * 2 - 1
                flip = rng.random(3 * L) < (0.04 if f == 2 else 0.12)
                soft[f] = np.where(flip, -cw, cw)
                soft[f, 4::5] = 0                                                  # punctured positions
            exp = np.stack([oracle.viterbi(s, 9, gens) for s in soft])
            for warm in (1, 0):
                E.tune(eng.TUNE_AM_WARM, warm)
                for G in segments:
                    E.tune(eng.TUNE_AM_SEGMENTS, G)
                    a = E.k9_stats()
                    got = E.stage_viterbi_k9(soft, L, gens)
                    b = E.k9_stats()
                    assert np.array_equal(got, exp), f"segmented K=9 Viterbi mismatch at len {L} gens {gens}, {G} segments, warm {warm}"
                    assert b[0] - a[0] > 0 and b[2] - a[2] > 0, (a, b)
                    if not warm:
                        assert b[1] - a[1] >= 2 and b[3] - a[3] >= 2, (L, G, a, b)   # cold starts on informative frames were repaired, not trusted
    E.tune(eng.TUNE_AM_SEGMENTS, 1)                                                # the single-wave form (in-order mode) still agrees
    soft = rng.integers(-1, 2, size=(2, 3 * 3750), dtype=np.int8)
    assert np.array_equal(E.stage_viterbi_k9(soft, 3750, E1_GENS), np.stack([oracle.viterbi(s, 9, E1_GENS) for s in soft]))
    E.close()


def check_am_decimator(lib, oracle, seed=5):
    """cu8 -> 32:1 cascade, ragged pushes (stage phases and the raw history carry over), exact."""
    rng = np.random.default_rng(seed)
    iq = rng.integers(0, 256, size=64 * 700 + 24, dtype=np.uint8)
    exp = oracle.am_decimate_cu8([iq])
    E = eng.Engine(max_streams=1, q15_capacity=2 * 71280, lib_path=lib, am_enable=True)
    E.set_mode(0, eng.MODE_AM)
    cuts = [0, 4, 8, 60, 64, 1000, 1004, 64 * 300 + 12, 64 * 301, iq.size]
    for a, b in zip(cuts[:-1], cuts[1:]):
        E.push_cu8(0, iq[a:b])
    got = _fetch_q15(E, exp.shape[0])
    assert exp.shape[0] == 700 and np.array_equal(got, exp)
    E.close()


def run_am_capture(lib, cap, chunk=32768):
    E = eng.Engine(max_streams=1, q15_capacity=400000, record_capacity=512, p1_slots=16, lib_path=lib, am_enable=True)
    E.set_mode(0, eng.MODE_AM)
    common.run_engine_streaming(E, 0, cap.iq, chunk=chunk)
    recs = E.drain(0)
    return E, recs, eng.am_records_to_log(E, 0, recs)


def am_symbol_agreement(E, ol):
    """Hard symbols are diagnostic (a cell on a decision boundary may flip with a 1-ulp libm difference)."""
    return None


def check_am_oracle_end_to_end(lib, oracle, kw, chunk=32768, max_sym_diff=0.002):
    from oracle import port
    from nrsc5_amd import synth_am
    cap = synth_am.am_ma1_capture(**kw)
    ol, _, _ = oracle.run(cap.iq, mode=1, taps=port.TAP_SOFT)
    E, recs, log = run_am_capture(lib, cap, chunk=chunk)
    diffs = common.compare_logs(common.strip_states(ol), common.strip_states(log))
    assert not diffs, diffs[:10]
    E.close()
    return log


def check_am_golden_end_to_end(lib, name, captures):
    g = golden(name)
    cap = captures(name)
    assert common.sha256(cap.iq) == str(g["iq_sha"])
    E, recs, log = run_am_capture(lib, cap)
    diffs = common.compare_logs(common.am_arrays_to_log(g), common.strip_states(log))
    assert not diffs, diffs[:10]
    E.close()
    return log


def check_am_batch_equals_streaming(lib, kws, p1_async=False):
    """Device-resident batch of AM captures (cs16 and cu8 lists) == the same captures through the streaming seam."""
    from nrsc5_amd import synth_am
    caps = [synth_am.am_ma1_capture(**kw) for kw in kws]
    logs = []
    for cap in caps:
        E, recs, log = run_am_capture(lib, cap)
        logs.append(log)
        E.close()
    n = len(caps)
    fmt = caps[0].iq.dtype
    stride = max(c.iq.size for c in caps)
    stride += (-stride) % 64
    E = eng.Engine(max_streams=n, q15_capacity=stride // 2 + 1024, record_capacity=512, p1_slots=16, lib_path=lib, am_enable=True, p1_async=p1_async)
    for k in range(n):
        E.set_mode(k, eng.MODE_AM)
    buf = np.zeros((n, stride), dtype=fmt)
    for k, c in enumerate(caps):
        buf[k, :c.iq.size] = c.iq
    dev = E.lib.nrsc5hip_debug_alloc_copy
    import ctypes
    dev.restype = ctypes.c_void_p
    dev.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    dptr = dev(buf.ctypes.data, buf.nbytes)
    assert dptr
    sizes = [c.iq.size - c.iq.size % 4 for c in caps]
    if fmt == np.uint8:
        E.batch_append_cu8(dptr, stride, sizes)
    else:
        E.batch_append_cs16(dptr, stride, sizes)
    E.batch_process(n)
    recs, counts, frames = E.batch_fetch_view(n) if p1_async else E.batch_fetch(n)
    for k in range(n):
        log = eng.am_records_to_log(E, k, recs[k, :counts[k]], frames[k])
        d = common.compare_logs(logs[k], log, rtol=0.0)
        assert not d, (k, d[:5])
    E.lib.nrsc5hip_debug_free.argtypes = [ctypes.c_void_p]
    E.lib.nrsc5hip_debug_free(dptr)
    E.close()


def check_am_reduced_bandwidth(lib, oracle):
    """RDBI = 1 (system control bit 15 of the AM reference carrier): the receiver decodes the eight P1 frames of an L1 frame and NO P3
    frame, and the frame's BER is over the P1 bits only (decode.c:520-545) -- in order, and in the window pipeline where the P3 segment
    waves, their checks and the frame accounting have to stand down (k_am_decode_*)."""
    from nrsc5_amd import synth_am
    kw = dict(n_frames=11, seed=21, cfo_hz=1.5, offset=700, rdbi=1)
    check_am_oracle_end_to_end(lib, oracle, kw)
    cap = synth_am.am_ma1_capture(**kw)
    ol, _, _ = oracle.run(cap.iq, mode=1)
    assert sum(1 for k, v in ol if k == "frame" and v["lc"] == 0) >= 16 and not any(k == "frame" and v["lc"] == 1 for k, v in ol)
    assert [v["rdbi"] for k, v in ol if k == "sync"] == [1]
    check_am_batch_equals_streaming(lib, [kw, dict(n_frames=9, seed=7, cfo_hz=2.0, offset=600)], p1_async=True)


def check_l2_feedback(lib, oracle, kw, am=False):
    """Engine-side L2 -> L1 feedback (RS(255,247) first-header check on the device, in-order decode) == the oracle driven
    by the restated frame_process decision, which itself is pinned against the unmodified reference incl. its L2."""
    from nrsc5_amd import synth_am
    if am:
        cap = synth_am.am_ma1_capture(**kw)
        ol, _, _ = oracle.run(cap.iq, mode=1, p1_hook=oracle.l2_hook())
        E = eng.Engine(max_streams=1, q15_capacity=400000, record_capacity=512, p1_slots=16, lib_path=lib, am_enable=True, l2_feedback=True)
        E.set_mode(0, eng.MODE_AM)
        common.run_engine_streaming(E, 0, cap.iq, chunk=32768)
        log = eng.am_records_to_log(E, 0, E.drain(0))
    else:
        cap = synth.fm_mp1_capture(**kw)
        ol, _, _ = oracle.run(cap.iq, p1_hook=oracle.l2_hook())
        E, recs, log = run_capture(lib, cap, l2_feedback=True)
    assert any(k == "lost_sync" for k, _ in ol), "capture does not exercise the feedback"
    diffs = common.compare_logs(common.strip_states(ol), common.strip_states(log))
    # frames decoded while falsely locked are Viterbi output on noise (see check_oracle_end_to_end)
    bad = {i for i, (k, v) in enumerate([r for r in common.strip_states(ol) if r[0] not in ("hdc", "soft", "vit", "amsym", "pxsoft")])
           if k == "ber" and v["cber"] > 0.02}
    diffs = [d for d in diffs if not any(d.startswith(f"#{i} ") or d.startswith(f"#{i + 1} frame") or d.startswith(f"#{i - 1} frame") for i in bad)]
    assert not diffs, diffs[:10]
    E.close()
    return log


def check_mode_switch(lib, reflib):
    """nrsc5_set_mode on a live session: FM capture, then AM, then FM again on the same stream -- each run equals the unmodified reference's on ONE session
    driven the same way (input_set_mode resets the stream, input.c:158-162; what the reset leaves in place -- FIR windows, sync_t.samperr / .angle / .bc --
    shows in the block records of the next capture's un-synchronised blocks and, after an FM capture, in the first synchronised AM block's angle)."""
    import pytest
    from nrsc5_amd import synth_am
    fm = synth.fm_mp1_capture(0, seed=71, cfo_hz=33.0, offset=640, snr_db=20, n_blocks=20)
    am = synth_am.am_ma1_capture(9, seed=72, cfo_hz=1.0, offset=900)
    fm_iq = fm.iq[:fm.iq.size - fm.iq.size % 4]
    exp = [common.strip_states(l) for l, _ in reflib.run_epochs([(eng.MODE_FM, fm_iq), (eng.MODE_AM, am.iq), (eng.MODE_FM, fm_iq)])]
    for k in (0, 1):                                               # LOST_SYNC fired inside the next nrsc5_set_mode (input_reset -> input_set_sync_state): the caller's own doing
        if exp[k] and exp[k][-1][0] == "lost_sync":
            exp[k] = exp[k][:-1]
    plain = eng.Engine(max_streams=1, q15_capacity=400000, lib_path=lib)
    with pytest.raises(eng.Nrsc5HipError):
        plain.set_mode(0, eng.MODE_AM)                        # engine created without am_enable
    with pytest.raises(eng.Nrsc5HipError):
        plain.set_mode(0, 7)
    plain.close()
    E = eng.Engine(max_streams=2, q15_capacity=400000, record_capacity=512, p1_slots=16, lib_path=lib, am_enable=True)
    for k, mode in enumerate((eng.MODE_FM, eng.MODE_AM, eng.MODE_FM)):
        E.set_mode(1, mode)
        iq = fm_iq if mode == eng.MODE_FM else am.iq
        common.run_engine_streaming(E, 1, iq, chunk=32768)
        (E.push_cu8 if iq.dtype == np.uint8 else E.push_cs16)(1, iq[:0])
        recs = E.drain(1)
        log = eng.records_to_log(E, 1, recs) if mode == eng.MODE_FM else eng.am_records_to_log(E, 1, recs)
        diffs = common.compare_logs(exp[k], common.strip_states(log))
        assert not diffs, (k, mode, diffs[:5])
    E.close()


def check_reset_keeps_fir_windows(lib, reflib, offsets=(0, 10)):
    """nrsc5hip_stream_reset on a USED stream == input_reset on a used session of the unmodified reference (nrsc5_set_mode on a live
    pipe session): firdecim_q15_reset rewinds the FIR windows without clearing them (firdecim_q15.c:53-56), so
      * the half-band's first 7 outputs see the 14 samples its window's last compaction left at the front (Q15 samples bit for bit),
      * the acquisition filter's first 31 outputs see 31 samples of the previous capture's last un-synchronised block -- visible in the
        first block's timing pick / angle when the new capture's symbol boundary falls into the first samples of the window.
    Capture A is full-scale noise (nothing locks: every block runs the acquisition filter), capture B a signal whose boundary sits at
    sample ~0 / ~5 of the window: in the REFERENCE the used session's log differs from a fresh session's (asserted, so the check cannot
    pass vacuously); the engine equals the used session after reset() and the fresh one after fresh()."""
    from oracle import ref
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, size=4 * 71280 * 3 + 4 * 5000, dtype=np.uint8)
    E = eng.Engine(max_streams=1, q15_capacity=400000, lib_path=lib)
    differing = 0
    for off in offsets:
        b = synth.fm_mp1_capture(0, seed=82, cfo_hz=120.0, offset=off, snr_db=20, n_blocks=20).iq
        b = b[:b.size - b.size % 4]
        _, used_log, used_q15 = reflib.run_with_mode_switch(a, b, taps=ref.TAP_Q15)
        fresh_log, fresh_q15, _ = reflib.run(b, taps=ref.TAP_Q15)
        assert (used_q15[:7] != fresh_q15[:7]).any() and np.array_equal(used_q15[7:20000], fresh_q15[7:20000])
        differing += bool(common.compare_logs(common.strip_states(fresh_log), common.strip_states(used_log)))
        for how, exp_log, exp_q15 in (("reset", used_log, used_q15), ("fresh", fresh_log, fresh_q15)):
            E.fresh(0)
            common.run_engine_streaming(E, 0, a, chunk=32768)
            E.drain(0)
            getattr(E, how)(0)
            E.push_cu8(0, b[:4 * 20000])                           # below one window: nothing is consumed, the FIFO starts at the reset
            assert np.array_equal(_fetch_q15(E, 20000), exp_q15[:20000]), (off, how)
            common.run_engine_streaming(E, 0, b[4 * 20000:], chunk=32768)
            E.push_cu8(0, np.zeros(0, dtype=np.uint8))
            log = eng.records_to_log(E, 0, E.drain(0))
            diffs = common.compare_logs(common.strip_states(exp_log), common.strip_states(log))
            assert not diffs, (off, how, diffs[:5])
    assert differing >= 1, "the reference's used session no longer differs from a fresh one: the captures lost their point"
    E.close()


def check_reset_keeps_fir_windows_am(lib, reflib, offsets=(132, 138)):
    """The same for filter_am (AM cs16: a noise capture, then a signal whose symbol boundary sits at the start of the window), and across a
    mode switch: an AM cu8 capture pushes decim[0] with samples >> 4 (input.c:70-76) -- an FM session on the same nrsc5_t afterwards starts
    its half-band from THOSE 14 samples, while its acquisition filter (filter_fm, never used so far) starts from zeros."""
    from oracle import ref
    from nrsc5_amd import synth_am
    rng = np.random.default_rng(6)
    a = rng.integers(-20000, 20000, size=2 * 8910 * 3 + 2 * 700, dtype=np.int16)
    E = eng.Engine(max_streams=1, q15_capacity=400000, record_capacity=512, p1_slots=16, lib_path=lib, am_enable=True)
    E.set_mode(0, eng.MODE_AM)
    differing = 0
    for off in offsets:
        b = synth_am.am_ma1_capture(9, seed=72, cfo_hz=1.0, offset=off).iq
        _, used_log, _ = reflib.run_with_mode_switch(a, b, mode=ref.MODE_AM)
        fresh_log, _, _ = reflib.run(b, mode=ref.MODE_AM)
        differing += bool(common.compare_logs(common.strip_states(fresh_log), common.strip_states(used_log)))
        for how, exp_log in (("reset", used_log), ("fresh", fresh_log)):
            E.fresh(0)
            common.run_engine_streaming(E, 0, a, chunk=32768)
            E.drain(0)
            getattr(E, how)(0)
            common.run_engine_streaming(E, 0, b, chunk=32768)
            E.push_cs16(0, np.zeros(0, dtype=np.int16))
            log = eng.am_records_to_log(E, 0, E.drain(0))
            diffs = common.compare_logs(common.strip_states(exp_log), common.strip_states(log))
            assert not diffs, (off, how, diffs[:5])
    assert differing == len(offsets), "the reference's used AM session no longer differs from a fresh one"
    # AM cu8 -> AM cu8: all five stages of the 32:1 cascade (input.c:70-88) start from what their windows' last compactions left
    a8 = rng.integers(0, 256, size=4 * 100000, dtype=np.uint8)                 # 200 000 raw samples: decim[4] has compacted (2034 x 16 = 32 544)
    b = synth_am.am_ma1_capture(9, seed=72, cfo_hz=1.0, offset=132, fmt="cu8").iq
    b = b[:b.size - b.size % 4]
    _, used_log, used_q15 = reflib.run_with_mode_switch(a8, b, mode=ref.MODE_AM, taps=ref.TAP_Q15)
    fresh_log, fresh_q15, _ = reflib.run(b, mode=ref.MODE_AM, taps=ref.TAP_Q15)
    assert (used_q15[:7] != fresh_q15[:7]).any() and np.array_equal(used_q15[40:8000], fresh_q15[40:8000])
    for how, exp_log, exp_q15 in (("reset", used_log, used_q15), ("fresh", fresh_log, fresh_q15)):
        for first in (4 * 64000, 4 * 8, 4 * 200):                                  # the seeds are applied however the caller cuts its first pushes
            E.fresh(0)
            common.run_engine_streaming(E, 0, a8, chunk=32768)
            E.drain(0)
            getattr(E, how)(0)
            E.push_cu8(0, b[:first])
            E.push_cu8(0, b[first:4 * 64000])                                      # 4000 outputs: below one window, the FIFO starts at the reset
            assert np.array_equal(_fetch_q15(E, 4000), exp_q15[:4000]), (how, first)
        common.run_engine_streaming(E, 0, b[4 * 64000:], chunk=32768)
        E.push_cu8(0, np.zeros(0, dtype=np.uint8))
        log = eng.am_records_to_log(E, 0, E.drain(0))
        diffs = common.compare_logs(common.strip_states(exp_log), common.strip_states(log))
        assert not diffs, (how, diffs[:5])
    # AM cu8 -> FM cu8 on one session
    a8 = rng.integers(0, 256, size=4 * 40000, dtype=np.uint8)
    b = synth.fm_mp1_capture(0, seed=82, cfo_hz=120.0, offset=0, snr_db=20, n_blocks=20).iq
    b = b[:b.size - b.size % 4]
    _, used_log, used_q15 = reflib.run_with_mode_switch(a8, b, mode=ref.MODE_AM, mode_b=ref.MODE_FM, taps=ref.TAP_Q15)
    fresh_log, fresh_q15, _ = reflib.run(b, taps=ref.TAP_Q15)
    assert (used_q15[:7] != fresh_q15[:7]).any() and np.array_equal(used_q15[7:20000], fresh_q15[7:20000])
    E.fresh(0)
    common.run_engine_streaming(E, 0, a8, chunk=32768)
    E.drain(0)
    E.set_mode(0, eng.MODE_FM)
    E.push_cu8(0, b[:4 * 20000])
    assert np.array_equal(_fetch_q15(E, 20000), used_q15[:20000])
    common.run_engine_streaming(E, 0, b[4 * 20000:], chunk=32768)
    E.push_cu8(0, np.zeros(0, dtype=np.uint8))
    log = eng.records_to_log(E, 0, E.drain(0))
    diffs = common.compare_logs(common.strip_states(used_log), common.strip_states(log))
    assert not diffs, diffs[:5]
    E.close()


def check_reset_window_boundaries(lib, reflib):
    """push() compacts a FIR window at push number k (2048 - (ntaps - 1)) (firdecim_q15.c:58-67): captures whose raw length sits just below, at
    and just above those counts -- for the FM half-band and for every stage of the AM cu8 cascade (stage l takes 2 floor(raw / 2^(l+1)) samples)
    -- cut into pushes that end at, straddle or are far smaller than the boundary; the decimated samples after the reset equal the reference's."""
    from oracle import ref
    from nrsc5_amd import synth_am
    rng = np.random.default_rng(11)
    b = synth.fm_mp1_capture(0, seed=82, cfo_hz=120.0, offset=0, snr_db=20, n_blocks=1).iq[:4 * 3000]
    E = eng.Engine(max_streams=1, q15_capacity=400000, record_capacity=512, p1_slots=16, lib_path=lib, am_enable=True)
    for n in (20, 2032, 2034, 2036, 4068, 2034 * 5 + 2):
        a = rng.integers(0, 256, size=2 * n, dtype=np.uint8)
        _, _, used = reflib.run_with_mode_switch(a, b, taps=ref.TAP_Q15)
        for cuts in ([], [4, 2 * n - 4], [2 * 2034], [2 * 2020, 2 * 2034 + 4], list(range(400, 2 * n, 400))):
            E.fresh(0)
            prev = 0
            for c in [c for c in cuts if 0 < c < 2 * n and c % 4 == 0] + [2 * n]:
                E.push_cu8(0, a[prev:c]); prev = c
            E.reset(0)
            E.push_cu8(0, b)
            assert np.array_equal(_fetch_q15(E, 3000), used[:3000]), ("fm", n, cuts[:4])
    bam = synth_am.am_ma1_capture(1, seed=72, cfo_hz=1.0, offset=132, fmt="cu8").iq[:4 * 32000]
    E.set_mode(0, eng.MODE_AM)
    for l in (1, 2, 3, 4):
        for d in (-2, 0, 2):
            n = 2034 * (1 << l) + d * (1 << l) + 64
            a = rng.integers(0, 256, size=2 * n, dtype=np.uint8)
            _, _, used = reflib.run_with_mode_switch(a, bam, mode=ref.MODE_AM, taps=ref.TAP_Q15)
            for chunk in (2 * n, 1000):
                E.fresh(0)
                for off in range(0, 2 * n, chunk):
                    E.push_cu8(0, a[off:off + chunk])
                E.reset(0)
                E.push_cu8(0, bam)
                assert np.array_equal(_fetch_q15(E, 1000), used[:1000]), ("am", l, d, chunk)
    E.close()


def check_pids_crc_flag(lib, oracle, am=False):
    """REC_PIDS_CRC == pids_frame_push's CRC-12 decision (restated in the oracle, pinned against the reference's
    STATION_ID events): frames with a fresh station id in every block, every fifth one with a broken CRC."""
    from nrsc5_amd import synth_am
    if am:
        cap = synth_am.am_ma1_capture(8, seed=12, cfo_hz=1.0, offset=700)
        E = eng.Engine(max_streams=1, q15_capacity=400000, record_capacity=512, p1_slots=16, lib_path=lib, am_enable=True)
        E.set_mode(0, eng.MODE_AM)
        common.run_engine_streaming(E, 0, cap.iq, chunk=32768)
    else:
        cap = synth.fm_mp1_capture(0, n_blocks=40, seed=77, cfo_hz=30.0, offset=777, snr_db=25, station_ids=True)
        E = eng.Engine(max_streams=1, q15_capacity=400000, lib_path=lib)
        common.run_engine_streaming(E, 0, cap.iq)
    recs = E.drain(0)
    n = good = 0
    for r in recs:
        if int(r["flags"]) & eng.REC_PIDS:
            ok = oracle.pids_crc_ok(eng.unpack_bits(r["pids"], eng.PIDS_BITS))
            assert bool(int(r["flags"]) & eng.REC_PIDS_CRC) == ok
            n += 1; good += ok
    assert n >= 20 and good >= 16 and (am or good < n)
    E.close()


def check_l2_index_stage(lib, oracle, lengths=None):
    """k_l2_index == orc_l2_index (itself pinned against the reference's frame_push / frame_process in
    tests/test_oracle_l2.py) on frames that walk every branch: index fields and RS-corrected PDU bytes, bit-exact."""
    from nrsc5_amd import synth_l2
    from oracle import port
    E = eng.Engine(max_streams=1, lib_path=lib)
    seen = set()
    for nbits in (lengths or sorted(synth_l2.LAYOUT)):
        cases = synth_l2.test_frames(nbits)
        frames = np.stack([b for _, b, _ in cases])
        got = E.stage_l2_index(frames)
        for (name, bits, _), (gi, gb) in zip(cases, got):
            oi, ob = oracle.l2_index(bits)
            assert gi == oi, (nbits, name, {k: (gi[k], oi[k]) for k in gi if gi[k] != oi[k]})
            assert np.array_equal(gb, ob), (nbits, name)
            seen.add(port.L2_STATUS[oi["status"]])
    if lengths is None:
        assert seen == {"end", "no_audio", "header_rs", "bad_locators", "too_many_pdus", "hef_overrun", "bad_stream"}, seen
        rng = np.random.default_rng(7)                         # randomised structures, batched per frame length
        by_len = {}
        for _ in range(120):
            nbits, bits = synth_l2.random_frame(rng)
            by_len.setdefault(nbits, []).append(bits)
        for nbits, frames in by_len.items():
            for bits, (gi, gb) in zip(frames, E.stage_l2_index(np.stack(frames))):
                oi, ob = oracle.l2_index(bits)
                assert gi == oi and np.array_equal(gb, ob), (nbits, {k: (gi[k], oi[k]) for k in gi if gi[k] != oi[k]})
    # unknown frame length: reported, not guessed
    gi, _ = E.stage_l2_index(np.zeros((1, 1000), dtype=np.uint8))[0]
    assert eng.L2_STATUS[gi["status"]] == "bad_length" and gi["n_pdu"] == 0
    E.close()


def _l2_job_bits(E, job):
    stream, slot, kind, which, nbits = job
    if kind == eng.L2_FM_P1:
        return E.p1_frame_bits(stream, slot)
    if kind == eng.L2_FM_PX:
        return E.px_frame_bits(stream, slot, which, nbits)
    return E.am_frame_bits(stream, slot, which, nbits)


def check_l2_index_end_to_end(lib, oracle, am=False, mode="MP3", p1_async=False):
    """IQ -> decoded frames -> nrsc5hip_l2_index on the frames still in HBM == the oracle's index of the same frames;
    for the FM signal source every audio packet the transmitter built comes back with a good CRC-8.  The engine runs with the
    l2_index option, so the same frames were also indexed inside the pipeline (P1, P3 / P4 and AM rings): those entries must be
    the same structs."""
    from nrsc5_amd import synth_am
    if am:
        cap = synth_am.am_ma1_capture(9, seed=31, cfo_hz=2.0, offset=300)
        E = eng.Engine(max_streams=1, q15_capacity=400000, record_capacity=512, p1_slots=16, lib_path=lib, am_enable=True, p1_async=p1_async, l2_index=True)
        E.set_mode(0, eng.MODE_AM)
    else:
        cap = synth.fm_mp1_capture(3, seed=52, cfo_hz=-40.0, offset=500, snr_db=25, mode=mode)
        E = eng.Engine(max_streams=1, q15_capacity=cap.iq.size // 4 + 200000, record_capacity=512, p1_slots=8, lib_path=lib, p1_async=p1_async, l2_index=True)
    common.run_engine_streaming(E, 0, cap.iq, chunk=32768 * 8)
    recs = E.drain(0)
    jobs = eng.l2_jobs_from_records(0, recs, eng.MODE_AM if am else eng.MODE_FM)
    assert len(jobs) >= (9 if am else 2), len(jobs)
    got = E.l2_index(jobs)
    kinds = set()
    for job, (gi, gb) in zip(jobs, got):
        oi, ob = oracle.l2_index(_l2_job_bits(E, job))
        assert gi == oi, (job, {k: (gi[k], oi[k]) for k in gi if gi[k] != oi[k]})
        assert np.array_equal(gb, ob), job
        kinds.add((job[2], job[4]))
        if not am and job[2] == eng.L2_FM_P1:
            assert gi["n_pdu"] == 1 and gi["pdus"][0]["nop"] == 32 and gi["pdus"][0]["crc_bad_lo"] == 0 and gi["lost_sync"] == 0
    assert len(kinds) >= 2, kinds
    # the pipeline's own index of the same frames (engine option l2_index)
    ring_p1, ring_px = E.batch_fetch_l2(1), E.batch_fetch_l2_px(1)
    ring_am = E.batch_fetch_l2_am(1) if am else None
    fused = set()
    for (stream, slot, kind, which, nbits), (gi, _) in zip(jobs, got):
        fr = ring_p1[0][slot] if kind == eng.L2_FM_P1 else ring_px[0][slot][which] if kind == eng.L2_FM_PX else ring_am[0][slot][which]
        assert eng.l2_frame_to_dict(fr) == gi, ("fused", kind, slot, which)
        fused.add(kind)
    assert fused == ({eng.L2_AM} if am else {eng.L2_FM_P1, eng.L2_FM_PX}), fused
    # argument errors are reported, not executed
    for bad in ((0, 99, eng.L2_FM_P1, 0, 146176), (0, 0, 7, 0, 146176), (0, 0, eng.L2_FM_PX, 2, 4608), (0, 0, eng.L2_AM, 8, 3750)):
        try:
            E.l2_index([bad])
            raise AssertionError(f"job {bad} accepted")
        except eng.Nrsc5HipError:
            pass
    E.close()


def check_deferred_feedback_recovers(lib, n_blocks=256):
    """Window pipeline + on-device L2 feedback on captures the reference algorithm falsely locks on (seeds 23 / 24) and one
    it does not.  The verdict of a deferred decode may arrive up to NWIN = 8 windows late; 256 blocks leave room for the
    worst case (128 blocks of latency + re-acquisition + one aligned L1 frame)."""
    caps = [synth.fm_mp1_capture(0, seed=sd, cfo_hz=c, offset=o, snr_db=20, n_blocks=n_blocks) for sd, c, o in ((23, 0.0, 1234), (24, 10.0, 2208), (25, 10.0, 777))]
    n = len(caps)
    stride = max(c.iq.size for c in caps); stride += (-stride) % 256
    buf = np.zeros((n, stride), dtype=np.uint8)
    for k, c in enumerate(caps):
        buf[k, :c.iq.size] = c.iq
    good = []
    for fb in (False, True):
        E = eng.Engine(max_streams=n, q15_capacity=stride // 4 + 1024, record_capacity=512, p1_slots=32, p1_async=True, l2_feedback=fb, lib_path=lib)
        dev = _to_device(E, buf)
        E.batch_append_cu8(dev, stride, [c.iq.size - c.iq.size % 4 for c in caps])
        E.batch_process(n)
        recs, counts, frames = E.batch_fetch(n)
        ok = []
        for k, c in enumerate(caps):
            truth = {np.packbits(f, bitorder="little").tobytes() for f in c.p1_frames}
            ok.append(sum(1 for r in recs[k, :counts[k]] if (int(r["flags"]) & eng.REC_P1) and frames[k, int(r["p1_slot"])].tobytes() in truth))
        good.append(ok)
        _free_device(E, dev)
        E.close()
    total = n_blocks // 16
    assert good[0][0] == 0 and good[0][1] == 0 and good[0][2] >= total - 2, good     # without feedback the two false locks never recover
    assert good[1][0] >= 1 and good[1][1] >= 1 and good[1][2] == good[0][2], good    # late (deferred decode), but they come back
    return good


FALSE_LOCK_CASES = ((23, 0.0, 1234), (24, 10.0, 2208), (25, 10.0, 777))     # seeds 23 / 24: the reference algorithm locks falsely first


def drift_replay_captures(n_blocks=96):
    """Streams that lose sync in the MIDDLE of a tracked stretch while the sample clock drifts (an interference burst over 10-12
    blocks breaks one P1 frame's first header): the replay checkpoint then holds a FIFO read position, a timing pick and 534
    loop phases that have been walking for 30-50 blocks (acquire.c:110-119,259; sync.c:455,769-777)."""
    from nrsc5_amd import channel
    return [synth.fm_mp1_capture(0, seed=sd, cfo_hz=c, offset=o, snr_db=20, n_blocks=n_blocks, chan=channel.Impairments(ppm=ppm), burst=burst)
            for sd, c, o, ppm, burst in ((51, 30.0, 1234, 55.0, (18.3, 12, 10.0)), (52, -120.0, 2208, -70.0, (34.0, 11, 8.0)), (54, 5.0, 700, 25.0, (33.5, 10.0, 6.0)))]


def check_deferred_feedback_equals_reference(lib, oracle, n_blocks=96, verdict_lag=0, extra=(), cases=FALSE_LOCK_CASES, caps=None):
    """THE benchmarked mode (batch, window pipeline, on-device L2 feedback) against the oracle driven by the restated
    frame_process decision (itself pinned against the unmodified reference incl. its L2): the complete ordered log --
    LOST_SYNC on the reference's block, re-acquisition, every PIDS / P1 frame, SYNC / MER / BER -- must be equal, no matter
    how late the verdict of the deferred decode arrives (verdict_lag forces it `lag` windows late: deep speculation)."""
    if caps is None:
        caps = [synth.fm_mp1_capture(0, seed=sd, cfo_hz=c, offset=o, snr_db=20, n_blocks=n_blocks) for sd, c, o in tuple(cases) + tuple(extra)]
    n = len(caps)
    stride = max(c.iq.size for c in caps); stride += (-stride) % 256
    buf = np.zeros((n, stride), dtype=np.uint8)
    for k, c in enumerate(caps):
        buf[k, :c.iq.size] = c.iq
    E = eng.Engine(max_streams=n, q15_capacity=stride // 4 + 1024, record_capacity=1024, p1_slots=48, p1_async=True, l2_feedback=True, lib_path=lib)
    E.tune(eng.TUNE_VERDICT_LAG, verdict_lag)
    dev = _to_device(E, buf)
    E.batch_append_cu8(dev, stride, [c.iq.size - c.iq.size % 4 for c in caps])
    E.batch_process(n)
    recs, counts, frames = E.batch_fetch_view(n)
    lost = 0
    for k, c in enumerate(caps):
        ol, _, _ = oracle.run(c.iq, p1_hook=oracle.l2_hook())
        r = recs[k, :counts[k]]
        assert not (r["flags"] & eng.REC_DISCARDED).any()
        log = eng.records_to_log(E, k, r, frames[k])
        diffs = common.compare_logs(common.strip_states(ol), common.strip_states(log))
        # frames decoded while falsely locked are Viterbi output on noise (see check_oracle_end_to_end): their bits / BER are
        # compared loosely, everything else -- including that the frame exists and fails its header check -- exactly
        kept = [x for x in common.strip_states(ol) if x[0] not in ("hdc", "soft", "vit", "amsym", "pxsoft")]
        bad = {i for i, (kk, v) in enumerate(kept) if kk == "ber" and v["cber"] > 0.02}
        diffs = [d for d in diffs if not any(d.startswith(f"#{i} ber") or d.startswith(f"#{i + 1} frame") for i in bad)]
        assert not diffs, (k, diffs[:10])
        lost += sum(1 for kk, _ in ol if kk == "lost_sync")
        # every decodable frame equals the transmitted bits
        truth = {np.packbits(f, bitorder="little").tobytes() for f in c.p1_frames}
        good = sum(1 for x in r if (int(x["flags"]) & eng.REC_P1) and frames[k, int(x["p1_slot"])].tobytes() in truth)
        assert good >= n_blocks // 16 - 2, (k, good)
    assert lost >= 2, "captures do not exercise the feedback"
    _free_device(E, dev)
    E.close()


def check_am_deferred_feedback_equals_reference(lib, oracle, verdict_lag=0, tunes=(), expect_k9_repairs=False, kws=None, min_lost=2):
    """AM twin of check_deferred_feedback_equals_reference: batch, 8-step decode windows, on-device L2 feedback.  Noise bursts
    break the first header of some P1 PDUs; the verdict of the deferred decode rewinds the stream to the block that delivered
    that PDU (k_rollback_am), so LOST_SYNC, the re-acquisition and everything after land on the reference's blocks."""
    from nrsc5_amd import synth_am
    if kws is None:
        kws = [dict(n_frames=16, seed=9, cfo_hz=2.0, offset=500, burst=(8.3, 0.5, 40.0)),
               dict(n_frames=16, seed=10, cfo_hz=-3.0, offset=900, burst=(9.6, 0.3, 40.0)),
               dict(n_frames=12, seed=11, cfo_hz=1.0, offset=100)]
    caps = [synth_am.am_ma1_capture(**kw) for kw in kws]
    n = len(caps)
    stride = max(c.iq.size for c in caps); stride += (-stride) % 64
    buf = np.zeros((n, stride), dtype=caps[0].iq.dtype)
    for k, c in enumerate(caps):
        buf[k, :c.iq.size] = c.iq
    E = eng.Engine(max_streams=n, q15_capacity=stride // 2 + 1024, record_capacity=1024, p1_slots=48, lib_path=lib, am_enable=True, p1_async=True, l2_feedback=True)
    E.tune(eng.TUNE_VERDICT_LAG, verdict_lag)
    for knob, value in tunes:
        E.tune(knob, value)
    for k in range(n):
        E.set_mode(k, eng.MODE_AM)
    dev = _to_device(E, buf)
    E.batch_append_cs16(dev, stride, [c.iq.size - c.iq.size % 4 for c in caps])
    E.batch_process(n)
    recs, counts, frames = E.batch_fetch_view(n)
    lost = 0
    for k, c in enumerate(caps):
        ol, _, _ = oracle.run(c.iq, mode=1, p1_hook=oracle.l2_hook())
        r = recs[k, :counts[k]]
        assert not (r["flags"] & eng.REC_DISCARDED).any()
        log = eng.am_records_to_log(E, k, r, frames[k])
        diffs = common.compare_logs(common.strip_states(ol), common.strip_states(log))
        kept = [x for x in common.strip_states(ol) if x[0] not in ("hdc", "soft", "vit", "amsym", "pxsoft")]
        bad = {i for i, (kk, v) in enumerate(kept) if kk == "ber" and v["cber"] > 0.02}
        diffs = [d for d in diffs if not any(d.startswith(f"#{i} ") or d.startswith(f"#{i + 1} frame") or d.startswith(f"#{i - 1} frame") for i in bad)]
        assert not diffs, (k, verdict_lag, diffs[:10])
        lost += sum(1 for kk, _ in ol if kk == "lost_sync")
    assert lost >= min_lost, "captures do not exercise the feedback"
    st = E.k9_stats()
    assert st[0] > 0 and st[2] > 0, st                         # the P3 frames went through segment waves
    if expect_k9_repairs:
        assert st[1] > 0 and st[3] > 0, st                     # ... and cold segment starts were re-run, in the production kernels
    _free_device(E, dev)
    E.close()


def check_mixed_batch_pipeline(lib, oracle, passes=2):
    """ONE engine, FM and AM streams with interleaved ids in one batch_process call (window pipeline + L2 feedback on the device, both
    replays active): every stream's log must equal the oracle's as if it had been alone; further passes over the same engine
    (reset_all) must give the same again."""
    from nrsc5_amd import synth_am
    fm_caps = [synth.fm_mp1_capture(0, seed=sd, cfo_hz=c, offset=o, snr_db=20, n_blocks=64) for sd, c, o in FALSE_LOCK_CASES[:1] + ((31, 80.0, 700),)]
    am_caps = [synth_am.am_ma1_capture(n_frames=12, seed=9, cfo_hz=2.0, offset=500, burst=(8.3, 0.5, 40.0)),
               synth_am.am_ma1_capture(n_frames=10, seed=11, cfo_hz=1.0, offset=100)]
    nf, na = len(fm_caps), len(am_caps)
    sf = max(c.iq.size for c in fm_caps); sf += (-sf) % 256
    sa = max(c.iq.size for c in am_caps); sa += (-sa) % 64
    bf = np.zeros((nf, sf), dtype=np.uint8); ba = np.zeros((na, sa), dtype=am_caps[0].iq.dtype)
    for k, c in enumerate(fm_caps): bf[k, :c.iq.size] = c.iq
    for k, c in enumerate(am_caps): ba[k, :c.iq.size] = c.iq
    E = eng.Engine(max_streams=nf + na, q15_capacity=max(sf // 4, sa // 2) + 1024, record_capacity=1024, p1_slots=48, lib_path=lib, am_enable=True, p1_async=True, l2_feedback=True)
    ids_f, ids_a = [0, 2][:nf], [1, 3][:na]                    # interleaved ids: the split by mode is the engine's job
    for s_ in ids_a:
        E.set_mode(s_, eng.MODE_AM)
    df, da = _to_device(E, bf), _to_device(E, ba)
    want = {}
    for k, c in enumerate(fm_caps):
        want[ids_f[k]] = (common.strip_states(oracle.run(c.iq, p1_hook=oracle.l2_hook())[0]), False)
    for k, c in enumerate(am_caps):
        want[ids_a[k]] = (common.strip_states(oracle.run(c.iq, mode=1, p1_hook=oracle.l2_hook())[0]), True)
    for _ in range(passes):
        E.reset_all()
        E.batch_append_cu8(df, sf, [c.iq.size - c.iq.size % 4 for c in fm_caps], stream_ids=ids_f)
        E.batch_append_cs16(da, sa, [c.iq.size - c.iq.size % 4 for c in am_caps], stream_ids=ids_a)
        E.batch_process(nf + na)
        recs, counts, frames = E.batch_fetch_view(nf + na)
        for s_, (ol, am) in want.items():
            r = recs[s_, :counts[s_]]
            assert not (r["flags"] & eng.REC_DISCARDED).any()
            log = (eng.am_records_to_log if am else eng.records_to_log)(E, s_, r, frames[s_])
            diffs = common.compare_logs(ol, common.strip_states(log))
            kept = [x for x in ol if x[0] not in ("hdc", "soft", "vit", "amsym", "pxsoft")]
            bad = {i for i, (kk, v) in enumerate(kept) if kk == "ber" and v["cber"] > 0.02}
            diffs = [d for d in diffs if not any(d.startswith(f"#{i} ") or d.startswith(f"#{i + 1} frame") or d.startswith(f"#{i - 1} frame") for i in bad)]
            assert not diffs, (s_, am, diffs[:8])
    _free_device(E, df); _free_device(E, da)
    E.close()


def check_hdc_consumer(lib, reflib, caps, p1_async=False):
    """IQ -> engine -> L2 index (device) -> nrsc5hip_hdc_* == the NRSC5_EVENT_HDC sequence of the unmodified reference."""
    from oracle import ref
    n = len(caps)
    E = eng.Engine(max_streams=n, q15_capacity=max(c.iq.size for c in caps) // 4 + 200000, record_capacity=1024, p1_slots=16, lib_path=lib,
                   p1_async=p1_async, l2_feedback=True)
    H = eng.HdcConsumer(n, lib=E.lib)
    total = 0
    for k, cap in enumerate(caps):
        common.run_engine_streaming(E, k, cap.iq, chunk=32768 * 8)
        recs = E.drain(k)
        eng.feed_hdc(E, H, k, recs)
        got = [(p, c, f, d) for (s, p, c, f, d) in H.events if s == k]
        log, _, _ = reflib.run(cap.iq, taps=ref.TAP_HDC)
        exp = [(v["program"], v["count"], v["flags"], bytes(v["data"])) for kk, v in log if kk == "hdc"]
        assert len(exp) >= 32, "capture yields too few HDC packets to mean anything"
        assert [(p, c, f) for p, c, f, _ in got] == [(p, c, f) for p, c, f, _ in exp], (k, len(got), len(exp))
        assert all(a[3] == b[3] for a, b in zip(got, exp)), k
        total += len(exp)
    assert H.host_bytes() < n * 256 * 1024
    H.close()
    E.close()
    return total


def check_l2_index_fused(lib, oracle, p1_async=False):
    """Engine option l2_index: the index written on the decode stream behind each P1 traceback == the post-pass index
    == the oracle's, for every P1 slot the records name (streaming getter and bulk fetch)."""
    caps = [synth.fm_mp1_capture(2, seed=61 + k, cfo_hz=c, offset=o, snr_db=22) for k, (c, o) in enumerate([(25.0, 400), (-310.0, 3100)])]
    n = len(caps)
    E = eng.Engine(max_streams=n, q15_capacity=max(c.iq.size for c in caps) // 4 + 200000, record_capacity=256, p1_slots=4,
                   lib_path=lib, p1_async=p1_async, l2_index=True, l2_feedback=True)
    for k, c in enumerate(caps):
        common.run_engine_streaming(E, k, c.iq, chunk=32768 * 8)
    bulk = E.batch_fetch_l2(n)
    seen = 0
    for k in range(n):
        recs = E.drain(k)
        jobs = eng.l2_jobs_from_records(k, recs)
        assert jobs, "no P1 frame decoded"
        post = E.l2_index(jobs, want_bytes=False)
        for job, (pi, _) in zip(jobs, post):
            fused = E.l2_frame(k, job[1])
            oi, _ = oracle.l2_index(E.p1_frame_bits(k, job[1]))
            assert fused == pi == oi, (job, {key: (fused[key], oi[key]) for key in oi if fused[key] != oi[key]})
            assert eng.l2_frame_to_dict(bulk[k][job[1]]) == oi
            assert fused["n_pdu"] == 1 and fused["pdus"][0]["nop"] == 32 and fused["pdus"][0]["crc_bad_lo"] == 0
            seen += 1
    assert seen >= 2 * n - 1
    E.close()
    # without the option the getters refuse, loudly
    E = eng.Engine(max_streams=1, lib_path=lib)
    for call in (lambda: E.l2_frame(0, 0), lambda: E.batch_fetch_l2(1)):
        try:
            call()
            raise AssertionError("l2 getter worked without l2_index")
        except eng.Nrsc5HipError:
            pass
    E.close()


def check_l2_index_vs_reference_golden(lib):
    """Device index -> the output_align / output_push calls it implies == the calls the UNMODIFIED reference made for the
    same frames (tests/golden/l2_reference_taps.json, recorded by tests/golden/make_golden_l2.py).  No oracle involved."""
    import hashlib
    import json
    from nrsc5_amd import synth_l2
    gold = json.load(open(os.path.join(GOLDEN_DIR, "l2_reference_taps.json")))
    E = eng.Engine(max_streams=1, lib_path=lib)
    checked = 0
    for nbits in sorted(synth_l2.LAYOUT):
        frames = {name: bits for name, bits, _ in synth_l2.test_frames(nbits)}
        cases = gold[str(nbits)]
        stack = np.stack([frames[c["name"]] for c in cases])
        for c, (gi, gb) in zip(cases, E.stage_l2_index(stack)):
            assert hashlib.sha1(frames[c["name"]].tobytes()).hexdigest() == c["bits_sha1"], "test-frame generator drifted: regenerate the golden"
            got = json.loads(json.dumps(common.l2_taps_digest(common.l2_expected_taps(gi, gb))))
            assert got == c["taps"], (nbits, c["name"])
            checked += len(got)
    assert checked > 1500
    E.close()


def check_frame_push_indexed_with_device_index(lib, reflib):
    """The device's index (C-ABI structs exactly as nrsc5hip_stage_l2_index returns them) fed to frame_push_indexed inside
    the unmodified reference (oracle/ref_shim/frame_indexed.c) makes it do what its own frame_push does with the bits:
    output_align / output_push / AAS packets / audio-service reports / HDC events / sync loss, over whole sessions."""
    from nrsc5_amd import synth_l2
    from tests.test_oracle_l2 import _all_l2_taps
    E = eng.Engine(max_streams=1, lib_path=lib)
    sessions = [synth_l2.psd_sequence(seed=1), synth_l2.psd_sequence(seed=2, nbits=24000, n_frames=4),
                [b for _, b, safe in synth_l2.test_frames(146176, seed=7) if safe], synth_l2.fixed_data_session(seed=2)]
    skip = {eng.L2_STATUS.index(s) for s in ("hef_overrun", "bad_stream", "too_many_pdus")}
    n_pkt = n_aas = 0
    for frames in sessions:
        structs, by = E.stage_l2_index_raw(np.stack(frames))
        keep = [k for k in range(len(frames)) if structs[k].status not in skip]
        direct = reflib.l2_frames([frames[k] for k in keep])
        indexed = reflib.l2_frames_indexed([(structs[k], by[k, :structs[k].nbytes]) for k in keep])
        for a, b in zip(direct, indexed):
            ta, tb = _all_l2_taps(a), _all_l2_taps(b)
            assert ta == tb
            n_pkt += sum(1 for t in ta if t[0] == "l2pkt"); n_aas += sum(1 for t in ta if t[0] == "l2aas")
    assert n_pkt >= 400 and n_aas >= 8, (n_pkt, n_aas)
    E.close()


def check_block_exact_pushes(lib, oracle, am=False):
    """The drop-in's feeding rule (integration/input_hip.c): every nrsc5_pipe_samples call (32768 bytes) is handed to the engine in
    pieces that end where nrsc5hip_bytes_to_next_block says the next block completes, so at most ONE block is processed per push
    -- also right after a CFO search, when the reference keeps up to 31 symbols (acquire_keep_extra) and the next window needs only
    two symbols of new samples -- and the L2 feedback of a block reaches the engine before the next one.  Engine: in-order P1,
    feedback applied by the engine's own in-order L2 check (== the reference's); log == oracle with the L2 hook."""
    if am:
        from nrsc5_amd import synth_am
        caps = [synth_am.am_ma1_capture(n_frames=9, seed=31, cfo_hz=-40.0, offset=1700, fmt="cu8"), synth_am.am_ma1_capture(n_frames=9, seed=32, cfo_hz=5.0, offset=300)]
    else:
        caps = [synth.fm_mp1_capture(**common.GOLDEN_CASES["fm_cu8_cfo-2400"]), synth.fm_mp1_capture(0, seed=23, cfo_hz=0.0, offset=1234, snr_db=20, n_blocks=50),
                synth.fm_mp1_capture(**common.GOLDEN_CASES["fm_cs16_cfo60"])]
    for cap in caps:
        cu8 = cap.iq.dtype == np.uint8
        E = eng.Engine(max_streams=1, q15_capacity=200000, record_capacity=256, p1_slots=8, lib_path=lib, am_enable=am, l2_feedback=True)
        if am:
            E.set_mode(0, eng.MODE_AM)
        item = 1 if cu8 else 2
        recs, pieces, most = [], 0, 0
        for off in range(0, cap.iq.size * item, 32768):
            call = cap.iq.view(np.uint8)[off:off + 32768]
            call = call[:call.size - call.size % 4]
            done = 0
            while done < call.size:
                room = E.bytes_to_next_block(0, cu8)
                assert room >= 4 and room % 4 == 0
                piece = call[done:done + room]
                if cu8:
                    E.push_cu8(0, piece)
                else:
                    E.push_cs16(0, piece.view(np.int16))
                new = E.drain(0)
                most = max(most, len(new)); recs.append(new); pieces += 1
                done += piece.size
        assert most == 1, most                                   # never two blocks in one push
        recs = np.concatenate(recs)
        assert pieces < 2 * (cap.iq.size * item // 32768 + 1) + len(recs) + 2     # ~one piece per call + one per block
        log = (eng.am_records_to_log if am else eng.records_to_log)(E, 0, recs)
        ol, _, _ = oracle.run(cap.iq, mode=1 if am else 0, p1_hook=oracle.l2_hook())
        diffs = common.compare_logs(common.strip_states(ol), common.strip_states(log))
        kept = [x for x in common.strip_states(ol) if x[0] not in ("hdc", "soft", "vit", "amsym", "pxsoft")]
        bad = {i for i, (kk, v) in enumerate(kept) if kk == "ber" and v["cber"] > 0.02}
        diffs = [d for d in diffs if not any(d.startswith(f"#{i} ber") or d.startswith(f"#{i + 1} frame") for i in bad)]
        assert not diffs, diffs[:8]
        E.close()


def check_viterbi_segmented(lib, oracle, lens=(2304, 4608), segments=(1, 2, 4, 16), seed=12):
    """Segmented forward pass (viterbi_v3.h): any number of segment waves gives the sequential decoder's bits -- on noise, on
    decodable frames, on the all-erasure frame (every ACS a tie) and on saturated input; and again with the warm-up of the
    segments switched off (test hook), when every speculative start on informative input is wrong and the segments must be
    REPAIRED (counted by the engine): the repair path, not luck, is what makes the result exact."""
    rng = np.random.default_rng(seed)
    E = eng.Engine(max_streams=1, q15_capacity=2 * 71280, lib_path=lib)
    for L in lens:
        soft = rng.integers(-127, 128, size=(5, 3 * L), dtype=np.int8)
        soft[0] = 0
        soft[1, :] = 127
        msg = rng.integers(0, 2, size=(1, L), dtype=np.uint8)
        coded = synth.conv_encode_k7(msg).reshape(3 * L).astype(np.int16) * 2 - 1
        soft[2] = np.clip(np.rint(coded * 40 + rng.normal(0, 14, size=coded.shape)), -127, 127).astype(np.int8)
        soft[:, 5::6] = 0
        exp = np.stack([oracle.viterbi_k7(s) for s in soft])
        for warm in (1, 0):
            E.tune(eng.TUNE_FWD_WARM, warm)
            for G in segments:
                E.tune(eng.TUNE_FWD_SEGMENTS, G)
                c0, r0 = E.fwd_stats()
                got = E.stage_viterbi_k7(soft, L)
                c1, r1 = E.fwd_stats()
                assert np.array_equal(got, exp), f"segmented Viterbi mismatch at len {L}, {G} segments, warm {warm}"
                if G > 1 and L >= 4608:
                    assert c1 > c0
                    if not warm:
                        assert r1 - r0 >= 3, (L, G, c1 - c0, r1 - r0)     # cold starts on informative frames were repaired, not trusted
    E.close()


def check_deferred_seam(lib, names=("ppm+60", "ppm-85_cs16", "ppm+100_cfo_search")):
    """Round 4's streaming seam.  (1) deferred wait: a block that starts FINE consumes 71280 - 2160 + next_samperr samples, so the
    host moves its mirror when it SUBMITS the step and takes the report later -- under a sample-clock error next_samperr != 0 on
    most blocks, so the prediction is really exercised (and counted: never wrong); (2) no P1 decode launches on blocks that cannot
    complete a frame (never a frame without its decode); (3) the drop-in's flow -- manual step: push the completing piece, drain
    the block before (waits), step, poll with drain_ready on every other call; (4) the same without polling, so that every step
    finds its predecessor still in flight and is queued AHEAD of that block's delivery wherever that is safe
    (nrsc5hip_stream_step_ahead).  All must leave the records bit-identical to
    the synchronous seam (NRSC5HIP_TUNE_DEFER_WAIT = 0, drain after every push)."""
    for name in names:
        cap = synth.fm_mp1_capture(**common.IMPAIRED_FM_CASES[name])
        cu8 = cap.iq.dtype == np.uint8
        raw = cap.iq.view(np.uint8)
        raw = raw[:raw.size - raw.size % 4]

        def push(E, piece):
            if cu8:
                E.push_cu8(0, piece)
            else:
                E.push_cs16(0, piece.view(np.int16))

        def run(mode):
            E = eng.Engine(max_streams=1, q15_capacity=200000, record_capacity=256, p1_slots=8, lib_path=lib)
            E.seam_counts(reset=True)
            if mode == "sync":                                   # round 3's seam: wait at once, H2D copy + decimator + commit, k_prepare as its own launch
                E.tune(eng.TUNE_DEFER_WAIT, 0); E.tune(eng.TUNE_DIRECT_DECIMATE, 0); E.tune(eng.TUNE_SEAM_PREPARE, 0); E.tune(eng.TUNE_HOST_CAPTURE, 0); E.tune(eng.TUNE_FOLD_REPORT, 0)
            if mode == "fifo":                                   # rounds 4 - 5: deferred wait over pinned staging + the direct decimator
                E.tune(eng.TUNE_HOST_CAPTURE, 0)
            if mode in ("dropin", "ahead"):
                E.set_manual_step(0, True)
            recs, frames = [], []

            def take(new):
                for r in new:
                    if int(r["flags"]) & eng.REC_P1:
                        frames.append(E.p1_frame_bits(0, int(r["p1_slot"])).copy())
                recs.append(new)
            for off in range(0, raw.size, 32768):
                call = raw[off:off + 32768]
                done = 0
                while done < call.size:
                    room = E.bytes_to_next_block(0, cu8)
                    assert room >= 4 and room % 4 == 0
                    piece = call[done:done + room]
                    push(E, piece)
                    done += piece.size
                    if mode in ("dropin", "ahead"):
                        if piece.size >= room:
                            ahead = E.stream_step_ahead(0)
                            take(E.drain(0))
                            if not ahead:
                                E.stream_step(0)
                            if mode == "dropin":
                                take(E.drain_ready(0))
                        elif mode == "dropin":
                            take(E.drain_ready(0))               # (mode "ahead": no polling -- on the emulator a poll always finds the step done,
                                                                 # and the path that queues a step behind one in flight would never run)
                    else:
                        take(E.drain(0))
            take(E.drain(0))
            counts = E.seam_counts()
            hcs = E.host_capture_stats()
            # round 6: a cu8 session of the default seam reads the pinned capture in place (every push one host copy); cs16 input never does
            assert (hcs["attaches"] == 1 and counts["host_capture_pushes"] > 0) == (cu8 and mode not in ("sync", "fifo")), (mode, hcs, counts)
            # ... and the sync kernel posts the report of every step that has nothing behind it (MP1: all but the blocks that may end a P1 frame and the un-synchronised ones)
            assert (hcs["reports_folded"] == 0) if mode == "sync" else (hcs["reports_folded"] >= counts["steps_without_p1_launches"] - 2 > 0), (mode, hcs, counts)
            E.close()
            return np.concatenate(recs), frames, counts
        ref, ref_frames, c0 = run("sync")
        assert c0["deferred_steps"] == 0 and len(ref) >= 30 and len(ref_frames) >= 1
        fine = sum(1 for r in ref[:-1] if int(r["state_after"]) == 2)
        assert sum(1 for r in ref if int(r["state_before"]) == 2 and int(r["samperr"]) != 1080) >= 5, "the capture does not move the timing pick"
        for mode in ("deferred", "fifo", "dropin", "ahead"):
            got, frames, c = run(mode)
            assert got.tobytes() == ref.tobytes(), (name, mode)
            assert len(frames) == len(ref_frames) and all(np.array_equal(a, b) for a, b in zip(frames, ref_frames))
            assert c["mispredicted_rd"] == 0 and c["late_p1_decodes"] == 0, c
            assert c["deferred_steps"] >= fine - 1 and c["steps_without_p1_launches"] >= fine - 1 - len(ref_frames), (c, fine)
            if mode == "ahead":                                  # every block behind a FINE block without a P1 decode was queued ahead
                assert c["steps_ahead"] >= fine - 2 - 2 * len(ref_frames), (c, fine)


def check_traceback_variants(lib):
    """The single-path traceback (every chunk walked once after a one-chunk run-in, the chain of chunk boundaries verified, wrong chunks
    re-walked; re-encode disagreements counted per chunk by the walk) against round 3's block-parallel one (all 64 candidates per chunk,
    one workgroup's loop for the BER): identical records -- the BER count to the last disagreement -- and frames, on a capture whose
    first lock is false: that frame is Viterbi output on noise (BER 0.136), where survivors merge slowly and hundreds of chunks must
    be re-walked."""
    cap = synth.fm_mp1_capture(0, seed=62, cfo_hz=-50.0, offset=700, snr_db=22, n_blocks=36)
    res = {}
    for walk in (0, 1):
        E = eng.Engine(max_streams=1, q15_capacity=1 << 20, record_capacity=256, p1_slots=8, lib_path=lib)
        E.tune(eng.TUNE_TRACEBACK_WALK, walk)
        common.run_engine_streaming(E, 0, cap.iq, chunk=32768 * 8)
        r = E.drain(0)
        frames = [E.p1_frame_bits(0, int(x["p1_slot"])).copy() for x in r if int(x["flags"]) & eng.REC_P1]
        res[walk] = (r.tobytes(), frames, E.tb_stats())
        E.close()
    assert res[0][0] == res[1][0] and len(res[0][1]) == len(res[1][1]) == 2
    assert all(np.array_equal(a, b) for a, b in zip(res[0][1], res[1][1]))
    assert res[0][2] == (0, 0) and res[1][2][0] == 2 * 2284 and 20 <= res[1][2][1] <= 2000, res[1][2]


def check_exact_oscillator_first_block(lib, reflib, bit_exact_min, policies=(0, 1, 2, 3), n=6):
    """Exact-oscillator mode (NRSC5HIP_TUNE_NCO_EXACT, k_nco_exact; DESIGN.md (c) limit 2).  The first block after a reset -- the block the
    CFO search runs on -- starts from acquire_t.phase = 1 and a coarse angle whose inputs the acquisition kernels reproduce bit for bit, so with
    policy >= 1 the NCO state the block leaves behind (69 120 float complex multiplications and 32 renormalisations later) must be the UNMODIFIED
    reference's, bit for bit, wherever the one libm call in between (atan2f of the coarse peak) returns the same float: always on the CPU emulator
    (same glibc), in >= bit_exact_min of n captures on the device.  With policy 0 (closed form) it is not.  Every policy keeps the complete log
    inside the float tolerances."""
    exact = {p: 0 for p in policies}
    worst = {p: 0.0 for p in policies}
    for k in range(n):
        cap = synth.fm_mp1_capture(0, seed=300 + k, cfo_hz=(-1, 1)[k & 1] * (190.0 + 17.0 * k), offset=211 * k + 5, snr_db=(15.0, 20.0, 25.0)[k % 3], n_blocks=20)
        ref_log = reflib.run(cap.iq)[0]
        rb = [v for kk, v in ref_log if kk == "block"]
        for pol in policies:
            E, recs, log = run_capture(lib, cap, tune=((eng.TUNE_NCO_EXACT, pol),))
            E.close()
            diffs = common.compare_logs(common.strip_states(ref_log), common.strip_states(log))
            assert not [d for d in diffs if " frame." in d or " pids." in d or " sync." in d], (k, pol, diffs[:5])
            gb = [v for kk, v in log if kk == "block"]
            a, b = rb[0], gb[0]
            same = np.float32(a["phase_re"]) == np.float32(b["phase_re"]) and np.float32(a["phase_im"]) == np.float32(b["phase_im"])
            exact[pol] += int(same)
            worst[pol] = max(worst[pol], abs(a["phase_re"] - b["phase_re"]), abs(a["phase_im"] - b["phase_im"]))
    print("first-block NCO state bit-identical to the reference (of %d captures):" % n, exact, "largest |difference|:", {p: float("%.2g" % w) for p, w in worst.items()})
    for pol in policies:
        if pol >= 1:
            assert exact[pol] >= bit_exact_min, (pol, exact, worst)
            assert worst[pol] < 1e-4, (pol, worst)
    if 0 in policies:
        assert exact[0] < n, "the closed-form phasor cannot reproduce the recurrence's rounding drift"
    return exact, worst


def run_zero_copy_batch(lib, caps, tune=(), l2_feedback=True, p1_slots=8):
    """the captures as ONE zero-copy batch with the window pipeline -> (records, counts, frames, flow stats)"""
    n = len(caps)
    stride = max(c.iq.size for c in caps); stride += (-stride) % 16
    host = np.zeros((n, stride), dtype=np.uint8)
    for k, c in enumerate(caps):
        host[k, :c.iq.size] = c.iq
    E = eng.Engine(max_streams=n, q15_capacity=2 * 71280, record_capacity=512, p1_slots=p1_slots, p1_async=True, l2_feedback=l2_feedback, batch_zero_copy=True, lib_path=lib)
    for knob, value in tune:
        E.tune(knob, value)
    dev = _to_device(E, host)
    E.batch_append_cu8(dev, stride, [c.iq.size - c.iq.size % 4 for c in caps])
    E.batch_process(n)
    recs, counts, frames = E.batch_fetch_view(n)
    out = (np.array(recs, copy=True), np.array(counts, copy=True), [[np.array(f, copy=True) for f in fr] if isinstance(fr, (list, tuple)) else np.array(fr, copy=True) for fr in frames], E.flow_stats(),
           [eng.records_to_log(E, k, recs[k, :counts[k]], frames[k]) for k in range(n)])
    _free_device(E, dev)
    E.close()
    return out


def check_flow_bursts(lib, caps, l2_feedback=True, min_flow_steps=8):
    """Dataflow bursts (k_flow: the block steps of a burst in which every stream is FINE as ONE launch of symbol-pair and block-step work items that hand over to each
    other) leave every record and every frame exactly as the two-kernel form with the same bodies does (k_mixfft<1, 2> + k_sync<256>: knobs MIXFFT_SYMS 16, SYNC_LANES 256),
    bit for bit; and as the default form (k_sync<768>: another summation grouping of the MER) does under the strict rule."""
    base = ((eng.TUNE_MIXFFT_SYMS, 16), (eng.TUNE_SYNC_LANES, 256))
    r0, c0, f0, s0, l0 = run_zero_copy_batch(lib, caps, tune=base + ((eng.TUNE_FLOW_MIN, 0),), l2_feedback=l2_feedback)
    r1, c1, f1, s1, l1 = run_zero_copy_batch(lib, caps, tune=base + ((eng.TUNE_FLOW_MIN, 1),), l2_feedback=l2_feedback)     # (the steps outside the bursts on the same bodies too)
    assert s0 == (0, 0) and s1[1] >= min_flow_steps, (s0, s1)
    assert np.array_equal(c0, c1), (c0, c1)
    for k in range(len(caps)):
        a, b = r0[k, :c0[k]], r1[k, :c1[k]]
        # every field of every record bit for bit -- but the ring slot a frame was filed in: with the replay on, slots taken by frames of discarded blocks depend on
        # where the bursts end (the rollback runs at burst ends); the frames themselves are compared through the logs below
        bad = [n for n in a.dtype.names if n != "p1_slot" and a[n].tobytes() != b[n].tobytes()]
        assert not bad, (k, bad)
        diffs = common.compare_logs(l0[k], l1[k], rtol=0.0)
        assert not diffs, (k, diffs[:10])
    # the shipped configuration: bursts as k_flow, the steps between them on k_mixfft<1, 1> + k_sync<768> -- against the same without bursts
    r2, c2, f2, s2, l2 = run_zero_copy_batch(lib, caps, tune=((eng.TUNE_FLOW_MIN, 0),), l2_feedback=l2_feedback)
    r3, c3, f3, s3, l3 = run_zero_copy_batch(lib, caps, tune=((eng.TUNE_FLOW_MIN, 1),), l2_feedback=l2_feedback)
    assert s3[1] >= min_flow_steps and np.array_equal(c2, c3)
    for k in range(len(caps)):
        diffs = common.compare_logs(l2[k], l3[k])
        assert not diffs, (k, diffs[:10])
    return s1


def check_host_capture_seam(lib, names=("ppm+60", "ppm+100_cfo_search"), capture_kib=640):
    """Round 6's seam for FM cu8: the session's bytes stay in a pinned capture that the stream reads in place (NRSC5HIP_TUNE_HOST_CAPTURE).  Against the FIFO seam
    (pinned staging + decimator kernel) on the same pushes, records and frames bit-identical, through: (1) a capture buffer barely larger than two windows, so
    that the live tail moves to the front of the buffer every other block (hc_rebase); (2) a session that turns to cs16 input in its middle (the capture becomes the
    FIFO again: hc_detach re-pushes the bytes behind the read position); (3) two captures on one stream with nrsc5hip_stream_reset between them: the second
    session's first decimator outputs see what the first left in decim[0]'s rewound window (StaleWindows), told from the capture instead of by the decimator kernel;
    (4) the batch entry point on a stream that holds a capture."""
    for name in names:
        cap = synth.fm_mp1_capture(**common.IMPAIRED_FM_CASES[name])
        assert cap.iq.dtype == np.uint8
        raw = cap.iq[:cap.iq.size - cap.iq.size % 4]

        def session(E, data, chunk, cs16_from=None):
            recs = []
            for off in range(0, data.size, chunk):
                piece = data[off:off + chunk]
                if cs16_from is not None and off >= cs16_from:
                    # (every other complex sample as Q15: not the half-band's output, but the same for both engines, and near enough to keep the receiver working)
                    E.push_cs16(0, np.ascontiguousarray(((piece.astype(np.int16) - 127) * 64).reshape(-1, 2)[::2]).reshape(-1))
                else:
                    E.push_cu8(0, piece)
                recs.append(E.drain(0))
            recs.append(E.drain(0))
            return np.concatenate(recs)

        def run(hc, what):
            E = eng.Engine(max_streams=1, q15_capacity=400000, record_capacity=512, p1_slots=8, lib_path=lib)
            E.tune(eng.TUNE_HOST_CAPTURE, capture_kib if hc else 0)
            out = []
            if what == "plain":
                out.append(session(E, raw, 32768))
            elif what == "cs16":
                out.append(session(E, raw, 65536, cs16_from=raw.size // 2 // 65536 * 65536))
            elif what == "reset":
                out.append(session(E, raw[:raw.size // 3 // 4 * 4 + 8], 50000))    # ends inside a block, 4-byte granular, with bytes that were never stepped
                E.reset(0)
                out.append(session(E, raw, 1 << 20))
            elif what == "batch":
                half = raw.size // 2 // 4 * 4
                out.append(session(E, raw[:half], 32768))
                E.batch_process(1)                                  # leaves the seam's mirror: the capture must become a FIFO of known length first
                out.append(E.drain(0))
                out.append(session(E, raw[half:], 32768))
            frames = [E.p1_frame_bits(0, int(r["p1_slot"])).copy() for r in out[-1] if int(r["flags"]) & eng.REC_P1][-1:]
            st = E.host_capture_stats()
            E.close()
            return [o.tobytes() for o in out], frames, st
        for what in ("plain", "cs16", "reset", "batch"):
            a, fa, _ = run(False, what)
            b, fb, st = run(True, what)
            assert st["attaches"] == (2 if what == "reset" else 1), (what, st)
            assert a == b, (name, what)
            assert len(fa) == len(fb) and all(np.array_equal(x, y) for x, y in zip(fa, fb)), (name, what)
            if what == "plain":
                assert st["rebases"] >= 5 and st["detaches"] == 0, st
            if what in ("cs16", "batch"):
                assert st["detaches"] == 1, st
