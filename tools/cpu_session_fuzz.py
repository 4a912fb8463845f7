"""CPU only: randomised SESSIONS -- several captures on one stream with nrsc5_set_mode / input_reset between them (FM cu8 / cs16, AM cs16 / cu8; noise of random length, or a
signal) -- on the CPU-emulated twin of the library against the UNMODIFIED reference driven the same way (oracle/ref.py: run_epochs).  After every reset the first decimated
samples must equal the reference's bit for bit (its FIR windows are rewound, not cleared: firdecim_q15.c:53-56) and the epoch's complete log must equal its log.
   python tools/cpu_session_fuzz.py [sessions=20] [seed0=70000]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def make_epoch(rng, long=False):
    from nrsc5_amd import synth, synth_am
    kind = ("fm_cu8_noise", "fm_cu8_sig", "fm_cs16_noise", "am_cs16_noise", "am_cs16_sig", "am_cu8_noise", "am_cu8_sig")[int(rng.integers(0, 7))]
    seed = int(rng.integers(1, 1 << 30))
    if kind == "fm_cu8_noise":
        return 0, np.random.default_rng(seed).integers(0, 256, size=4 * int(rng.integers(10, 120000)), dtype=np.uint8), kind
    if kind == "fm_cs16_noise":
        return 0, np.random.default_rng(seed).integers(-20000, 20000, size=2 * int(rng.integers(10, 200000)), dtype=np.int16), kind
    if kind == "am_cs16_noise":
        return 1, np.random.default_rng(seed).integers(-20000, 20000, size=2 * int(rng.integers(10, 40000)), dtype=np.int16), kind
    if kind == "am_cu8_noise":
        return 1, np.random.default_rng(seed).integers(0, 256, size=4 * int(rng.integers(10, 300000)), dtype=np.uint8), kind
    if kind == "fm_cu8_sig":
        iq = synth.fm_mp1_capture(0, seed=seed, cfo_hz=float(rng.uniform(-300, 300)), offset=int(rng.integers(0, 60)), snr_db=20, n_blocks=int(rng.integers(20, 44) if long else rng.integers(4, 20))).iq
        return 0, iq[:iq.size - iq.size % 4], kind
    fmt = "cs16" if kind == "am_cs16_sig" else "cu8"
    iq = synth_am.am_ma1_capture(int(rng.integers(6, 10) if long else rng.integers(1, 3)), seed=seed, cfo_hz=float(rng.uniform(-5, 5)), offset=int(rng.integers(120, 150)), fmt=fmt).iq
    return 1, iq[:iq.size - iq.size % 4], kind


def first_piece(mode, iq):
    """elements of the first push: fewer decimated samples than one acquisition window, so that nothing is consumed before the FIFO head is read"""
    if iq.dtype == np.uint8:
        return min(iq.size, 4 * 16000 if mode == 0 else 4 * 64000)
    return min(iq.size, 2 * (16000 if mode == 0 else 4000))


def run_session(lib, reflib, seed, verbose=False):
    from nrsc5_amd import engine as eng
    from oracle import ref
    import common
    import engine_checks as ec
    rng = np.random.default_rng(seed)
    epochs = [make_epoch(rng) for _ in range(int(rng.integers(2, 5)))]
    exp = reflib.run_epochs([(m, iq) for m, iq, _ in epochs], taps=ref.TAP_Q15)
    E = eng.Engine(max_streams=1, q15_capacity=400000, record_capacity=1024, p1_slots=16, lib_path=lib, am_enable=True, l2_feedback=True)   # frame.c sends a stream whose first L2 header fails back to acquisition: the engine decides that on the device
    problems = []
    for k, ((mode, iq, kind), (exp_log, exp_q15)) in enumerate(zip(epochs, exp)):
        E.set_mode(0, mode)                                        # input_set_mode -> input_reset (the first one on a fresh stream)
        push = E.push_cu8 if iq.dtype == np.uint8 else E.push_cs16
        n0 = first_piece(mode, iq)
        push(0, iq[:n0])
        nq = n0 // 4 if (iq.dtype == np.uint8 and mode == 0) else n0 // 64 if iq.dtype == np.uint8 else n0 // 2
        nq = min(nq, len(exp_q15))
        if nq and not np.array_equal(ec._fetch_q15(E, nq), exp_q15[:nq]):
            bad = np.nonzero((ec._fetch_q15(E, nq) != exp_q15[:nq]).any(axis=1))[0]
            problems.append((k, kind, "q15", bad[:5].tolist()))
        common.run_engine_streaming(E, 0, iq[n0:], chunk=32768)
        push(0, iq[:0])
        recs = E.drain(0)
        log = eng.records_to_log(E, 0, recs) if mode == 0 else eng.am_records_to_log(E, 0, recs)
        exp_log = common.strip_states(exp_log)
        if k + 1 < len(epochs) and exp_log and exp_log[-1][0] == "lost_sync":
            exp_log = exp_log[:-1]                                 # fired inside the NEXT nrsc5_set_mode (input_reset -> input_set_sync_state): the caller's own doing, not a record
        diffs = common.compare_logs(exp_log, common.strip_states(log))
        if diffs:
            # the known float classes of a first lock (DESIGN.md (c) limit 2) show in single captures too: is this one of them, or the reset's doing?
            fresh_exp, _, _ = reflib.run(iq, mode=mode)
            F = eng.Engine(max_streams=1, q15_capacity=400000, record_capacity=1024, p1_slots=16, lib_path=lib, am_enable=True, l2_feedback=True)
            F.set_mode(0, mode)
            common.run_engine_streaming(F, 0, iq, chunk=32768)
            push_f = F.push_cu8 if iq.dtype == np.uint8 else F.push_cs16
            push_f(0, iq[:0])
            frecs = F.drain(0)
            flog = eng.records_to_log(F, 0, frecs) if mode == 0 else eng.am_records_to_log(F, 0, frecs)
            F.close()
            fdiffs = common.compare_logs(common.strip_states(fresh_exp), common.strip_states(flog))
            problems.append((k, kind, "log: the same capture differs in a FRESH session too (first-lock class)" if fdiffs else "log: ONLY AFTER THE RESET", diffs[:3]))
        if verbose:
            print(f"  epoch {k} {kind:14s} {iq.size:8d} {'bytes' if iq.dtype == np.uint8 else 'int16'}: {nq} decimated samples compared, {len(exp_log)} log entries, frames {sum(1 for r in exp_log if r[0] == 'frame')}")
    E.close()
    return [e[2] for e in epochs], problems


if __name__ == "__main__":
    from nrsc5_amd import build
    from oracle import ref
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 70000
    lib, R = build.build_emu(), ref.RefLib()
    bad = reset_related = epochs_total = 0
    for i in range(n):
        kinds, problems = run_session(lib, R, seed0 + i, verbose="-v" in sys.argv)
        bad += bool(problems)
        epochs_total += len(kinds)
        reset_related += sum(1 for p in problems if p[2] == "q15" or "ONLY AFTER" in p[2])
        print(f"session {seed0 + i}: {' -> '.join(kinds)}: {'OK' if not problems else problems}", flush=True)
    print(f"{n} sessions, {epochs_total} captures, {bad} sessions with a difference; differences that a fresh session of the same capture does not show: {reset_related}")
