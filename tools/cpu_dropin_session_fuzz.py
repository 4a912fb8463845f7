"""CPU only: randomised two-capture sessions through the PUBLIC pipe API (nrsc5_open_pipe, nrsc5_set_mode, nrsc5_pipe_samples_*, nrsc5_set_mode on the live session, nrsc5_pipe_samples_*,
nrsc5_close) -- the reference's own L4 / L2 code over integration/input_hip.c and the CPU-emulated twin of the library, against the plain reference library; events compared one by one
(SYNC / LOST_SYNC / MER / BER / HDC packets).  Captures as in tools/cpu_session_fuzz.py.    python tools/cpu_dropin_session_fuzz.py [sessions=40] [seed0=60000]"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

if __name__ == "__main__":
    import cpu_session_fuzz as fz
    from nrsc5_amd import build
    from tests import test_emu_dropin as td
    from tests import common
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
    build.build_emu()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "integration"), "emu"], stdout=subprocess.DEVNULL)
    dropin = os.path.join(ROOT, "integration", "_build", "libnrsc5_emudropin.so")
    plain = os.path.join(ROOT, "oracle", "_ref", "libnrsc5_plain.so")
    bad = after_reset_only = events = 0
    for i in range(n):
        rng = np.random.default_rng(seed0 + i)
        (ma, a, ka), (mb, b, kb) = fz.make_epoch(rng, long=True), fz.make_epoch(rng, long=True)   # long enough for SYNC / MER / BER / audio packets
        exp = td.run_two(plain, a, b, mode_a=ma, mode_b=mb)
        got = td.run_two(dropin, a, b, mode_a=ma, mode_b=mb)
        events += len(exp[0]) + len(exp[1])
        verdict = "OK"
        for k, (e, g, mode, iq) in enumerate(((exp[0], got[0], ma, a), (exp[1], got[1], mb, b))):
            try:
                td._compare_events(e, g)
            except AssertionError as ex:
                # a class that single captures show too (DESIGN.md (c) limit 2), or the session's doing?
                try:
                    td._compare_events(td._run(plain, iq, mode=mode), td._run(dropin, iq, mode=mode))
                    fresh_differs = False
                except AssertionError:
                    fresh_differs = True
                verdict = f"capture {k} ({(ka, kb)[k]}): {str(ex)[:160]} -- {'the same capture differs in a fresh session too' if fresh_differs else 'ONLY IN THIS SESSION'}"
                bad += 1
                after_reset_only += not fresh_differs
                break
        print(f"session {seed0 + i}: {ka} -> {kb}: {len(exp[0])} + {len(exp[1])} events: {verdict}", flush=True)
    print(f"{n} sessions, {events} reference events, {bad} sessions with a difference; of those only in the session (not in a fresh one of the same capture): {after_reset_only}")
