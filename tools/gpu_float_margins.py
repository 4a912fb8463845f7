"""Measured deviation of every float the engine reports from the CPU oracle's (== the unmodified reference's, tests/test_oracle*.py),
per field: largest absolute and largest RELATIVE difference (|a - b| / |a|, no floor) over a set of captures -- the margins behind
the tolerances of tests/common.py.   gpurun -- 'python tools/gpu_float_margins.py > gpurun_out/float_margins.txt'"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nrsc5_amd import engine as eng, synth, synth_am
from oracle import port
from tests import common

eng.check_fresh()
O = port.Oracle()
stats = {}


def note(field, a, b, where):
    d = abs(a - b)
    rel = d / abs(a) if a != 0 else (0.0 if d == 0 else float("inf"))
    s = stats.setdefault(field, {"n": 0, "max_abs": 0.0, "max_rel": 0.0, "where_abs": "", "where_rel": "", "min_mag": float("inf"), "max_rel_over_1e-2": 0.0})
    s["n"] += 1
    if d > s["max_abs"]:
        s["max_abs"], s["where_abs"] = d, f"{where}: {a!r} vs {b!r}"
    if rel > s["max_rel"] and rel != float("inf"):
        s["max_rel"], s["where_rel"] = rel, f"{where}: {a!r} vs {b!r}"
    if abs(a) >= 1e-2 and rel > s["max_rel_over_1e-2"]:
        s["max_rel_over_1e-2"] = rel
    if a != 0:
        s["min_mag"] = min(s["min_mag"], abs(a))


def run(name, cap, am=False, p1_async=False, l2=False):
    E = eng.Engine(max_streams=1, q15_capacity=400000, record_capacity=1024, p1_slots=24, am_enable=am, p1_async=p1_async, l2_feedback=l2)
    if am:
        E.set_mode(0, eng.MODE_AM)
    common.run_engine_streaming(E, 0, cap.iq, chunk=32768 * 8)
    recs = E.drain(0)
    log = common.strip_states((eng.am_records_to_log if am else eng.records_to_log)(E, 0, recs))
    ol = common.strip_states(O.run(cap.iq, mode=1 if am else 0, p1_hook=O.l2_hook() if l2 else None)[0])
    skip = ("hdc", "soft", "vit", "amsym", "pxsoft", "station")
    a = [r for r in ol if r[0] not in skip]; b = [r for r in log if r[0] not in skip]
    assert [k for k, _ in a] == [k for k, _ in b], name
    # stretches in which the reference is falsely locked (sync .. lost-sync with a frame of cber > 0.02)
    loose, start, bad = set(), None, False
    for i, (k, v) in enumerate(a):
        if k == "sync":
            start, bad = i, False
        elif k == "ber" and v["cber"] > 0.02:
            bad = True
        elif k == "lost_sync" and start is not None:
            if bad:
                loose.update(range(start, i + 1))
            start = None
    for i, ((k, va), (_, vb)) in enumerate(zip(a, b)):
        for f, x in va.items():
            if isinstance(x, float):
                fine = k != "block" or va.get("state_after") == 2
                tag = " (falsely locked)" if i in loose else ("" if fine else " (not FINE)")
                if l2 and i not in loose:
                    tag += " [cfo ~ 0, after a false lock]"
                note(f"{k}.{f}" + tag, x, vb[f], f"{name}#{i}")
    E.close()


cases = [(n, synth.fm_mp1_capture(**kw), False) for n, kw in common.GOLDEN_CASES.items()]
cases += [(n, synth_am.am_ma1_capture(**kw), True) for n, kw in common.GOLDEN_AM_CASES.items()]
rng = np.random.default_rng(7)
for k in range(10):
    cases.append((f"fm_rand{k}", synth.fm_mp1_capture(0, seed=200 + k, cfo_hz=float(rng.uniform(-300, 300)), offset=int(rng.integers(0, 4320)),
                                                      snr_db=(15.0, 20.0, 25.0)[k % 3], n_blocks=72), False))
for k in range(3):
    cases.append((f"am_rand{k}", synth_am.am_ma1_capture(n_frames=10, seed=300 + k, cfo_hz=float(rng.uniform(-100, 100)), offset=int(rng.integers(0, 2400))), True))
for mode, kw in (("MP2", dict(n_blocks=52, seed=31, cfo_hz=20.0, offset=300, snr_db=25)), ("MP3", dict(n_blocks=54, seed=32, cfo_hz=-150.0, offset=500, snr_db=18, fmt="cs16")),
                 ("MP11", dict(n_blocks=52, seed=33, cfo_hz=0.0, offset=64, snr_db=14))):
    cases.append((f"fm_{mode}_snr{kw['snr_db']}_cfo{kw['cfo_hz']}", synth.fm_mp1_capture(n_frames=0, mode=mode, **kw), False))
for name, cap, am in cases:
    run(name, cap, am)
for sd, c, o in ((23, 0.0, 1234), (24, 10.0, 2208), (25, 10.0, 777), (26, 0.0, 100), (27, 0.5, 3000)):
    run(f"fm_lowcfo{sd}", synth.fm_mp1_capture(0, seed=sd, cfo_hz=c, offset=o, snr_db=20, n_blocks=96), False, l2=True)
print(f"# {len(cases)} + 5 captures: 5 FM + 3 AM goldens, 10 FM random CFO / offset / SNR 15-25 dB x 72 blocks, 3 AM x 10 L1 frames, MP2 / MP3 / MP11 (the last at 14 dB SNR and CFO 0),")
print("# and 5 captures with CFO 0 .. 10 Hz of which 3 lock falsely first (tagged); engine (streaming seam, in-order) vs oracle")
print(f"{'field':34s} {'n':>6s} {'max |a-b|':>12s} {'max |a-b|/|a|':>14s} {'(|a|>=1e-2)':>12s} {'min |a|':>10s}")
for f, s in sorted(stats.items()):
    print(f"{f:34s} {s['n']:6d} {s['max_abs']:12.3e} {s['max_rel']:14.3e} {s['max_rel_over_1e-2']:12.3e} {s['min_mag']:10.2e}")
print()
for f, s in sorted(stats.items()):
    print(f"{f}: worst abs at {s['where_abs']}; worst rel at {s['where_rel']}")
