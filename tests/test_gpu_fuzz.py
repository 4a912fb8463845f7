"""`-m gpu`: a slice of the randomised sweeps ON THE MI355X (rounds 4 - 5 ran them on the CPU twin only -- whose libm is the reference's), every run against
the UNMODIFIED reference (oracle/_ref incl. its L2) under the strict rule of tests/common.py + the counted classes of bench.py, seeds rotated by the committed counter
tests/fuzz_seed_counter.txt (tools/bump_fuzz_counter.py):

  * 256 FM captures in one zero-copy batch with CFO uniform in +-3 kHz (nearly every stream locks through detect_cfo), impaired channels on every 4th -- other streams than
    tests/test_gpu_batch256.py's fixed set;
  * 128 AM (MA1, cs16) captures through the window pipeline with the replay, interference bursts on every 16th;
  * 64 two-capture sessions through the PUBLIC pipe API on the drop-in (nrsc5_open_pipe, nrsc5_set_mode, samples, nrsc5_set_mode on the live session, samples, nrsc5_close:
    FM -> FM, FM -> AM, AM -> FM, AM -> AM with stale FIR windows and kept sync fields) against the plain reference, event by event.

Budget: under three GPU-minutes.  The assertion messages carry the counts: strict / counted-transient / failed."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import common
from oracle import ref

pytestmark = pytest.mark.gpu

COUNTER = int(open(os.path.join(common.ROOT, "tests", "fuzz_seed_counter.txt")).read().split()[0])


def test_gpu_fuzz_fm_cfo_search_batch(hip_lib):
    if not ref.available(sse=True):
        pytest.skip("oracle/_ref not prebuilt")
    import torch
    from tests import test_gpu_batch256 as t
    base = 100000 + 256 * COUNTER
    out = t.run_batch_against_reference(hip_lib, torch.device("cuda", 0), list(range(base, base + 256)))
    strict, transient, failing = out["streams_equal_under_the_strict_rule"], out["streams_with_transient_loop_state_deviation"], out["streams_failing_by_class"]
    msg = f"FM fuzz, streams {base}..{base + 255}: {out['first_locks_with_integer_cfo']} CFO-search locks, strict {strict}, counted transient {transient}, failing {failing}; {out['transient_details'][:4]} {out['first_diffs'][:2]}"
    print(msg)
    assert not failing and out["streams_compared"] == 256, msg
    assert transient <= 4, msg                                 # round 6 measured 0 of 768 streams on three such batches (profiles/r06_nco_policy_decision.txt)


def _bench(args, timeout=600):
    r = subprocess.run([sys.executable, os.path.join(common.ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_gpu_fuzz_am_batch(hip_lib):
    if not ref.available(sse=True):
        pytest.skip("oracle/_ref not prebuilt")
    base = 50000 + 128 * COUNTER
    r, line = _bench(["--workload", "am-cs16", "--streams", "128", "--stream-base", str(base), "--am-frames", "14", "--steps", "1", "--warmup", "0", "--cpu-baseline-seconds", "1", "--no-extra-legs"])
    assert line is not None, r.stderr[-2000:]
    eq = line["parity"]["reference_equality_rank0"]
    msg = (f"AM fuzz, streams {base}..{base + 127}: compared {eq['streams_compared']}, strict {eq['streams_equal_under_the_strict_rule']}, counted transient "
           f"{eq['streams_with_transient_loop_state_deviation']}, failing {eq['streams_failing_by_class']}, lost sync {eq['streams_with_lost_sync_this_pass']}; failures {line['parity_failures']}")
    print(msg)
    assert r.returncode == 0 and line["parity_failures"] == [] and eq["streams_compared"] == 128, msg
    assert eq["streams_equal_under_the_strict_rule"] + eq["streams_with_transient_loop_state_deviation"] == 128 and eq["streams_with_transient_loop_state_deviation"] <= 2, msg


def test_gpu_fuzz_two_capture_sessions(hip_lib):
    from tests import test_gpu_dropin as gd, test_emu_dropin as td
    sys.path.insert(0, os.path.join(common.ROOT, "tools"))
    import cpu_session_fuzz as fz
    dropin = os.path.join(gd.BUILD["libnrsc5_hipdropin.so"], "libnrsc5_hipdropin.so")
    plain = os.path.join(common.ROOT, "oracle", "_ref", "libnrsc5_plain.so")
    if not (os.path.exists(dropin) and os.path.exists(plain)):
        pytest.skip("drop-in / plain reference not prebuilt")
    n, seed0 = 64, 700000 + 64 * COUNTER
    bad, events, kinds = [], 0, {}
    for i in range(n):
        rng = np.random.default_rng(seed0 + i)
        (ma, a, ka), (mb, b, kb) = fz.make_epoch(rng, long=(i % 2 == 0)), fz.make_epoch(rng, long=(i % 2 == 0))
        exp = td.run_two(plain, a, b, mode_a=ma, mode_b=mb)
        got = td.run_two(dropin, a, b, mode_a=ma, mode_b=mb)
        events += len(exp[0]) + len(exp[1])
        kinds[(ma, mb)] = kinds.get((ma, mb), 0) + 1
        for k in range(2):
            try:
                td._compare_events(exp[k], got[k])
            except AssertionError as ex:
                bad.append(f"session {seed0 + i} capture {k} ({(ka, kb)[k]}): {str(ex)[:200]}")
                break
    msg = f"session fuzz {seed0}..{seed0 + n - 1}: {n} sessions ({kinds}), {events} reference events, {len(bad)} sessions with a difference: {bad[:3]}"
    print(msg)
    assert not bad, msg


def test_gpu_fuzz_fm_service_modes_streaming(hip_lib):
    """The batches above are MP1 cu8; this slice walks what they leave out: MP2 / MP3 / MP11 (P3 / P4 through interleaver IV), cs16 input, impaired channels of every kind, SNR 14 - 30 dB,
    CFO +-3 kHz -- through the STREAMING seam with the in-order L2 feedback (what the drop-in uses), 24 captures per run, each against the unmodified reference incl. its L2
    (tools/cpu_parity_fuzz.py's generator; leftovers classified by its rule: a differing PIDS frame whose CRC the reference accepts, a P1 / P3 / P4 frame bit, an event: failures)."""
    if not ref.available(sse=True):
        pytest.skip("oracle/_ref not prebuilt")
    sys.path.insert(0, os.path.join(common.ROOT, "tools"))
    import bench
    import cpu_parity_fuzz as pf
    from nrsc5_amd import engine as eng, synth
    run, kind = bench._checker(0, True)
    assert kind == "reference"
    seed0 = 900000 + 24 * COUNTER
    strict = counted = 0
    bad, modes = [], {}
    for i in range(24):
        kw = pf.params(i, seed0)
        kw["snr_db"] = max(kw["snr_db"], 14.0)
        if i % 3 == 0:
            kw["mode"] = ("MP2", "MP3", "MP11")[(i // 3) % 3]         # a third of the captures on the extended modes (the generator draws them 3 times in 10)
        if kw["mode"] != "MP1":
            kw["n_blocks"] = max(kw["n_blocks"], 56)                  # interleaver IV needs two block pairs before a P3 frame appears
        modes[(kw["mode"], kw["fmt"])] = modes.get((kw["mode"], kw["fmt"]), 0) + 1
        cap = synth.fm_mp1_capture(0, **kw)
        ref_log = run(cap.iq)
        E = eng.Engine(max_streams=1, q15_capacity=max(2 * 71280 + 32768 * 8, 400000), lib_path=hip_lib, l2_feedback=True)
        common.run_engine_streaming(E, 0, cap.iq)
        log = eng.records_to_log(E, 0, E.drain(0))
        E.close()
        fatal, nex, max_bits, ntr = bench.compare_with_reference(ref_log, log, False)
        if fatal:
            cls = pf.classify(ref_log, fatal)
            if cls["other"] or cls["pids_valid"] or cls["timing"] or cls["mer"]:
                bad.append(f"capture {seed0 + i} {kw['mode']} {kw['fmt']} cfo {kw['cfo_hz']:.0f} snr {kw['snr_db']:.0f}: {cls} {fatal[:3]}")
            else:
                counted += 1                                          # only numbers formed on blocks the reference itself could not demodulate (discarded PIDS frames, MER below 0 dB, loop state there)
        elif ntr:
            counted += 1
        else:
            strict += 1
    msg = f"service-mode fuzz {seed0}..{seed0 + 23}: {modes}: strict {strict}, counted {counted}, failing {len(bad)}: {bad[:3]}"
    print(msg)
    assert not bad, msg
