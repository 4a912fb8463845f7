"""Locks after a CFO search (integer CFO != 0) on the CPU-emulated twin of the library against the UNMODIFIED reference: the one place where the two may differ
beyond 1e-4, in loop-internal state only (DESIGN.md (c) limit 2: the reference's float NCO recurrence drifts, the symbol kernel's closed-form phase does not;
measured over 900 such captures in profiles/r04_cfo_lock_transients.txt).  Pinned here, on the captures that showed the largest deviations there plus ordinary ones:
every frame, PIDS word, BER, SYNC / LOST_SYNC event and freq_offset is equal in ALL of them; whatever else differs is a block's timing pick (by <= 3 samples), its
loop / NCO diagnostics, or a MER report within 40 records of the SYNC event -- and nothing differs in most captures."""
import os
import re

import pytest

from nrsc5_amd import synth
from oracle import ref
from tests import common, engine_checks as ec

pytestmark = pytest.mark.skipif(not ref.available(sse=True), reason="oracle/_ref not built (needs /root/reference)")

# (seed, cfo, offset, snr): the first four deviated in the 900-capture sweep (MER by 2.3 / 0.49 / 0.41 / 0.35 dB), the others did not
CASES = [(1370, 186.07498555527155, 1634, 20.0), (1493, 294.72944310475714, 4311, 20.0), (1625, -245.78306178496598, 2745, 20.0), (1602, -206.126519085935, 3997, 25.0),
         (1001, 233.0, 100, 25.0), (1002, -291.0, 4000, 20.0), (1003, 199.0, 2160, 15.0), (1004, -188.0, 17, 20.0)]
LOOP_FIELDS = {"samperr", "keep", "next_samperr", "next_angle", "prev_angle", "phase_re", "phase_im"}


@pytest.mark.parametrize("seed,cfo,offset,snr", CASES)
def test_cfo_search_lock_differs_in_loop_state_only(emu_lib, seed, cfo, offset, snr):
    cap = synth.fm_mp1_capture(0, seed=seed, cfo_hz=cfo, offset=offset, snr_db=snr, n_blocks=40)
    exp = common.strip_states(ref.RefLib(sse=True).run(cap.iq)[0])
    E, recs, log = ec.run_capture(emu_lib, cap)
    E.close()
    got = common.strip_states(log)
    kept = [x for x in exp if x[0] not in ("hdc", "soft", "vit", "amsym", "pxsoft", "station")]
    assert any(k == "block" and v["cfo"] != 0 for k, v in kept), "the capture was meant to lock through the CFO search"
    syncs = [i for i, (k, _) in enumerate(kept) if k == "sync"]
    for d in common.compare_logs(exp, got):
        m = re.match(r"#(\d+) (\w+)\.(\w+): expected (\S+) got (\S+)", d)
        assert m, d                                              # a differing frame / bit array / record sequence: never
        idx, kind, field, a, b = int(m.group(1)), m.group(2), m.group(3), float(m.group(4)), float(m.group(5))
        assert kind in ("block", "mer"), d                       # sync events, freq_offset, BER, PIDS: never
        if kind == "mer":
            assert any(0 <= idx - j <= 40 for j in syncs) and abs(a - b) <= 3.0, d
        else:
            assert field in LOOP_FIELDS, d
            if field in ("samperr", "keep", "next_samperr"):
                assert abs(a - b) <= 3, d
