"""GPU-box diagnostic: compares the fast Viterbi's forward decisions and output on the device with the
CPU-emulated build of the same source (tests/simt), to localise device-only discrepancies."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nrsc5_amd import engine as eng, build

emu = os.path.join(ROOT, "tests", "simt", "libnrsc5hip_emu.so")
G = eng.Engine(max_streams=1, q15_capacity=2 * 71280)
print("selftest failures:", G.stage_selftest())
C = eng.Engine(max_streams=1, q15_capacity=2 * 71280, lib_path=emu)
rng = np.random.default_rng(4)
for L in (2304, 4608):
    soft = rng.integers(-127, 128, size=3 * L, dtype=np.int8); soft[5::6] = 0
    gb, gd = G.stage_viterbi_k7_debug(soft, L)
    cb, cd = C.stage_viterbi_k7_debug(soft, L)
    bad = np.nonzero(gd != cd)[0]
    print(f"len {L}: decision words differing {bad.size}/{gd.size}; output bits differing {(gb != cb).sum()}")
    if bad.size:
        print(" first bad steps", bad[:20], "phases", np.bincount(bad % 6, minlength=6), "s in chunk", np.bincount(bad % 64, minlength=64))
        for t in bad[:6]:
            x = int(gd[t]) ^ int(cd[t])
            print(f"  step {t} (phase {t % 6}, s {t % 64}): gpu {int(gd[t]):016x} cpu {int(cd[t]):016x} xor {x:016x}")
