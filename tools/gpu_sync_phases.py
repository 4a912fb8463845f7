import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nrsc5_amd import engine as eng, synth
from tests import common
cap = synth.fm_mp1_capture(0, seed=1, cfo_hz=100.0, offset=700, snr_db=20, n_blocks=40)
E = eng.Engine(max_streams=1, q15_capacity=cap.iq.size // 4 + 200000)
E.tune(eng.TUNE_SYNC_PHASES, 1)
common.run_engine_streaming(E, 0, cap.iq)
recs = E.drain(0)
c = E.debug_sync_phases()
names = ["sync_adjust", "costas", "coarse", "(gap)", "smag+regress", "equalise+MER", "soft bits", "PIDS", "finish"]
n = len(recs)
print("blocks", n)
for nm, v in zip(["sync_adjust", "costas", "coarse", "samperr/angle", "equalise+MER", "soft bits", "PIDS gather+viterbi", "finish"], c):
    print(f"{nm:22s} {v / n:10.0f} cycles/block  ({v / n / 100:7.1f} us @100MHz if wall clock)")
