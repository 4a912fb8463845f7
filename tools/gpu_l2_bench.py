"""L2 index kernel timing: 1024 FM P1 frames (5 audio PDUs, 87 packets each) + 2048 AM P1 frames in one launch each.
Run under rocprofv3 --kernel-trace --stats (tools/gpu_l2.sh); prints host wall times as a cross-check."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from nrsc5_amd import engine as eng, synth_l2

E = eng.Engine(max_streams=1, q15_capacity=2 * 71280)
for nbits, name, reps in ((146176, "multi", 1024), (146176, "single", 1024), (3750, "two", 2048)):
    bits = [b for n, b, _ in synth_l2.test_frames(nbits) if n == name][0]
    frames = np.broadcast_to(bits, (reps, nbits)).copy()
    E.stage_l2_index(frames[:4], want_bytes=False)
    t = time.time()
    out = E.stage_l2_index(frames, want_bytes=False)
    dt = time.time() - t
    idx = out[-1][0]
    print(f"{nbits}-bit '{name}' x {reps}: host wall {dt * 1e3:.1f} ms incl. packing + copies; n_pdu {idx['n_pdu']} packets {sum(p['nop'] for p in idx['pdus'])} status {eng.L2_STATUS[idx['status']]}")
E.close()
