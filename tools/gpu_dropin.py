"""the drop-in leg of bench.py alone (stream 0 of the FM batch through nrsc5_pipe_samples_cu8 in 32768-byte calls), a few repetitions;
NRSC5HIP_SYNC_DELIVERY=1 in the environment gives the synchronous delivery for comparison.   python tools/gpu_dropin.py [reps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from nrsc5_amd import engine as eng
eng.check_fresh()
args = bench.parse(["--no-cpu-baseline"])
W = bench.Fm(args, torch.device("cuda", 0), 0, [0])
iq = np.ascontiguousarray(W.stream_iq(0))
W.E.close(); del W
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    out = bench.dropin_leg(iq, bench.FS)
    print(json.dumps({k: out.get(k) for k in ("dropin", "dropin_strict_delivery", "plain", "events_equal", "events_equal_strict_delivery", "events", "breakdown")}))
