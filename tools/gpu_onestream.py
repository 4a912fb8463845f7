"""one bench stream alone (batch API and streaming seam) against the reference: first differing records, with context"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from nrsc5_amd import engine as eng
from tests import common
eng.check_fresh()
ids = [int(x) for x in sys.argv[1].split(",")]
args = bench.parse(["--no-cpu-baseline"])
dev = torch.device("cuda", 0)
run, kind = bench._checker(0, True)
for gs in ids:
    W = bench.Fm(args, dev, 0, [gs])
    iq = W.stream_iq(0)
    ref = common.strip_states(run(iq))
    steps, (recs, counts, frames) = W.one_pass()
    logb = common.strip_states(W.to_log(0, recs[0, :counts[0]], frames[0]))
    E = eng.Engine(max_streams=1, q15_capacity=400000, record_capacity=1024, p1_slots=24, l2_feedback=True)
    common.run_engine_streaming(E, 0, iq, chunk=32768 * 8)
    logs = common.strip_states(eng.records_to_log(E, 0, E.drain(0)))
    for name, lg in (("batch", logb), ("streaming", logs)):
        d = common.compare_logs(ref, lg)
        print(f"stream {gs} {name}: {len(d)} diffs; first: {d[:4]}")
    skip = ("hdc", "soft", "vit", "amsym", "pxsoft", "station")
    a = [r for r in ref if r[0] not in skip]; b = [r for r in logs if r[0] not in skip]
    for i in range(min(8, len(a))):
        if a[i][0] == "block":
            print(i, "ref ", {k: (round(v, 7) if isinstance(v, float) else v) for k, v in a[i][1].items()})
            print(i, "eng ", {k: (round(v, 7) if isinstance(v, float) else v) for k, v in b[i][1].items()})
        else:
            print(i, a[i][0], {k: v for k, v in a[i][1].items() if not hasattr(v, "shape")}, "|", {k: v for k, v in b[i][1].items() if not hasattr(v, "shape")})
    E.close(); W.E.close()
