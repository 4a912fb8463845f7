"""Trellis micro-benchmark on the GPU box: P1-size frames, forward pass (by number of segment waves per frame) and forward +
block-parallel traceback (the traceback consumes the decision words in place, so it is timed as the difference)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nrsc5_amd import engine as eng
G = eng.Engine(max_streams=1, q15_capacity=2 * 71280)
L = 146176
for nf, segs in ((1, (1, 4, 16)), (4, (1, 16)), (16, (1, 16)), (64, (1, 4, 16)), (256, (1, 2, 4, 8)), (512, (1, 2, 4)), (1024, (1, 2)), (2048, (1,))):
    for sg in segs:
        G.tune(eng.TUNE_FWD_SEGMENTS, sg)
        full = G.stage_viterbi_bench(L, nf, 3, reps=5); fwd = G.stage_viterbi_bench(L, nf, 1, reps=5)
        print(f"frames {nf:5d} x {sg:2d} segments: fwd+parallel-tb {full:8.3f} ms | fwd {fwd:8.3f} ms ({fwd*1e6/(L+64):6.2f} ns per step and launch, {fwd*1e6/(L+64)/nf*min(nf*sg,1024)/sg if False else fwd*1e6/(L+64)/nf:8.4f} ns/step/frame) | parallel tb {full - fwd:8.3f} ms", flush=True)
print("fwd stats (boundaries checked, segments repaired):", G.fwd_stats())
