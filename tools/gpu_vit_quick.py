"""short trellis micro-benchmark: forward (segmented) and traceback, a few batch sizes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nrsc5_amd import engine as eng
G = eng.Engine(max_streams=1, q15_capacity=2 * 71280)
L = 146176
for nf, sg in ((1, 16), (16, 16), (64, 16), (256, 4), (512, 2), (1024, 1)):
    G.tune(eng.TUNE_FWD_SEGMENTS, sg)
    full = G.stage_viterbi_bench(L, nf, 3, reps=5); fwd = G.stage_viterbi_bench(L, nf, 1, reps=5)
    print(f"frames {nf:5d} x {sg:2d} segments: fwd+tb {full:8.3f} ms | fwd {fwd:8.3f} ms | traceback {full - fwd:8.3f} ms", flush=True)
