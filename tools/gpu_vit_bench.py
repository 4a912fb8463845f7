"""Trellis micro-benchmark on the GPU box: P1-size frames, forward pass alone and forward + block-parallel traceback
(the traceback consumes the decision words in place, so it is timed as the difference)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nrsc5_amd import engine as eng
G = eng.Engine(max_streams=1, q15_capacity=2 * 71280)
L = 146176
for nf in (1, 256, 768, 1024):
    full = G.stage_viterbi_bench(L, nf, 3, reps=5); fwd = G.stage_viterbi_bench(L, nf, 1, reps=5)
    print(f"frames {nf:5d}: fwd+parallel-tb {full:8.3f} ms | fwd {fwd:8.3f} ms ({fwd*1e6/(L+64):6.2f} ns/step) | parallel tb {full - fwd:8.3f} ms")
