import sys; sys.path.insert(0, ".")
from nrsc5_amd import engine as eng
E = eng.Engine(max_streams=1, q15_capacity=2 * 71280)
for L in (3750 // 2 * 2 + 2, 24000):
    for nf in (256, 2304):
        f = E.stage_viterbi_k9_bench(L, nf, 1); t = E.stage_viterbi_k9_bench(L, nf, 2); b = E.stage_viterbi_k9_bench(L, nf, 3)
        print(f"len {L} frames {nf}: forward {f:.3f} ms ({f * 1e6 / (L + 64):.0f} ns/step)  traceback {t:.3f} ms ({t * 1e6 / (L + 64):.0f} ns/step)  both {b:.3f} ms")
