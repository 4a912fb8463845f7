"""K=9 decode micro-benchmark: the single-wave form against the segment-wave form (noisy code words, one sign in 16 flipped).
gpurun -- 'python tools/gpu_k9_bench.py'"""
import sys; sys.path.insert(0, ".")
from nrsc5_amd import engine as eng
E = eng.Engine(max_streams=1, q15_capacity=2 * 71280)
for L in (3750, 24000, 30000):
    for nf in (1, 256):
        for G in (1, 4, 8):
            if L == 3750 and G == 8: continue
            E.tune(eng.TUNE_AM_SEGMENTS, G)
            a = E.k9_stats()
            f = E.stage_viterbi_k9_bench(L, nf, 1); t = E.stage_viterbi_k9_bench(L, nf, 2); b = E.stage_viterbi_k9_bench(L, nf, 3)
            z = E.k9_stats()
            print(f"len {L:5d} frames {nf:4d} segments {G}: forward {f:.3f} ms ({f * 1e6 / (L + 64):.0f} ns/step)  traceback {t:.3f} ms ({t * 1e6 / (L + 64):.0f} ns/step)  both {b:.3f} ms"
                  f"   boundaries fwd {z[0] - a[0]} re-run {z[1] - a[1]}, tb {z[2] - a[2]} re-walked {z[3] - a[3]}", flush=True)
