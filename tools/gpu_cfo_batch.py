"""GPU: the 256-stream CFO-search batch of tests/test_gpu_batch256.py (4 L1 frames per stream, CFO uniform in +-3 kHz: ~235 of 256 streams lock through detect_cfo)
under every NCO policy (NRSC5HIP_TUNE_NCO_EXACT 0..3), each stream against the unmodified reference; prints, per policy and batch seed, how many streams
deviate and in what.   python tools/gpu_cfo_batch.py [bases=0,256] [policies=0,1,2,3] [loop_exact=1]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from nrsc5_amd import engine as eng
    from tests import test_gpu_batch256 as t
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    bases = [int(x) for x in (argv[0] if len(argv) > 0 else "0,256").split(",")]
    pols = [int(x) for x in (argv[1] if len(argv) > 1 else "0,1,2,3").split(",")]
    loops = [int(x) for x in (argv[2] if len(argv) > 2 else "1").split(",")]      # NRSC5HIP_TUNE_LOOP_EXACT: 0 fast loop arithmetic, 1 exact while un-synchronised (default), 2 always
    dev = torch.device("cuda", 0)
    lib = eng.DEFAULT_LIB
    if "--unfused" in sys.argv:                                 # diagnostic twin: the FFT / mix complex products unfused (python -m nrsc5_amd.build --cmul-unfused)
        lib = os.path.join(ROOT, "nrsc5_amd", "libnrsc5hip_unfused.so")
    if "--accurate" in sys.argv:                                # diagnostic twin: double-precision sine / cosine / arc tangent in the Costas loops (python -m nrsc5_amd.build --accurate-trig)
        lib = os.path.join(ROOT, "nrsc5_amd", "libnrsc5hip_acctrig.so")
    for b in bases:
        for p, lp in [(p, lp) for p in pols for lp in loops]:
            t0 = time.time()
            out = t.run_batch_against_reference(lib, dev, list(range(b, b + t.S)), tune=((eng.TUNE_NCO_EXACT, p), (eng.TUNE_LOOP_EXACT, lp)))
            print(json.dumps({"base": b, "policy": p, "loop_exact": lp, "cfo_search_locks": out["first_locks_with_integer_cfo"], "strict": out["streams_equal_under_the_strict_rule"],
                              "transient_streams": out["streams_with_transient_loop_state_deviation"], "failing_by_class": out["streams_failing_by_class"],
                              "block_steps": out["block_steps"], "seconds": round(time.time() - t0, 1),
                              "details": out["transient_details"][:8], "first_diffs": out["first_diffs"][:4]}))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
