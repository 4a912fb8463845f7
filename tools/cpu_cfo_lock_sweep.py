"""CPU only: how often, and by how much, does a receiver whose float pipeline is NOT bit-identical to the reference's deviate from it at a lock after a CFO
search (integer CFO != 0)?  Synthetic MP1 captures with |CFO| in (185, 300) Hz (integer CFO = +-1), random timing offsets, SNR 15 / 20 / 25 dB, 40 blocks each,
through (a) the unmodified reference (oracle/_ref) and (b) the CPU-emulated twin of the library (tests/simt: the kernels' logic with glibc's libm and no fused
multiply-adds -- a third float sequence beside the reference's and the GPU's); complete logs compared under the strict rule (tests/common.py: integers exact,
floats 1e-4).  DESIGN.md (c) limit 2; result of the round-4 run: profiles/r04_cfo_lock_transients.txt.
    python tools/cpu_cfo_lock_sweep.py [--self | --nco | --emu-lib PATH | --policy N] [--loop-exact N] [processes=8] [captures=900]
--self: both sides are the unmodified reference, (b) linked with another FFT (oracle/_ref/libnrsc5_ref_sse_dp.so; `make -C oracle _ref/libnrsc5_ref_sse_dp.so`).
--nco:  (b) is the reference with an ideal (double-precision) oscillator inside each symbol instead of its float recurrence (tools/build_ref_ideal_nco.py)."""
import json, os, re, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# --self: the reference (Stockham radix-4 float FFT) against the reference on a double-precision FFT; --nco: ... against the reference with an ideal oscillator inside each
# symbol (tools/build_ref_ideal_nco.py) -- instead of the reference against the library
SELF = next((a for a in ("--self", "--nco") if a in sys.argv), "")
if SELF:
    sys.argv.remove(SELF)
EMU_OVERRIDE = None                         # --emu-lib PATH: another emulated twin (tools/build_emu_nco_growth.py) instead of tests/simt/libnrsc5hip_emu.so
POLICY = None                               # --policy N: NRSC5HIP_TUNE_NCO_EXACT of the emulated twin (0 closed form [default], 1 first block exact, 2 until FINE, 3 always)
if "--policy" in sys.argv:
    k = sys.argv.index("--policy"); POLICY = int(sys.argv[k + 1]); del sys.argv[k:k + 2]
LOOP = None                                 # --loop-exact N: NRSC5HIP_TUNE_LOOP_EXACT of the emulated twin (0 fast loop arithmetic, 1 exact while un-synchronised [default], 2 always)
if "--loop-exact" in sys.argv:
    k = sys.argv.index("--loop-exact"); LOOP = int(sys.argv[k + 1]); del sys.argv[k:k + 2]
if "--emu-lib" in sys.argv:
    k = sys.argv.index("--emu-lib"); EMU_OVERRIDE = sys.argv[k + 1]; del sys.argv[k:k + 2]


def work(args):
    seed, cfo, offset, snr = args
    from nrsc5_amd import synth, build
    from tests import common, engine_checks as ec
    import bench
    run, kind = bench._checker(0, True)
    assert kind == "reference", "oracle/_ref is not built (python -c 'import __graft_entry__ as g; g.build()')"
    cap = synth.fm_mp1_capture(0, seed=seed, cfo_hz=cfo, offset=offset, snr_db=snr, n_blocks=40)
    ref_log = run(cap.iq)
    if SELF:
        # the unmodified reference against ITSELF on a different FFT (oracle/cpu_fft_dp.c: double precision, rounded once)
        from oracle import ref
        R2 = ref.RefLib(path=os.path.join(ROOT, "oracle", "_ref", "libnrsc5_ref_sse_dp.so" if SELF == "--self" else "libnrsc5_ref_sse_nco.so"))
        log = R2.run(cap.iq, mode=0)[0]
    else:
        from nrsc5_amd import engine as eng
        E, recs, log = ec.run_capture(EMU_OVERRIDE or build.EMU_LIB, cap, tune=(() if POLICY is None else ((eng.TUNE_NCO_EXACT, POLICY),)) + (() if LOOP is None else ((eng.TUNE_LOOP_EXACT, LOOP),)))
        E.close()
    exp, got = common.strip_states(ref_log), common.strip_states(log)
    diffs = common.compare_logs(exp, got)
    mer, dm, dsamp = [d for d in diffs if " mer." in d], 0.0, 0
    fmax = {}                                   # largest deviation per float field of the block records (next_angle, phase_re / _im: absolute; prev_angle: relative)
    for d in diffs:
        m = re.search(r"(\w+)\.(\w+): expected (\S+) got (\S+)", d)
        if not m:
            continue
        if m.group(1) == "mer":
            dm = max(dm, abs(float(m.group(3)) - float(m.group(4))))
        elif m.group(2) in ("samperr", "next_samperr", "keep"):
            dsamp = max(dsamp, abs(int(float(m.group(3))) - int(float(m.group(4)))))
        elif m.group(1) in ("block", "sync"):
            a, b = float(m.group(3)), float(m.group(4))
            dev = abs(a - b) / max(abs(a), 1e-30) if m.group(2) in ("prev_angle", "freq_offset") else abs(a - b)
            fmax[m.group(2)] = max(fmax.get(m.group(2), 0.0), dev)
    frames_equal = not any(d for d in diffs if " frame." in d or " pids." in d or " ber." in d)
    sync_equal = not any(d for d in diffs if " sync." in d)
    return seed, round(cfo, 1), len(diffs), len(mer), round(dm, 3), dsamp, frames_equal, sync_equal, {k: float('%.3g' % v) for k, v in fmax.items()}


if __name__ == "__main__":
    from multiprocessing import Pool
    from nrsc5_amd import build
    build.build_emu()
    nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 900
    rng = np.random.default_rng(4242)
    jobs = []
    for i in range(n):
        cfo = float(rng.uniform(185, 300)) * (1 if rng.integers(0, 2) else -1)
        jobs.append((1000 + i, cfo, int(rng.integers(0, 4320)), (15.0, 20.0, 25.0)[i % 3]))
    t = time.time()
    with Pool(nproc) as p:
        res = p.map(work, jobs, chunksize=4)
    dev = [r for r in res if r[2]]
    out = {"captures": len(res), "seconds": round(time.time() - t, 1), "with_any_difference": len(dev), "with_a_mer_difference": sum(1 for r in res if r[3]),
           "mer_deviation_dB_sorted": sorted([r[4] for r in res if r[3]], reverse=True),
           "timing_pick_deviation_samples_sorted": sorted([r[5] for r in dev], reverse=True),
           "deviating_captures_with_frames_pids_ber_equal": sum(1 for r in dev if r[6]), "deviating_captures_with_sync_events_equal": sum(1 for r in dev if r[7]),
           "largest_float_deviation_per_field (prev_angle, freq_offset relative; others absolute)": {k: max(r[8].get(k, 0.0) for r in dev) for k in sorted({k for r in dev for k in r[8]})},
           "worst (seed, cfo, diffs, mer diffs, max mer dB, max timing samples, frames equal, sync equal, float deviations)": sorted(dev, key=lambda r: -r[4])[:8]}
    print(json.dumps(out, indent=1))
