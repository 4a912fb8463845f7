"""CPU only: randomised parity sweep -- the CPU-emulated twin of the library (tests/simt) against the UNMODIFIED reference (oracle/_ref) on synthetic hybrid-FM captures (MP1, some MP2 / MP3 / MP11)
with random carrier offset (+-3 kHz: CFO searches up to +-8 bins), timing offset, SNR (8 .. 30 dB), input format and, in half of them, an impaired channel
(sample-clock error, echoes, analog host, fading, clipping), judged by bench.py's own rule (compare_with_reference: frames / events / estimates strict, the two
counted exemption classes of DESIGN.md (c)).  Prints what is left over.   python tools/cpu_parity_fuzz.py [--am] [processes=8] [captures=400] [seed0=50000]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
AM = "--am" in sys.argv                     # hybrid AM (MA1 / MA3, cs16) instead of FM
if AM:
    sys.argv.remove("--am")


def params(i, seed0):
    rng = np.random.default_rng(seed0 + i)
    from nrsc5_amd.channel import Impairments
    chan = None
    if rng.integers(0, 2):
        kind = int(rng.integers(0, 5))
        ppm = float(rng.uniform(-100, 100))
        if kind == 0: chan = Impairments(ppm=ppm)
        elif kind == 1: chan = Impairments(ppm=ppm, paths=((float(rng.uniform(5e-6, 40e-6)), float(rng.uniform(-10, -3)), float(rng.uniform(-2, 2)), float(rng.uniform(0, 6.28))),))
        elif kind == 2: chan = Impairments(ppm=ppm, host_db=20.0)
        elif kind == 3: chan = Impairments(ppm=ppm, fade_db=float(rng.uniform(3, 10)), fade_period_s=float(rng.uniform(0.7, 3.0)))
        else: chan = Impairments(ppm=ppm, clip_rms=float(rng.uniform(2.0, 3.5)))
    return dict(seed=seed0 + i, cfo_hz=float(rng.uniform(-3000, 3000)), offset=int(rng.integers(0, 4320)), snr_db=float(rng.uniform(8, 30)),
                fmt=("cu8", "cs16")[int(rng.integers(0, 2))], n_blocks=int(rng.integers(36, 56)), chan=chan,
                mode=("MP1", "MP1", "MP1", "MP1", "MP1", "MP1", "MP1", "MP2", "MP3", "MP11")[int(rng.integers(0, 10))],
                rms_lsb=9.0 if (chan is not None and chan.host_db is not None) else 20.0)


def params_am(i, seed0):
    rng = np.random.default_rng(seed0 + i)
    from nrsc5_amd.channel import Impairments
    chan = None
    if rng.integers(0, 2):
        kind = int(rng.integers(0, 3))
        ppm = float(rng.uniform(-70, 70))
        if kind == 0: chan = Impairments(ppm=ppm)
        elif kind == 1: chan = Impairments(ppm=ppm, paths=((float(rng.uniform(20e-6, 120e-6)), float(rng.uniform(-10, -4)), float(rng.uniform(-0.5, 0.5)), float(rng.uniform(0, 6.28))),))
        else: chan = Impairments(ppm=ppm, fade_db=float(rng.uniform(2, 6)), fade_period_s=float(rng.uniform(1.5, 4.0)))
    return dict(n_frames=int(rng.integers(10, 14)), seed=seed0 + i, cfo_hz=float(rng.uniform(-40, 40)), offset=int(rng.integers(0, 8640)), noise=float(rng.uniform(0.2, 0.9)),
                fmt="cs16", mode=("MA1", "MA3")[int(rng.integers(0, 4) == 0)], chan=chan)


def work(args):
    i, seed0 = args
    from nrsc5_amd import synth, build, engine as eng
    from tests import common, engine_checks as ec
    import bench
    run, kind = bench._checker(1 if AM else 0, True)
    assert kind == "reference"
    if AM:
        from nrsc5_amd import synth_am
        kw = params_am(i, seed0)
        cap = synth_am.am_ma1_capture(**kw)
        ref_log = run(cap.iq)
        E = eng.Engine(max_streams=1, q15_capacity=400000, record_capacity=512, p1_slots=16, lib_path=build.EMU_LIB, am_enable=True, l2_feedback=True)
        E.set_mode(0, eng.MODE_AM)
        common.run_engine_streaming(E, 0, cap.iq, chunk=32768)
        log = eng.am_records_to_log(E, 0, E.drain(0))
        E.close()
    else:
        kw = params(i, seed0)
        cap = synth.fm_mp1_capture(0, **kw)
        ref_log = run(cap.iq)
        E, recs, log = ec.run_capture(build.EMU_LIB, cap, l2_feedback=True)
        E.close()
    fatal, nex, max_bits, ntr = bench.compare_with_reference(ref_log, log, AM)
    nframes = sum(1 for k, _ in ref_log if k in ("frame", "p3"))
    return i, {k: (v if k != "chan" else repr(v)) for k, v in kw.items()}, fatal[:4], nex, max_bits, ntr, nframes


if __name__ == "__main__":
    from multiprocessing import Pool
    from nrsc5_amd import build
    build.build_emu()
    nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
    t = time.time()
    with Pool(nproc) as p:
        res = p.map(work, [(i, seed0) for i in range(n)], chunksize=2)
    bad = [r for r in res if r[2]]
    out = {"captures": n, "seconds": round(time.time() - t, 1), "reference_frames_in_all": sum(r[6] for r in res), "captures_with_frames": sum(1 for r in res if r[6]),
           "captures_with_fatal_differences": len(bad), "captures_with_counted_transients": sum(1 for r in res if r[5]), "transient_fields": sum(r[5] for r in res),
           "frames_exempt_cber": sum(r[3] for r in res), "fatal": [(r[0], r[1], r[2]) for r in bad[:12]]}
    print(json.dumps(out, indent=1, default=str))
