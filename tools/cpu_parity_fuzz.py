"""CPU only: randomised parity sweep -- the CPU-emulated twin of the library (tests/simt) against the UNMODIFIED reference (oracle/_ref) on synthetic hybrid-FM captures (MP1, some MP2 / MP3 / MP11)
with random carrier offset (+-3 kHz: CFO searches up to +-8 bins), timing offset, SNR (8 .. 30 dB), input format and, in half of them, an impaired channel
(sample-clock error, echoes, analog host, fading, clipping), judged by bench.py's own rule (compare_with_reference: frames / events / estimates strict, the two
counted exemption classes of DESIGN.md (c)).  Prints what is left over.   python tools/cpu_parity_fuzz.py [--am | --batch] [processes=8] [captures=400] [seed0=50000]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
AM = "--am" in sys.argv                     # hybrid AM (MA1 / MA3, cs16) instead of FM
if AM:
    sys.argv.remove("--am")
EMU_OVERRIDE = None                         # --emu-lib PATH: another emulated twin (tools/build_emu_nco_growth.py) instead of tests/simt/libnrsc5hip_emu.so
if "--emu-lib" in sys.argv:
    _k = sys.argv.index("--emu-lib"); EMU_OVERRIDE = sys.argv[_k + 1]; del sys.argv[_k:_k + 2]
BATCH = "--batch" in sys.argv               # FM through the zero-copy batch with the window pipeline and replay, six captures per engine
if BATCH:
    sys.argv.remove("--batch")


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


_ORACLE = None


def classify(ref_log, fatal):
    """What the rule's leftovers are made of: ('pids_valid', n) = differing PIDS frames whose CRC is VALID in the reference (information lost: must be 0),
    'pids_garbage' = differing PIDS frames the reference discards (CRC fails: Viterbi output on a block it could not demodulate), 'mer_noise' = MER reports below 0 dB
    (a block of noise), 'timing' / 'loop' = block fields, 'other' = anything else (frames, events, estimates: must be 0)."""
    global _ORACLE
    import re
    from tests import common
    from oracle import port
    if _ORACLE is None:
        _ORACLE = port.Oracle()
    kept = [x for x in common.strip_states(ref_log) if x[0] not in ("hdc", "soft", "vit", "amsym", "pxsoft", "station")]
    out = {"pids_valid": 0, "pids_garbage": 0, "mer_noise": 0, "mer": 0, "timing": 0, "loop": 0, "other": 0}
    for d in fatal:
        m = re.match(r"#(\d+) (\w+)\.(\w+): (?:(\d+) elements differ)?(?:expected (\S+) got (\S+))?", d)
        if not m:
            out["other"] += 1; continue
        idx, kind, field = int(m.group(1)), m.group(2), m.group(3)
        if kind == "pids":
            ok = bool(_ORACLE.pids_crc_ok(np.asarray(kept[idx][1]["bits"], dtype=np.uint8)))
            out["pids_valid" if ok else "pids_garbage"] += 1
        elif kind == "mer":
            out["mer_noise" if float(m.group(5)) < 0.0 else "mer"] += 1
        elif kind == "block":
            out["timing" if field in ("samperr", "keep", "next_samperr") else "loop"] += 1
        else:
            out["other"] += 1
    return out


def work_batch(args):
    """--batch: six FM cu8 captures per engine through the PRODUCT's headline path -- zero-copy batch, window pipeline (decodes overlapped with the block steps), L2 -> L1
    feedback on the device with replay -- a third of them with an interference burst that breaks a P1 frame's first L2 header in mid-stream (LOST_SYNC + re-acquisition)."""
    i, seed0 = args
    from nrsc5_amd import synth, build, engine as eng
    from tests import engine_checks as ec
    import bench
    run, kind = bench._checker(0, True)
    assert kind == "reference"
    caps, kws = [], []
    for k in range(6):
        kw = params(6 * i + k, seed0)
        kw["fmt"] = "cu8"; kw["mode"] = "MP1"; kw["n_blocks"] = int(52 + (kw["n_blocks"] % 20))
        rng = np.random.default_rng(seed0 + 7 * (6 * i + k) + 3)
        if rng.integers(0, 3) == 0:                           # an interference burst over 8 - 12 blocks: breaks one P1 frame's first header (tests/engine_checks.py: drift_replay_captures)
            kw["burst"] = (float(rng.uniform(18.0, 40.0)), int(rng.integers(8, 13)), float(rng.uniform(5.0, 10.0)))
            kw["n_blocks"] = int(rng.integers(90, 100))
        caps.append(synth.fm_mp1_capture(0, **kw)); kws.append(kw)
    n = len(caps)
    stride = max(c.iq.size for c in caps); stride += (-stride) % 16
    host = np.zeros((n, stride), dtype=np.uint8)
    for k, c in enumerate(caps):
        host[k, :c.iq.size] = c.iq
    E = eng.Engine(max_streams=n, q15_capacity=2 * 71280, record_capacity=512, p1_slots=8, p1_async=True, l2_feedback=True, batch_zero_copy=True, lib_path=(EMU_OVERRIDE or build.EMU_LIB))
    dev = ec._to_device(E, host)
    E.batch_append_cu8(dev, stride, [c.iq.size - c.iq.size % 4 for c in caps])
    E.batch_process(n)
    recs, counts, frames = E.batch_fetch_view(n)
    out = []
    for k in range(n):
        ref_log = run(caps[k].iq)
        log = eng.records_to_log(E, k, recs[k, :counts[k]], frames[k])
        fatal, nex, max_bits, ntr = bench.compare_with_reference(ref_log, log, False)
        lost = sum(1 for kk, _ in ref_log if kk == "lost_sync")
        out.append((6 * i + k, {kk: (v if kk != "chan" else repr(v)) for kk, v in kws[k].items()}, fatal[:4], nex, max_bits, ntr, sum(1 for kk, _ in ref_log if kk == "frame"), lost,
                    classify(ref_log, fatal) if fatal else None))
    ec._free_device(E, dev)
    E.close()
    return out


def work_batch_am(args):
    """--am --batch: four AM cs16 captures per engine through the window pipeline (the nine trellis passes of an L1 frame on a decode stream) with the L2 -> L1 feedback and
    replay on the device; a third of them with an interference burst that breaks a P1 PDU's first header."""
    i, seed0 = args
    from nrsc5_amd import synth_am, build, engine as eng
    from tests import engine_checks as ec
    import bench
    run, kind = bench._checker(1, True)
    assert kind == "reference"
    caps, kws = [], []
    for k in range(4):
        kw = params_am(4 * i + k, seed0)
        rng = np.random.default_rng(seed0 + 11 * (4 * i + k) + 5)
        if rng.integers(0, 3) == 0:
            kw["burst"] = (float(rng.uniform(5.5, 9.5)), float(rng.uniform(0.3, 0.6)), 40.0)      # (first L1 frame, frames, sigma): tests/engine_checks.py
            kw["n_frames"] = max(kw["n_frames"], 13)
        caps.append(synth_am.am_ma1_capture(**kw)); kws.append(kw)
    n = len(caps)
    stride = max(c.iq.size for c in caps); stride += (-stride) % 64
    E = eng.Engine(max_streams=n, q15_capacity=stride // 2 + 1024, record_capacity=1024, p1_slots=48, lib_path=(EMU_OVERRIDE or build.EMU_LIB), am_enable=True, p1_async=True, l2_feedback=True)
    for k in range(n):
        E.set_mode(k, eng.MODE_AM)
    buf = np.zeros((n, stride), dtype=np.int16)
    for k, c in enumerate(caps):
        buf[k, :c.iq.size] = c.iq
    dev = ec._to_device(E, buf)
    E.batch_append_cs16(dev, stride, [c.iq.size - c.iq.size % 4 for c in caps])
    E.batch_process(n)
    recs, counts, frames = E.batch_fetch_view(n)
    out = []
    for k in range(n):
        ref_log = run(caps[k].iq)
        log = eng.am_records_to_log(E, k, recs[k, :counts[k]], frames[k])
        fatal, nex, max_bits, ntr = bench.compare_with_reference(ref_log, log, True)
        lost = sum(1 for kk, _ in ref_log if kk == "lost_sync")
        out.append((4 * i + k, {kk: (v if kk != "chan" else repr(v)) for kk, v in kws[k].items()}, fatal[:4], nex, max_bits, ntr, sum(1 for kk, _ in ref_log if kk == "frame"), lost,
                    classify(ref_log, fatal) if fatal else None))
    ec._free_device(E, dev)
    E.close()
    return out


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
        E = eng.Engine(max_streams=1, q15_capacity=400000, record_capacity=512, p1_slots=16, lib_path=(EMU_OVERRIDE or build.EMU_LIB), am_enable=True, l2_feedback=True)
        E.set_mode(0, eng.MODE_AM)
        common.run_engine_streaming(E, 0, cap.iq, chunk=32768)
        log = eng.am_records_to_log(E, 0, E.drain(0))
        E.close()
    else:
        kw = params(i, seed0)
        cap = synth.fm_mp1_capture(0, **kw)
        ref_log = run(cap.iq)
        E, recs, log = ec.run_capture((EMU_OVERRIDE or build.EMU_LIB), cap, l2_feedback=True)
        E.close()
    fatal, nex, max_bits, ntr = bench.compare_with_reference(ref_log, log, AM)
    nframes = sum(1 for k, _ in ref_log if k in ("frame", "p3"))
    return i, {k: (v if k != "chan" else repr(v)) for k, v in kw.items()}, fatal[:4], nex, max_bits, ntr, nframes, 0, (classify(ref_log, fatal) if fatal else None)


if __name__ == "__main__":
    from multiprocessing import Pool
    from nrsc5_amd import build
    build.build_emu()
    nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
    t = time.time()
    with Pool(nproc) as p:
        if BATCH and AM:
            res = [r for rs in p.map(work_batch_am, [(i, seed0) for i in range(n // 4)], chunksize=1) for r in rs]
        elif BATCH:
            res = [r for rs in p.map(work_batch, [(i, seed0) for i in range(n // 6)], chunksize=1) for r in rs]
        else:
            res = p.map(work, [(i, seed0) for i in range(n)], chunksize=2)
    bad = [r for r in res if r[2]]
    out = {"captures": n, "seconds": round(time.time() - t, 1), "reference_frames_in_all": sum(r[6] for r in res), "captures_with_frames": sum(1 for r in res if r[6]),
           "captures_with_fatal_differences": len(bad), "captures_with_counted_transients": sum(1 for r in res if r[5]), "transient_fields": sum(r[5] for r in res),
           "frames_exempt_cber": sum(r[3] for r in res), "fatal": [(r[0], r[1], r[2]) for r in bad[:12]]}
    tot = {}
    for r in bad:
        for k, v in (r[8] or {}).items():
            tot[k] = tot.get(k, 0) + v
    out["what_the_fatal_differences_are"] = tot
    out["fatal_captures_classified"] = [(r[0], {k: v for k, v in (r[8] or {}).items() if v}) for r in bad]
    if BATCH:
        out["captures_with_lost_sync_in_the_reference"] = sum(1 for r in res if r[7]); out["lost_sync_events"] = sum(r[7] for r in res)
    print(json.dumps(out, indent=1, default=str))
