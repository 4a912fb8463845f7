#!/usr/bin/env python
"""AM half of BASELINE config 5 on one MI355X: N hybrid-AM MA1 streams resident in HBM (cs16 @46511.71875 S/s and/or
cu8 @1488375 S/s) through the batch API; reports IQ MS/s, x real-time, per-kernel-class device time and the fraction
of P1/P3 frames that equal the transmitted bits.  Prints one JSON line (side measurement; bench.py stays the FM metric)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the engine drives 1 main + up to 5 decode streams next to torch's: give each its own hardware queue (read at HIP init)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=256)
    ap.add_argument("--frames", type=int, default=41, help="L1 frames per stream (41 = 61 s)")
    ap.add_argument("--fmt", default="cs16", choices=["cs16", "cu8"])
    ap.add_argument("--in-order", action="store_true", help="decode every frame in order on the main stream (reference event timing)")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    args = ap.parse_args()
    import torch
    from nrsc5_amd import engine as eng, synth_am
    dev = torch.device("cuda", 0)
    S = args.streams
    cap = synth_am.am_ma1_capture(args.frames, seed=77, cfo_hz=4.0, offset=3000 * (32 if args.fmt == "cu8" else 1), fmt=args.fmt)
    fs = synth_am.FS_CS16 if args.fmt == "cs16" else synth_am.FS_CU8
    base = torch.from_numpy(cap.iq).to(dev)
    per = 4 if args.fmt == "cs16" else 64                       # stagger the streams' timing
    n = (cap.iq.size - 2 * per * S) // 4 * 4
    iq = torch.empty((S, n), dtype=base.dtype, device=dev)
    for k in range(S):
        iq[k] = base[2 * per * (k % 97): 2 * per * (k % 97) + n]
    torch.cuda.synchronize()
    nsamp = n / 2
    E = eng.Engine(max_streams=S, q15_capacity=int(nsamp / (1 if args.fmt == "cs16" else 32)) + 4096, record_capacity=8 * args.frames + 16,
                   p1_slots=args.frames, am_enable=True, p1_async=not args.in_order)
    for k in range(S):
        E.set_mode(k, eng.MODE_AM)
    sizes = np.full(S, n, dtype=np.uint32)

    def one_pass():
        E.reset_all()
        if args.fmt == "cs16":
            E.batch_append_cs16(iq.data_ptr(), n, sizes)
        else:
            E.batch_append_cu8(iq.data_ptr(), n, sizes)
        steps = E.batch_process(S)
        return steps, (E.batch_fetch(S) if args.in_order else E.batch_fetch_view(S))

    for _ in range(args.warmup):
        one_pass()
    E.profile(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        steps, (recs, counts, frames) = one_pass()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    prof = E.profile(0)
    # truth check on the last pass: every decoded P1 / P3 frame equals a transmitted frame
    truth1 = {np.packbits(b, bitorder="little").tobytes() for fr in cap.p1_frames for b in fr}
    truth3 = {np.packbits(b, bitorder="little").tobytes() for b in cap.p3_frames}
    n1 = ok1 = n3 = ok3 = 0
    for k in range(0, S, max(1, S // 16)):
        for r in recs[k, :counts[k]]:
            fl, slot, bc = int(r["flags"]), int(r["p1_slot"]), int(r["bc_decoded"])
            if fl & eng.REC_P1:
                n1 += 1
                w = frames[k, slot, bc * 118:(bc + 1) * 118]
                ok1 += np.packbits(eng.unpack_bits(w, 3750), bitorder="little").tobytes() in truth1
            if fl & eng.REC_P3:
                n3 += 1
                w = frames[k, slot, 944:944 + 750]
                ok3 += np.packbits(eng.unpack_bits(w, 24000), bitorder="little").tobytes() in truth3
    # CPU baseline beside it: the unmodified reference (oracle/_ref, SSE build) or the restatement, one stream on one host core
    cpu = None
    try:
        from oracle import ref, port
        one = np.ascontiguousarray(cap.iq[:n])
        if ref.available(sse=True):
            R = ref.RefLib(sse=True); run = lambda: R.run(one, mode=ref.MODE_AM); kind = "reference"
        else:
            O = port.Oracle(); run = lambda: O.run(one, mode=1); kind = "port"
        run()
        reps, t1 = 0, time.perf_counter()
        while time.perf_counter() - t1 < 8.0:
            run(); reps += 1
        dtc = (time.perf_counter() - t1) / reps
        cpu = {"value": round(nsamp / dtc / 1e6, 3), "unit": "IQ MS/s", "x_realtime": round(nsamp / fs / dtc, 1), "cores": 1, "kind": kind,
               "sample": f"{reps}x one {nsamp / fs:.0f}-s stream of this workload, 32768-byte pushes"}
    except Exception as ex:                                     # the checker is optional here
        cpu = {"error": str(ex)}
    out = {"metric": "AM IQ MS/s demodulated and decoded", "fmt": args.fmt, "streams": S, "seconds_per_stream": round(nsamp / fs, 2),
           "value": round(S * nsamp / dt / 1e6, 3), "x_realtime": round(S * nsamp / fs / dt, 1), "ms_per_pass": round(dt * 1e3, 2),
           "decode": "in-order" if args.in_order else "window pipeline", "block_steps": steps, "device_ms_per_pass": {k: round(v[0] / args.steps, 3) for k, v in prof.items() if v[1]},
           "launches_per_pass": {k: v[1] // args.steps for k, v in prof.items() if v[1]},
           "truth": {"p1_checked": n1, "p1_exact": int(ok1), "p3_checked": n3, "p3_exact": int(ok3)}, "cpu_baseline": cpu}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
