#!/usr/bin/env python
"""BASELINE config 5 on one MI355X: a mixed batch of 128 hybrid-FM MP1 cu8 streams + 128 hybrid-AM MA1 streams (64 cs16
@46511.71875 S/s, 64 cu8 @1488375 S/s through the 32:1 cascade) in ONE engine, all captures resident in HBM.  Prints one
JSON line (side measurement; bench.py stays the FM headline metric)."""
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
FS = 1488375.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fm", type=int, default=128)
    ap.add_argument("--am", type=int, default=128)
    ap.add_argument("--seconds", type=float, default=20.0)
    ap.add_argument("--am-frames", type=int, default=41)
    ap.add_argument("--steps", type=int, default=2)
    args = ap.parse_args()
    import torch
    from nrsc5_amd import engine as eng, synth_am, synth_torch as stt
    dev = torch.device("cuda", 0)
    nfm, nam = args.fm, args.am
    n_frames = max(2, int(np.ceil(args.seconds * FS / (16 * 138240))))
    # FM half: the bench's generator
    pool = [stt.modulate(stt.payload_stream(n_frames, seed=p)[2], dev) for p in range(4)]
    tail = 8640
    stride_fm = (2 * (4320 + pool[0].shape[0] + tail) + 255) // 256 * 256
    fm = torch.zeros((nfm, stride_fm), dtype=torch.uint8, device=dev)
    fm_bytes = np.zeros(nfm, dtype=np.uint32)
    for k in range(nfm):
        prm = stt.stream_params(k)
        out = stt.channel_cu8(pool[k % 4], prm["cfo_hz"], prm["offset"], prm["snr_db"], prm["seed"], tail=tail, out=fm[k])
        fm_bytes[k] = out.shape[0] - out.shape[0] % 4
    # AM half: one MA1 transmission per format, streams staggered in time
    n16, n8 = nam // 2, nam - nam // 2
    c16 = synth_am.am_ma1_capture(args.am_frames, seed=77, cfo_hz=4.0, offset=3000, fmt="cs16")
    c8 = synth_am.am_ma1_capture(args.am_frames, seed=78, cfo_hz=-3.0, offset=3000 * 32, fmt="cu8")
    b16, b8 = torch.from_numpy(c16.iq).to(dev), torch.from_numpy(c8.iq).to(dev)
    len16 = (c16.iq.size - 8 * 97) // 4 * 4
    len8 = (c8.iq.size - 128 * 97) // 4 * 4
    am16 = torch.stack([b16[8 * (k % 97): 8 * (k % 97) + len16] for k in range(n16)])
    am8 = torch.stack([b8[128 * (k % 97): 128 * (k % 97) + len8] for k in range(n8)])
    torch.cuda.synchronize()
    S = nfm + nam
    cap = max(stride_fm // 4, len16 // 2, len8 // 64) + 4096
    E = eng.Engine(max_streams=S, q15_capacity=int(cap), record_capacity=max(16 * n_frames, 8 * args.am_frames) + 32,
                   p1_slots=max(n_frames + 1, args.am_frames), p1_async=True, am_enable=True, l2_feedback=True)
    ids_fm = np.arange(nfm, dtype=np.int32)
    ids16 = np.arange(nfm, nfm + n16, dtype=np.int32)
    ids8 = np.arange(nfm + n16, S, dtype=np.int32)
    for s in list(ids16) + list(ids8):
        E.set_mode(int(s), eng.MODE_AM)

    def one_pass():
        E.reset_all()
        E.batch_append_cu8(fm.data_ptr(), stride_fm, fm_bytes, stream_ids=ids_fm)
        E.batch_append_cs16(am16.data_ptr(), len16, np.full(n16, len16, dtype=np.uint32), stream_ids=ids16)
        E.batch_append_cu8(am8.data_ptr(), len8, np.full(n8, len8, dtype=np.uint32), stream_ids=ids8)
        steps = E.batch_process(S)
        return steps, E.batch_fetch_view(S)

    one_pass()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        steps, (recs, counts, frames) = one_pass()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    sec_fm = float(fm_bytes.sum()) / 2 / FS
    sec_am = n16 * (len16 / 2) / synth_am.FS_CS16 + n8 * (len8 / 2) / synth_am.FS_CU8
    samples = float(fm_bytes.sum()) / 2 + n16 * len16 / 2 + n8 * len8 / 2
    p1_fm = int(sum(((recs[k, :counts[k]]["flags"] & eng.REC_P1) != 0).sum() for k in range(nfm)))
    p1_am = int(sum(((recs[k, :counts[k]]["flags"] & eng.REC_P1) != 0).sum() for k in range(nfm, S)))
    print(json.dumps({"metric": "mixed FM+AM batch, IQ MS/s demodulated and decoded", "config": "configs[4]: 128 FM cu8 + 64 AM cs16 + 64 AM cu8",
                      "value": round(samples / dt / 1e6, 1), "signal_seconds_per_pass": round(sec_fm + sec_am, 1),
                      "x_realtime": round((sec_fm + sec_am) / dt, 1), "ms_per_pass": round(dt * 1e3, 2),
                      "fm_streams": nfm, "fm_seconds_each": round(sec_fm / nfm, 2), "am_streams": nam, "am_seconds_each": round(sec_am / nam, 2),
                      "p1_frames_fm": p1_fm, "p1_frames_am": p1_am}))


if __name__ == "__main__":
    main()
