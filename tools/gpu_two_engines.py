"""Experiment: the block-step chain is two dependent kernels per step -- k_mixfft (VALU-heavy) and k_sync (one latency-bound workgroup per stream, ~12 % VALU) -- that never
overlap on one queue.  Two ENGINES with half the streams each, driven from two host threads, let one half's k_sync run beside the other half's k_mixfft.
Same captures, same engine options as the bench's fm workload (window pipeline, on-device L2 feedback, zero-copy), shortened to N_FRAMES L1 frames.
  GPU_MAX_HW_QUEUES=16 python tools/gpu_two_engines.py [groups ...]"""
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nrsc5_amd import engine as eng                     # noqa: E402
from nrsc5_amd import synth_torch as stt                # noqa: E402

S, N_FRAMES, PASSES = 256, int(os.environ.get("N_FRAMES", "6")), 6
dev = torch.device("cuda", 0)
pool = []
for p in range(8):
    p1, pids, m = stt.payload_stream(N_FRAMES, seed=900 + p)
    pool.append(stt.modulate(m, dev))
nsig, tail = pool[0].shape[0], 8640
stride = (2 * (4320 + nsig + stt.STRIDE_SLACK + tail) + 255) // 256 * 256
iq = torch.zeros((S, stride), dtype=torch.uint8, device=dev)
nbytes = np.zeros(S, dtype=np.uint32)
for k in range(S):
    out = stt.receive_cu8(pool[k % 8], stt.stream_params(k), tail=tail, out=iq[k])
    nbytes[k] = out.shape[0] - out.shape[0] % 4
torch.cuda.synchronize()


def make(n):
    return eng.Engine(max_streams=n, q15_capacity=2 * 71280, record_capacity=max(512, 2 * 16 * N_FRAMES + 64), p1_slots=N_FRAMES + 12, p1_async=True, l2_feedback=True, batch_zero_copy=True)


def one_pass(E, lo, n):
    E.reset_all()
    E.batch_append_cu8(iq[lo].data_ptr(), stride, nbytes[lo:lo + n])
    return E.batch_process(n)


def fetch(E, n):
    recs, counts, frames = E.batch_fetch_view(n)
    return [(recs[k, :counts[k]].tobytes(), frames[k].tobytes()) for k in range(n)]


results = {}
for groups in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 1]:
    n = S // groups
    engines = [make(n) for _ in range(groups)]
    times = []
    for ps in range(PASSES + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if groups == 1:
            one_pass(engines[0], 0, n)
        else:
            th = [threading.Thread(target=one_pass, args=(engines[g], g * n, n)) for g in range(groups)]
            for t in th: t.start()
            for t in th: t.join()
        torch.cuda.synchronize()
        if ps: times.append((time.perf_counter() - t0) * 1e3)
    got = [x for g in range(groups) for x in fetch(engines[g], n)]
    results[groups] = got
    same = (got == results[1]) if 1 in results else None
    print(f"groups={groups} streams/engine={n} ms per pass: median {np.median(times):.2f} min {min(times):.2f} max {max(times):.2f} | records+frames equal to one engine: {same}", flush=True)
    for E in engines: E.close()
