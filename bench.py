#!/usr/bin/env python
"""Headline benchmark: IQ MS/s demodulated AND decoded (x real-time) on MI355X.

A "step" = one complete pass of the hot path over one batch of synthetic captures that are already
resident in HBM: reset stream state -> K1 decimate -> per 32-symbol block {acquire | mix+FFT | sync /
equalise / soft-demod / PIDS Viterbi} -> per L1 frame {de-interleave, P1 Viterbi, BER, descramble} ->
D2H of every block record and decoded P1 frame.  Workload = BASELINE.json configs[2]:
`--streams` (default 256) independent hybrid-FM MP1 cu8 streams @1.488375 MS/s per GPU (weak scaling
for --gpus N: 256 per GPU, the configs[3] family).  configs[1] (one stream) is a parity-test case.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the engine drives 1 main + 3 decode streams next to torch's: give each its own hardware queue
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

FS = 1488375.0
ALG_BYTES_PER_SAMPLE = 2.008            # 2 B cu8 in + 18432 B decoded bits per 2211840-sample L1 frame
HBM_PEAK_GBPS = 8000.0                  # MI355X_MICROARCH.md: 8 TB/s spec
BLOCK_SAMPLES = 138240                  # cu8 complex samples per 32-symbol block
FRAME_SAMPLES = 16 * BLOCK_SAMPLES

# input samples one launch of each kernel class accounts for, per processed stream
SAMPLES_PER_LAUNCH = {"decimate": None, "acquire": BLOCK_SAMPLES, "prepare": BLOCK_SAMPLES, "mixfft": BLOCK_SAMPLES,
                      "sync": BLOCK_SAMPLES, "p1_deint": BLOCK_SAMPLES, "p1_viterbi": None, "pids": BLOCK_SAMPLES}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", default="fm", choices=["fm", "am-cs16", "am-cu8", "mixed"],
                    help="fm (default): the headline metric, configs[2]; the others run the side measurements of BASELINE "
                         "configs[4] (tools/gpu_am_bench.py, tools/gpu_mixed_bench.py; single GPU, their own JSON line)")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streams", type=int, default=256, help="streams per GPU")
    ap.add_argument("--seconds", type=float, default=20.0, help="capture length per stream (SURVEY 8d: 20 s)")
    ap.add_argument("--payloads", type=int, default=8, help="distinct transmissions shared by the streams (each stream has its own CFO/offset/noise)")
    ap.add_argument("--sync-p1", action="store_true", help="decode P1 frames in order on the main stream (exact reference event timing) instead of the overlapped window pipeline")
    ap.add_argument("--l2-feedback", type=int, default=1, help="1: the engine applies the reference's L2 -> L1 sync-loss feedback itself (RS check of the first L2 header on the device), as the CPU baseline's frame.c does; 0: off")
    ap.add_argument("--copy-input", action="store_true", help="decimate the captures into the engine's Q15 FIFO first (K1 as its own kernel) instead of reading them in place")
    ap.add_argument("--no-profile", action="store_true", help="diagnostic: no HIP-event kernel timing inside the timed region (roofline fields become 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--l2-index-inline", action="store_true", help="engine option l2_index: index every P1 frame on the decode streams inside the timed region (default: untimed post-pass)")
    ap.add_argument("--no-l2-index", action="store_true", help="skip the (untimed) L2 audio-index property check of the decoded frames")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0)
    ap.add_argument("--oracle-streams", type=int, default=4, help="parity block: how many falsely-locking streams of the last pass are compared with the oracle (plus half as many others)")
    ap.add_argument("--launch-check", action="store_true", help="only start the ranks, form the process group (RCCL; gloo without a GPU) and print what it sees -- no workload")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic_latest.json"),
                    help="optional PMC-derived HBM bytes per launch for the dominant kernel (written by profiles/collect_pmc.py)")
    return ap.parse_args()


def cpu_baseline(stream_iq: np.ndarray, budget_s: float):
    """Reference (oracle/_ref, the unmodified C sources, SSE build) or, if that build did not travel,
    the C restatement ('port'), single thread, fed in 32768-byte calls like src/main.c:1097-1120."""
    from oracle import ref, port
    kind = "port"
    runner = None
    if ref.available(sse=True):
        try:
            R = ref.RefLib(sse=True)
            runner = lambda iq: R.run(iq)
            kind = "reference"
        except OSError:
            runner = None
    if runner is None:
        O = port.Oracle()
        runner = lambda iq: O.run(iq)
    t0 = time.perf_counter()
    runner(stream_iq)
    dt1 = time.perf_counter() - t0
    reps = max(1, min(64, int(budget_s / max(dt1, 1e-3)) - 1))
    t0 = time.perf_counter()
    for _ in range(reps):
        runner(stream_iq)
    dt = (time.perf_counter() - t0) / reps
    nsamp = stream_iq.size / 2
    return {"value": round(nsamp / dt / 1e6, 3), "unit": "IQ MS/s", "x_realtime": round(nsamp / dt / FS, 2), "cores": 1,
            "kind": kind, "sample": f"{reps}x one {nsamp / FS:.1f}-s stream of this workload on 1 host core, 32768-byte pushes"}


def side_workload(args):
    """BASELINE configs[4] side measurements live in tools/ (same engine, same C ABI); dispatch to them."""
    import runpy
    if args.workload == "mixed":
        sys.argv = [os.path.join(ROOT, "tools", "gpu_mixed_bench.py"), "--steps", str(args.steps)]
    else:
        sys.argv = [os.path.join(ROOT, "tools", "gpu_am_bench.py"), "--fmt", args.workload.split("-")[1],
                    "--streams", str(256 if args.workload == "am-cs16" else 128), "--steps", str(args.steps), "--warmup", str(args.warmup)]
    runpy.run_path(sys.argv[0], run_name="__main__")


def launch_check(args):
    """--launch-check: the multi-rank start-up of --gpus N without the workload (runs on CPU with gloo too)."""
    import torch
    from nrsc5_amd import shard
    rank, world, local = shard.init_from_env(expect_world=args.gpus)
    cuda = torch.cuda.is_available()
    dev = torch.device("cuda", local) if cuda else torch.device("cpu")
    if cuda:
        torch.cuda.set_device(dev)
    shard.barrier(dev)
    ranks = shard.sum_over_ranks([1.0, float(rank)], dev)
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_in_process_group": int(ranks[0]), "rank_sum": int(ranks[1]),
                          "backend": "nccl(RCCL)" if cuda else "gloo", "launched_by": os.environ.get("NRSC5_BENCH_LAUNCHER", "external torchrun" if world > 1 else "single process")}))


def main():
    args = parse()
    from nrsc5_amd import shard
    if args.gpus > 1 and not shard.launched_by_torchrun():
        # `python bench.py --gpus N` on its own: start the N ranks here, the way the driver's torchrun line does
        sys.exit(shard.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus, {"NRSC5_BENCH_LAUNCHER": "bench.py"}))
    if args.launch_check:
        return launch_check(args)
    if args.workload != "fm":
        return side_workload(args)
    import torch
    from nrsc5_amd import engine as eng, synth_torch as stt

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (MI355X); there is no CPU fallback")
    if torch.cuda.device_count() < max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))):
        raise SystemExit(f"--gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible")
    rank, world, local = shard.init_from_env(expect_world=args.gpus)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    S = args.streams
    n_frames = max(2, int(np.ceil(args.seconds * FS / FRAME_SAMPLES)))
    my_streams = list(shard.stream_range(S * world, world, rank))

    # ---- synthetic captures, resident in HBM before timing starts -----------------------------------
    t_gen = time.perf_counter()
    pool = []
    for p in range(args.payloads):
        p1, pids, m = stt.payload_stream(n_frames, seed=p)
        pool.append((np.packbits(p1, axis=1, bitorder="little"), stt.modulate(m, dev)))
    nsig = pool[0][1].shape[0]
    tail = 8640
    stride = (2 * (4320 + nsig + tail) + 255) // 256 * 256
    iq = torch.zeros((S, stride), dtype=torch.uint8, device=dev)
    nbytes = np.zeros(S, dtype=np.uint32)
    params = []
    for k, gs in enumerate(my_streams):
        prm = stt.stream_params(gs)
        out = stt.channel_cu8(pool[gs % args.payloads][1], prm["cfo_hz"], prm["offset"], prm["snr_db"], prm["seed"], tail=tail, out=iq[k])
        nbytes[k] = out.shape[0] - out.shape[0] % 4
        params.append(prm)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    total_samples_rank = float(nbytes.astype(np.float64).sum() / 2)

    # replay (window pipeline + L2 feedback): blocks that ran behind a failed P1 frame keep their records / ring slots
    # (marked void, never delivered), so both rings carry head-room for the speculated stretch
    # zero-copy batch: the captures are read where they are (half-band fused into the symbol kernel): no decimated copy,
    # so the FIFO stays at its minimum size
    zero_copy = not args.copy_input
    E = eng.Engine(max_streams=S, q15_capacity=2 * 71280 if zero_copy else int(stride // 4 + 1024), record_capacity=max(512, 2 * 16 * n_frames + 64),
                   p1_slots=n_frames + 12, p1_async=not args.sync_p1, device=local, l2_feedback=bool(args.l2_feedback), l2_index=bool(args.l2_index_inline),
                   batch_zero_copy=zero_copy)

    host_ms = {"reset": 0.0, "append": 0.0, "process": 0.0, "fetch": 0.0}

    def one_pass(fetch=True):
        t = [time.perf_counter()]
        E.reset_all(); t.append(time.perf_counter())
        E.batch_append_cu8(iq.data_ptr(), stride, nbytes); t.append(time.perf_counter())
        steps = E.batch_process(S); t.append(time.perf_counter())
        out = E.batch_fetch_view(S) if fetch else None; t.append(time.perf_counter())
        for k, name in enumerate(("reset", "append", "process", "fetch")):
            host_ms[name] += (t[k + 1] - t[k]) * 1e3
        return steps, out

    for _ in range(args.warmup):
        one_pass()
    E.profile(0 if args.no_profile else 1)
    for k in host_ms:
        host_ms[k] = 0.0
    shard.barrier(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        block_steps, (recs, counts, frames) = one_pass()
    torch.cuda.synchronize()
    shard.barrier(dev)
    dt = time.perf_counter() - t0
    prof = E.profile(0)
    dt = shard.max_over_ranks(dt, dev)
    tot = shard.sum_over_ranks([total_samples_rank, 1.0], dev)
    total_samples, ranks_seen = float(tot[0]), int(tot[1])     # ranks_seen: counted through the collective itself

    # ---- verification of the last pass against the transmitted truth ----------------------------------
    rows = []
    n_locked = 0
    l2_jobs, l2_exact = [], []
    for k, gs in enumerate(my_streams):
        r = recs[k, :counts[k]]
        truth = pool[gs % args.payloads][0]
        p1r = r[(r["flags"] & eng.REC_P1) != 0]
        ok = 0
        h = 0
        first = None
        for j, rr in enumerate(p1r):
            w = frames[k, int(rr["p1_slot"])]
            h = zlib.crc32(w.tobytes(), h)
            b = w.view(np.uint8)
            exact = 0
            if first is None:
                match = np.nonzero((truth == b[None, :]).all(axis=1))[0]
                if match.size:
                    first = int(match[0]) - j
                    exact = 1
            else:
                idx = first + j
                exact = int(0 <= idx < truth.shape[0] and np.array_equal(truth[idx], b))
            ok += exact
            l2_jobs.append((k, int(rr["p1_slot"]), eng.L2_FM_P1, 0, eng.P1_BITS)); l2_exact.append(exact)
        fine = int((r["state_after"] == eng.SYNC_FINE).sum())
        locked = len(p1r) > 0 and ok >= len(p1r) - 1
        n_locked += int(locked)
        rows.append([gs, len(r), len(p1r), ok, int(((r["flags"] & eng.REC_PIDS) != 0).sum()), fine, h])
    allrows = shard.gather_summaries(np.array(rows, dtype=np.int64), dev)
    # ---- full-size property check on the device: the L2 audio index of every decoded P1 frame (frame_push + RS header +
    # CRC-8 of all 32 audio packets, k_l2_index) must be clean exactly for the frames that equal the transmitted bits
    l2 = None
    if l2_jobs and not args.no_l2_index:
        try:
            t_l2 = time.perf_counter()
            if args.l2_index_inline:
                ring = E.batch_fetch_l2(S)
                idx = [(eng.l2_frame_to_dict(ring[j[0]][j[1]]), None) for j in l2_jobs]
            else:
                idx = E.l2_index(l2_jobs, want_bytes=False)
            t_l2 = time.perf_counter() - t_l2
            clean = [int(d["n_pdu"] == 1 and d["pdus"][0]["nop"] == 32 and d["pdus"][0]["crc_bad_lo"] == 0 and d["lost_sync"] == 0) for d, _ in idx]
            l2 = {"where": "decode streams, inside the timed region" if args.l2_index_inline else "post-pass, untimed",
                  "frames_indexed": len(idx), "host_ms_incl_copies": round(t_l2 * 1e3, 2),
                  "audio_packets_crc_ok": int(sum(sum(p["nop"] - bin(p["crc_bad_lo"] | (p["crc_bad_hi"] << 32)).count("1") for p in d["pdus"]) for d, _ in idx)),
                  "frames_clean": int(sum(clean)), "clean_and_bit_exact": int(sum(c & e for c, e in zip(clean, l2_exact))),
                  "bit_exact": int(sum(l2_exact)), "frames_flagged_lost_sync": int(sum(d["lost_sync"] for d, _ in idx))}
        except Exception as ex:                       # informational, never allowed to take the bench line down
            l2 = {"error": repr(ex)}

    if rank != 0:
        return
    # ---- reference-equality of the benchmarked mode on a sample of this pass's streams (untimed checker leg) ----------
    # streams the reference algorithm first locks falsely on (their log has LOST_SYNC) + the first streams that did not:
    # the complete ordered log -- sync / lost-sync blocks, every PIDS and P1 frame, MER / BER / CFO -- against the oracle
    # driven by the restated frame_process decision
    ref_eq = None
    if not args.no_cpu_baseline and world == 1:
        try:
            from oracle import port
            from tests import common
            O = port.Oracle()
            lost = [k for k in range(len(my_streams)) if ((recs[k, :counts[k]]["flags"] & eng.REC_LOST_SYNC) != 0).any()]
            sample = lost[:args.oracle_streams] + [k for k in range(len(my_streams)) if k not in lost][:max(1, args.oracle_streams // 2)]
            t_or = time.perf_counter()
            equal, first_diffs = 0, []
            for k in sample:
                ol, _, _ = O.run(iq[k, :int(nbytes[k])].cpu().numpy(), p1_hook=O.l2_hook())
                log = eng.records_to_log(E, k, recs[k, :counts[k]], frames[k])
                diffs = common.compare_logs(common.strip_states(ol), common.strip_states(log))
                kept = [x for x in common.strip_states(ol) if x[0] not in ("hdc", "soft", "vit", "amsym", "pxsoft")]
                bad = {i for i, (kk, v) in enumerate(kept) if kk == "ber" and v["cber"] > 0.02}     # frames decoded while falsely locked: noise in, noise out
                diffs = [d for d in diffs if not any(d.startswith(f"#{i} ber") or d.startswith(f"#{i + 1} frame") for i in bad)]
                equal += not diffs
                if diffs and len(first_diffs) < 3:
                    first_diffs.append({"stream": int(my_streams[k]), "diff": diffs[0]})
            ref_eq = {"streams_with_lost_sync_this_pass": len(lost), "streams_checked": len(sample), "of_which_with_lost_sync": len([k for k in sample if k in lost]),
                      "logs_equal_to_oracle_with_l2_hook": int(equal), "first_diffs": first_diffs, "seconds": round(time.perf_counter() - t_or, 1),
                      "compared": "ordered log: state/sync/lost_sync blocks, PIDS + P1 frames bit-exact (frames with cber > 0.02 = decoded while falsely locked excepted), floats 1e-4"}
        except Exception as ex:
            ref_eq = {"error": repr(ex)}
    value = total_samples * args.steps / dt / 1e6
    # ---- roofline of the dominant kernel (by device time, HIP events on its launch stream) -----------
    blocks_rank = int(counts.sum())
    frames_rank = int(sum(row[2] for row in rows))
    dom = max(prof, key=lambda k: prof[k][0])
    dom_ms, dom_launches = prof[dom]
    per_pass_ms = {k: v[0] / args.steps for k, v in prof.items()}
    if dom == "decimate":
        dom_samples = total_samples_rank * args.steps
    elif dom == "p1_viterbi":
        dom_samples = frames_rank * FRAME_SAMPLES * args.steps
    else:
        dom_samples = blocks_rank * BLOCK_SAMPLES * args.steps
    alg_bytes_per_launch = dom_samples * ALG_BYTES_PER_SAMPLE / max(dom_launches, 1)
    avg_launch_s = dom_ms / 1e3 / max(dom_launches, 1)
    achieved = alg_bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
    traffic = None
    if os.path.exists(args.traffic_json):
        try:
            tj = json.load(open(args.traffic_json))
            if tj.get("kernel_class") == dom:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic,
                "avg_launch_ms": round(avg_launch_s * 1e3, 4), "launches": dom_launches,
                "alg_bytes_per_launch": int(alg_bytes_per_launch),
                "whole_path_GBps": round(value * ALG_BYTES_PER_SAMPLE / 1e3, 3),
                "device_ms_per_pass": {k: round(v, 3) for k, v in per_pass_ms.items()},
                "host_ms_per_pass": {k: round(v / args.steps, 3) for k, v in host_ms.items()}}
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        k0 = 0
        cpu = cpu_baseline(iq[k0, :int(nbytes[k0])].cpu().numpy(), args.cpu_baseline_seconds)
    n_total = allrows.shape[0]
    good = int(((allrows[:, 2] > 0) & (allrows[:, 3] >= allrows[:, 2] - 1)).sum())
    line = {
        "metric": "IQ MS/s demod+decoded", "value": round(value, 2), "unit": "IQ MS/s",
        "x_realtime": round(value * 1e6 / FS, 1), "n_gpus": world, "ranks_in_process_group": ranks_seen, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int16 Q15 front end / f32 OFDM+sync / int32 Viterbi metrics", "data": "synthetic",
        "config": {"workload": f"configs[2]: batch={S} independent hybrid-FM MP1 cu8 streams @1.488375 MS/s per GPU, "
                               f"{nbytes[0] / 2 / FS:.2f} s each ({n_frames} L1 frames), CFO +-300 Hz, offset [0,4320), SNR 15/20/25 dB",
                   "streams_per_gpu": S, "seconds_per_stream": round(float(nbytes[0]) / 2 / FS, 3),
                   "p1_decode": "in-order" if args.sync_p1 else "windowed-overlap", "l2_feedback": "on-device" if args.l2_feedback else "off", "input": "read in place (half-band fused into the symbol kernel)" if zero_copy else "decimated copy in the Q15 FIFO", "block_steps_per_pass": int(block_steps),
                   "distinct_payloads": args.payloads, "hbm_resident_input_GB": round(float(nbytes.sum()) / 1e9, 2)},
        "roofline": roofline, "cpu_baseline": cpu,
        "parity": {"streams": n_total, "streams_locked_and_all_p1_frames_equal_transmitted_bits": good,
                   "p1_frames_decoded": int(allrows[:, 2].sum()), "p1_frames_bit_exact_vs_truth": int(allrows[:, 3].sum()),
                   "pids_frames_decoded": int(allrows[:, 4].sum()), "reference_equality_rank0": ref_eq, "l2_index_rank0": l2,
                   "note": "streams whose timing offset falls in the reference algorithm's false-lock zone (sync.c phase-slope ambiguity, ~6 % of uniform offsets) decode one garbage frame in the reference too, whose L2 then forces a re-acquisition; with l2_feedback the engine does the same on the device: the verdict of the deferred decode rewinds the stream to the end of that frame's block (k_replay.hip), so LOST_SYNC and the re-acquisition land on the reference's blocks"},
        "gen_seconds": round(t_gen, 1),
    }
    print(json.dumps(line))


if __name__ == "__main__":
    main()
