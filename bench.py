#!/usr/bin/env python
"""Headline benchmark: IQ MS/s demodulated AND decoded (x real-time) on MI355X.

A "step" = one complete pass of the hot path over one batch of synthetic captures that are already resident in HBM:
reset stream state -> per 32-symbol block {acquire | half-band + mix + FFT | sync / equalise / soft-demod} -> per L1 frame
{de-interleave, Viterbi, BER, descramble, first-header RS check} -> D2H of every block record and decoded frame.

Workloads (`--workload`), all with the same JSON schema:
  fm        BASELINE.json configs[2] (default, the headline metric): `--streams` (256) independent hybrid-FM MP1 cu8 streams
            @1.488375 MS/s per GPU; with --gpus N the configs[3] family: 256 per GPU (weak scaling, default) or
            `--scaling strong --total-streams 2048` (a fixed set of streams split over the ranks)
  am-cs16   configs[4], AM half: 256 hybrid-AM MA1 cs16 streams @46511.71875 S/s, 61 s each
  am-cu8    configs[4], AM half through the 5-stage 32:1 decimator: 128 MA1 cu8 streams @1.488375 MS/s
  mixed     configs[4]: 128 FM cu8 + 64 AM cs16 + 64 AM cu8 streams in one engine
The default fm line also carries: configs[1] (one FM stream, `single_stream`), the in-order (reference event timing) figure of
the same batch (`in_order`), compact configs[4] legs (`config4`: am-cs16 and mixed, 3 passes each) and the DROP-IN as a user of
`nrsc5 -r` sees it (`dropin`: the reference's public pipe API on libnrsc5_hipdropin.so vs the plain reference).

Parity is enforced, not reported: every stream that lost sync in the last pass plus `--oracle-streams` others is compared with
the UNMODIFIED reference (oracle/_ref, its own L2) -- complete ordered log, frames bit-exact -- and the process exits non-zero
when any compared log differs or a checker leg raises (the JSON line is still printed, with the failure in it).

`python bench.py --gpus N` starts its N ranks itself (one process per GPU, torch.distributed over RCCL) unless it already
runs under torchrun.  Prints ONE JSON line on rank 0 with `roofline` and `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the engine drives 1 main + 3-4 decode streams next to torch's: give each its own hardware queue
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# dmabuf IPC: RCCL across processes needs it on this driver -- also when an external torchrun started the ranks without it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

FS = 1488375.0
FS_AM = 46511.71875
HBM_PEAK_GBPS = 8000.0                  # MI355X_MICROARCH.md: 8 TB/s spec
N_SIMD = 1024                           # 256 CUs x 4 SIMDs
BLOCK_SAMPLES = 138240                  # cu8 complex samples per 32-symbol FM block
FRAME_SAMPLES = 16 * BLOCK_SAMPLES
# algorithmic bytes per input complex sample: input bytes + packed decoded bits
ALG_FM_CU8 = 2.0 + 18432.0 / FRAME_SAMPLES                      # 2.008
ALG_AM_CS16 = 4.0 + 6830.0 / 69120.0                            # 4.10
ALG_AM_CU8 = 2.0 + 6830.0 / (69120.0 * 32)                      # 2.003
FAILURES = []                            # parity / checker failures: the line is printed, then the process exits 3


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", default="fm", choices=["fm", "am-cs16", "am-cu8", "mixed"])
    ap.add_argument("--steps", type=int, default=10, help="timed passes (SURVEY 8d: median of >= 10 next to the mean)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=0, help="streams per GPU (default: 256; am-cu8: 128)")
    ap.add_argument("--stream-base", type=int, default=0, help="first global stream id (a stream's CFO / offset / noise / channel are seeded by its id: another base = another batch)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak", help="--gpus N: weak = --streams per GPU; strong = --total-streams split over the ranks (configs[3]: 2048)")
    ap.add_argument("--total-streams", type=int, default=2048, help="--scaling strong: streams of the whole job")
    ap.add_argument("--seconds", type=float, default=20.0, help="FM capture length per stream (SURVEY 8d: 20 s)")
    ap.add_argument("--am-frames", type=int, default=41, help="AM L1 frames per stream (41 = 61 s)")
    ap.add_argument("--payloads", type=int, default=64, help="distinct FM transmissions shared by the streams (each stream has its own CFO/offset/noise)")
    ap.add_argument("--sync-p1", action="store_true", help="decode frames in order on the main stream (reference event timing) instead of the overlapped window pipeline")
    ap.add_argument("--l2-feedback", type=int, default=1, help="1: the engine applies the reference's L2 -> L1 sync-loss feedback itself (RS check of the first L2 header on the device), as the CPU baseline's frame.c does; 0: off")
    ap.add_argument("--copy-input", action="store_true", help="fm: decimate the captures into the engine's Q15 FIFO first (K1 as its own kernel) instead of reading them in place")
    ap.add_argument("--no-profile", action="store_true", help="diagnostic: no HIP-event kernel timing inside the timed region (roofline fields become 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip every checker leg that runs on the host cores (CPU baseline, reference equality)")
    ap.add_argument("--no-extra-legs", action="store_true", help="fm: skip the single-stream (configs[1]), in-order, configs[4] and drop-in legs")
    ap.add_argument("--l2-index-inline", action="store_true", help="fm: engine option l2_index: index every P1 frame on the decode streams inside the timed region (default: untimed post-pass)")
    ap.add_argument("--no-l2-index", action="store_true", help="fm: skip the (untimed) L2 audio-index property check of the decoded frames")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-processes", type=int, default=0, help="processes of the N-core CPU aggregate (default: all host cores, at most 64)")
    ap.add_argument("--oracle-streams", type=int, default=-1, help="parity block: streams that did NOT lose sync compared with the reference (every stream that lost sync is compared); -1 (default) = ALL streams")
    ap.add_argument("--parity-processes", type=int, default=0, help="host processes of the parity checker (default: host cores / ranks on this node, at most 64)")
    ap.add_argument("--oracle-lost-max", type=int, default=64, help="parity block: upper bound on the lost-sync streams compared (reported when it bites)")
    ap.add_argument("--tune", action="append", default=[], metavar="KNOB=VALUE", help="nrsc5hip_debug_tune before the run: decode_streams / am_decode_streams = 1..5, fwd_segments = 0..16")
    ap.add_argument("--force-process-group", action="store_true", help="with --launch-check: form a process group for a single rank too (a one-rank RCCL group on a 1-GPU box)")
    ap.add_argument("--launch-check", action="store_true", help="only start the ranks, form the process group (RCCL; gloo without a GPU) and print what it sees -- no workload")
    ap.add_argument("--ingest", choices=("local", "scatter"), default="local",
                    help="fm, --gpus N: local = every rank synthesises its own captures; scatter = rank 0 synthesises all of them and sends each rank its shard (RCCL point-to-point, before the timed region)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic_latest.json"),
                    help="PMC-derived HBM bytes (profiles/collect_pmc.py); used only if it was collected from THIS source tree")
    ap.add_argument("--sq-json", default=os.path.join(ROOT, "profiles", "sq_latest.json"),
                    help="PMC-derived VALU issue statistics (profiles/collect_sq.py); used only if collected from THIS source tree")
    args = ap.parse_args(argv)
    if args.ingest == "scatter" and args.workload != "fm":
        ap.error("--ingest scatter is implemented for --workload fm only")
    return args


# ---- checker legs: the unmodified reference (oracle/_ref, SSE build, its own L2) or, where that is absent, the restatement -----
def _checker(mode: int, with_l2: bool):
    """(run(iq) -> ordered log, kind).  kind "reference": oracle/_ref/libnrsc5_ref_sse.so = the reference's own translation units
    incl. frame.c, so the L2 -> L1 sync-loss feedback is the reference's; kind "port": oracle/ restatement (+ restated decision)."""
    from oracle import ref, port
    if ref.available(sse=True):
        try:
            R = ref.RefLib(sse=True)
            return (lambda iq: R.run(iq, mode=mode)[0]), "reference"
        except OSError:
            pass
    O = port.Oracle()
    if with_l2:
        return (lambda iq: O.run(iq, mode=mode, p1_hook=O.l2_hook())[0]), "port"
    return (lambda iq: O.run(iq, mode=mode)[0]), "port"


def _cpu_worker(path: str, mode: int, reps: int, q):
    sys.path.insert(0, ROOT)
    iq = np.load(path, mmap_mode="r")
    run, _ = _checker(mode, False)
    iq = np.ascontiguousarray(iq)
    t0 = time.perf_counter()
    for _ in range(reps):
        run(iq)
    q.put(time.perf_counter() - t0)


def cpu_baseline(stream_iq: np.ndarray, fs: float, mode: int, budget_s: float, nproc: int):
    """One stream of the workload through the CPU path, fed in 32768-byte pushes like src/main.c:1097-1120:
    (a) one host core, (b) one process per host core, each with its own session and the same stream (aggregate)."""
    import multiprocessing as mp
    import tempfile
    run, kind = _checker(mode, False)
    t0 = time.perf_counter(); run(stream_iq); dt1 = time.perf_counter() - t0
    reps = max(1, min(64, int(0.5 * budget_s / max(dt1, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(reps):
        run(stream_iq)
    dt = (time.perf_counter() - t0) / reps
    nsamp = stream_iq.size / 2
    out = {"value": round(nsamp / dt / 1e6, 3), "unit": "IQ MS/s", "x_realtime": round(nsamp / dt / fs, 2), "cores": 1, "kind": kind,
           "fft": "oracle/cpu_fft.c (radix-4 Stockham shim; fftw3f is not installed on this box)",
           "sample": f"{reps}x one {nsamp / fs:.1f}-s stream of this workload on 1 host core, 32768-byte pushes"}
    ncpu = os.cpu_count() or 1
    n = nproc if nproc > 0 else min(ncpu, 64)
    if n > 1:
        try:
            fd, path = tempfile.mkstemp(suffix=".npy", dir="/tmp"); os.close(fd)
            np.save(path, stream_iq)
            per = max(1, min(4, int(0.5 * budget_s / max(dt, 1e-3))))      # contention (shared caches, SMT) stretches these passes 3-4x
            ctx = mp.get_context("spawn")
            q = ctx.Queue()
            procs = [ctx.Process(target=_cpu_worker, args=(path, mode, per, q)) for _ in range(n)]
            t0 = time.perf_counter()
            for p in procs:
                p.start()
            times = [q.get(timeout=600) for _ in procs]
            for p in procs:
                p.join(timeout=60)
            wall = time.perf_counter() - t0               # includes process start-up: a lower bound on the aggregate
            os.unlink(path)
            out["all_cores"] = {"value": round(n * per * nsamp / max(times) / 1e6, 2), "unit": "IQ MS/s", "x_realtime": round(n * per * nsamp / max(times) / fs, 1),
                                "cores": n, "host_cores": ncpu, "sample": f"{n} processes x {per} passes of that stream, one session each; slowest process {max(times):.1f} s (wall incl. start-up {wall:.1f} s)"}
        except Exception as ex:                           # the aggregate is a side figure of a baseline: reported, not fatal
            out["all_cores"] = {"error": repr(ex)}
    return out


_DIFF_RE = re.compile(r"#(\d+) (\w+)\.(\w+): (?:(\d+) elements differ)?(?:expected (\S+) got (\S+))?")
# Loop-internal state that is NOT an estimate north_star names: the tracking loop's residual error signal, the NCO phase (an
# integrator) and the integer timing pick of a block (sub-sample timing is tracked by the phase slope).  The symbol kernel evaluates the
# NCO phase in closed form; the reference advances it by a float recurrence whose rounding drift (~3e-6 rad per symbol) is part of what it
# hands to its FFT (measured in round 4 to be the trigger -- not the FFT: profiles/r04_cfo_lock_transients.txt).  On a WEAK edge reference
# carrier that difference is a few 1e-4 relative, and the CFO search (sync.c:292-337: three garbage-tracking Costas passes over that bin at phases of ~1000 rad)
# amplifies it chaotically: for a few blocks after a lock with integer CFO != 0 that one carrier's loop state differs, and about
# once in 30 000 blocks a float lands within rounding distance of the threshold of roundf() (sync.c:455).  Frames, events, the CFO
# estimates (freq_offset, prev_angle), MER and BER are unaffected and stay under the strict rule; these fields are COUNTED when they
# deviate (a timing pick by at most 1 sample, the floats by at most TRANSIENT_ABS) instead of failing the run.
# In the 16 blocks after such a lock the partition next to that carrier is equalised with the deviating reference: the first MER report
# (an average over those blocks) may differ by a few tenths of a dB and prev_angle by a few 1e-4 relative -- counted the same way, only
# within AFTER_LOCK records of a SYNC event.  (Measured on the 256-stream batch, all streams compared: 250 streams equal under the strict
# rule, 6 with transient deviations -- 4 roundf() flips, 2 CFO-search locks: MER 0.24 dB, prev_angle 3.6e-4 -- 0 with anything else.)
TRANSIENT_INT = {"samperr", "keep", "next_samperr"}
# round 5: next_angle 2e-3 (5e-3 in round 4), NCO phase 5e-2 (unchanged): the largest deviations among 2400 CFO-search locks on the CPU twin are 1.7e-3 / 4.3e-2 (ONE capture, #1903;
# the next largest 1.7e-4 / 4.4e-3), among ~800 on the MI355X 1.2e-4 / 3.2e-3 -- a heavy tail: the bounds are set just above the largest counted member, not at twice the typical one
TRANSIENT_ABS = {"next_angle": 2e-3, "phase_re": 5e-2, "phase_im": 5e-2}
TRANSIENT_DETAILS = []                # the first deviations counted as transient, verbatim (per process)
MER_EXEMPT_BUDGET_PCT, MER_NOISE_BUDGET_PCT_X10 = 1, 5      # 1 % / 0.5 % of the MER values of the compared streams
TRANSIENT_STREAM_BUDGET_PCT = 2      # round 6 (exact loop arithmetic while un-synchronised + the first block's oscillator by the reference's recurrence): 0 of 256 in the bench batch, 0 of 768 in three fresh CFO-search batches, 2 of 256 in tests/test_gpu_batch256.py's (round 5: 6 of 256, budget 5 %)
# round 5: 0.2 dB (0.5 in round 4): with the oscillator's amplitude and the exact first block on the device the largest first-MER deviation of a counted lock is
# 0.078 dB in 2400 CFO-search locks on the CPU twin and 0.069 dB on the MI355X; the tail beyond (one lock in ~500 on the device: 0.57 dB, a timing pick by 3 samples) FAILS the run
# round 6: 0.05 dB at first (0.2 in round 5): with the Costas loops and the CFO search on the reference's own operations (NRSC5HIP_TUNE_LOOP_EXACT, default) the largest first-MER deviation
# of 476 CFO-search locks on the MI355X is 0.015 dB (a sideband at 0.47 dB MER: profiles/r06_cfo_batch_gpu_loop_exact_policy0.txt); round 5's members (0.069 dB; 0.57 dB outside) are gone
# ... and 0.1 dB after fifteen slices of the GPU fuzz (3840 fresh streams, ~3600 CFO-search locks: profiles/r06_gpu_fuzz_and_2048.txt): ONE lock's first report is 0.084 dB off
# (stream 103547: the loop state of its lock block deviates by 0.5 %), two more by 0.012 / 0.010 dB
AFTER_LOCK, AFTER_LOCK_ABS, AFTER_LOCK_REL = 40, {"lower": 0.1, "upper": 0.1}, {"prev_angle": 1e-3}


def compare_with_reference(ref_log, got_log, am: bool):
    """Complete ordered log of one stream against the checker's.  Frames the reference decodes while falsely locked (its own
    BER estimate cber > 0.02: Viterbi output on noise, which fails its L2 header on both sides and produces no HDC) are the one
    documented exemption from bit-exactness: their bits and BER are compared loosely and COUNTED here instead of being filtered
    silently.  -> (fatal diffs, exempt frames, largest number of differing bits in an exempt frame, transient loop-state deviations)"""
    from tests import common
    exp, got = common.strip_states(ref_log), common.strip_states(got_log)
    diffs = common.compare_logs(exp, got)
    kept = [x for x in exp if x[0] not in ("hdc", "soft", "vit", "amsym", "pxsoft", "station")]
    bad = {i for i, (k, v) in enumerate(kept) if k == "ber" and v["cber"] > 0.02}
    # AM: the BER of an L1 frame is reported once, after its P3 frame (block 7; decode.c:507-554), and covers the whole L1 frame: the eight P1 PDUs delivered in the blocks before it
    # were demodulated from the same signal.  A PDU of an L1 frame whose own BER estimate is above 0.02 is exempt like the P3 frame beside it (round 6: stream 52981 of the GPU
    # fuzz -- a PDU whose first L2 header fails in the reference and here alike, LOST_SYNC on the same block, 69 bits of a 40-bit-wide burst apart: one hard QAM decision on a
    # boundary; the record sits one LOST_SYNC away from the BER and the old index rule missed it)
    am_bad_frames = set()
    if am:
        nxt_ber = None
        for i in range(len(kept) - 1, -1, -1):
            if kept[i][0] == "ber":
                nxt_ber = i
            elif kept[i][0] == "frame" and nxt_ber is not None and nxt_ber in bad and nxt_ber - i <= 40:
                am_bad_frames.add(i)
    remaining, max_bits, transient = [], 0, 0
    global TRANSIENT_DETAILS
    syncs = [i for i, (k, _) in enumerate(kept) if k == "sync"]
    for d in diffs:
        m = _DIFF_RE.match(d)
        if m:
            idx, kind, field = int(m.group(1)), m.group(2), m.group(3)
            # FM: "ber" then the frame; AM: the P3 frame then "ber" (after block 7)
            if (kind == "ber" and idx in bad) or (kind == "frame" and ((idx - 1) in bad or (am and ((idx + 1) in bad or idx in am_bad_frames)))):
                if m.group(4):
                    max_bits = max(max_bits, int(m.group(4)))
                continue
            if kind in ("block", "mer") and m.group(5) is not None:
                try:
                    a, b = float(m.group(5)), float(m.group(6))
                    if kind == "block" and ((field in TRANSIENT_INT and abs(a - b) <= 1) or (field in TRANSIENT_ABS and abs(a - b) <= TRANSIENT_ABS[field])):
                        transient += 1
                        if len(TRANSIENT_DETAILS) < 40: TRANSIENT_DETAILS.append(d)
                        continue
                    if any(0 <= idx - j <= AFTER_LOCK for j in syncs) and (abs(a - b) <= AFTER_LOCK_ABS.get(field, -1.0) or abs(a - b) <= AFTER_LOCK_REL.get(field, -1.0) * abs(a)):
                        transient += 1
                        if len(TRANSIENT_DETAILS) < 40: TRANSIENT_DETAILS.append("after lock: " + d)
                        continue
                except ValueError:
                    pass
        remaining.append(d)
    return remaining, len(bad), max_bits, transient


# ---- the checker fanned out over the host cores ---------------------------------------------------------------------------------------
# One reference session per process (`spawn`: the parent holds a HIP context).  A task = (capture of one stream, written to a scratch file the
# worker unlinks once it is in memory; the engine's ordered log of that stream); a result = compare_with_reference's verdict.  Round 4 ran the
# checker in-process on a deterministic sample of 80 of the 256 streams (20 - 35 s) -- and the one stream of the batch that failed was not in the
# sample.  Comparing EVERY stream costs 256 x 0.34 s of host time: seconds on the box's cores.
class ParityPool:
    def __init__(self, nproc: int):
        import multiprocessing as mp
        import tempfile
        self.ctx = mp.get_context("spawn")
        self.tasks, self.results = self.ctx.Queue(), self.ctx.Queue()
        self.dir = tempfile.mkdtemp(prefix="nrsc5_parity_", dir=_scratch_dir())
        self.procs = [self.ctx.Process(target=_parity_worker, args=(self.tasks, self.results), daemon=True) for _ in range(nproc)]
        for p in self.procs:
            p.start()
        self.n = nproc

    def run(self, jobs):
        """jobs: iterable of (key, iq ndarray, am, got_log) produced lazily (a capture is 62 MB: at most 2 x nproc files wait at a time).
        -> {key: result tuple}"""
        out, pending = {}, 0
        def take():
            nonlocal pending
            import queue
            waited = 0.0
            while True:
                try:
                    r = self.results.get(timeout=5.0)
                    break
                except queue.Empty:
                    waited += 5.0
                    if not any(p.is_alive() for p in self.procs):
                        raise RuntimeError("parity checker: every worker process has exited with results outstanding")
                    if waited > 900:
                        raise RuntimeError("parity checker: no result for 900 s")
            out[r[0]] = r
            pending -= 1
        for key, iq, am, got_log in jobs:
            path = os.path.join(self.dir, f"{key}.npy")
            np.save(path, np.ascontiguousarray(iq))
            self.tasks.put((key, path, bool(am), got_log))
            pending += 1
            while pending >= 2 * self.n:
                take()
        while pending:
            take()
        return out

    def close(self):
        import shutil
        for _ in self.procs:
            self.tasks.put(None)
        for p in self.procs:
            p.join(timeout=20)
            if p.is_alive():
                p.terminate()
        shutil.rmtree(self.dir, ignore_errors=True)


def _scratch_dir():
    """/dev/shm when it has room for the files in flight (128 x 62 MB), else /tmp"""
    import shutil
    try:
        if shutil.disk_usage("/dev/shm").free > 12 << 30:
            return "/dev/shm"
    except OSError:
        pass
    return "/tmp"


def _parity_worker(tasks, results):
    sys.path.insert(0, ROOT)
    from tests import common
    checkers = {}
    while True:
        t = tasks.get()
        if t is None:
            return
        key, path, am, got_log = t
        try:
            iq = np.load(path)
            os.unlink(path)
            if am not in checkers:
                checkers[am] = _checker(1 if am else 0, True)
            run, kind = checkers[am]
            ref_log = run(iq)
            del TRANSIENT_DETAILS[:]
            e0, n0 = common.EXEMPT["mer_within_0.01dB"], common.EXEMPT["mer_below_0dB_within_0.1dB"]
            diffs, nex, mb, ntr = compare_with_reference(ref_log, got_log, am)
            results.put((key, kind, diffs, nex, mb, ntr, (int(common.EXEMPT["mer_within_0.01dB"] - e0), int(common.EXEMPT["mer_below_0dB_within_0.1dB"] - n0), 2 * sum(1 for k_, _ in ref_log if k_ == "mer")), list(TRANSIENT_DETAILS)))
        except Exception as ex:                                     # a checker that raises is a failure of the run, never a silent skip
            results.put((key, "error", [f"checker raised {ex!r}"], 0, 0, 0, (0, 0, 0), []))


_POOL = None


def parity_pool(args):
    """created on first use, shared by every parity block of the run (fm, am-cs16, mixed / fm, mixed / am)"""
    global _POOL
    if _POOL is None:
        import atexit
        world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
        n = args.parity_processes or max(1, min(64, (os.cpu_count() or 1) // world))
        _POOL = ParityPool(n)
        atexit.register(_POOL.close)
    return _POOL


def _diff_class(d: str) -> str:
    """what kind of record a fatal difference sits in: a decoded-frame or event difference is the thing north_star calls bit-exact,
    a timing pick off by more than one sample or a float beyond its bound is a tracking difference"""
    m = _DIFF_RE.match(d)
    if not m:
        return "log_structure"
    kind, field = m.group(2), m.group(3)
    if kind == "frame":
        return "p1_px_frame_bits"
    if kind == "pids":
        return "pids_frame_bits"
    if kind == "block" and field in TRANSIENT_INT:
        return "timing_pick_beyond_1_sample"
    if kind in ("sync", "mer", "ber", "block"):
        return f"{kind}_{field}"
    return kind


def _maybe_broken(W, k, log):
    """TEST HOOK (tests/test_gpu_two_ranks.py): NRSC5_BENCH_BREAK_STREAM=<rank>:<local stream> flips one bit of that stream's first decoded frame
    (PIDS or P1) in the DEVICE log before the comparison -- the run must then fail, and say on which rank."""
    spec = os.environ.get("NRSC5_BENCH_BREAK_STREAM")
    if not spec:
        return log
    r, ks = spec.split(":")
    if int(r) != int(os.environ.get("RANK", "0")) or int(ks) != int(k):
        return log
    for kind, v in log:
        if kind in ("pids", "frame"):
            bits = np.array(v["bits"], copy=True)
            bits[3] ^= 1
            v["bits"] = bits
            break
    return log


def reference_equality(W, recs, counts, frames, to_log, am: bool):
    """EVERY stream of the workload (--oracle-streams -1, the default) -- or every stream that lost sync in the last pass (bounded by
    --oracle-lost-max) + --oracle-streams others -- against the checker, one reference session per host process."""
    eng, a = W.eng, W.args
    lost = [k for k in W.checkable if ((recs[k, :counts[k]]["flags"] & eng.REC_LOST_SYNC) != 0).any()]
    others = [k for k in W.checkable if k not in set(lost)]
    if a.oracle_streams < 0 or a.oracle_streams >= len(others):
        pick, lost_checked = others, lost
    else:
        # a sample: spread the others over the batch (different CFO / offset / SNR classes); half of them from the streams with an
        # impaired channel (sample-clock error + echoes / analog host / fading: synth_torch.stream_params, am_stream_params)
        imp = [k for k in others if W.impaired(k)]
        clean = [k for k in others if not W.impaired(k)]
        n_imp = min(len(imp), a.oracle_streams // 2)
        n_clean = min(len(clean), a.oracle_streams - n_imp)
        spread = lambda xs, n: [xs[int(i * len(xs) / n)] for i in range(n)] if n else []
        pick = spread(imp, n_imp) + spread(clean, n_clean)
        lost_checked = lost[:a.oracle_lost_max]
    t0 = time.perf_counter()
    pool = parity_pool(a)
    todo = lost_checked + pick
    res = pool.run((k, W.stream_iq(k), am, _maybe_broken(W, k, to_log(k, recs[k, :counts[k]], frames[k]))) for k in todo)
    eq_lost = eq_other = strict = exempt = max_bits = tr_streams = tr_fields = imp_checked = imp_equal = mer_exempt = mer_noise = mer_values = 0
    first_diffs, tr_details, classes, kind = [], [], {}, "reference"
    lost_set = set(lost_checked)
    for k in todo:
        _, knd, diffs, nex, mb, ntr, nmer, details = res[k]
        if knd != "error":
            kind = knd
        exempt += nex; max_bits = max(max_bits, mb); tr_streams += ntr > 0; tr_fields += ntr; strict += (not diffs and ntr == 0); mer_exempt += nmer[0]; mer_noise += nmer[1]; mer_values += nmer[2]
        imp_checked += W.impaired(k); imp_equal += (W.impaired(k) and not diffs)
        tr_details += [f"stream {int(W.my_streams[k])}: {d}" for d in details[:3]]
        if not diffs:
            if k in lost_set:
                eq_lost += 1
            else:
                eq_other += 1
        else:
            for c in {_diff_class(d) for d in diffs}:
                classes[c] = classes.get(c, 0) + 1
            if len(first_diffs) < 6:
                first_diffs.append({"stream": int(W.my_streams[k]), "diffs": diffs[:3]})
    out = {"kind": kind, "checker": "oracle/_ref/libnrsc5_ref_sse.so: the unmodified reference incl. its L2 (frame.c)" if kind == "reference" else "oracle/ restatement + restated frame_process decision (oracle/_ref not present)",
           "streams": len(W.checkable), "streams_compared": len(todo), "all_streams_compared": len(todo) == len(W.checkable), "checker_processes": pool.n,
           "streams_with_lost_sync_this_pass": len(lost), "lost_sync_streams_checked": len(lost_checked), "lost_sync_streams_equal": eq_lost,
           "other_streams_checked": len(pick), "other_streams_equal": eq_other,
           "impaired_channel_streams_checked": int(imp_checked), "impaired_channel_streams_equal": int(imp_equal),
           "mer_values_compared": int(mer_values), "mer_reports_beyond_1e-4_within_0.01dB": int(mer_exempt), "mer_reports_below_0dB_within_0.1dB": int(mer_noise),
           "mer_exemption_budgets": {"within_0.01dB": max(8, mer_values * MER_EXEMPT_BUDGET_PCT // 100), "below_0dB_within_0.1dB": max(4, mer_values * MER_NOISE_BUDGET_PCT_X10 // 1000)},
           "frames_exempt_cber": exempt, "exempt_max_bit_differences": max_bits,
           "streams_equal_under_the_strict_rule": strict,
           "streams_with_transient_loop_state_deviation": tr_streams, "transient_loop_state_fields": tr_fields,
           "streams_failing_by_class": classes,
           "first_diffs": first_diffs, "transient_details": tr_details[:12], "seconds": round(time.perf_counter() - t0, 1),
           "compared": "complete ordered log: sync / lost-sync blocks, every PIDS / P1 (/ P3) frame bit-exact, integers exact, floats 1e-4 (tests/common.py). Exemptions, all COUNTED above: "
                       "frames_exempt_cber = frames the reference itself decodes while falsely locked (cber > 0.02, no HDC on either side): bits and BER compared loosely; "
                       f"transient_loop_state = block fields samperr / keep / next_samperr off by at most 1 sample, next_angle by at most {TRANSIENT_ABS['next_angle']:g}, the NCO phase by at most {TRANSIENT_ABS['phase_re']:g}, and -- only within {AFTER_LOCK} "
                       f"records after a SYNC event -- a MER report by at most {AFTER_LOCK_ABS['lower']:g} dB and prev_angle by at most {AFTER_LOCK_REL['prev_angle']:g} relative (a CFO-search lock or a roundf() threshold flip, DESIGN (c) limit 2); "
                       "mer_reports_beyond_1e-4_within_0.01dB = MER reports (a sum of squared equaliser errors, printed with one decimal by the reference) that differ by more than 1e-4 of the "
                       "power ratio but less than 0.01 dB: near-singular equaliser cells in channel notches / interference (tests/common.py); mer_reports_below_0dB_within_0.1dB = reports of a sideband the reference itself rates below 0 dB (noise to the receiver), within 0.1 dB. The run FAILS when more than "
                       f"{TRANSIENT_STREAM_BUDGET_PCT} % of the compared streams carry a transient deviation; streams_failing_by_class names what a failing stream differs in "
                       "(p1_px_frame_bits / pids_frame_bits: a decoded frame; timing_pick_beyond_1_sample; <record>_<field>: a float beyond its bound)."}
    # the counted MER exemptions have budgets over the whole batch as well as per log (tests/common.py): measured 24 / 11 / 16 of ~6 600 values (0.36 %) within
    # 0.01 dB and 14 of ~13 000 (0.11 %) below 0 dB within 0.1 dB -- an equaliser that is a little off everywhere exceeds both at once
    if mer_exempt > max(8, mer_values * MER_EXEMPT_BUDGET_PCT // 100):
        FAILURES.append(f"{W.name}: {mer_exempt} of {mer_values} MER values needed the 0.01 dB bound (budget {MER_EXEMPT_BUDGET_PCT} %)")
    if mer_noise > max(4, mer_values * MER_NOISE_BUDGET_PCT_X10 // 1000):
        FAILURES.append(f"{W.name}: {mer_noise} of {mer_values} MER values needed the 0.1 dB below-0-dB bound (budget {MER_NOISE_BUDGET_PCT_X10 / 10} %)")
    if tr_streams * 100 > TRANSIENT_STREAM_BUDGET_PCT * max(1, len(todo)):
        FAILURES.append(f"{W.name}: {tr_streams} of {len(todo)} compared streams with transient loop-state deviations (budget {TRANSIENT_STREAM_BUDGET_PCT} %)")
    if len(lost) > len(lost_checked):
        out["lost_sync_streams_not_checked"] = len(lost) - len(lost_checked)
    if eq_lost != len(lost_checked) or eq_other != len(pick):
        FAILURES.append(f"{W.name}: {len(lost_checked) - eq_lost} lost-sync + {len(pick) - eq_other} other stream logs differ from the {kind} ({len(todo)} of {len(W.checkable)} streams compared): {classes}")
    return out


def launch_check(args):
    """--launch-check: the multi-rank start-up of --gpus N without the workload (runs on CPU with gloo too)."""
    import torch
    from nrsc5_amd import shard
    cuda = torch.cuda.is_available()
    if cuda:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    rank, world, local = shard.init_from_env(expect_world=args.gpus, always=args.force_process_group)
    dev = torch.device("cuda", local) if cuda else torch.device("cpu")
    shard.barrier(dev)
    # the ingest scatter of --ingest scatter in miniature: rank 0 makes every rank's rows, each rank must receive its own
    got = shard.scatter_rows(lambda r: torch.full((2, 4), r, dtype=torch.uint8, device=dev), 2, (4,), torch.uint8, dev)
    ranks = shard.sum_over_ranks([1.0, float(rank), float(bool((got == rank).all()))], dev)
    mine = my_stream_ids(args, world, rank)
    per_rank = shard.gather_floats(float(len(mine)), dev)
    # every collective the real run uses, on this backend's tensors (RCCL: device tensors): the max / gather of the timing, the per-rank verdict vectors and
    # failure texts, the per-stream summary rows
    tmax = shard.max_over_ranks(1.0 + rank, dev)
    vec = shard.gather_vectors([float(rank), float(len(mine))], dev)
    texts = shard.gather_texts(json.dumps([f"rank {rank} \u2713"]), dev)
    rows = shard.gather_summaries(np.array([[s, 1, 2, 2, 3, 4, 1000 + s] for s in mine[:3]], dtype=np.int64), dev)
    collectives_ok = (tmax == float(world) and vec.shape == (world, 2) and [int(v[0]) for v in vec] == list(range(world))
                      and [json.loads(t) for t in texts] == [[f"rank {r} \u2713"] for r in range(world)] and rows.shape == (3 * world, len(shard.SUMMARY_FIELDS))
                      and bool((rows[:, 6] == 1000 + rows[:, 0]).all()))
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_in_process_group": int(ranks[0]), "rank_sum": int(ranks[1]), "ingest_scatter_ok": int(ranks[2]), "collectives_ok": bool(collectives_ok),
                          "scaling": args.scaling, "streams_per_rank": [int(x) for x in per_rank], "ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                          "backend": "nccl(RCCL)" if cuda else "gloo", "launched_by": os.environ.get("NRSC5_BENCH_LAUNCHER", "external torchrun" if world > 1 else "single process")}))
        sys.stdout.flush()
    shard.shutdown(dev)


def my_stream_ids(args, world, rank):
    """contiguous stream ranges per rank, no data-path collective: weak = --streams per GPU, strong = --total-streams over all ranks"""
    from nrsc5_amd import shard
    b = args.stream_base
    if args.scaling == "strong":
        return [b + k for k in shard.stream_range(args.total_streams, world, rank)]
    S = args.streams or (128 if args.workload == "am-cu8" else 256)
    return [b + k for k in shard.stream_range(S * world, world, rank)]


# ---- workloads ----------------------------------------------------------------------------------------------------------
class Fm:
    """configs[2] / configs[3]: hybrid-FM MP1 cu8 streams (seed 1000+k, CFO +-300 Hz, offset [0, 4320), SNR 15/20/25 dB)"""
    name, fs, mode, alg = "fm", FS, 0, ALG_FM_CU8
    dtype = "exact float32 half-band (== int16 Q15) / f32 OFDM + sync / int32 Viterbi metrics"

    def __init__(self, args, dev, local, my_streams):
        import torch
        from nrsc5_amd import engine as eng, synth_torch as stt
        self.args, self.dev, self.eng, self.my_streams = args, dev, eng, my_streams
        S = self.S = len(my_streams)
        self.checkable = list(range(S))
        self.n_frames = n_frames = max(2, int(np.ceil(args.seconds * FS / FRAME_SAMPLES)))
        self.pool = []
        for p in range(args.payloads):
            p1, pids, m = stt.payload_stream(n_frames, seed=p)
            self.pool.append((np.packbits(p1, axis=1, bitorder="little"), stt.modulate(m, dev)))
        nsig = self.pool[0][1].shape[0]
        tail = 8640
        self.stride = (2 * (4320 + nsig + stt.STRIDE_SLACK + tail) + 255) // 256 * 256
        def generate(streams):
            iq = torch.zeros((len(streams), self.stride), dtype=torch.uint8, device=dev)
            nb = torch.zeros((len(streams), 1), dtype=torch.int64, device=dev)
            for k, gs in enumerate(streams):
                out = stt.receive_cu8(self.pool[gs % args.payloads][1], stt.stream_params(gs), tail=tail, out=iq[k])
                nb[k, 0] = out.shape[0] - out.shape[0] % 4
            return iq, nb
        if args.ingest == "scatter":
            # the captures arrive on rank 0 and go out root -> peer, one shard per point-to-point send (untimed set-up)
            from nrsc5_amd import shard
            import torch.distributed as dist
            world = dist.get_world_size() if dist.is_initialized() else 1
            self.iq, nb = shard.scatter_shards(lambda r: generate(my_stream_ids(args, world, r)),
                                               [((S, self.stride), torch.uint8), ((S, 1), torch.int64)], dev)
        else:
            self.iq, nb = generate(my_streams)
        self.nbytes = nb[:, 0].cpu().numpy().astype(np.uint32)
        torch.cuda.synchronize()
        self.samples = float(self.nbytes.astype(np.float64).sum() / 2)
        self.signal_seconds = self.samples / FS
        self.zero_copy = not args.copy_input
        self.pass_no = 0
        self.perm = np.arange(S, dtype=np.int32)
        self.E = self.make_engine(S, local, in_order=args.sync_p1)

    def make_engine(self, S, local, in_order):
        a = self.args
        # replay (window pipeline + L2 feedback): blocks that ran behind a failed P1 frame keep their records / ring slots
        # (marked void, never delivered), so both rings carry head-room for the speculated stretch.  Zero-copy: the captures
        # are read where they are, so the FIFO stays at its minimum size.
        E = self.eng.Engine(max_streams=S, q15_capacity=2 * 71280 if self.zero_copy else int(self.stride // 4 + 1024),
                            record_capacity=max(512, 2 * 16 * self.n_frames + 64), p1_slots=self.n_frames + 12, p1_async=not in_order, device=local,
                            l2_feedback=bool(a.l2_feedback), l2_index=bool(a.l2_index_inline), batch_zero_copy=self.zero_copy)
        apply_tune(E, a)
        return E

    def one_pass(self, E=None, S=None, host_ms=None):
        """Capture row r goes to engine stream (r + pass number) mod S: every pass decodes every stream slot from ANOTHER capture
        (its neighbour carries another payload), so a frame or record left over from the previous pass can never satisfy verify()
        -- the rings and the pinned host mirror are not cleared between passes (that memset would cost more than the pass)."""
        E = E or self.E; S = S or self.S
        self.perm = ((np.arange(S) + self.pass_no) % S).astype(np.int32)
        self.pass_no += 1
        t = [time.perf_counter()]
        E.reset_all(); t.append(time.perf_counter())
        E.batch_append_cu8(self.iq.data_ptr(), self.stride, self.nbytes[:S], stream_ids=self.perm); t.append(time.perf_counter())
        steps = E.batch_process(S); t.append(time.perf_counter())
        out = E.batch_fetch_view(S) if E.cfg.p1_async else E.batch_fetch(S); t.append(time.perf_counter())
        if host_ms is not None:
            for k, name in enumerate(("reset", "append", "process", "fetch")):
                host_ms[name] += (t[k + 1] - t[k]) * 1e3
        return steps, out

    def describe(self, steps):
        a = self.args
        return {"workload": f"configs[2]: batch={self.S} independent hybrid-FM MP1 cu8 streams @1.488375 MS/s per GPU, "
                            f"{self.nbytes[0] / 2 / FS:.2f} s each ({self.n_frames} L1 frames), CFO +-300 Hz, offset [0,4320), SNR 15/20/25 dB; every 4th stream through an "
                            f"impaired channel: sample clock +-20...+-100 ppm, and by turns echoes / an analog FM host at +20 dB / 8 dB fading (synth_torch.stream_params)",
                "impaired_channel_streams": int(sum(1 for gs in self.my_streams if gs % 4 == 1)),
                "streams_per_gpu": self.S, "seconds_per_stream": round(float(self.nbytes[0]) / 2 / FS, 3),
                "p1_decode": "in-order" if a.sync_p1 else "windowed-overlap", "l2_feedback": "on-device" if a.l2_feedback else "off",
                "input": "read in place (half-band fused into the symbol kernel)" if self.zero_copy else "decimated copy in the Q15 FIFO",
                "block_steps_per_pass": int(steps), "ingest": a.ingest, "distinct_payloads": a.payloads,
                "payload_note": f"{a.payloads} transmitted payloads shared by the streams; CFO / timing offset / noise realisation / channel are per stream (seed 1000 + stream id)",
                "stale_result_guard": "capture row r feeds engine stream (r + pass number) mod S: a slot's previous-pass frames belong to another payload and cannot satisfy the truth check",
                "hbm_resident_input_GB": round(float(self.nbytes.sum()) / 1e9, 2)}

    def verify(self, recs, counts, frames):
        """last pass vs the transmitted truth: per stream [id, blocks, P1 frames, exact frames, PIDS frames, FINE blocks, CRC of frames]"""
        eng = self.eng
        rows, self.l2_jobs, self.l2_exact = [], [], []
        for k, gs in enumerate(self.my_streams):
            r = recs[k, :counts[k]]
            truth = self.pool[gs % self.args.payloads][0]
            p1r = r[(r["flags"] & eng.REC_P1) != 0]
            ok, h, first = 0, 0, None
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
                self.l2_jobs.append((int(self.perm[k]), int(rr["p1_slot"]), eng.L2_FM_P1, 0, eng.P1_BITS)); self.l2_exact.append(exact)
            rows.append([gs, len(r), len(p1r), ok, int(((r["flags"] & eng.REC_PIDS) != 0).sum()), int((r["state_after"] == eng.SYNC_FINE).sum()), h])
        return rows

    def parity(self, allrows):
        good = int(((allrows[:, 2] > 0) & (allrows[:, 3] >= allrows[:, 2] - 1)).sum())
        return {"streams": int(allrows.shape[0]), "streams_locked_and_all_p1_frames_equal_transmitted_bits": good,
                "p1_frames_decoded": int(allrows[:, 2].sum()), "p1_frames_bit_exact_vs_truth": int(allrows[:, 3].sum()),
                "pids_frames_decoded": int(allrows[:, 4].sum()),
                "note": "streams whose timing offset falls in the reference algorithm's false-lock zone (sync.c phase-slope ambiguity, ~6 % of uniform offsets) "
                        "decode one garbage frame in the reference too (not a transmitted frame: decoded - bit_exact_vs_truth = those), whose L2 then forces a "
                        "re-acquisition; with l2_feedback the engine does the same on the device: the verdict of the deferred decode rewinds the stream to the "
                        "end of that frame's block (k_replay.hip), so LOST_SYNC and the re-acquisition land on the reference's blocks"}

    def stream_iq(self, k):
        return self.iq[k, :int(self.nbytes[k])].cpu().numpy()

    def cpu_sample(self):
        return self.stream_iq(0)

    def to_log(self, k, r, fr):
        return self.eng.records_to_log(self.E, int(self.perm[k]), r, fr)

    def unpermute(self, recs, counts, frames):
        """row order (capture k) from engine-stream order; untimed"""
        return recs[self.perm], counts[self.perm], frames[self.perm]

    def is_am(self, k):
        return False

    def impaired(self, k):
        return self.my_streams[k] % 4 == 1

    # ---- fm-only side legs (rank 0, N = 1) ---------------------------------------------------------------------------------
    def l2_property(self):
        """full-size property check on the device: the L2 audio index of every decoded P1 frame (frame_push + RS header + CRC-8 of
        all 32 audio packets, k_l2_index) must be clean exactly for the frames that equal the transmitted bits"""
        eng, a = self.eng, self.args
        t0 = time.perf_counter()
        if a.l2_index_inline:
            ring = self.E.batch_fetch_l2(self.S)
            idx = [(eng.l2_frame_to_dict(ring[j[0]][j[1]]), None) for j in self.l2_jobs]
        else:
            idx = self.E.l2_index(self.l2_jobs, want_bytes=False)
        dt = time.perf_counter() - t0
        clean = [int(d["n_pdu"] == 1 and d["pdus"][0]["nop"] == 32 and d["pdus"][0]["crc_bad_lo"] == 0 and d["lost_sync"] == 0) for d, _ in idx]
        out = {"where": "decode streams, inside the timed region" if a.l2_index_inline else "post-pass, untimed",
               "frames_indexed": len(idx), "host_ms_incl_copies": round(dt * 1e3, 2),
               "audio_packets_crc_ok": int(sum(sum(p["nop"] - bin(p["crc_bad_lo"] | (p["crc_bad_hi"] << 32)).count("1") for p in d["pdus"]) for d, _ in idx)),
               "frames_clean": int(sum(clean)), "clean_and_bit_exact": int(sum(c & e for c, e in zip(clean, self.l2_exact))),
               "bit_exact": int(sum(self.l2_exact)), "frames_flagged_lost_sync": int(sum(d["lost_sync"] for d, _ in idx))}
        if not (out["frames_clean"] == out["clean_and_bit_exact"] == out["bit_exact"]):
            FAILURES.append("fm: L2 index property (clean <=> bit-exact) violated")
        return out

    def extra_legs(self, local):
        """configs[1] (one stream through the same engine build) and the in-order mode (reference event timing) of the whole batch"""
        import torch
        out = {}
        self.E.close(); self.E = None
        E1 = self.make_engine(1, local, in_order=False)
        self.one_pass(E1, 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            steps, (recs, counts, frames) = self.one_pass(E1, 1)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        n1 = float(self.nbytes[0]) / 2
        p1 = int(((recs[0, :counts[0]]["flags"] & self.eng.REC_P1) != 0).sum())
        out["single_stream"] = {"workload": "configs[1]: stream 0 of the batch alone on the GPU (batch API: capture resident in HBM, window pipeline, L2 feedback, zero-copy)",
                                "ms_per_pass": round(dt * 1e3, 3), "x_realtime": round(n1 / FS / dt, 1), "value_MSps": round(n1 / dt / 1e6, 2),
                                "us_per_block": round(dt * 1e6 / max(int(counts[0]), 1), 1), "blocks": int(counts[0]), "p1_frames": p1}
        E1.close()
        if not self.args.sync_p1:
            E2 = self.make_engine(self.S, local, in_order=True)
            self.one_pass(E2)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            steps, (recs, counts, frames) = self.one_pass(E2)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            out["in_order"] = {"what": "same batch with p1_async = 0: every frame decoded before the next block of any stream (the reference's event timing "
                                       "without replay), 1 pass", "ms_per_step": round(dt * 1e3, 2), "value_MSps": round(self.samples / dt / 1e6, 1),
                               "x_realtime": round(self.signal_seconds / dt, 1), "block_steps": int(steps),
                               "p1_frames": int(sum(((recs[k, :counts[k]]["flags"] & self.eng.REC_P1) != 0).sum() for k in range(self.S)))}
            E2.close()
        return out


def am_batch(args, dev, streams, fmt, n_frames, signal_seed):
    """Per-stream receivers of ONE clean hybrid-AM MA1 transmission: own CFO, timing offset, noise realisation and level (seed
    5000 + stream id); every 16th stream is hit by an interference burst that breaks first L2 headers (the reference drops to
    SYNC_STATE_NONE there: the replay path).  -> (iq [S, stride] on the device, sizes in elements, truth)"""
    import torch
    from nrsc5_amd import synth_am, synth_torch as stt
    sig, p1, p3, _ = synth_am.am_ma1_signal(n_frames, seed=signal_seed, fmt=fmt)
    sd = torch.from_numpy(sig.astype(np.complex64)).to(dev)
    over = 1 if fmt == "cs16" else 32
    from nrsc5_amd import channel
    stride = (2 * (9 * 270 * over + sig.shape[0] + 128 * over + 1080 * over) + 255) // 256 * 256     # + 128: +20 ppm stretch 61 s by 57 samples
    iq = torch.zeros((len(streams), stride), dtype=torch.int16 if fmt == "cs16" else torch.uint8, device=dev)
    sizes = np.zeros(len(streams), dtype=np.uint32)
    for k, gs in enumerate(streams):
        prm = stt.am_stream_params(gs, n_frames)
        rx = channel.apply_torch(sd, synth_am.FS_CS16 if fmt == "cs16" else synth_am.FS_CU8, prm["chan"])
        out = stt.channel_am(rx, prm["cfo_hz"], prm["offset"] * over, prm["noise"], prm["seed"], fmt, burst=prm["burst"], out=iq[k])
        sizes[k] = out.shape[0] - out.shape[0] % 4
    truth1 = {np.packbits(b, bitorder="little").tobytes() for fr in p1 for b in fr}
    truth3 = {np.packbits(b, bitorder="little").tobytes() for b in p3}
    del sd
    return iq, stride, sizes, truth1, truth3


def am_frame_truth_check(eng, r, frames_k, truth1, truth3):
    n1 = ok1 = n3 = ok3 = 0
    for rr in r:
        fl, slot, bc = int(rr["flags"]), int(rr["p1_slot"]), int(rr["bc_decoded"])
        if fl & eng.REC_P1:
            n1 += 1
            ok1 += np.packbits(eng.unpack_bits(frames_k[slot, bc * 118:(bc + 1) * 118], 3750), bitorder="little").tobytes() in truth1
        if fl & eng.REC_P3:
            n3 += 1
            ok3 += np.packbits(eng.unpack_bits(frames_k[slot, 944:944 + 750], 24000), bitorder="little").tobytes() in truth3
    return n1, ok1, n3, ok3


class Am:
    """configs[4], AM half: hybrid-AM MA1 streams, cs16 @46511.71875 S/s or cu8 @1.488375 MS/s (32:1 cascade); one transmission seen
    by per-stream receivers (CFO +-100 Hz, timing offset over 9 OFDM symbols, three noise levels, interference bursts on every 16th)"""
    mode = 1
    dtype = "int16 Q15 decimator (cu8) / f32 OFDM + sync / int32 K=9 Viterbi metrics"

    def __init__(self, args, dev, local, my_streams, fmt):
        import torch
        from nrsc5_amd import engine as eng, synth_am
        self.args, self.dev, self.eng, self.my_streams, self.fmt = args, dev, eng, my_streams, fmt
        self.name = "am-" + fmt
        S = self.S = len(my_streams)
        self.checkable = list(range(S))
        self.fs = synth_am.FS_CS16 if fmt == "cs16" else synth_am.FS_CU8
        self.alg = ALG_AM_CS16 if fmt == "cs16" else ALG_AM_CU8
        self.iq, self.stride, self.sizes, self._t1, self._t3 = am_batch(args, dev, my_streams, fmt, args.am_frames, 77)
        torch.cuda.synchronize()
        self.samples = float(self.sizes.astype(np.float64).sum()) / 2.0
        self.signal_seconds = self.samples / self.fs
        self.E = eng.Engine(max_streams=S, q15_capacity=int(self.stride / 2 / (1 if fmt == "cs16" else 32)) + 4096, record_capacity=max(512, 2 * 8 * args.am_frames + 64),
                            p1_slots=args.am_frames + 12, am_enable=True, p1_async=not args.sync_p1, l2_feedback=bool(args.l2_feedback), device=local)
        apply_tune(self.E, args)
        for k in range(S):
            self.E.set_mode(k, eng.MODE_AM)

    def one_pass(self, host_ms=None):
        E = self.E
        t = [time.perf_counter()]
        E.reset_all(); t.append(time.perf_counter())
        if self.fmt == "cs16":
            E.batch_append_cs16(self.iq.data_ptr(), self.stride, self.sizes)
        else:
            E.batch_append_cu8(self.iq.data_ptr(), self.stride, self.sizes)
        t.append(time.perf_counter())
        steps = E.batch_process(self.S); t.append(time.perf_counter())
        out = E.batch_fetch(self.S) if self.args.sync_p1 else E.batch_fetch_view(self.S); t.append(time.perf_counter())
        if host_ms is not None:
            for k, name in enumerate(("reset", "append", "process", "fetch")):
                host_ms[name] += (t[k + 1] - t[k]) * 1e3
        return steps, out

    def describe(self, steps):
        return {"workload": f"configs[4], AM half: batch={self.S} hybrid-AM MA1 {self.fmt} streams @{self.fs:.5f} S/s per GPU, {self.sizes[0] / 2 / self.fs:.1f} s each "
                            f"({self.args.am_frames} L1 frames), per-stream CFO +-100 Hz / timing offset / noise (seed 5000 + stream id), interference bursts on every 16th stream"
                            + (", through the 5-stage 32:1 decimator" if self.fmt == "cu8" else ""),
                "streams_per_gpu": self.S, "seconds_per_stream": round(float(self.sizes[0]) / 2 / self.fs, 2), "p1_decode": "in-order" if self.args.sync_p1 else "windowed-overlap",
                "l2_feedback": "on-device" if self.args.l2_feedback else "off", "block_steps_per_pass": int(steps), "hbm_resident_input_GB": round(self.iq.numel() * self.iq.element_size() / 1e9, 2)}

    def verify(self, recs, counts, frames):
        eng = self.eng
        rows = []
        for k, gs in enumerate(self.my_streams):
            r = recs[k, :counts[k]]
            check = (k % max(1, self.S // 16)) == 0             # frame-by-frame truth check on every 16th stream; counts on all
            if check:
                n1, ok1, n3, ok3 = am_frame_truth_check(eng, r, frames[k], self._t1, self._t3)
            else:
                n1 = int(((r["flags"] & eng.REC_P1) != 0).sum()); n3 = int(((r["flags"] & eng.REC_P3) != 0).sum()); ok1 = ok3 = 0
            rows.append([gs, len(r), n1 + n3, (ok1 + ok3) if check else -1, int(((r["flags"] & eng.REC_PIDS) != 0).sum()), int((r["state_after"] == eng.SYNC_FINE).sum()), n3])
        return rows

    def parity(self, allrows):
        chk = allrows[allrows[:, 3] >= 0]
        return {"streams": int(allrows.shape[0]), "frames_decoded_p1_plus_p3": int(allrows[:, 2].sum()), "p3_frames_decoded": int(allrows[:, 6].sum()),
                "pids_frames_decoded": int(allrows[:, 4].sum()), "streams_checked_frame_by_frame": int(chk.shape[0]),
                "frames_checked": int(chk[:, 2].sum()), "frames_equal_transmitted_bits": int(chk[:, 3].sum())}

    def stream_iq(self, k):
        return self.iq[k, :int(self.sizes[k])].cpu().numpy()

    def cpu_sample(self):
        return self.stream_iq(0)

    def to_log(self, k, r, fr):
        return self.eng.am_records_to_log(self.E, k, r, fr)

    def is_am(self, k):
        return True

    def impaired(self, k):
        return self.my_streams[k] % 4 == 2


class Mixed:
    """configs[4]: 128 FM cu8 + 64 AM cs16 + 64 AM cu8 streams in ONE engine (the FM and AM halves run back to back)"""
    name = "mixed"
    dtype = "as the fm and am workloads"
    mode = None

    def __init__(self, args, dev, local, my_streams):
        import torch
        from nrsc5_amd import engine as eng, synth_am, synth_torch as stt
        self.args, self.dev, self.eng, self.my_streams = args, dev, eng, my_streams
        S = self.S = len(my_streams)
        self.checkable = list(range(S))
        self.nfm = nfm = S // 2
        nam = S - nfm
        self.n16, self.n8 = nam // 2, nam - nam // 2
        n_frames = self.n_frames = max(2, int(np.ceil(args.seconds * FS / FRAME_SAMPLES)))
        self.pool = []
        for p in range(4):
            p1, pids, m = stt.payload_stream(n_frames, seed=p)
            self.pool.append((np.packbits(p1, axis=1, bitorder="little"), stt.modulate(m, dev)))
        tail = 8640
        self.stride_fm = (2 * (4320 + self.pool[0][1].shape[0] + stt.STRIDE_SLACK + tail) + 255) // 256 * 256
        self.fm = torch.zeros((nfm, self.stride_fm), dtype=torch.uint8, device=dev)
        self.fm_bytes = np.zeros(nfm, dtype=np.uint32)
        for k in range(nfm):
            out = stt.receive_cu8(self.pool[k % 4][1], stt.stream_params(my_streams[k]), tail=tail, out=self.fm[k])
            self.fm_bytes[k] = out.shape[0] - out.shape[0] % 4
        self.am16, self.len16, self.sizes16, self._t1_16, self._t3_16 = am_batch(args, dev, [my_streams[nfm + k] for k in range(self.n16)], "cs16", args.am_frames, 77)
        self.am8, self.len8, self.sizes8, self._t1_8, self._t3_8 = am_batch(args, dev, [my_streams[nfm + self.n16 + k] for k in range(self.n8)], "cu8", args.am_frames, 78)
        torch.cuda.synchronize()
        s16, s8 = float(self.sizes16.astype(np.float64).sum()) / 2, float(self.sizes8.astype(np.float64).sum()) / 2
        self.samples = float(self.fm_bytes.sum()) / 2 + s16 + s8
        self.signal_seconds = float(self.fm_bytes.sum()) / 2 / FS + s16 / synth_am.FS_CS16 + s8 / synth_am.FS_CU8
        self.alg_bytes = float(self.fm_bytes.sum()) / 2 * ALG_FM_CU8 + s16 * ALG_AM_CS16 + s8 * ALG_AM_CU8
        self.alg = self.alg_bytes / self.samples
        self.fs = self.samples / self.signal_seconds            # for the x real-time of the line
        cap = max(self.len16 // 2, self.len8 // 64, 2 * 71280) + 4096
        self.E = eng.Engine(max_streams=S, q15_capacity=int(cap), record_capacity=max(2 * 16 * n_frames + 64, 2 * 8 * args.am_frames + 64, 512),
                            p1_slots=max(n_frames, args.am_frames) + 12, p1_async=True, am_enable=True, l2_feedback=bool(args.l2_feedback),
                            batch_zero_copy=True, device=local)
        apply_tune(self.E, args)
        self.ids_fm = np.arange(nfm, dtype=np.int32)
        self.ids16 = np.arange(nfm, nfm + self.n16, dtype=np.int32)
        self.ids8 = np.arange(nfm + self.n16, S, dtype=np.int32)
        for s in list(self.ids16) + list(self.ids8):
            self.E.set_mode(int(s), eng.MODE_AM)

    def one_pass(self, host_ms=None):
        E = self.E
        t = [time.perf_counter()]
        E.reset_all(); t.append(time.perf_counter())
        E.batch_append_cu8(self.fm.data_ptr(), self.stride_fm, self.fm_bytes, stream_ids=self.ids_fm)
        E.batch_append_cs16(self.am16.data_ptr(), self.len16, self.sizes16, stream_ids=self.ids16)
        E.batch_append_cu8(self.am8.data_ptr(), self.len8, self.sizes8, stream_ids=self.ids8)
        t.append(time.perf_counter())
        steps = E.batch_process(self.S); t.append(time.perf_counter())
        out = E.batch_fetch_view(self.S); t.append(time.perf_counter())
        if host_ms is not None:
            for k, name in enumerate(("reset", "append", "process", "fetch")):
                host_ms[name] += (t[k + 1] - t[k]) * 1e3
        return steps, out

    def describe(self, steps):
        return {"workload": f"configs[4]: mixed batch of {self.nfm} hybrid-FM MP1 cu8 streams ({self.n_frames} L1 frames each, read in place) + {self.n16} AM MA1 cs16 + "
                            f"{self.n8} AM MA1 cu8 streams ({self.args.am_frames} L1 frames each) in one engine; per-stream CFO / offset / noise everywhere",
                "streams_per_gpu": self.S, "signal_seconds_per_pass": round(self.signal_seconds, 1), "p1_decode": "windowed-overlap",
                "l2_feedback": "on-device" if self.args.l2_feedback else "off", "block_steps_per_pass": int(steps)}

    def verify(self, recs, counts, frames):
        eng = self.eng
        rows = []
        for k, gs in enumerate(self.my_streams):
            r = recs[k, :counts[k]]
            nfr = int(((r["flags"] & eng.REC_P1) != 0).sum())
            ok = -1
            if k < self.nfm:
                truth = {t.tobytes() for t in self.pool[k % 4][0]}
                ok = sum(1 for rr in r if (int(rr["flags"]) & eng.REC_P1) and frames[k, int(rr["p1_slot"])].view(np.uint8).tobytes() in truth)
            elif (k - self.nfm) % 8 == 0:
                t1, t3 = (self._t1_16, self._t3_16) if k < self.nfm + self.n16 else (self._t1_8, self._t3_8)
                n1, ok1, n3, ok3 = am_frame_truth_check(eng, r, frames[k], t1, t3)
                ok = ok1
            rows.append([gs, len(r), nfr, ok, int(((r["flags"] & eng.REC_PIDS) != 0).sum()), int((r["state_after"] == eng.SYNC_FINE).sum()), int(k < self.nfm)])
        return rows

    def parity(self, allrows):
        fm = allrows[allrows[:, 6] == 1]; am = allrows[allrows[:, 6] == 0]
        amc = am[am[:, 3] >= 0]
        return {"streams": int(allrows.shape[0]), "fm_p1_frames_decoded": int(fm[:, 2].sum()), "fm_p1_frames_bit_exact_vs_truth": int(fm[:, 3].sum()),
                "am_p1_frames_decoded": int(am[:, 2].sum()), "am_streams_checked_frame_by_frame": int(amc.shape[0]), "am_p1_frames_checked": int(amc[:, 2].sum()),
                "am_p1_frames_equal_transmitted_bits": int(amc[:, 3].sum()), "pids_frames_decoded": int(allrows[:, 4].sum())}

    def stream_iq(self, k):
        if k < self.nfm:
            return self.fm[k, :int(self.fm_bytes[k])].cpu().numpy()
        if k < self.nfm + self.n16:
            j = k - self.nfm
            return self.am16[j, :int(self.sizes16[j])].cpu().numpy()
        j = k - self.nfm - self.n16
        return self.am8[j, :int(self.sizes8[j])].cpu().numpy()

    def cpu_sample(self):
        return None

    def to_log(self, k, r, fr):
        return self.eng.am_records_to_log(self.E, k, r, fr) if k >= self.nfm else self.eng.records_to_log(self.E, k, r, fr)

    def is_am(self, k):
        return k >= self.nfm

    def impaired(self, k):
        return self.my_streams[k] % 4 == (2 if k >= self.nfm else 1)


TUNE_KNOBS = {"decode_streams": 0, "am_decode_streams": 1, "fwd_segments": 4, "am_segments": 6, "decode_cus": 7, "decode_priority": 8, "mixfft_syms": 10, "traceback_walk": 12, "sync_lanes": 13, "nco_exact": 17, "flow_min": 18, "loop_exact": 19}


def apply_tune(E, args):
    for kv in args.tune:
        k, v = kv.split("=")
        E.tune(TUNE_KNOBS[k], int(v))


def mixed_reference_equality(W, recs, counts, frames):
    """Mixed batch: the FM and the AM members are checked by their own checker sessions"""
    out = {}
    for label, am in (("fm", False), ("am", True)):
        sub = type("Sub", (), {})()
        sub.eng, sub.args, sub.my_streams, sub.name = W.eng, W.args, W.my_streams, f"mixed/{label}"
        sub.checkable = [k for k in range(W.S) if W.is_am(k) == am]
        sub.stream_iq = W.stream_iq
        sub.impaired = W.impaired
        out[label] = reference_equality(sub, recs, counts, frames, W.to_log, am)
    return out


def dropin_leg(iq: np.ndarray, fs: float):
    """configs[0]/[1] as a user of `nrsc5 -r` sees them: one capture through the reference's PUBLIC pipe API (nrsc5_open_pipe,
    nrsc5_pipe_samples_cu8 in 32768-byte calls: src/main.c:1095-1121) on the drop-in (integration/input_hip.c + libnrsc5hip.so
    under the reference's untouched L4 / L2 code) and on the plain reference build; events must be equal."""
    import ctypes
    from oracle import ref
    from tests import common
    paths = {"dropin": os.path.join(ROOT, "integration", "_build", "libnrsc5_hipdropin.so"), "plain": os.path.join(ROOT, "oracle", "_ref", "libnrsc5_plain.so")}
    if not all(os.path.exists(p) for p in paths.values()):
        return {"skipped": "integration/_build/libnrsc5_hipdropin.so or oracle/_ref/libnrsc5_plain.so not prebuilt (need /root/reference in the build container)"}
    out, logs = {"feed": "nrsc5_pipe_samples_cu8, 32768-byte calls (src/main.c:1095-1121)", "seconds_of_signal": round(iq.size / 2 / fs, 2)}, {}
    from nrsc5_amd import engine as eng
    hip = eng.load_library()                                 # the same loaded libnrsc5hip.so the drop-in is linked with
    RUNS = 5

    def timed_runs(lib, name, reps):
        """reps x pipe_run -> (feed seconds of every run, wall of the median run, log of the last run, seam breakdown of the median run)"""
        feeds, walls, brk, log = [], [], [], None
        for rep in range(reps):
            p = ctypes.c_void_p()
            hip.nrsc5hip_debug_seam_totals(None, 1); hip.nrsc5hip_debug_seam_counts(None, 1)
            t0 = time.perf_counter()
            n = lib.pipe_run(iq.ctypes.data, iq.size, 32768, 0, 0, ctypes.byref(p))
            walls.append(time.perf_counter() - t0)
            f = float(lib.pipe_last_feed_seconds())
            feeds.append(f)
            log = ref.parse_log(ctypes.string_at(p, n))
            if name != "plain":
                tot = (ctypes.c_double * 8)()
                hip.nrsc5hip_debug_seam_totals(tot, 0)
                blocks = max(tot[6], 1.0)
                cnt = (ctypes.c_double * 6)()
                hip.nrsc5hip_debug_seam_counts(cnt, 0)
                brk.append({"blocks": int(tot[6]), "pushes": int(tot[4]), "submissions_h2d_plus_decimator": int(tot[5]),
                            "block_steps_left_in_flight_deferred_wait": int(cnt[0]), "read_positions_mispredicted": int(cnt[1]),
                            "block_steps_without_p1_decode_launches": int(cnt[2]), "p1_decodes_launched_late": int(cnt[3]), "block_steps_queued_ahead_of_the_previous_delivery": int(cnt[4]), "pushes_copied_into_the_host_resident_capture": int(cnt[5]),
                            "us_per_block": {"host_copy_into_pinned_staging": round(tot[0] / blocks * 1e6, 1), "host_enqueue_h2d_and_decimator": round(tot[1] / blocks * 1e6, 1),
                                             "host_enqueue_block_step": round(tot[2] / blocks * 1e6, 1), "wait_for_device": round(tot[3] / blocks * 1e6, 1),
                                             "fetch_p1_frames": round(tot[7] / blocks * 1e6, 1),
                                             "reference_host_code_L2_and_callbacks_and_rest": round((f - tot[0] - tot[1] - tot[2] - tot[3] - tot[7]) / blocks * 1e6, 1)},
                            "total_us_per_block": round(f / blocks * 1e6, 1)})
        med = int(np.argsort(feeds)[len(feeds) // 2])
        return feeds, walls[med], log, (brk[med] if brk else None)

    def entry(feeds, wall):
        f = float(np.median(feeds))
        return {"feed_seconds": round(f, 4), "x_realtime": round(iq.size / 2 / fs / f, 1), "runs": len(feeds), "statistic": "median",
                "x_realtime_min_max": [round(iq.size / 2 / fs / max(feeds), 1), round(iq.size / 2 / fs / min(feeds), 1)], "wall_seconds_incl_open_close": round(wall, 3)}

    libs = {}
    for name, path in paths.items():
        lib = ctypes.CDLL(path)
        lib.pipe_run.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        lib.pipe_run.restype = ctypes.c_size_t
        lib.pipe_last_feed_seconds.restype = ctypes.c_double
        libs[name] = lib
    env0 = os.environ.get("NRSC5HIP_SYNC_DELIVERY")
    try:
        # opt-in (NRSC5HIP_OVERLAP_DELIVERY=1 / NRSC5HIP_SYNC_DELIVERY=0): overlapped delivery (events of block n during the first call after the device has finished it, at the latest in the call that
        # completes block n + 1 / a zero-length call / nrsc5_close)
        os.environ["NRSC5HIP_SYNC_DELIVERY"] = "0"
        feeds, wall, logs["dropin"], out["breakdown"] = timed_runs(libs["dropin"], "dropin", RUNS)
        out["dropin"] = entry(feeds, wall)
        out["dropin"]["delivery"] = "overlapped (opt-in, NRSC5HIP_OVERLAP_DELIVERY=1): a block's events arrive up to one block (93 ms of signal) after the call that completed it; order preserved; flushed by a zero-length call or nrsc5_close"
        # the drop-in's DEFAULT since round 6: the reference's contract -- every event inside the nrsc5_pipe_samples_* call that completes its block
        os.environ["NRSC5HIP_SYNC_DELIVERY"] = "1"
        feeds, wall, logs["dropin_strict"], brk = timed_runs(libs["dropin"], "dropin_strict", RUNS)
        out["dropin_strict_delivery"] = entry(feeds, wall)
        out["dropin_strict_delivery"]["delivery"] = "DEFAULT: events inside the call that completes their block, as src/input.c delivers them"
        out["dropin_strict_delivery"]["breakdown_us_per_block"] = brk["us_per_block"] if brk else None
    finally:
        if env0 is None:
            os.environ.pop("NRSC5HIP_SYNC_DELIVERY", None)
        else:
            os.environ["NRSC5HIP_SYNC_DELIVERY"] = env0
    feeds, wall, logs["plain"], _ = timed_runs(libs["plain"], "plain", 2)
    out["plain"] = entry(feeds, wall)

    def events_equal(exp, got):
        same = [k for k, _ in exp] == [k for k, _ in got]
        if same:
            for (k, a), (_, b) in zip(exp, got):
                if k == "hdc":
                    same = same and a["program"] == b["program"] and a["flags"] == b["flags"] and a["data"] == b["data"]
                elif k in ("sync", "mer", "ber"):
                    same = same and all(common.float_close(f, float(a[f]), float(b[f])) for f in a)
        return bool(same)
    exp = logs["plain"]
    same = events_equal(exp, logs["dropin"])
    same_strict = events_equal(exp, logs["dropin_strict"])
    out["events"] = len(exp); out["hdc_packets"] = sum(k == "hdc" for k, _ in exp); out["events_equal"] = same; out["events_equal_strict_delivery"] = same_strict
    out["speedup_vs_plain"] = round(out["plain"]["feed_seconds"] / out["dropin"]["feed_seconds"], 2)
    if not same or not same_strict:
        FAILURES.append("dropin: public-API event log differs from the plain reference" + ("" if same_strict else " (strict delivery)"))
    return out


def rocprof_fraction(args, dom, alg_bytes_per_launch):
    """The same roofline fraction with the launch duration rocprofv3 measured (no HIP-event overhead), when profiles/ carries a kernel-stats
    summary of THIS tree (profiles/kernel_stats_latest.json, written by tools/stamp_kernel_stats.py from a `rocprofv3 --kernel-trace --stats`
    CSV of this command); None otherwise."""
    path = os.path.join(ROOT, "profiles", "kernel_stats_latest.json")
    try:
        ks = json.load(open(path))
        if ks.get("source_sha") != source_fingerprint() or ks.get("workload") != args.workload:
            return None
        k = ks["kernels"].get(LEAD_KERNEL.get(dom, ""))
        if not k:
            return None
        avg_s = k["average_ns"] * 1e-9
        ach = alg_bytes_per_launch / avg_s / 1e9
        return {"kernel": LEAD_KERNEL[dom], "avg_launch_us": round(avg_s * 1e6, 2), "calls": k["calls"], "achieved_GBps": round(ach, 1), "frac": round(ach / HBM_PEAK_GBPS, 4),
                "file": "profiles/kernel_stats_latest.json (from " + ks.get("csv", "?") + ")"}
    except Exception:
        return None


LEAD_KERNEL = {"mixfft": "k_mixfft", "sync": "k_sync", "p1_viterbi": "k_p1_forward", "p1_traceback": "k_p1_tbwalk", "am": "k_am_block", "am_decode": "k_am_decode_fwd"}
KERNELS_OF_CLASS = {"p1_viterbi": "k_p1_forward (K=7 forward trellis pass of one decode window's P1 frames)", "p1_traceback": "k_p1_traceback (+ k_l2_index_window)",
                    "p1_deint": "k_p1_deint", "mixfft": "k_mixfft", "sync": "k_sync (+ k_px_deint, k_px_commit)", "pids": "k_pids_decode (+ k_px_decode)",
                    "am": "k_am_block + k_am_interleave", "am_decode": "k_am_decode (8 x P1 + P3 + 8 x PIDS trellis passes of one AM L1 frame per stream)",
                    "acquire": "k_acq_list / _decimate / _fir / _corr / _peak", "prepare": "k_prepare, k_rollback", "decimate": "k_decimate_* / k_append_cs16 / k_attach_raw"}
DOMINANT_DEFAULT = {"fm": "mixfft", "mixed": "mixfft", "am-cs16": "am", "am-cu8": "am"}
# the classes launched on the block-step chain -- the one queue a pass cannot be shorter than; the decode classes run beside it on up to three other queues and their
# summed launch durations overlap each other and the chain
CHAIN_CLASSES = ("decimate", "acquire", "prepare", "mixfft", "sync", "am")
DOMINANT_RULE = ("largest device time per pass among the kernel classes of the block-step chain (decimate, acquire, prepare, mixfft, sync, am): the chain is the pass's critical path; "
                 "the decode classes (p1_*, pids, am_decode) overlap it on other queues -- their summed durations are listed in device_ms_per_pass, their VALU load in `valu`")


def dominant_class(prof_all):
    chain = {k: v for k, v in prof_all.items() if k in CHAIN_CLASSES}
    pool = chain or prof_all
    return max(pool, key=lambda k: pool[k][0])


def source_fingerprint():
    from nrsc5_amd import build
    return build.source_sha()


def make_workload(name, args, dev, local, my_streams):
    if name == "fm":
        return Fm(args, dev, local, my_streams)
    if name == "mixed":
        return Mixed(args, dev, local, my_streams)
    return Am(args, dev, local, my_streams, name.split("-")[1])


def checked_parity(W, recs, counts, frames, allrows, checker: bool):
    parity = W.parity(allrows)
    if checker:
        try:
            if isinstance(W, Mixed):
                parity["reference_equality_rank0"] = mixed_reference_equality(W, recs, counts, frames)
            else:
                parity["reference_equality_rank0"] = reference_equality(W, recs, counts, frames, W.to_log, W.mode == 1)
        except Exception as ex:
            parity["reference_equality_rank0"] = {"error": repr(ex)}
            FAILURES.append(f"{W.name}: reference-equality leg raised {ex!r}")
    return parity


def compact_leg(name, args, dev, local, steps=3):
    """One configs[4] workload inside the default fm line: 1 warm-up + `steps` timed passes, truth + reference equality."""
    import torch
    a = argparse.Namespace(**vars(args)); a.workload = name; a.sync_p1 = False; a.ingest = "local"
    S = 128 if name == "am-cu8" else 256
    t0 = time.perf_counter()
    W = make_workload(name, a, dev, local, list(range(S)))
    gen = time.perf_counter() - t0
    W.one_pass()
    dt = 0.0
    for _ in range(steps):
        W.E.poison_results()             # untimed: frame / record rings and their host mirrors get a pattern no decode produces, so the
        torch.cuda.synchronize()         # checks below can only pass on what the LAST pass wrote (every AM stream carries the same payload)
        t0 = time.perf_counter()
        block_steps, (recs, counts, frames) = W.one_pass()
        torch.cuda.synchronize(); dt += time.perf_counter() - t0
    dt /= steps
    if hasattr(W, "unpermute"):
        recs, counts, frames = W.unpermute(recs, counts, frames)
    rows = np.array(W.verify(recs, counts, frames), dtype=np.int64)
    parity = checked_parity(W, recs, counts, frames, rows, not args.no_cpu_baseline)
    out = {"config": W.describe(block_steps), "steps": steps, "ms_per_step": round(dt * 1e3, 3), "value_MSps": round(W.samples / dt / 1e6, 2),
           "x_realtime": round(W.signal_seconds / dt, 1), "parity": parity, "gen_seconds": round(gen, 1)}
    W.E.close()
    del W
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    from nrsc5_amd import shard
    if args.gpus > 1 and not shard.launched_by_torchrun():
        # `python bench.py --gpus N` on its own: start the N ranks here, the way the driver's torchrun line does
        sys.exit(shard.launch_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus, {"NRSC5_BENCH_LAUNCHER": "bench.py"}))
    if args.launch_check:
        return launch_check(args)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (MI355X); there is no CPU fallback")
    share = os.environ.get("NRSC5_BENCH_SHARE_GPU") == "1"     # TEST MODE: every rank on GPU 0 (with NRSC5_SHARD_BACKEND=gloo): the N-rank flow on a one-GPU box; its numbers mean nothing
    if not share and torch.cuda.device_count() < max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))):
        raise SystemExit(f"--gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible")
    rank, world, local = shard.init_from_env(expect_world=args.gpus)
    if share:
        local = 0
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    cdev = torch.device("cpu") if os.environ.get("NRSC5_SHARD_BACKEND") == "gloo" else dev      # where the collectives' tensors live
    from nrsc5_amd import engine as _eng
    _eng.check_fresh()                                           # never measure a library that was built from other sources

    my_streams = my_stream_ids(args, world, rank)
    t_gen = time.perf_counter()
    W = make_workload(args.workload, args, dev, local, my_streams)
    t_gen = time.perf_counter() - t_gen
    E = W.E

    host_ms = {"reset": 0.0, "append": 0.0, "process": 0.0, "fetch": 0.0}
    # Device time per kernel class comes from the LAST WARM-UP pass, instrumented with HIP events around every launch (that
    # costs the block-step chain ~9 % of a pass, so it stays out of the timed region); the timed passes keep the events of the
    # dominant class only -- the roofline's launch duration is measured there, live, on the kernel's own stream.
    prof_all, dom = {}, DOMINANT_DEFAULT[args.workload]
    for w in range(args.warmup):
        last = w == args.warmup - 1 and not args.no_profile
        if last:
            E.profile(1)
        W.one_pass()
        if last:
            prof_all = {k: v for k, v in E.profile(0).items() if v[1]}
            if prof_all:
                dom = dominant_class(prof_all)
    E.profile(0 if args.no_profile else dom)
    shard.barrier(cdev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pass_ms, tp = [], t0
    for _ in range(args.steps):
        block_steps, (recs, counts, frames) = W.one_pass(host_ms=host_ms)
        tn = time.perf_counter(); pass_ms.append((tn - tp) * 1e3); tp = tn      # a pass ends with its results on the host
    torch.cuda.synchronize()
    dt_rank = time.perf_counter() - t0
    shard.barrier(cdev)
    dt = time.perf_counter() - t0
    prof = E.profile(0)
    fwd_checked, fwd_repaired = E.fwd_stats() if hasattr(E, "fwd_stats") else (0, 0)
    tb_checked, tb_rewalked = E.tb_stats() if hasattr(E, "tb_stats") else (0, 0)
    dt = shard.max_over_ranks(dt, cdev)
    per_rank_ms = [round(x / args.steps * 1e3, 3) for x in shard.gather_floats(dt_rank, cdev)]
    tot = shard.sum_over_ranks([W.samples, W.signal_seconds, 1.0], cdev)
    total_samples, total_seconds, ranks_seen = float(tot[0]), float(tot[1]), int(tot[2])     # ranks_seen: counted through the collective itself

    if hasattr(W, "unpermute"):
        recs, counts, frames = W.unpermute(recs, counts, frames)
    rows = W.verify(recs, counts, frames)
    allrows = shard.gather_summaries(np.array(rows, dtype=np.int64), cdev)
    # Parity against the unmodified reference, on EVERY rank for its own streams (untimed): with --gpus N each rank fans its checker out
    # over its share of the host cores and the verdicts are gathered, so that one multi-GPU run proves all N x 256 streams, not rank 0's.
    checker = not args.no_cpu_baseline
    n_fail0 = len(FAILURES)
    parity = checked_parity(W, recs, counts, frames, allrows if rank == 0 else np.array(rows, dtype=np.int64), checker)
    per_rank_parity = None
    if world > 1:
        re_ = parity.get("reference_equality_rank0") or {}
        subs = [re_] if "streams_compared" in re_ else [v for v in re_.values() if isinstance(v, dict) and "streams_compared" in v]
        tot = lambda key: float(sum(x.get(key, 0) for x in subs))
        vec = [float(rank), float(len(my_streams)), tot("streams_compared"), tot("lost_sync_streams_equal") + tot("other_streams_equal"),
               tot("streams_equal_under_the_strict_rule"), tot("streams_with_transient_loop_state_deviation"), float(len(FAILURES) - n_fail0), float(bool(checker))]
        g = shard.gather_vectors(vec, cdev)
        per_rank_parity = [{"rank": int(r[0]), "streams": int(r[1]), "streams_compared": int(r[2]), "streams_equal": int(r[3]), "streams_equal_under_the_strict_rule": int(r[4]),
                            "streams_with_transient_loop_state_deviation": int(r[5]), "parity_failures": int(r[6]), "checker_ran": bool(r[7])} for r in g]
        # every rank's own list of failures travels through the process group too: rank 0's stdout line carries ALL verdicts verbatim (a line on a
        # stderr shared with the launcher, the peers and their checker processes can be lost or torn: GPUTEST_r05)
        texts = shard.gather_texts(json.dumps(FAILURES[n_fail0:]), cdev)
        for r_, t_ in zip(per_rank_parity, texts):
            r_["failures"] = json.loads(t_)
        # ... and, where asked for, into a file of this rank's own (NRSC5_BENCH_VERDICT_DIR/rank-parity-<rank>.json, written whole then renamed)
        vdir = os.environ.get("NRSC5_BENCH_VERDICT_DIR")
        if vdir:
            os.makedirs(vdir, exist_ok=True)
            tmp = os.path.join(vdir, f".rank-parity-{rank}.json.{os.getpid()}")
            with open(tmp, "w") as f:
                json.dump(per_rank_parity[rank], f)
            os.replace(tmp, os.path.join(vdir, f"rank-parity-{rank}.json"))
        print("rank-parity " + json.dumps(per_rank_parity[rank]), file=sys.stderr)     # informational only
        sys.stderr.flush()
    shard.shutdown(cdev)                  # every collective of the run is behind us: the ranks leave the group together
    if rank != 0:
        if _POOL is not None:
            _POOL.close()
        sys.exit(3 if len(FAILURES) > n_fail0 else 0)
    if per_rank_parity is not None:
        parity["per_rank_reference_equality"] = per_rank_parity
        for r in per_rank_parity[1:]:
            if r["parity_failures"] or (checker and r["streams_compared"] != r["streams"]):
                FAILURES.append(f"rank {r['rank']}: {r['parity_failures']} parity failure(s), {r['streams_compared']} of {r['streams']} streams compared: {r['failures'][:3]}")
    checker = checker and world == 1     # the host-core legs below (CPU baseline, drop-in, configs[4]) belong to the single-GPU line
    if args.workload == "fm" and not args.no_l2_index:
        try:
            parity["l2_index_rank0"] = W.l2_property()
        except Exception as ex:
            parity["l2_index_rank0"] = {"error": repr(ex)}
            FAILURES.append(f"fm: L2 index leg raised {ex!r}")

    value = total_samples * args.steps / dt / 1e6
    # ---- roofline of the dominant kernel class (by device time, HIP events on its launch stream) --------------------------
    used = {k: v for k, v in prof.items() if v[1]}
    roofline = None
    if used:
        dom_ms, dom_launches = used[dom]
        # every kernel class sees each input sample of the pass once: algorithmic bytes per launch = bytes of the pass / launches per pass
        alg_bytes_per_launch = W.samples * W.alg * args.steps / max(dom_launches, 1)
        avg_launch_s = dom_ms / 1e3 / max(dom_launches, 1)
        achieved = alg_bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        traffic, whole, valu = None, None, None
        fp = source_fingerprint()
        if os.path.exists(args.traffic_json):
            try:
                tj = json.load(open(args.traffic_json))
                if tj.get("source_sha") == fp and tj.get("workload") == args.workload:
                    traffic = tj.get("per_class_hbm_bytes_per_launch", {}).get(dom)
                    whole = {"hbm_bytes_per_pass": tj.get("whole_path_hbm_bytes_per_pass"), "over_algorithmic": tj.get("whole_path_over_algorithmic"),
                             "fetch_correction": tj.get("fetch_correction"), "file": os.path.relpath(args.traffic_json, ROOT)}
                    if whole["hbm_bytes_per_pass"]:
                        # counter bytes of one pass (all nrsc5 kernels, reads corrected as the guide prescribes) over this run's time per pass
                        whole["counter_GBps"] = round(whole["hbm_bytes_per_pass"] / (dt / args.steps) / 1e9, 1)
                        whole["counter_frac_of_peak"] = round(whole["counter_GBps"] / HBM_PEAK_GBPS, 4)
            except Exception:
                traffic = None
        if os.path.exists(args.sq_json):
            try:
                sj = json.load(open(args.sq_json))
                if sj.get("source_sha") == fp and sj.get("workload") == args.workload:
                    # VALU roofline: SIMD cycles in which a VALU instruction issued (SQ_ACTIVE_INST_VALU x 4: the counter ticks in quad-cycles),
                    # summed over every kernel of one pass, over the SIMD cycles the chip has in this run's pass time
                    clk = sj["clock_ghz"]
                    avail = N_SIMD * clk * 1e9 * (dt / args.steps)
                    valu = {"issued_valu_simd_cycles_per_pass": sj["valu_simd_cycles_per_pass"], "simd_cycles_in_pass": round(avail), "frac": round(sj["valu_simd_cycles_per_pass"] / avail, 4),
                            "clock_ghz": clk, "clock_note": sj.get("clock_note"), "per_class_frac_of_pass": {k: round(v / avail, 4) for k, v in sj.get("per_class_valu_simd_cycles", {}).items()},
                            "per_class_valu_busy_while_resident": sj.get("per_class_valu_busy_while_resident"), "file": os.path.relpath(args.sq_json, ROOT)}
            except Exception:
                valu = None
        # the kernel class with the largest device time of ALL classes (the decode classes overlap the chain on other queues): from the instrumented warm-up pass
        by_time = None
        if prof_all:
            kt = max(prof_all, key=lambda k: prof_all[k][0])
            kt_ms, kt_n = prof_all[kt]
            if kt_n and kt_ms > 0:
                kt_bytes = W.samples * W.alg / kt_n
                kt_ach = kt_bytes / (kt_ms / 1e3 / kt_n) / 1e9
                by_time = {"kernel": kt, "kernel_functions": KERNELS_OF_CLASS.get(kt, kt), "device_ms_per_pass": round(kt_ms, 3), "launches_per_pass": int(kt_n), "avg_launch_ms": round(kt_ms / kt_n, 4),
                           "alg_bytes_per_launch": int(kt_bytes), "achieved": round(kt_ach, 3), "frac": round(kt_ach / HBM_PEAK_GBPS, 6), "unit": "GB/s",
                           "note": "largest summed device time among all kernel classes in one instrumented pass (HIP events around every launch); equals `kernel` when the chain's dominant class is also the largest"}
        whole_frac = value * W.alg / 1e3 / HBM_PEAK_GBPS
        roofline = {"bound": "hbm", "measured_bound": "valu issue + dependency latency, not HBM: the path moves %.1f %% of the HBM peak algorithmically and issues VALU instructions on %s of the chip's SIMD cycles (`valu`); "
                                                       "`bound` stays \"hbm\" because that is the roof the metric is quoted against (BASELINE.json)" % (100 * whole_frac, ("%.0f %%" % (100 * valu["frac"])) if valu else "(no stamped SQ record for this tree)"),
                    "frac_valu": valu["frac"] if valu else None, "whole_pass": {"algorithmic_GBps": round(value * W.alg / 1e3, 3), "frac_of_hbm_peak": round(whole_frac, 6)},
                    "kernel_by_device_time": by_time, "kernel": dom, "kernel_functions": KERNELS_OF_CLASS.get(dom, dom), "dominant_rule": DOMINANT_RULE, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic, "whole_path_traffic": whole,
                    "practical_bound": "valu issue + the trellis' serial dependency chain (SURVEY 8d): see `valu`", "valu": valu,
                    "rocprof": rocprof_fraction(args, dom, alg_bytes_per_launch),
                    "avg_launch_ms": round(avg_launch_s * 1e3, 4), "launches": dom_launches,
                    "avg_launch_ms_note": "HIP events on the kernel's own launch stream; up to three decode streams and the block-step chain run concurrently, so this is a per-launch latency under contention, not an exclusive-occupancy figure",
                    "alg_bytes_per_launch": int(alg_bytes_per_launch), "alg_bytes_per_sample": round(W.alg, 4),
                    "whole_path_GBps": round(value * W.alg / 1e3, 3),
                    "device_ms_per_pass": {k: round(v[0], 3) for k, v in prof_all.items()},
                    "device_ms_per_pass_note": "one separately instrumented warm-up pass (events around every launch); the timed passes time the dominant class only",
                    "host_ms_per_pass": {k: round(v / args.steps, 3) for k, v in host_ms.items()}}
    cpu = None
    if checker:
        sample = W.cpu_sample()
        if sample is not None:
            try:
                cpu = cpu_baseline(sample, W.fs, W.mode, args.cpu_baseline_seconds, args.cpu_processes)
            except Exception as ex:
                cpu = {"error": repr(ex)}
                FAILURES.append(f"cpu baseline leg raised {ex!r}")
    extra = {}
    if args.workload == "fm" and world == 1 and not args.no_extra_legs:
        try:
            sample = W.stream_iq(0)
            extra = W.extra_legs(local)
            del W.iq, W.pool
            torch.cuda.empty_cache()
            extra["dropin"] = dropin_leg(np.ascontiguousarray(sample), FS)
            extra["config4"] = {"am_cs16": compact_leg("am-cs16", args, dev, local), "mixed": compact_leg("mixed", args, dev, local)}
        except Exception as ex:
            extra["extra_legs_error"] = repr(ex)
            FAILURES.append(f"extra legs raised {ex!r}")
    config = W.describe(block_steps)
    if world > 1:
        config["parallelism"] = f"{world} ranks x {len(my_streams)} streams ({args.scaling} scaling), no data-path collective"
        config["total_streams"] = int(allrows.shape[0])
    line = {
        "metric": "IQ MS/s demod+decoded", "value": round(value, 2), "unit": "IQ MS/s",
        "x_realtime": round(total_seconds * args.steps / dt, 1), "n_gpus": world, "ranks_in_process_group": ranks_seen, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "ms_per_step_median": round(float(np.median(pass_ms)), 3),
        "ms_per_step_min_max": [round(min(pass_ms), 3), round(max(pass_ms), 3)], "per_rank_ms_per_step": per_rank_ms, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": W.dtype, "data": "synthetic",
        "config": config, "roofline": roofline, "cpu_baseline": cpu, "parity": parity,
        "gen_seconds": round(t_gen, 1),
        "traceback_walk": {"chunk_boundaries_checked": tb_checked, "chunks_rewalked": tb_rewalked,
                           "what": "single-path traceback: every 64-step chunk walked once after a 64-step run-in through the chunk above (speculative), the chain of chunk boundaries verified, wrong chunks re-walked (viterbi_v3.h); counts since the engine was created, rank 0"},
        "forward_pass_segments": {"boundaries_checked": fwd_checked, "segments_repaired": fwd_repaired,
                                  "what": "K=7 forward trellis pass cut into concurrently running segment waves per frame (speculative start, verified, repaired when wrong: viterbi_v3.h); counts since the engine was created, rank 0"},
    }
    line.update(extra)
    line["parity_failures"] = list(FAILURES)
    print(json.dumps(line))
    sys.stdout.flush()
    if FAILURES:
        print("PARITY / CHECKER FAILURE: " + "; ".join(FAILURES), file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
