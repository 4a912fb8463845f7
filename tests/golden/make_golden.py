"""Regenerates tests/golden/*.npz from the UNMODIFIED reference (oracle/_ref, built from
/root/reference by oracle/Makefile).  Run in the build container only:

    python tests/golden/make_golden.py

Each fixture pins: the sha256 of the synthetic capture (so a host that regenerates different
bytes is detected), the exact Q15 decimator output hash, every per-block soft-bit hash, every
decoded PIDS / P1 frame, and the per-block timing/CFO trace of the reference."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests import common  # noqa: E402
from nrsc5_amd import synth, synth_am  # noqa: E402
from oracle import ref  # noqa: E402


def main():
    R = ref.RefLib(sse=False)
    Rs = ref.RefLib(sse=True)
    only = [a for a in sys.argv[2:]]
    for name, kw in common.GOLDEN_CASES.items():
        if only and name not in only:
            continue
        cap = synth.fm_mp1_capture(**kw)
        log, q15, _ = R.run(cap.iq, taps=ref.TAP_Q15 | ref.TAP_SOFT)
        log_sse, _, _ = Rs.run(cap.iq, taps=ref.TAP_SOFT)
        assert not common.compare_logs(log, log_sse, rtol=0.0, skip_kinds=("hdc",)), "generic and SSE reference builds disagree"
        arrs = common.log_to_arrays(log)
        soft = [v for k, v in log if k == "soft"]
        arrs["soft_bc"] = np.array([v["bc"] for v in soft], dtype=np.int32)
        arrs["soft_sha"] = np.array([common.sha256(v["bits"]) for v in soft])
        arrs["q15_sha"] = np.array(common.sha256(q15))
        arrs["q15_head"] = q15[:4096].copy()
        arrs["iq_sha"] = np.array(common.sha256(cap.iq))
        arrs["truth_p1"] = np.packbits(np.array(cap.p1_frames, dtype=np.uint8).reshape(-1, 146176), axis=1, bitorder="little")
        if cap.p3_frames:
            arrs["truth_p3"] = np.packbits(np.array(cap.p3_frames, dtype=np.uint8), axis=1, bitorder="little")
        if cap.p4_frames:
            arrs["truth_p4"] = np.packbits(np.array(cap.p4_frames, dtype=np.uint8), axis=1, bitorder="little")
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrs)
        nfr = int(arrs["p1"].shape[0])
        print(f"{name}: {len(arrs['block_int'])} blocks, {nfr} P1 frames, {arrs['pids'].shape[0]} PIDS, "
              f"sync {arrs['sync'].tolist()}, {os.path.getsize(os.path.join(HERE, name + '.npz'))} bytes")


def main_am():
    R = ref.RefLib(sse=False)
    Rs = ref.RefLib(sse=True)
    only = [a for a in sys.argv[2:]]
    for name, kw in common.GOLDEN_AM_CASES.items():
        if only and name not in only:
            continue
        cap = synth_am.am_ma1_capture(**kw)
        log, q15, _ = R.run(cap.iq, mode=ref.MODE_AM, taps=ref.TAP_Q15 | ref.TAP_SOFT)
        log_sse, _, _ = Rs.run(cap.iq, mode=ref.MODE_AM, taps=ref.TAP_SOFT)
        assert not common.compare_logs(log, log_sse, rtol=0.0, skip_kinds=("hdc",)), "generic and SSE reference builds disagree"
        arrs = common.am_log_to_arrays(log)
        sym = [v for k, v in log if k == "amsym"]
        arrs["sym_bc"] = np.array([v["bc"] for v in sym], dtype=np.int32)
        arrs["sym_sha"] = np.array([common.sha256(np.concatenate([v["pl"], v["pu"], v["s"], v["t"]])) for v in sym])
        arrs["q15_sha"] = np.array(common.sha256(q15))
        arrs["q15_head"] = q15[:4096].copy()
        arrs["iq_sha"] = np.array(common.sha256(cap.iq))
        arrs["truth_p1"] = np.packbits(np.array(cap.p1_frames, dtype=np.uint8).reshape(-1, 3750), axis=1, bitorder="little")
        arrs["truth_p3"] = np.packbits(np.array(cap.p3_frames, dtype=np.uint8), axis=1, bitorder="little")
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrs)
        print(f"{name}: {len(arrs['block_int'])} blocks, {arrs['p1'].shape[0]} P1 + {arrs['p3'].shape[0]} P3 frames, "
              f"{arrs['pids'].shape[0]} PIDS, ber {arrs['ber'].tolist()}, sync {arrs['sync'].tolist()}, "
              f"{os.path.getsize(os.path.join(HERE, name + '.npz'))} bytes")


if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] == "fm":
        main()
    if len(sys.argv) < 2 or sys.argv[1] == "am":
        main_am()
