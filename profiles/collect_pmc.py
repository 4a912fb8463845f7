"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes) of
`python bench.py --steps 1 --warmup 0` into HBM bytes per launch of every kernel class and of the whole path
-> profiles/traffic_latest.json, stamped with the fingerprint of the device sources it was collected from
(bench.py ignores the file when that differs from the tree it runs in).

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB-units of 1024 B
(hbm_bytes = counter * 1024) and FETCH_SIZE reports exactly 1/2 of the bytes of wide coalesced streaming reads (128-B requests
tallied at 64 B): doubled here before it is compared with a byte count.  The correction is CALIBRATED on this path's own access
pattern: k_mixfft reads the cu8 captures exactly once (the algorithmic input bytes of the pass) and nothing else of that size, so
its corrected fetch over the capture bytes must come out near 1 (recorded as `fetch_calibration`).  WRITE_SIZE is taken as is.
Per-launch / whole-path figures use the corrected reads + writes; the raw read count is kept beside them."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# kernel-name fragments -> the classes of nrsc5hip_profile (include/nrsc5hip.h)
CLASSES = {"decimate": ("k_decimate_fm_cu8", "k_decimate_commit", "k_append_cs16", "k_attach_raw", "k_am_decimate"),
           "acquire": ("k_acq_",), "prepare": ("k_prepare", "k_rollback"), "mixfft": ("k_mixfft",), "sync": ("k_sync", "k_px_deint", "k_px_commit"),
           "p1_deint": ("k_p1_deint",), "p1_viterbi": ("k_p1_forward", "k_p1_fix"), "p1_traceback": ("k_p1_traceback", "k_p1_tbwalk", "k_p1_tbmap", "k_l2_index_window"), "pids": ("k_pids_decode", "k_px_decode"),
           "am": ("k_am_block", "k_am_viterbi", "k_am_interleave"), "am_decode": ("k_am_decode",)}
LEAD = {"decimate": "k_decimate_fm_cu8", "acquire": "k_acq_fir", "prepare": "k_prepare", "mixfft": "k_mixfft", "sync": "k_sync",
        "p1_deint": "k_p1_deint", "p1_viterbi": "k_p1_forward", "p1_traceback": "k_p1_traceback", "pids": "k_pids_decode", "am": "k_am_block", "am_decode": "k_am_decode"}


def kernel_base(kernel_name: str) -> str:
    """'void nrsc5::k_sync<768>(nrsc5::DevTables, ...)' -> 'k_sync' (template instances print with their return type and arguments)"""
    n = kernel_name.split("(")[0].strip()
    if n.startswith("void "):
        n = n[5:]
    n = n.split("<")[0]
    return n.replace("nrsc5::", "")


def load(dirname, counter):
    tot, calls = {}, {}
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter or "nrsc5::" not in row["Kernel_Name"]:
                continue
            name = kernel_base(row["Kernel_Name"])
            tot[name] = tot.get(name, 0.0) + float(row["Counter_Value"])
            calls[name] = calls.get(name, 0) + 1
    return tot, calls


def main(fetch_dir, write_dir, workload, alg_bytes_per_pass, out):
    from nrsc5_amd import build
    ft, fc = load(fetch_dir, "FETCH_SIZE")
    wt, wc = load(write_dir, "WRITE_SIZE")
    per_class, per_class_detail = {}, {}
    FX = 2.0                                                   # the guide's correction for wide coalesced reads
    for cls, frags in CLASSES.items():
        rd = sum(v for k, v in ft.items() if any(k.startswith(x) for x in frags)) * 1024
        wr = sum(v for k, v in wt.items() if any(k.startswith(x) for x in frags)) * 1024
        launches = sum(c for k, c in fc.items() if k.startswith(LEAD[cls]))
        if launches:
            per_class[cls] = (FX * rd + wr) / launches
            per_class_detail[cls] = {"launches": launches, "fetch_bytes_raw": rd, "fetch_bytes_corrected": FX * rd, "write_bytes": wr}
    rd_all, wr_all = sum(ft.values()) * 1024, sum(wt.values()) * 1024
    cal = None
    if workload == "fm" and ft.get("k_mixfft"):
        in_bytes = float(alg_bytes_per_pass) * 2.0 / (2.0 + 18432.0 / 2211840.0)      # the cu8 input share of the algorithmic bytes
        cal = {"kernel": "k_mixfft", "corrected_fetch_bytes": FX * ft["k_mixfft"] * 1024, "capture_bytes_read_once": in_bytes,
               "ratio": FX * ft["k_mixfft"] * 1024 / in_bytes}
    res = {"source_sha": build.source_sha(), "workload": workload, "passes": 1,
           "fetch_correction": "FETCH_SIZE x 2 (MI355X_MICROARCH.md, HBM: 128-B requests tallied at 64 B), calibrated on k_mixfft", "fetch_calibration": cal,
           "per_class_hbm_bytes_per_launch": per_class, "per_class": per_class_detail,
           "whole_path_fetch_bytes_raw": rd_all, "whole_path_fetch_bytes_corrected": FX * rd_all, "whole_path_write_bytes": wr_all,
           "whole_path_hbm_bytes_per_pass": FX * rd_all + wr_all, "algorithmic_bytes_per_pass": float(alg_bytes_per_pass),
           "whole_path_over_algorithmic": (FX * rd_all + wr_all) / float(alg_bytes_per_pass),
           "per_kernel_fetch_MB_raw": {k: round(v * 1024 / 1e6, 1) for k, v in sorted(ft.items(), key=lambda kv: -kv[1])[:12]},
           "per_kernel_write_MB": {k: round(v * 1024 / 1e6, 1) for k, v in sorted(wt.items(), key=lambda kv: -kv[1])[:12]}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: res[k] for k in ("source_sha", "fetch_calibration", "whole_path_hbm_bytes_per_pass", "whole_path_over_algorithmic", "per_class_hbm_bytes_per_launch")}))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
