"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes)
of `python bench.py` into HBM bytes per launch of the dominant kernel class -> profiles/traffic_latest.json.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB-units of 1024 B
(hbm_bytes = counter * 1024) and FETCH_SIZE under-reports wide coalesced streaming reads by 2x (128-B requests
tallied at 64 B).  The trellis kernels read bytes / 8-byte words, not 16-B lanes, so both the raw and the
2x-corrected read figures are recorded; `hbm_bytes_per_launch` uses the raw read count + writes (lower bound)."""
import csv
import glob
import json
import os
import sys

CLASS_KERNELS = {"p1_viterbi": ("k_p1_forward", "k_p1_traceback", "k_p1_deint"), "sync": ("k_sync",), "mixfft": ("k_mixfft",),
                 "decimate": ("k_decimate_fm_cu8",), "p1_deint": ("k_p1_deint",)}


def load(dirname, counter):
    tot, calls = {}, {}
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            tot[name] = tot.get(name, 0.0) + float(row["Counter_Value"])
            calls[name] = calls.get(name, 0) + 1
    return tot, calls


def main(fetch_dir, write_dir, kernel_class, out):
    ft, fc = load(fetch_dir, "FETCH_SIZE")
    wt, wc = load(write_dir, "WRITE_SIZE")
    keys = CLASS_KERNELS[kernel_class]
    rd = sum(v for k, v in ft.items() if any(x in k for x in keys)) * 1024
    wr = sum(v for k, v in wt.items() if any(x in k for x in keys)) * 1024
    launches = sum(c for k, c in fc.items() if keys[0] in k)
    res = {"kernel_class": kernel_class, "launches": launches,
           "fetch_bytes_per_launch_raw": rd / max(launches, 1), "fetch_bytes_per_launch_x2": 2 * rd / max(launches, 1),
           "write_bytes_per_launch": wr / max(launches, 1),
           "hbm_bytes_per_launch": (rd + wr) / max(launches, 1),
           "per_kernel_fetch_KiB": {k[:60]: v for k, v in sorted(ft.items(), key=lambda kv: -kv[1])[:8]},
           "per_kernel_write_KiB": {k[:60]: v for k, v in sorted(wt.items(), key=lambda kv: -kv[1])[:8]}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res)[:600])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4])
