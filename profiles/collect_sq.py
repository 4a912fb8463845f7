"""Aggregates one rocprofv3 --pmc pass of SQ counters over `python bench.py` into per-kernel issue statistics
(profiles/r01_sq_issue_stats.json).  MI355X_MICROARCH.md: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~= WAVE_CYCLES
(disjoint, same units), so the three fractions say where a wave's life goes: parked on s_waitcnt / barriers, stalled at
issue (dependency / pipe), or issuing; ACTIVE_INST_VALU / WAVE_CYCLES is the VALU share of it."""
import csv
import glob
import json
import os
import sys

COUNTERS = ("SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_INSTS_VALU", "SQ_WAVES", "SQ_BUSY_CYCLES")


def main(dirname, out):
    tot, calls = {}, {}
    for f in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0]
            if not name.startswith("nrsc5::"):
                continue
            c = row["Counter_Name"]
            d = tot.setdefault(name, {})
            d[c] = d.get(c, 0.0) + float(row["Counter_Value"])
            if c == "SQ_WAVES":
                calls[name] = calls.get(name, 0) + 1
    res = {}
    for name, d in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0)):
        wc = d.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        res[name] = {"launches": calls.get(name, 0), "waves": d.get("SQ_WAVES", 0.0),
                     "wave_cycles": d.get("SQ_WAVE_CYCLES", 0.0),
                     "frac_issuing": round(d.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 4),
                     "frac_issuing_valu": round(d.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4),
                     "frac_parked_waitcnt_barrier": round(d.get("SQ_WAIT_ANY", 0.0) / wc, 4),
                     "frac_issue_stall": round(d.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4),
                     "valu_insts_per_wave": round(d.get("SQ_INSTS_VALU", 0.0) / max(d.get("SQ_WAVES", 0.0), 1.0), 1),
                     "sq_busy_cycles": d.get("SQ_BUSY_CYCLES", 0.0)}
    json.dump({"counters": COUNTERS, "kernels": res}, open(out, "w"), indent=1)
    for k, v in list(res.items())[:14]:
        print(k, v)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
