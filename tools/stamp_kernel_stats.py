"""profiles/kernel_stats_latest.json from a `rocprofv3 --kernel-trace --stats` CSV of `bench.py --workload W`: per kernel (template
instances folded) calls / average ns, stamped with the fingerprint of the device sources it was collected from.
    python tools/stamp_kernel_stats.py profiles/r04_kernel_stats_fm_256x20s.csv fm"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nrsc5_amd import build
from profiles.collect_pmc import kernel_base
src, workload = sys.argv[1], sys.argv[2]
ker = {}
for r in csv.DictReader(open(src)):
    if "nrsc5::" not in r["Name"]:
        continue
    k = ker.setdefault(kernel_base(r["Name"]), {"calls": 0, "total_ns": 0})
    k["calls"] += int(r["Calls"]); k["total_ns"] += int(r["TotalDurationNs"])
for k in ker.values():
    k["average_ns"] = k["total_ns"] / max(k["calls"], 1)
out = {"source_sha": build.source_sha(), "workload": workload, "csv": os.path.relpath(src, ROOT), "kernels": ker}
json.dump(out, open(os.path.join(ROOT, "profiles", "kernel_stats_latest.json"), "w"), indent=1)
print(out["source_sha"], {k: round(v["average_ns"] / 1e3, 1) for k, v in sorted(ker.items(), key=lambda kv: -kv[1]["total_ns"])[:6]})
