"""After `gpurun -- bash tools/gpu_r5_final.sh TAG`: copy the records of that call from gpurun_out/ (scratch) into profiles/ (tracked) under the round's
names, refusing anything whose stamp is not the fingerprint of the device sources in this tree.   python tools/collect_final_records.py TAG"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nrsc5_amd import build  # noqa: E402

tag = sys.argv[1]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
sha = build.source_sha()
for src, dst in (("kernel_stats_latest.json", "kernel_stats_latest.json"), ("traffic_fm.json", "traffic_latest.json"), ("sq_fm.json", "sq_latest.json")):
    d = json.load(open(os.path.join(G, src)))
    assert d["source_sha"] == sha, (src, d["source_sha"], sha)
    shutil.copy(os.path.join(G, src), os.path.join(P, dst))
shutil.copy(os.path.join(G, f"{tag}_kernel_stats_fm.csv"), os.path.join(P, "r05_kernel_stats_fm_256x20s.csv"))
shutil.copy(os.path.join(G, f"{tag}_kernel_stats_am-cs16.csv"), os.path.join(P, "r05_kernel_stats_am-cs16_256x61s.csv"))
shutil.copy(os.path.join(G, f"{tag}_trace_summary.txt"), os.path.join(P, "r05_trace_final.txt"))
bench = json.load(open(os.path.join(G, f"{tag}_bench.json")))
assert not bench["parity_failures"], bench["parity_failures"]
json.dump(bench, open(os.path.join(P, "r05_bench_fm.json"), "w"))
out = {"stream_base_0": {k: bench[k] for k in ("ms_per_step", "parity_failures")} | {"reference_equality": bench["parity"]["reference_equality_rank0"]}}
for base in (256, 512):
    b = json.load(open(os.path.join(G, f"{tag}_parity_base{base}.json")))
    out[f"stream_base_{base}"] = {"ms_per_step": b["ms_per_step"], "parity_failures": b["parity_failures"], "reference_equality": b["parity"]["reference_equality_rank0"]}
json.dump(out, open(os.path.join(P, "r05_parity_all_256_streams.json"), "w"), indent=1)
print("fingerprint", sha, "| fm", bench["ms_per_step"], "ms per pass,", bench["x_realtime"], "x real time | strict-rule equal streams per seed:",
      [v["reference_equality"].get("streams_equal_under_the_strict_rule") for v in out.values()])
