#!/bin/bash
# PMC traffic + SQ issue counters of one FM pass, then -- with the stamped summaries of THIS tree in place -- the default bench line
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r04y}
bash tools/gpu_pmc.sh fm 2>&1 | tail -2 | cut -c1-700
bash tools/gpu_sq.sh fm 2>&1 | tail -2 | cut -c1-200
cp gpurun_out/traffic_fm.json profiles/traffic_latest.json; cp gpurun_out/sq_fm.json profiles/sq_latest.json
( time timeout 600 python bench.py ) > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc=$?"
grep "^{" gpurun_out/${TAG}_bench.log | tail -1 > gpurun_out/${TAG}_bench.json
python - "$TAG" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/{sys.argv[1]}_bench.json")); r = d["roofline"]
print(d["ms_per_step"], d["x_realtime"], r["kernel"], r["frac"], r["traffic"], (r.get("valu") or {}).get("frac"), d["parity_failures"])
print("single", d["single_stream"]["x_realtime"], "dropin", d["dropin"]["dropin"]["x_realtime"], d["dropin"]["events_equal"], "inorder", d["in_order"]["ms_per_step"])
for k, v in d["config4"].items(): print(k, v["ms_per_step"], v["x_realtime"])
PY
