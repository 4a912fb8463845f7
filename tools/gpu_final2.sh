#!/bin/bash
# the bench line of the final tree with the stamped PMC / SQ summaries in place, and the AM batch with EVERY stream checked
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
( time timeout 600 python bench.py ) > gpurun_out/r03h_bench.log 2>&1; echo "bench rc=$?"
grep "^{" gpurun_out/r03h_bench.log | tail -1 > gpurun_out/r03h_bench.json
( time timeout 600 python bench.py --workload am-cs16 --oracle-streams 256 --no-extra-legs --steps 1 --warmup 1 ) > gpurun_out/r03h_am_all.log 2>&1; echo "am all rc=$?"
grep "^{" gpurun_out/r03h_am_all.log | tail -1 > gpurun_out/r03h_am_all.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03h_bench.json")); r = d["roofline"]
print(d["ms_per_step"], d["x_realtime"], r["kernel"], r["frac"], r["traffic"], (r.get("valu") or {}).get("frac"), d["parity_failures"])
d = json.load(open("gpurun_out/r03h_am_all.json")); print(d["ms_per_step"], json.dumps(d["parity"].get("reference_equality_rank0"))[:700], d["parity_failures"])
PY
