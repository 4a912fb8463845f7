#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for args in "--no-profile" "" "--no-profile --l2-feedback 0" "--l2-feedback 0"; do
( timeout 300 python bench.py --no-cpu-baseline $args ) > gpurun_out/b.log 2>&1
python - <<PY
import json
l=[x for x in open("gpurun_out/b.log") if x.startswith("{")]
j=json.loads(l[-1]); print("$args", j["value"], j["ms_per_step"], j["roofline"]["host_ms_per_pass"])
PY
done
