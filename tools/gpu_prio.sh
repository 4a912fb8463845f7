#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { ( timeout 300 python bench.py --no-cpu-baseline --l2-feedback 0 --steps 4 ) > gpurun_out/b.log 2>&1
python - <<PY
import json
l=[x for x in open("gpurun_out/b.log") if x.startswith("{")]
j=json.loads(l[-1]); d=j["roofline"]["device_ms_per_pass"]; print("$1", j["value"], j["ms_per_step"], "vit", d["p1_viterbi"], "sync", d["sync"], "mix", d["mixfft"])
PY
}
NRSC5HIP_PRIO_FWD=2 NRSC5HIP_PRIO_TB=3 run "fwd2 tb3"
NRSC5HIP_PRIO_FWD=2 NRSC5HIP_PRIO_TB=2 run "fwd2 tb2"
NRSC5HIP_PRIO_FWD=1 NRSC5HIP_PRIO_TB=2 run "fwd1 tb2"
NRSC5HIP_PRIO_FWD=2 NRSC5HIP_PRIO_TB=3 NRSC5HIP_NAUX=2 run "fwd2 tb3 naux2"
NRSC5HIP_PRIO_FWD=3 NRSC5HIP_PRIO_TB=3 NRSC5HIP_NAUX=2 run "fwd3 tb3 naux2"
