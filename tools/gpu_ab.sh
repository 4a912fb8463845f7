#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { ( timeout 300 python bench.py --no-cpu-baseline --steps 4 ) > gpurun_out/b.log 2>&1
python - <<PY
import json
l=[x for x in open("gpurun_out/b.log") if x.startswith("{")]
j=json.loads(l[-1]); d=j["roofline"]["device_ms_per_pass"]; print("$1", j["value"], j["ms_per_step"], "vit", d["p1_viterbi"], "acq", d["acquire"], "sync", d["sync"], "mix", d["mixfft"], j["parity"]["p1_frames_bit_exact_vs_truth"])
PY
}
run wide; NRSC5HIP_SYNC_NARROW=1 run narrow; run wide; NRSC5HIP_SYNC_NARROW=1 run narrow
( time timeout 600 python -m pytest tests -m gpu -q -k "golden or oracle_end_to_end or l2 or batch" ) > gpurun_out/pytest_ab.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_ab.log
