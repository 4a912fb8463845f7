#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for n in 512 1024; do
( timeout 400 python bench.py --no-cpu-baseline --steps 2 --streams $n ) > gpurun_out/b$n.log 2>&1
python - <<PY
import json
l=[x for x in open("gpurun_out/b$n.log") if x.startswith("{")]
if l:
    j=json.loads(l[-1]); d=j["roofline"]["device_ms_per_pass"]; print($n, j["value"], j["x_realtime"], j["ms_per_step"], d, j["parity"]["p1_frames_bit_exact_vs_truth"], j["parity"]["p1_frames_decoded"])
else:
    print($n, "failed"); print(open("gpurun_out/b$n.log").read()[-600:])
PY
done
