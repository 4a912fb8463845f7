#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 600 python -m pytest tests -m gpu -q -k "l2 or ma3 or MA3 or am_oracle" ) > gpurun_out/pytest_l2.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/pytest_l2.log
for fb in 1 0; do
( timeout 300 python bench.py --l2-feedback $fb --no-cpu-baseline ) > gpurun_out/bench_fb$fb.log 2>&1; echo "bench fb=$fb rc=$?"
python - <<PY
import json
l=[x for x in open("gpurun_out/bench_fb$fb.log") if x.startswith("{")]
j=json.loads(l[-1]); print(j["value"], j["ms_per_step"], j["parity"]["streams_locked_and_all_p1_frames_equal_transmitted_bits"], j["parity"]["p1_frames_decoded"], j["parity"]["p1_frames_bit_exact_vs_truth"], j["roofline"]["device_ms_per_pass"])
PY
done
