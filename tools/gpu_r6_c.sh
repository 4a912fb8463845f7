#!/bin/bash
# round 6, third GPU call: the whole GPU suite (with the fuzz slice and the dataflow test), what the exact loop arithmetic costs (FM batch under LOOP_EXACT 0 / 1), the AM batch
#   gpurun --timeout 2400 -- 'bash tools/gpu_r6_c.sh TAG'
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r06c}
( time timeout 1500 python -m pytest tests -m gpu -x -q -s ) > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; grep -a "fuzz\|flow bursts\|passed\|failed" gpurun_out/${TAG}_tests.log | cut -c1-400 | tail -12
for LE in 0 1 0 1; do
  ( timeout 420 python bench.py --workload fm --no-extra-legs --steps 10 --warmup 2 --tune loop_exact=$LE ) > gpurun_out/${TAG}_bench_le$LE.log 2>gpurun_out/${TAG}_bench_le$LE.err; echo "bench loop_exact=$LE rc=$?"
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/${TAG}_bench_le$LE.log") if l.startswith("{")][-1])
    r = d["parity"]["reference_equality_rank0"]
    print("loop_exact=$LE", d["ms_per_step"], d["ms_per_step_min_max"], "failures", d["parity_failures"], "strict", r["streams_equal_under_the_strict_rule"], "transient", r["streams_with_transient_loop_state_deviation"], r["streams_failing_by_class"])
except Exception as ex:
    print("loop_exact=$LE: no line", ex)
PY
done
( timeout 600 python bench.py --workload am-cs16 --steps 5 --warmup 1 ) > gpurun_out/${TAG}_bench_am.log 2>gpurun_out/${TAG}_bench_am.err; echo "bench am rc=$?"
python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/${TAG}_bench_am.log") if l.startswith("{")][-1])
    r = d["parity"]["reference_equality_rank0"]
    print("am-cs16", d["ms_per_step"], d["ms_per_step_min_max"], "failures", d["parity_failures"], "strict", r["streams_equal_under_the_strict_rule"], d["roofline"]["device_ms_per_pass"])
except Exception as ex:
    print("am: no line", ex)
PY
