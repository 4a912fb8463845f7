#!/bin/bash
# round 6: dataflow bursts (k_flow) on the MI355X -- the bit-identity test first (bounded: a hand-off that is never seen gives up after ~1 s per poll), then the FM batch
# with and without the bursts, every stream against the unmodified reference each time.
#   gpurun --timeout 1500 -- 'bash tools/gpu_r6_flow.sh TAG'
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r06b}
( time timeout 420 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dataflow" -s ) > gpurun_out/${TAG}_flow_test.log 2>&1; echo "flow test rc=$?"; tail -6 gpurun_out/${TAG}_flow_test.log | cut -c1-300
for FM in 0 32 0 32; do
  ( timeout 420 python bench.py --workload fm --no-extra-legs --steps 10 --warmup 2 --tune flow_min=$FM ) > gpurun_out/${TAG}_bench_flow$FM.log 2>gpurun_out/${TAG}_bench_flow$FM.err; echo "bench flow_min=$FM rc=$?"
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/${TAG}_bench_flow$FM.log") if l.startswith("{")][-1])
    r = d["parity"]["reference_equality_rank0"]
    print("flow_min=$FM", d["ms_per_step"], d["ms_per_step_min_max"], "failures", d["parity_failures"], "strict", r["streams_equal_under_the_strict_rule"], "transient", r["streams_with_transient_loop_state_deviation"], d["roofline"]["device_ms_per_pass"])
except Exception as ex:
    print("flow_min=$FM: no line", ex)
PY
done
