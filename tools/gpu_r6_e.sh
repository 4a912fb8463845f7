#!/bin/bash
# round 6: which NCO policy for the batch default?  Three fresh 256-stream CFO-search batches under policies 0 / 1 / 2 (exact loops on), then what policies 1 and 2 cost the bench pass.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r06e}
timeout 600 python tools/gpu_cfo_batch.py 100256,100512,100768 0,1,2 1 > gpurun_out/${TAG}_policies.txt 2>&1; echo "policies rc=$?"
python - <<PY
import json
for l in open("gpurun_out/${TAG}_policies.txt"):
    if l.startswith("{"):
        d = json.loads(l); print(d["base"], "policy", d["policy"], "locks", d["cfo_search_locks"], "strict", d["strict"], "transient", d["transient_streams"], "failing", d["failing_by_class"], "steps", d["block_steps"], d["seconds"])
PY
for NP in 0 1 2; do
  ( timeout 420 python bench.py --workload fm --no-extra-legs --steps 10 --warmup 2 --tune nco_exact=$NP ) > gpurun_out/${TAG}_bench_nco$NP.log 2>gpurun_out/${TAG}_bench_nco$NP.err; echo "bench nco_exact=$NP rc=$?"
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/${TAG}_bench_nco$NP.log") if l.startswith("{")][-1])
    r = d["parity"]["reference_equality_rank0"]
    print("nco_exact=$NP", d["ms_per_step"], d["ms_per_step_min_max"], "failures", d["parity_failures"], "strict", r["streams_equal_under_the_strict_rule"], "transient", r["streams_with_transient_loop_state_deviation"], r["streams_failing_by_class"], d["roofline"]["device_ms_per_pass"].get("prepare"))
except Exception as ex:
    print("nco_exact=$NP: no line", ex)
PY
done
