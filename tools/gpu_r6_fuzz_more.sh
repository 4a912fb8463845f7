#!/bin/bash
# more slices of the GPU fuzz (tests/test_gpu_fuzz.py) under other seed counters, then the N = 1 point of configs[3]: 2048 streams on ONE GPU
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r06g}
for C in ${2:-2 3 4 5}; do
  echo $C > tests/fuzz_seed_counter.txt
  ( time timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -s ) > gpurun_out/${TAG}_fuzz_counter$C.log 2>&1; echo "fuzz counter $C rc=$?"; grep -a "fuzz" gpurun_out/${TAG}_fuzz_counter$C.log | cut -c1-420
done
( time timeout 900 python bench.py --streams 2048 --no-extra-legs --steps 3 --warmup 1 --cpu-baseline-seconds 2 ) > gpurun_out/${TAG}_bench_2048.log 2>gpurun_out/${TAG}_bench_2048.err; echo "bench 2048 rc=$?"
python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/${TAG}_bench_2048.log") if l.startswith("{")][-1])
    r = d["parity"]["reference_equality_rank0"]
    print("2048 streams", d["ms_per_step"], d["x_realtime"], d["value"], "failures", d["parity_failures"], "compared", r["streams_compared"], "strict", r["streams_equal_under_the_strict_rule"], "transient", r["streams_with_transient_loop_state_deviation"], r["streams_failing_by_class"], r["seconds"])
except Exception as ex:
    print("2048: no line", ex)
PY
