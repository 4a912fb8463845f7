#!/bin/bash
# round 3: GPU parity suite, the default bench line (all legs), rocprofv3 kernel stats (fm + am-cs16), PMC traffic + SQ issue
# counters of one fm pass.   gpurun --timeout 1500 -- 'bash tools/gpu_r3_check.sh TAG [notests]'
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r03a}
if [ "$2" != "notests" ]; then
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/${TAG}_tests.log
fi
( time timeout 600 python bench.py ) > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc=$?"
grep "^{" gpurun_out/${TAG}_bench.log | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-600 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.log | cut -c1-400
python - "$TAG" <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/{sys.argv[1]}_bench.json"))
    for k in ("single_stream", "in_order", "dropin", "parity_failures"):
        print(k, json.dumps(d.get(k))[:700])
    print("parity", json.dumps(d["parity"].get("reference_equality_rank0"))[:900])
    for k, v in (d.get("config4") or {}).items():
        print("config4", k, v["ms_per_step"], v["x_realtime"], json.dumps(v["parity"])[:900])
    print("extra err", d.get("extra_legs_error"))
except Exception as ex:
    print("no bench json", ex)
PY
for WL in fm am-cs16; do
rm -rf gpurun_out/${TAG}_prof_$WL
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_$WL -o p -- python $R/bench.py --workload $WL --no-cpu-baseline --no-extra-legs --steps 3 --warmup 1 ) > gpurun_out/${TAG}_prof_$WL.log 2>&1; echo "prof $WL rc=$?"
f=$(find gpurun_out/${TAG}_prof_$WL -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_kernel_stats_$WL.csv && head -16 "$f" | cut -c1-200
grep "^{" gpurun_out/${TAG}_prof_$WL.log | tail -1 > gpurun_out/${TAG}_bench_under_rocprof_$WL.json
rm -rf gpurun_out/${TAG}_prof_$WL
done
bash tools/gpu_pmc.sh fm 2>&1 | tail -4 | cut -c1-1500
bash tools/gpu_sq.sh fm 2>&1 | tail -12 | cut -c1-400
