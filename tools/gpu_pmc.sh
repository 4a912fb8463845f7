#!/bin/bash
# HBM traffic of the whole path from PMC counters: two passes (counters only, no trace domains), then the stamped summary.
# gpurun --timeout 900 -- 'bash tools/gpu_pmc.sh [workload]'
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
WL=${1:-fm}
CMD="python $R/bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-extra-legs --no-l2-index"
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
( cd /tmp && time timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o f --output-format csv -- $CMD ) > gpurun_out/pmc_fetch.log 2>&1; echo "fetch rc=$?"
( cd /tmp && time timeout 400 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o w --output-format csv -- $CMD ) > gpurun_out/pmc_write.log 2>&1; echo "write rc=$?"
ALG=$(grep "^{" gpurun_out/pmc_fetch.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['alg_bytes_per_launch'] * d['roofline']['launches'])")
python profiles/collect_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write $WL $ALG gpurun_out/traffic_${WL}.json
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name '*counter_collection.csv' -delete 2>/dev/null
