#!/bin/bash
# one SQ-counter PMC pass (counters only, no trace domains) over one bench pass -> small per-kernel summary
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out; rm -rf gpurun_out/pmc_sq
( cd /tmp && time timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES -d $R/gpurun_out/pmc_sq -o sq --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-l2-index ) > gpurun_out/pmc_sq.log 2>&1; echo "sq rc=$?"
tail -4 gpurun_out/pmc_sq.log | cut -c1-300
python profiles/collect_sq.py gpurun_out/pmc_sq gpurun_out/sq_issue_stats.json
find gpurun_out/pmc_sq -name '*counter_collection.csv' -delete
