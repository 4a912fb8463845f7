#!/bin/bash
# one SQ-counter PMC pass (counters + kernel trace for the clock, no other trace domain) over one bench pass -> profiles-style summary
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out; rm -rf gpurun_out/pmc_sq
WL=${1:-fm}
CMD="python $R/bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --no-l2-index --no-extra-legs"
( cd /tmp && time timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_sq -o sq --output-format csv -- $CMD ) > gpurun_out/pmc_sq.log 2>&1; rc=$?; echo "sq rc=$rc"
if [ $rc -ne 0 ] || [ -z "$(find gpurun_out/pmc_sq -name '*counter_collection.csv' | head -1)" ]; then
  # a counter the tool refuses (or too many for one pass): the SQ set alone
  rm -rf gpurun_out/pmc_sq
  ( cd /tmp && time timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES -d $R/gpurun_out/pmc_sq -o sq --output-format csv -- $CMD ) > gpurun_out/pmc_sq2.log 2>&1; echo "sq (no GRBM, no trace) rc=$?"
fi
tail -4 gpurun_out/pmc_sq.log | cut -c1-300
python profiles/collect_sq.py gpurun_out/pmc_sq gpurun_out/sq_${WL}.json $WL
find gpurun_out/pmc_sq -name '*.csv' -delete
