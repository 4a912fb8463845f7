#!/bin/bash
# the round's records from ONE box, to be run AFTER the last change of a device source: GPU tests, rocprofv3 kernel stats (fm, am-cs16; the fm summary
# stamped with the source fingerprint), PMC traffic, SQ counters, the timeline -- then, with the stamped summaries of THIS tree in place, the default
# bench line (every stream of fm / am-cs16 / mixed against the unmodified reference) and the FM batch on two more seeds.
#   gpurun --timeout 2700 -- 'bash tools/gpu_r6_final.sh TAG'
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r06z}
# (second argument "notests": the GPU suite has just run on this very tree in a call of its own)
if [ "$2" != notests ]; then ( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/${TAG}_tests.log; fi
for WL in fm am-cs16; do
  rm -rf gpurun_out/${TAG}_prof_$WL
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof_$WL -o p -- python $R/bench.py --workload $WL --no-cpu-baseline --no-extra-legs --steps 3 --warmup 1 ) > gpurun_out/${TAG}_prof_$WL.log 2>&1; echo "prof $WL rc=$?"
  f=$(find gpurun_out/${TAG}_prof_$WL -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_kernel_stats_$WL.csv && grep "nrsc5::" "$f" | head -4 | cut -c1-160
  if [ "$WL" = fm ] && [ -f gpurun_out/${TAG}_kernel_stats_fm.csv ]; then cp gpurun_out/${TAG}_kernel_stats_fm.csv profiles/r06_kernel_stats_fm_256x20s.csv; python tools/stamp_kernel_stats.py profiles/r06_kernel_stats_fm_256x20s.csv fm; cp profiles/kernel_stats_latest.json gpurun_out/kernel_stats_latest.json; fi
  rm -rf gpurun_out/${TAG}_prof_$WL
done
bash tools/gpu_pmc.sh fm 2>&1 | tail -2 | cut -c1-500
bash tools/gpu_sq.sh fm 2>&1 | tail -3 | cut -c1-300
cp gpurun_out/traffic_fm.json profiles/traffic_latest.json; cp gpurun_out/sq_fm.json profiles/sq_latest.json
bash tools/gpu_trace.sh ${TAG}_trace > /dev/null 2>&1; head -8 gpurun_out/${TAG}_trace_summary.txt; rm -rf gpurun_out/${TAG}_trace_raw
bash tools/gpu_r6_step.sh $TAG "256 512" notests
