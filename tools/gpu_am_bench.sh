#!/bin/bash
# gpurun --timeout 900 -- 'bash tools/gpu_am_bench.sh'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 300 python tools/gpu_am_bench.py --streams 256 --frames 41 --fmt cs16 ) > gpurun_out/am_bench_cs16.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/am_bench_cs16.log
( time timeout 400 python tools/gpu_am_bench.py --streams 128 --frames 41 --fmt cu8 --steps 2 ) > gpurun_out/am_bench_cu8.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/am_bench_cu8.log
cd /tmp && ( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_am -o am -- python $GRAFT_REPO_ROOT/tools/gpu_am_bench.py --streams 256 --frames 20 --fmt cs16 --steps 2 ) > $GRAFT_REPO_ROOT/gpurun_out/am_rocprof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT; find gpurun_out/prof_am -name "*kernel_stats*" | head -2; head -12 $(find gpurun_out/prof_am -name "*kernel_stats*" | head -1)
