#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 300 python tools/gpu_am_bench.py --streams 256 --frames 41 --fmt cs16 ) > gpurun_out/am_bench_cs16.log 2>&1; grep "^{" gpurun_out/am_bench_cs16.log
( timeout 300 python tools/gpu_am_bench.py --streams 128 --frames 41 --fmt cu8 --steps 2 ) > gpurun_out/am_bench_cu8.log 2>&1; grep "^{" gpurun_out/am_bench_cu8.log
( timeout 300 python tools/gpu_mixed_bench.py ) > gpurun_out/mixed.log 2>&1; grep "^{" gpurun_out/mixed.log
cd /tmp && ( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_am -o am -- python $GRAFT_REPO_ROOT/tools/gpu_am_bench.py --streams 256 --frames 41 --fmt cs16 --steps 2 ) > $GRAFT_REPO_ROOT/gpurun_out/am_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT; head -5 $(find gpurun_out/prof_am -name "*kernel_stats*" | head -1) | cut -c1-160
