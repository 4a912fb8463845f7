#!/bin/bash
# round 2: GPU parity suite, headline bench, rocprofv3 kernel stats of the same command
# gpurun --timeout 900 -- 'bash tools/gpu_r2_check.sh'
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-r02a}
( time timeout 600 python -m pytest tests -m gpu -x -q ) > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/${TAG}_tests.log
( time timeout 300 python bench.py ) > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc=$?"
grep "^{" gpurun_out/${TAG}_bench.log | tail -1 > gpurun_out/${TAG}_bench.json; cut -c1-1500 gpurun_out/${TAG}_bench.json
rm -rf gpurun_out/${TAG}_prof
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o p -- python $R/bench.py --no-cpu-baseline --no-extra-legs --steps 3 --warmup 1 ) > gpurun_out/${TAG}_prof.log 2>&1; echo "prof rc=$?"
f=$(find gpurun_out/${TAG}_prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_kernel_stats.csv && head -30 "$f"
grep "^{" gpurun_out/${TAG}_prof.log | tail -1 > gpurun_out/${TAG}_bench_under_rocprof.json
find gpurun_out/${TAG}_prof -name '*kernel_trace.csv' -delete 2>/dev/null
python tools/gpu_vit_bench.py > gpurun_out/${TAG}_vit_bench.txt 2>&1; cat gpurun_out/${TAG}_vit_bench.txt
