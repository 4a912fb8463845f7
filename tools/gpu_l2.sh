#!/bin/bash
# L2 index: parity tests first, kernel timing under rocprofv3, then the full suite + smoke + bench.
# gpurun --timeout 1500 -- 'bash tools/gpu_l2.sh'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k l2_index ) > gpurun_out/pytest_l2.log 2>&1; echo "l2 pytest rc=$?"
tail -12 gpurun_out/pytest_l2.log
rm -rf gpurun_out/l2prof
R=$PWD; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/l2prof -o l2 -- python $R/tools/gpu_l2_bench.py ) > gpurun_out/l2_bench.log 2>&1; echo "l2 bench rc=$?"
grep "x " gpurun_out/l2_bench.log
f=$(find gpurun_out/l2prof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/l2_kernel_stats.csv && cat "$f"
find gpurun_out/l2prof -name '*kernel_trace.csv' -delete 2>/dev/null
bash tools/gpu_full.sh
