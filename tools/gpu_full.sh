#!/bin/bash
# full GPU suite + smoke + headline bench.  gpurun --timeout 1500 -- 'bash tools/gpu_full.sh'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 300 python __graft_entry__.py --smoke ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
( time timeout 900 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/pytest_gpu.log
( time timeout 400 python bench.py ) > gpurun_out/bench_full.log 2>&1; echo "bench rc=$?"
grep "^{" gpurun_out/bench_full.log | tail -1
