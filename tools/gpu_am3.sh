#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/gpu_k9_bench.py 2>&1 | grep len
( time timeout 600 python -m pytest tests -m gpu -q -x -k "am" ) > gpurun_out/pytest_am.log 2>&1; echo "pytest am rc=$?"
grep -E "passed|failed" gpurun_out/pytest_am.log
( timeout 300 python tools/gpu_am_bench.py --streams 256 --frames 41 --fmt cs16 ) > gpurun_out/am_bench_cs16.log 2>&1; echo "rc=$?"; grep "^{" gpurun_out/am_bench_cs16.log
