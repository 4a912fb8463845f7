#!/bin/bash
# quick re-check after host-side changes: L2 index tests + smoke
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "l2_index" ) > gpurun_out/pytest_l2.log 2>&1; echo "l2 pytest rc=$?"; tail -3 gpurun_out/pytest_l2.log
( timeout 120 python __graft_entry__.py --smoke ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
