#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 600 python -m pytest tests -m gpu -q -x -k "am or mode_switch or dropin or pids_crc" ) > gpurun_out/pytest_am.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_am.log
( timeout 300 python tools/gpu_mixed_bench.py ) > gpurun_out/mixed.log 2>&1; grep "^{" gpurun_out/mixed.log
( timeout 300 python tools/gpu_am_bench.py --streams 256 --frames 41 --fmt cs16 ) 2>/dev/null | grep "^{" | cut -c1-330
