#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
( NRSC5HIP_LANES=2 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "batch_equals" ) > gpurun_out/lanes2.log 2>&1; echo "lanes2 rc=$?"; tail -12 gpurun_out/lanes2.log
( timeout 300 python tools/gpu_sync_phases.py ) > gpurun_out/sync_phases.log 2>&1; echo "rc=$?"; tail -10 gpurun_out/sync_phases.log
