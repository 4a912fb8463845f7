#!/bin/bash
# round 4 work loop: full GPU suite, drop-in timeline, k_sync phases, FM bench line without the extra legs.   gpurun --timeout 1200 -- 'bash tools/gpu_r4_step.sh TAG'
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r04s}
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${TAG}_tests.log | cut -c1-300
bash tools/gpu_dropin_trace.sh ${TAG}_dtrace 2>&1 | tail -45 | cut -c1-200
python tools/gpu_sync_phases_batch.py 2>&1 | tail -9
bash tools/gpu_r4_ab.sh ${AB:-"-"} 2>&1 | tail -4 | cut -c1-400
