#!/bin/bash
# round 6: the lone stream's block step on the streaming seam -- seam + decoder tests, phase timers of its two kernels, the drop-in leg three times
#   gpurun --timeout 1500 -- 'bash tools/gpu_r6_pids.sh TAG'
# (profiles/r06_seam_launch_plans.txt (5) was taken with this script while the tree also held a position-hinted symbol kernel, NRSC5HIP_SEAM_HINT=1 / 0: measured, no gain, never committed)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r06p}
( time timeout 900 python -m pytest tests -m gpu -x -q -k "viterbi or pids or dropin or deferred_seam or host_capture or golden or oracle_end_to_end or selftest" ) > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/${TAG}_tests.log
python tools/gpu_seam_phases.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_seam_phases.log | head -19
for K in 1 2 3; do timeout 600 python tools/gpu_dropin.py 1 2>&1 | grep "^{" | tee -a gpurun_out/${TAG}_dropin.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); s=d['dropin_strict_delivery']; o=d['dropin']
    print('strict', s['x_realtime'], s['x_realtime_min_max'], s['breakdown_us_per_block'], '| overlapped', o['x_realtime'], o['x_realtime_min_max'], '| equal', d['events_equal'], d['events_equal_strict_delivery'])"; done
