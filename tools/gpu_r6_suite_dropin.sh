#!/bin/bash
# the whole GPU suite as the driver runs it, then the drop-in leg twice
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r06s}
( time timeout 1700 python -m pytest tests -m gpu -x -q ) > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${TAG}_tests.log
for K in 1 2; do timeout 600 python tools/gpu_dropin.py 1 2>&1 | grep "^{" | tee -a gpurun_out/${TAG}_dropin.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); s=d['dropin_strict_delivery']; o=d['dropin']
    print('strict', s['x_realtime'], s['x_realtime_min_max'], s['breakdown_us_per_block'], '| overlapped', o['x_realtime'], o['x_realtime_min_max'], '| equal', d['events_equal'], d['events_equal_strict_delivery'])"; done
