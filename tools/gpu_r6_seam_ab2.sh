#!/bin/bash
# round 6: the host-resident capture's buffer size and mapping, one box, alternating (strict + overlapped drop-in legs)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r06ac}
( time timeout 900 python -m pytest tests -m gpu -x -q -k "host_capture or deferred_seam" ) > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${TAG}_tests.log
for ROUND in 1 2; do
for CFG in "1 0" "1 1" "1 2" "1024 0" "4096 0" "65536 0" "4096 1"; do
  set -- $CFG
  echo "== HOST_CAPTURE=$1 MEM=$2"
  NRSC5HIP_HOST_CAPTURE=$1 NRSC5HIP_HOST_CAPTURE_MEM=$2 timeout 600 python tools/gpu_dropin.py 1 2>&1 | grep "^{" | tee -a gpurun_out/${TAG}_cap$1_mem$2.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); s=d['dropin_strict_delivery']; o=d['dropin']
    print('strict', s['x_realtime'], s['x_realtime_min_max'], s['breakdown_us_per_block'], '| overlapped', o['x_realtime'], o['x_realtime_min_max'], '| equal', d['events_equal'], d['events_equal_strict_delivery'])"
done; done
