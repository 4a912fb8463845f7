#!/bin/bash
# round 6: launch plans of the fast seam on one box, alternating (strict + overlapped drop-in legs of tools/gpu_dropin.py)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r06ab}
( time timeout 900 python -m pytest tests -m gpu -x -q -k "host_capture or deferred_seam or dropin" ) > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${TAG}_tests.log
for ROUND in 1 2; do
for CFG in "1 0" "0 0" "1 24" "1 28" "0 24" "1 16"; do
  set -- $CFG
  echo "== FOLD_REPORT=$1 EARLY_SYMBOLS=$2"
  NRSC5HIP_FOLD_REPORT=$1 NRSC5HIP_EARLY_SYMBOLS=$2 timeout 600 python tools/gpu_dropin.py 1 2>&1 | grep "^{" | tee -a gpurun_out/${TAG}_fold$1_early$2.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); s=d['dropin_strict_delivery']; o=d['dropin']
    print('strict', s['x_realtime'], s['x_realtime_min_max'], s['breakdown_us_per_block'], '| overlapped', o['x_realtime'], o['x_realtime_min_max'], '| equal', d['events_equal'], d['events_equal_strict_delivery'])"
done; done
