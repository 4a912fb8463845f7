#!/bin/bash
# round 6: the lone stream's chain confined to a few CUs (NRSC5HIP_CHAIN_CUS = 100 * CUs + mask-bit stride), one box, alternating
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r06ad}
for ROUND in 1 2; do
for CFG in 0 801 808 408 1608 3208 1601 3201 6401; do
  echo "== CHAIN_CUS=$CFG"
  NRSC5HIP_CHAIN_CUS=$CFG timeout 600 python tools/gpu_dropin.py 1 2>&1 | grep "^{" | tee -a gpurun_out/${TAG}_cus$CFG.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); s=d['dropin_strict_delivery']; o=d['dropin']
    print('strict', s['x_realtime'], s['x_realtime_min_max'], s['breakdown_us_per_block'], '| overlapped', o['x_realtime'], o['x_realtime_min_max'], '| equal', d['events_equal'], d['events_equal_strict_delivery'])"
done; done
