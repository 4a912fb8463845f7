#!/bin/bash
# round 6: the block step's two kernels side by side (NRSC5HIP_CONCURRENT_STEP, default 1) against one behind the other, one box, alternating; then the kernel timeline
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r06ae}
( time timeout 900 python -m pytest tests -m gpu -x -q -k "host_capture or deferred_seam or dropin or block_exact or push_size or fuzz_two_capture" ) > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${TAG}_tests.log
for ROUND in 1 2 3; do
for CFG in 1 0; do
  echo "== CONCURRENT_STEP=$CFG"
  NRSC5HIP_CONCURRENT_STEP=$CFG timeout 600 python tools/gpu_dropin.py 1 2>&1 | grep "^{" | tee -a gpurun_out/${TAG}_conc$CFG.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); s=d['dropin_strict_delivery']; o=d['dropin']
    print('strict', s['x_realtime'], s['x_realtime_min_max'], s['breakdown_us_per_block'], '| overlapped', o['x_realtime'], o['x_realtime_min_max'], '| equal', d['events_equal'], d['events_equal_strict_delivery'])"
done; done
bash tools/gpu_dropin_trace.sh ${TAG}_trace 2>&1 | tail -32
