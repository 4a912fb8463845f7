#!/bin/bash
# round 4: the streaming seam on the device -- its tests, then the drop-in leg with deferred and with synchronous delivery.
#   gpurun --timeout 900 -- 'bash tools/gpu_r4_dropin.sh TAG'
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r04b}
( time timeout 600 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_parity.py -m gpu -x -q -k "dropin or deferred_seam or poisoned or block_exact or push_size or small_fifo or golden_end_to_end or halfband or mode_switch or l2_feedback" ) > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${TAG}_tests.log
( timeout 300 python tools/gpu_dropin.py 3 ) > gpurun_out/${TAG}_dropin.log 2>&1; echo "dropin rc=$?"; grep "^{" gpurun_out/${TAG}_dropin.log | cut -c1-1200
( NRSC5HIP_SYNC_DELIVERY=1 timeout 300 python tools/gpu_dropin.py 2 ) > gpurun_out/${TAG}_dropin_sync.log 2>&1; echo "dropin(sync delivery) rc=$?"; grep "^{" gpurun_out/${TAG}_dropin_sync.log | cut -c1-1200
