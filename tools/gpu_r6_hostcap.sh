#!/bin/bash
# round 6, host-resident capture of the fast seam: the seam tests, then the drop-in leg with the capture (default) and with the FIFO seam (NRSC5HIP_HOST_CAPTURE=0), same box
#   gpurun --timeout 1500 -- 'bash tools/gpu_r6_hostcap.sh TAG'
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r06hc}
( time timeout 900 python -m pytest tests -m gpu -x -q -k "host_capture or deferred_seam or dropin or block_exact or push_size or reset_of_a_used or halfband" ) > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${TAG}_tests.log
for HC in 1 0 1 0; do
  echo "== NRSC5HIP_HOST_CAPTURE=$HC"; NRSC5HIP_HOST_CAPTURE=$HC timeout 600 python tools/gpu_dropin.py 1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/${TAG}_dropin_hc$HC.log | cut -c1-1500
done
