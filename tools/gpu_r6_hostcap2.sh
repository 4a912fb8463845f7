#!/bin/bash
# round 6: seam tests, the drop-in leg twice, its kernel timeline
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r06hd}
( time timeout 900 python -m pytest tests -m gpu -x -q -k "host_capture or deferred_seam or dropin or block_exact or push_size or reset_of_a_used or halfband or fuzz_two_capture" ) > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${TAG}_tests.log
for K in 1 2; do timeout 600 python tools/gpu_dropin.py 1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/${TAG}_dropin.log | cut -c1-1200; done
bash tools/gpu_dropin_trace.sh ${TAG}_trace 2>&1 | tail -45
