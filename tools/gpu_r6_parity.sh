#!/bin/bash
# round 6, first GPU call: the GPU suite on the tree with ref_sincosf / exact loops, then the two 256-stream CFO-search batches under
# NRSC5HIP_TUNE_LOOP_EXACT 0 / 1 / 2 (NCO policy 0 and 1), every stream against the unmodified reference.
#   gpurun --timeout 2400 -- 'bash tools/gpu_r6_parity.sh TAG'
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r06a}
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/${TAG}_tests.log
( time timeout 900 python tools/gpu_cfo_batch.py 0,256 0 0,1,2 ) > gpurun_out/${TAG}_cfo_batch_policy0.txt 2>&1; echo "cfo p0 rc=$?"; cut -c1-400 gpurun_out/${TAG}_cfo_batch_policy0.txt
( time timeout 600 python tools/gpu_cfo_batch.py 0,256 1 1,2 ) > gpurun_out/${TAG}_cfo_batch_policy1.txt 2>&1; echo "cfo p1 rc=$?"; cut -c1-400 gpurun_out/${TAG}_cfo_batch_policy1.txt
( time timeout 600 python bench.py ) > gpurun_out/${TAG}_bench.log 2>gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; tail -1 gpurun_out/${TAG}_bench.log | cut -c1-1500
