#!/bin/bash
# diagnostic: captures of given global stream ids of a tests/test_gpu_batch256.py-style batch (the 256 streams from BASE) as .npy under gpurun_out/, with the batch's verdict
#   gpurun -- 'bash tools/gpu_dump_streams.sh BASE "id id ..."'
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
NRSC5_DUMP_STREAMS=$(echo $2 | tr ' ' ',') timeout 300 python tools/gpu_cfo_batch.py $1 1 1 2>&1 | grep -a "^{" | cut -c1-600
ls -la gpurun_out/*.npy
