#!/bin/bash
# slices of the GPU fuzz under the seed counters given:  gpurun --timeout 1800 -- 'bash tools/gpu_r6_fuzz_only.sh TAG "6 7 8"'
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r06i}
for C in ${2:-6 7 8 9 10}; do
  echo $C > tests/fuzz_seed_counter.txt
  ( time timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -s ) > gpurun_out/${TAG}_fuzz_counter$C.log 2>&1; echo "fuzz counter $C rc=$?"; grep -a "fuzz" gpurun_out/${TAG}_fuzz_counter$C.log | cut -c1-420
done
