#!/bin/bash
# more slices of the GPU fuzz under other seed counters (no 2048-stream point)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r06J}
for C in ${2:-40 41 42 43}; do
  echo $C > tests/fuzz_seed_counter.txt
  ( time timeout 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -s ) > gpurun_out/${TAG}_fuzz_counter$C.log 2>&1; echo "fuzz counter $C rc=$?"; grep -a "fuzz" gpurun_out/${TAG}_fuzz_counter$C.log | grep -v "^E\|assert\|def \|msg\|test_gpu\|FAILED\|^___" | cut -c1-420
done
