#!/bin/bash
# round-end verification: L2 index tests + timing, full suite + smoke + bench, bench with the index fused into the pipeline.
# gpurun --timeout 1500 -- 'bash tools/gpu_final.sh'
cd "${GRAFT_REPO_ROOT:-.}"
bash tools/gpu_l2.sh
( time timeout 300 python bench.py --l2-index-inline --no-cpu-baseline ) > gpurun_out/bench_l2_inline.log 2>&1; echo "bench inline rc=$?"
grep "^{" gpurun_out/bench_l2_inline.log | tail -1
