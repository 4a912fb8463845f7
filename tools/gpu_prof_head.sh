#!/bin/bash
# per-kernel rocprofv3 summary of the headline bench at HEAD, with the L2 index fused into the decode streams
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=$PWD; mkdir -p gpurun_out; rm -rf gpurun_out/prof_head
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_head -o head -- python $R/bench.py --l2-index-inline --no-cpu-baseline --steps 2 --warmup 1 ) > gpurun_out/prof_head.log 2>&1; echo "rc=$?"
grep "^{" gpurun_out/prof_head.log | tail -1 | cut -c1-300
f=$(find gpurun_out/prof_head -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/head_kernel_stats.csv && cat "$f"
find gpurun_out/prof_head -name '*kernel_trace.csv' -delete 2>/dev/null
