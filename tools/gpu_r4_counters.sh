#!/bin/bash
# PMC traffic + SQ issue counters + the timeline of one FM pass (after the collectors learnt to read template instances' names)
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r04y}
bash tools/gpu_pmc.sh fm 2>&1 | tail -2 | cut -c1-600
bash tools/gpu_sq.sh fm 2>&1 | tail -3 | cut -c1-300
bash tools/gpu_trace.sh ${TAG}_trace > /dev/null 2>&1; head -24 gpurun_out/${TAG}_trace_summary.txt | cut -c1-200
