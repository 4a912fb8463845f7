#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 17 python bench.py --no-cpu-baseline --no-l2-index ) > gpurun_out/bench_fwdfix.log 2>&1; echo "bench rc=$?"; grep "^{" gpurun_out/bench_fwdfix.log | tail -1
