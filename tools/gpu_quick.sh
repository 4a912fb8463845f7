#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline ) > gpurun_out/bench_bd.log 2>&1
grep -o '"host_ms_per_pass": {[^}]*}' gpurun_out/bench_bd.log; grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_bd.log
