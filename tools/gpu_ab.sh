#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
for A in 3 4 5; do
  ( NRSC5HIP_NAUX=$A timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > gpurun_out/bench_a$A.log 2>&1
  echo "naux=$A $(grep -o '"value": [0-9.]*' gpurun_out/bench_a$A.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_a$A.log) $(grep -o '"p1_frames_bit_exact_vs_truth": [0-9]*' gpurun_out/bench_a$A.log)"
  grep -o '"device_ms_per_pass": {[^}]*}' gpurun_out/bench_a$A.log
done
