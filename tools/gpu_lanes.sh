#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
( time timeout 900 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
for L in 1 2 3 4; do
  ( NRSC5HIP_LANES=$L timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline ) > gpurun_out/bench_l$L.log 2>&1
  echo "lanes=$L $(grep -o '"value": [0-9.]*' gpurun_out/bench_l$L.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_l$L.log) $(grep -o '"p1_frames_bit_exact_vs_truth": [0-9]*' gpurun_out/bench_l$L.log)"
  grep -o '"device_ms_per_pass": {[^}]*}' gpurun_out/bench_l$L.log
done
