#!/bin/bash
# Round-1 GPU session A: smoke, GPU parity suite, small bench, kernel trace. Run via:
#   gpurun --timeout 1500 -- 'bash tools/gpu_check.sh'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
{
  rocminfo | grep -E "Marketing Name|gfx9" | head -4
  echo "nproc=$(nproc)"; lscpu | grep "Model name"
} > gpurun_out/box.log 2>&1
( time timeout 400 python __graft_entry__.py --smoke ) > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"
tail -4 gpurun_out/smoke.log
( time timeout 900 python -m pytest tests -m gpu -q ) > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/pytest_gpu.log
( time timeout 600 python bench.py --steps 2 --warmup 1 ) > gpurun_out/bench_full.log 2>&1; echo "bench full rc=$?"
tail -3 gpurun_out/bench_full.log
cd /tmp && ( time timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -o r01 -- python $GRAFT_REPO_ROOT/bench.py --streams 64 --seconds 6 --steps 2 --warmup 1 --no-cpu-baseline ) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT; tail -3 gpurun_out/rocprof.log; find gpurun_out/prof_r01 -name "*stats*" | head; ls -la gpurun_out/prof_r01 | head
