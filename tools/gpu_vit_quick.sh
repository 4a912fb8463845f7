#!/bin/bash
# quick trellis check on the GPU box: stage-level exactness + micro-benchmark (+ optional extra pytest -k expression)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
TAG=${1:-vit}
( timeout 600 python -m pytest tests -m gpu -x -q -k "${2:-viterbi or selftest or golden_end_to_end}" ) > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${TAG}_tests.log
python tools/gpu_vit_bench.py > gpurun_out/${TAG}_vit_bench.txt 2>&1; cat gpurun_out/${TAG}_vit_bench.txt
