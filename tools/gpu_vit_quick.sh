#!/bin/bash
# forward-pass change: trellis micro-benchmark + the bit-exact Viterbi / end-to-end parity tests that exercise it
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 60 python tools/gpu_vit_bench.py ) > gpurun_out/vit_bench.log 2>&1; echo "bench rc=$?"; cat gpurun_out/vit_bench.log | tail -4
( timeout 150 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "viterbi_exact or viterbi_roundtrip or golden_end_to_end or full_size_truth" ) > gpurun_out/pytest_vit.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_vit.log
