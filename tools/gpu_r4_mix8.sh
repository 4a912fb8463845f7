#!/bin/bash
# the 256-lane symbol kernel: its parity tests, then A/B against the default on the FM pass
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "256_lanes or fft2048 or selftest" 2>&1 | tail -5
bash tools/gpu_r4_ab.sh "$@"
