#!/bin/bash
# after an arithmetic change in the symbol kernel: the parity tests that see its floats, then the FM pass
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "256_lanes or fft2048 or symbol_kernel or zero_copy or golden or oracle_end_to_end or batch_equals_streaming or impaired or channel" 2>&1 | tail -3
bash tools/gpu_r4_ab.sh "$@"
