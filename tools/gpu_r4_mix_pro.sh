#!/bin/bash
# k_mixfft with its prologue in one burst: phase timers (diagnostic build), parity tests of the symbol kernel, A/B of the knob forms
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
python tools/gpu_mixfft_phases.py 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "256_lanes or fft2048 or symbol_kernel or zero_copy or golden" 2>&1 | tail -3
bash tools/gpu_r4_ab.sh "$@"
