#!/bin/bash
# round 6: the fuzz batch whose three streams failed (100256..100511), under the arithmetic variants that separate the candidates: default; exact oscillator in every block;
# + every complex product of mix / FFT unfused (the CPU twin's sequence); double-precision sine / cosine in the closed-form phasor.  Captures of the failing streams dumped.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r06d}
NRSC5_DUMP_STREAMS=100465,100357,100262 timeout 300 python tools/gpu_cfo_batch.py 100256 0,3 1,2 > gpurun_out/${TAG}_default.txt 2>&1; echo "default rc=$?"; cut -c1-700 gpurun_out/${TAG}_default.txt | grep -a "^{"
timeout 300 python tools/gpu_cfo_batch.py 100256 0,3 2 --unfused > gpurun_out/${TAG}_unfused.txt 2>&1; echo "unfused rc=$?"; cut -c1-700 gpurun_out/${TAG}_unfused.txt | grep -a "^{"
timeout 300 python tools/gpu_cfo_batch.py 100256 0 1 --accurate > gpurun_out/${TAG}_acctrig.txt 2>&1; echo "acctrig rc=$?"; cut -c1-700 gpurun_out/${TAG}_acctrig.txt | grep -a "^{"
ls -la gpurun_out/*.npy
