#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 300 python tools/gpu_vit_bench.py ) > gpurun_out/vit_bench.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/vit_bench.log
