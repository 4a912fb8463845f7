#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 300 python tools/gpu_vit_debug.py ) > gpurun_out/vit_debug.log 2>&1; echo "rc=$?"; tail -40 gpurun_out/vit_debug.log
