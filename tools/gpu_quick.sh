#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 300 python tools/gpu_vit_debug.py ) > gpurun_out/vit_debug.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/vit_debug.log
( timeout 300 python tools/gpu_vit_bench.py ) > gpurun_out/vit_bench.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/vit_bench.log
( timeout 300 python tools/gpu_sync_phases.py ) > gpurun_out/sync_phases.log 2>&1; echo "rc=$?"; tail -12 gpurun_out/sync_phases.log
