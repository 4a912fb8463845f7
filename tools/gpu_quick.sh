#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_dropin.py -m gpu -q ) > gpurun_out/dropin.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/dropin.log
