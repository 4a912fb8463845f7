#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 600 python -m pytest tests -m gpu -q -k "l2" ) > gpurun_out/pytest_l2.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/pytest_l2.log
