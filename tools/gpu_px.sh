#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 600 python -m pytest tests -m gpu -q -k "extended or dropin or golden" ) > gpurun_out/pytest_px.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/pytest_px.log
