#!/bin/bash
# AM path on the device: parity tests + the public-API drop-in.  gpurun --timeout 900 -- 'bash tools/gpu_am.sh'
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 700 python -m pytest tests -m gpu -q -x -k "am or selftest or dropin" ) > gpurun_out/pytest_am.log 2>&1; echo "pytest am rc=$?"
tail -30 gpurun_out/pytest_am.log
