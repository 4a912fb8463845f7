#!/bin/bash
# the whole GPU suite on the tree with the new defaults, then the default bench line
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r06f}
( time timeout 1800 python -m pytest tests -m gpu -q -s ) > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; grep -a "fuzz\|flow bursts\|passed\|failed\|^FAILED\|^ERROR" gpurun_out/${TAG}_tests.log | cut -c1-500 | tail -14
( time timeout 900 python bench.py ) > gpurun_out/${TAG}_bench.log 2>gpurun_out/${TAG}_bench.err; echo "bench rc=$?"; tail -1 gpurun_out/${TAG}_bench.log | cut -c1-600
