#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 600 python -m pytest tests -m gpu -q -x -k "am" ) > gpurun_out/pytest_am.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_am.log
for na in 3 4; do
NRSC5HIP_NAUX=$na timeout 300 python tools/gpu_am_bench.py --streams 256 --frames 41 --fmt cs16 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('naux', $na, j['x_realtime'], j['ms_per_pass'], j['device_ms_per_pass'])
"
done
cd /tmp && ( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_am -o am -- python $GRAFT_REPO_ROOT/tools/gpu_am_bench.py --streams 256 --frames 41 --fmt cs16 --steps 2 ) > $GRAFT_REPO_ROOT/gpurun_out/am_rocprof.log 2>&1
cd $GRAFT_REPO_ROOT; head -5 $(find gpurun_out/prof_am -name "*kernel_stats*" | head -1) | cut -c1-200
