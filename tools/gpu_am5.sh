#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/gpu_k9_bench.py 2>&1 | grep "len 24000"
( timeout 600 python -m pytest tests -m gpu -q -x -k "am" ) > gpurun_out/pytest_am.log 2>&1; grep -E "passed|failed" gpurun_out/pytest_am.log
for na in 4 5; do
NRSC5HIP_NAUX_AM=$na timeout 300 python tools/gpu_am_bench.py --streams 256 --frames 41 --fmt cs16 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('naux_am', $na, j['x_realtime'], j['ms_per_pass'], j['device_ms_per_pass'])
"
done
