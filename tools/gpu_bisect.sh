#!/bin/bash
# correctness bisect: bench variants, print decoded / exact frame counts
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=$1; shift
i=0
while [ $# -gt 0 ]; do
  v="$1"; shift; i=$((i+1))
  ( timeout 600 python bench.py --no-cpu-baseline --no-extra-legs --no-l2-index --steps 1 --warmup 1 --no-profile $v ) > gpurun_out/${TAG}_$i.log 2>&1
  grep "^{" gpurun_out/${TAG}_$i.log | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); p = d['parity']
    print('[$v]', 'ms', d['ms_per_step'], 'steps', d['config'].get('block_steps_per_pass'), 'decoded', p.get('p1_frames_decoded'), 'exact', p.get('p1_frames_bit_exact_vs_truth'), 'good streams', p.get('streams_locked_and_all_p1_frames_equal_transmitted_bits'), '/', p.get('streams'))
except Exception as ex:
    print('[$v] failed', ex)
"
done
