#!/bin/bash
# quick A/B on one box: the FM pass without checker legs, each argument one run's extra bench.py flags (quote them), e.g.
#   gpurun --timeout 600 -- 'bash tools/gpu_r5_ab.sh "" "--tune nco_exact=1" ""'
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
[ $# -eq 0 ] && set -- ""
for A in "$@"; do
  python bench.py --no-extra-legs --no-cpu-baseline --no-l2-index --steps 6 --warmup 2 $A 2>gpurun_out/ab_err.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('[%s]' % '$A', d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min_max'], 'steps', d['config']['block_steps_per_pass'], {k: v for k, v in r['device_ms_per_pass'].items()}, 'dominant launch us', round(r['avg_launch_ms']*1e3,1))"
done
