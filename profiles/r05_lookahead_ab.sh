#!/bin/bash
# same-box A/B of the step lookahead (NRSC5HIP_TUNE_STEP_LOOKAHEAD): fm with 0 / 2 / 3 steps queued ahead, am-cs16 with 0 / 2, twice, alternating;
# every run compares 32 streams with the unmodified reference (the full comparison is the final records run's)
#   gpurun --timeout 900 -- 'bash profiles/r05_lookahead_ab.sh (needs the patch applied)'
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
run() { python bench.py --workload $1 --no-extra-legs --no-cpu-baseline --no-l2-index --oracle-streams 32 --steps 8 --warmup 2 --tune step_lookahead=$2 2>gpurun_out/la_err.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1 lookahead=$2', d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min_max'], 'failures', d['parity_failures'])"; }
for i in 1 2; do
  for la in 0 2 3; do run fm $la; done
  for la in 0 2; do run am-cs16 $la; done
done
