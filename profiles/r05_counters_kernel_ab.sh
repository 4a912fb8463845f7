#!/bin/bash
# same-box A/B of NRSC5HIP_TUNE_COUNTERS_KERNEL (burst counters by kernel instead of hipMemcpyAsync / hipMemsetAsync): fm, am-cs16, mixed, alternating; 32 streams
# against the unmodified reference per run (the full comparison is the final records run's)
#   gpurun --timeout 900 -- 'bash tools/gpu_r5_counters.sh'
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
run() { python bench.py --workload $1 --no-extra-legs --no-cpu-baseline --no-l2-index --oracle-streams 32 --steps 8 --warmup 2 --tune counters_kernel=$2 2>gpurun_out/ck_err.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1 counters_kernel=$2', d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min_max'], 'failures', d['parity_failures'])"; }
for i in 1 2 3; do for v in 0 1; do run fm $v; done; done
for i in 1 2; do for v in 0 1; do run am-cs16 $v; done; done
for v in 0 1; do run mixed $v; done
