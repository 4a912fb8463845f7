#!/bin/bash
# same-box A/B of NRSC5HIP_TUNE_DECODE_SIDE (a window's short decode kernels on a side stream): fm and mixed, alternating, 32 streams against the reference per run
#   gpurun --timeout 900 -- 'bash tools/gpu_r5_side.sh'
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
run() { python bench.py --workload $1 --no-extra-legs --no-cpu-baseline --no-l2-index --oracle-streams 32 --steps 8 --warmup 2 --tune decode_side=$2 2>gpurun_out/side_err.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$1 decode_side=$2', d['ms_per_step'], d['ms_per_step_median'], d['ms_per_step_min_max'], 'failures', d['parity_failures'])"; }
for i in 1 2 3; do for v in 0 1; do run fm $v; done; done
for v in 0 1; do run mixed $v; done
