#!/bin/bash
# same-box A/B of two builds of the library: nrsc5_amd/libnrsc5hip_base.so (built from an earlier commit, see DESIGN (f)) against the tree's own, alternating.
#   gpurun --timeout 600 -- 'bash tools/gpu_r5_ab2.sh 3 "--tune nco_exact=0"'
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; N=${1:-2}; FLAGS=${2:-}
for i in $(seq $N); do
  for L in base new; do
    if [ $L = base ]; then export NRSC5HIP_AB_LIB=$PWD/nrsc5_amd/libnrsc5hip_base.so; else unset NRSC5HIP_AB_LIB; fi
    python bench.py --no-extra-legs --no-cpu-baseline --no-l2-index --steps 6 --warmup 2 $FLAGS 2>gpurun_out/ab_err.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$L', d['ms_per_step'], d['ms_per_step_median'], {k: v for k, v in r['device_ms_per_pass'].items() if k in ('mixfft','sync','acquire','prepare')}, round(r['avg_launch_ms']*1e3,1))"
  done
done
