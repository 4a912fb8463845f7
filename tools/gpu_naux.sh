#!/bin/bash
# decode-stream count / priority sweep of the headline bench (no CPU baseline, no L2 post-pass)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for n in 2 3 4 5; do
  NRSC5HIP_NAUX=$n python bench.py --no-cpu-baseline --no-l2-index --steps 3 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('naux', $n, 'ms_per_step', d['ms_per_step'], 'fwd avg', d['roofline']['avg_launch_ms'], 'steps', d['config']['block_steps_per_pass'])"
done
for p in 1 2; do
  NRSC5HIP_PRIO_FWD=$p python bench.py --no-cpu-baseline --no-l2-index --steps 3 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prio_fwd', $p, 'ms_per_step', d['ms_per_step'], 'fwd avg', d['roofline']['avg_launch_ms'])"
done
