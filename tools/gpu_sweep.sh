#!/bin/bash
# streams-per-GPU sweep of the fm workload (2048 = the N = 1 point of configs[3]'s strong-scaling curve)
# gpurun --timeout 1500 -- 'bash tools/gpu_sweep.sh TAG 512 1024 2048'
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-sweep}; shift
for S in "$@"; do
  ( time timeout 900 python bench.py --streams $S --no-cpu-baseline --no-extra-legs --no-l2-index --steps 3 --warmup 1 ) > gpurun_out/${TAG}_s$S.log 2>&1
  grep "^{" gpurun_out/${TAG}_s$S.log | tail -1 > gpurun_out/${TAG}_s$S.json
  python - "$TAG" "$S" <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/{sys.argv[1]}_s{sys.argv[2]}.json")); r = d["roofline"]
    print(f"streams {sys.argv[2]:>5s}: {d['ms_per_step']:9.3f} ms/pass  {d['value']:12.1f} MS/s  {d['x_realtime']:10.1f} x  steps {d['config']['block_steps_per_pass']}  dom {r['kernel']} {r['avg_launch_ms']} ms  exact {d['parity']['p1_frames_bit_exact_vs_truth']}/{d['parity']['p1_frames_decoded']}  gen {d['gen_seconds']} s  dev {r['device_ms_per_pass']}")
except Exception as ex:
    print(sys.argv[2], "failed", ex); print(open(f"gpurun_out/{sys.argv[1]}_s{sys.argv[2]}.log").read()[-1500:])
PY
done
