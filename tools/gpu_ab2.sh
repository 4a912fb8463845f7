#!/bin/bash
# bench lines for a list of bench.py argument sets: gpurun -- 'bash tools/gpu_ab2.sh TAG "--no-profile" "--steps 5"'
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-ab2}; shift
i=0
for v in "" "$@"; do
  ( timeout 300 python bench.py --no-cpu-baseline --no-extra-legs $v ) > gpurun_out/${TAG}_v$i.log 2>&1
  grep "^{" gpurun_out/${TAG}_v$i.log | tail -1 > gpurun_out/${TAG}_v$i.json
  python - "$TAG" "v$i" "$v" <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/{sys.argv[1]}_{sys.argv[2]}.json")); r = d["roofline"]
    print(f"[{sys.argv[3]}]", d["ms_per_step"], "steps", d["config"].get("block_steps_per_pass"), "fwd", r.get("avg_launch_ms"), r.get("device_ms_per_pass"), r.get("host_ms_per_pass"))
except Exception as ex:
    print(sys.argv[2], "no json", ex); print(open(f"gpurun_out/{sys.argv[1]}_{sys.argv[2]}.log").read()[-1500:])
PY
  i=$((i+1))
done
