#!/bin/bash
# the round's closing call: tools/gpu_r4_final.sh (GPU tests, kernel summaries, PMC, SQ, timeline, the default bench line) and then EVERY stream of the FM batch
# and of the AM batch against the unmodified reference (bench.py --oracle-streams 256).   gpurun --timeout 2400 -- 'bash tools/gpu_r4_final2.sh TAG'
cd "${GRAFT_REPO_ROOT:-.}"; TAG=${1:-r04g}
bash tools/gpu_r4_final.sh $TAG
for WL in fm am-cs16; do
  ( time timeout 420 python bench.py --workload $WL --no-extra-legs --oracle-streams 256 --oracle-lost-max 256 --steps 2 --warmup 1 --cpu-baseline-seconds 2 ) > gpurun_out/${TAG}_parity_$WL.log 2>&1; echo "parity $WL rc=$?"
  grep "^{" gpurun_out/${TAG}_parity_$WL.log | tail -1 > gpurun_out/${TAG}_parity_$WL.json
  python - gpurun_out/${TAG}_parity_$WL.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); p = d["parity"]["reference_equality_rank0"]
print(d["ms_per_step"], {k: v for k, v in p.items() if isinstance(v, (int, float))}, d["parity_failures"])
PY
done
