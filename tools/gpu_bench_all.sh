#!/bin/bash
# the PMC traffic of the headline workload first (so that the bench lines carry the counters of this very source), then every
# bench workload once (same JSON schema)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
TAG=${1:-r02}
bash tools/gpu_pmc.sh fm > gpurun_out/${TAG}_pmc.log 2>&1; tail -3 gpurun_out/${TAG}_pmc.log
cp gpurun_out/traffic_fm.json profiles/traffic_latest.json
for wl in fm am-cs16 am-cu8 mixed; do
  ( time timeout 400 python bench.py --workload $wl ) > gpurun_out/${TAG}_bench_${wl}.log 2>&1; echo "$wl rc=$?"
  grep "^{" gpurun_out/${TAG}_bench_${wl}.log | tail -1 > gpurun_out/${TAG}_bench_${wl}.json
  python - "$TAG" "$wl" <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/{sys.argv[1]}_bench_{sys.argv[2]}.json"))
    print({k: d.get(k) for k in ("value", "x_realtime", "ms_per_step")}, (d.get("roofline") or {}).get("kernel"), (d.get("roofline") or {}).get("frac"), "cpu", (d.get("cpu_baseline") or {}).get("x_realtime"), ((d.get("cpu_baseline") or {}).get("all_cores") or {}).get("x_realtime"))
    for k in ("single_stream", "in_order"):
        if k in d: print("  ", k, d[k])
    print("  parity", {k: v for k, v in d["parity"].items() if not isinstance(v, (dict, str))})
except Exception as ex:
    print("no json", ex); print(open(f"gpurun_out/{sys.argv[1]}_bench_{sys.argv[2]}.log").read()[-1500:])
PY
done
