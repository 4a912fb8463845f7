#!/bin/bash
# quick round trip: GPU parity suite, one bench line without the CPU legs, kernel timeline of one pass
# gpurun --timeout 900 -- 'bash tools/gpu_quick.sh TAG'
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-q}
( time timeout 600 python -m pytest tests -m gpu -x -q ) > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/${TAG}_tests.log
( time timeout 300 python bench.py --no-cpu-baseline ) > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc=$?"
grep "^{" gpurun_out/${TAG}_bench.log | tail -1 > gpurun_out/${TAG}_bench.json
python - "$TAG" <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/{sys.argv[1]}_bench.json"))
    print({k: d.get(k) for k in ("value", "x_realtime", "ms_per_step")})
    r = d["roofline"]; print("roofline", r["kernel"], r["frac"], r["avg_launch_ms"], r.get("device_ms_per_pass"), r.get("host_ms_per_pass"))
    for k in ("single_stream", "in_order"):
        if k in d: print("  ", k, d[k])
    print("  parity", {k: v for k, v in d["parity"].items() if not isinstance(v, (dict, str))}, d["parity"].get("reference_equality_rank0"))
    print("  config", d["config"].get("block_steps_per_pass"))
except Exception as ex:
    print("no json", ex); print(open(f"gpurun_out/{sys.argv[1]}_bench.log").read()[-3000:])
PY
bash tools/gpu_trace.sh ${TAG}_trace > /dev/null 2>&1; head -24 gpurun_out/${TAG}_trace_summary.txt; grep -A40 "decode kernels" gpurun_out/${TAG}_trace_summary.txt
