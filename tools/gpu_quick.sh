#!/bin/bash
# quick GPU loop: full GPU test suite + default bench line (summary only).   gpurun --timeout 900 -- 'bash tools/gpu_quick.sh TAG [bench args]'
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-q}; shift
( time timeout 900 python -m pytest tests -m gpu -q ) > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/${TAG}_tests.log | cut -c1-300
( time timeout 600 python bench.py "$@" ) > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc=$?"
grep "^{" gpurun_out/${TAG}_bench.log | tail -1 > gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.log | cut -c1-300
python - "$TAG" <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/{sys.argv[1]}_bench.json"))
    r = d["roofline"]
    print("ms_per_step", d["ms_per_step"], "x", d["x_realtime"], "steps", d["config"].get("block_steps_per_pass"), "fwd ms", r["avg_launch_ms"], "frac", r["frac"], "valu", (r.get("valu") or {}).get("frac"))
    print("device_ms", r.get("device_ms_per_pass"), "host", r.get("host_ms_per_pass"))
    for k in ("single_stream", "in_order", "dropin", "parity_failures"):
        print(k, json.dumps(d.get(k))[:600])
    pe = d["parity"].get("reference_equality_rank0") or {}
    print("parity", {k: pe.get(k) for k in ("kind", "streams_with_lost_sync_this_pass", "lost_sync_streams_equal", "other_streams_equal", "frames_exempt_cber", "exempt_max_bit_differences", "first_diffs", "error")})
    for k, v in (d.get("config4") or {}).items():
        print("config4", k, v["ms_per_step"], v["x_realtime"])
    print("extra err", d.get("extra_legs_error"))
except Exception as ex:
    print("no bench json", ex)
PY
