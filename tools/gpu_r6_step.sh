#!/bin/bash
# round 6 work loop, one box: GPU tests, the default bench line (every stream of fm / am-cs16 / mixed against the unmodified reference, both drop-in
# delivery modes), then the FM batch on further seeds (--stream-base: another 256 streams each).   gpurun --timeout 1500 -- 'bash tools/gpu_r6_step.sh TAG [bases]'
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r06a}; BASES=${2:-"256 512"}
if [ "${3:-}" != notests ]; then ( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${TAG}_tests.log; fi
( time timeout 900 python bench.py ) > gpurun_out/${TAG}_bench.log 2>&1; echo "bench rc=$?"
grep "^{" gpurun_out/${TAG}_bench.log | tail -1 > gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.log | cut -c1-600
python - "$TAG" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/{sys.argv[1]}_bench.json")); r = d["roofline"]
print(d["ms_per_step"], d["x_realtime"], r["kernel"], r["frac"], d["parity_failures"])
num = lambda p: {k: v for k, v in p.items() if isinstance(v, (int, float, bool)) or k in ("streams_failing_by_class",)}
print("fm parity", num(d["parity"]["reference_equality_rank0"]))
print("  first_diffs", d["parity"]["reference_equality_rank0"]["first_diffs"][:3], d["parity"]["reference_equality_rank0"]["transient_details"][:6])
print("single", d["single_stream"]["x_realtime"], "dropin", d["dropin"]["dropin"], "strict", d["dropin"]["dropin_strict_delivery"], d["dropin"]["events_equal"], d["dropin"]["events_equal_strict_delivery"], "inorder", d["in_order"]["ms_per_step"])
for k, v in d["config4"].items():
    p = v["parity"]["reference_equality_rank0"]
    print(k, v["ms_per_step"], v["x_realtime"], num(p) if "streams_compared" in p else {kk: num(vv) for kk, vv in p.items()})
PY
for B in $BASES; do
  ( time timeout 420 python bench.py --stream-base $B --no-extra-legs --steps 2 --warmup 1 --cpu-baseline-seconds 2 ) > gpurun_out/${TAG}_parity_base$B.log 2>&1; echo "parity base $B rc=$?"
  grep "^{" gpurun_out/${TAG}_parity_base$B.log | tail -1 > gpurun_out/${TAG}_parity_base$B.json
  python - gpurun_out/${TAG}_parity_base$B.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); p = d["parity"]["reference_equality_rank0"]
print(d["ms_per_step"], {k: v for k, v in p.items() if isinstance(v, (int, float, bool)) or k == "streams_failing_by_class"}, p["first_diffs"][:3], p["transient_details"][:6], d["parity_failures"])
PY
done
