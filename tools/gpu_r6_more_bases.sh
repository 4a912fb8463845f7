#!/bin/bash
# the bench's own batches on further stream bases (other CFO / offset / noise / channel draws), every stream against the unmodified reference, all three workloads
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r06m}
for SPEC in "fm 1024" "fm 1280" "fm 1536" "fm 1792" "am-cs16 1024" "am-cs16 1280" "am-cu8 1024" "mixed 1024" "mixed 1280"; do
  set -- $SPEC; WL=$1; B=$2
  ( timeout 600 python bench.py --workload $WL --stream-base $B --no-extra-legs --steps 2 --warmup 1 --cpu-baseline-seconds 2 ) > gpurun_out/${TAG}_${WL}_base$B.log 2>gpurun_out/${TAG}_${WL}_base$B.err; rc=$?
  python - $WL $B $rc gpurun_out/${TAG}_${WL}_base$B.log <<'PY'
import json, sys
wl, b, rc, path = sys.argv[1:5]
try:
    d = json.loads([l for l in open(path) if l.startswith("{")][-1]); p = d["parity"]["reference_equality_rank0"]
    subs = [p] if "streams_compared" in p else [v for v in p.values() if isinstance(v, dict) and "streams_compared" in v]
    tot = lambda k: sum(x.get(k, 0) for x in subs)
    print(wl, "base", b, "rc", rc, d["ms_per_step"], "compared", tot("streams_compared"), "strict", tot("streams_equal_under_the_strict_rule"), "transient", tot("streams_with_transient_loop_state_deviation"),
          "mer_exempt", tot("mer_reports_beyond_1e-4_within_0.01dB"), tot("mer_reports_below_0dB_within_0.1dB"), "of", tot("mer_values_compared"), "failures", d["parity_failures"], [x.get("streams_failing_by_class") for x in subs], [x.get("transient_details", [])[:2] for x in subs])
except Exception as ex:
    print(wl, "base", b, "rc", rc, "no line", ex)
PY
done
