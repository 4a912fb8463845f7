#!/bin/bash
# same-box A/B of bench.py tune sets:  gpurun -- 'bash tools/gpu_r6_ab_generic.sh TAG "setA" "setB" ...'   (a set = space-separated knob=value pairs, "-" = defaults)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=$1; shift
for ROUND in 1 2; do
for SET in "$@"; do
  ARGS=""; if [ "$SET" != "-" ]; then for kv in $SET; do ARGS="$ARGS --tune $kv"; done; fi
  ( timeout 420 python bench.py --workload ${WL:-fm} --no-extra-legs --steps 8 --warmup 2 $ARGS ) > gpurun_out/${TAG}_ab.log 2>gpurun_out/${TAG}_ab.err
  python - "$SET" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open("gpurun_out/" + __import__("os").environ.get("TAGX", "") + "") if False] or "null")
except Exception:
    d = None
PY
  python - "$SET" "$TAG" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(f"gpurun_out/{sys.argv[2]}_ab.log") if l.startswith("{")][-1])
    r = d["parity"]["reference_equality_rank0"]
    print(f"[{sys.argv[1]}]", d["ms_per_step"], d["ms_per_step_min_max"], "failures", d["parity_failures"], "strict", r.get("streams_equal_under_the_strict_rule"), "transient", r.get("streams_with_transient_loop_state_deviation"), {k: v for k, v in d["roofline"]["device_ms_per_pass"].items() if k in ("mixfft", "sync", "p1_viterbi", "p1_traceback", "am", "am_decode")})
except Exception as ex:
    print(f"[{sys.argv[1]}] no line", ex)
PY
done
done
