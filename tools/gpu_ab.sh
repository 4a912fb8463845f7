#!/bin/bash
# A/B of engine tuning knobs (bench.py --tune KNOB=VALUE ...): headline numbers per variant, then a kernel timeline of the default
# gpurun --timeout 900 -- 'bash tools/gpu_ab.sh TAG "decode_streams=2" "fwd_segments=8 decode_streams=2" ...'
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-ab}; shift
run() {
  name=$1; shift
  args=""; for kv in "$@"; do args="$args --tune $kv"; done
  ( timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --no-l2-index --steps 4 $BENCH_EXTRA $args ) > gpurun_out/${TAG}_${name}.log 2>&1
  grep "^{" gpurun_out/${TAG}_${name}.log | tail -1 > gpurun_out/${TAG}_${name}.json
  python - "$TAG" "$name" "$*" <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/{sys.argv[1]}_{sys.argv[2]}.json"))
    r = d["roofline"]
    print(f"{sys.argv[2]:5s} [{sys.argv[3]}] ms/pass", d["ms_per_step"], "steps", d["config"].get("block_steps_per_pass"), "dom", r["kernel"], r["avg_launch_ms"], "dev", r.get("device_ms_per_pass"), "seg", d.get("forward_pass_segments", {}).get("segments_repaired"), "exact", d["parity"].get("p1_frames_bit_exact_vs_truth"))
except Exception as ex:
    print(sys.argv[2], "no json", ex); print(open(f"gpurun_out/{sys.argv[1]}_{sys.argv[2]}.log").read()[-1500:])
PY
}
run base
i=0
for v in "$@"; do i=$((i+1)); run v$i $v; done
if [ -z "$NO_TRACE" ]; then bash tools/gpu_trace.sh ${TAG}_trace $BENCH_EXTRA > /dev/null 2>&1; head -40 gpurun_out/${TAG}_trace_summary.txt; grep -A30 "^   window" gpurun_out/${TAG}_trace_summary.txt | head -40; fi
