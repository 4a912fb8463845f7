#!/bin/bash
# A/B of engine variants selected by the create-time test knobs: headline bench line (no CPU legs) per variant, then a timeline
# gpurun --timeout 900 -- 'bash tools/gpu_ab.sh TAG "VAR=1" "VAR2=0" ...'
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-ab}; shift
run() {
  name=$1; shift
  ( env "$@" timeout 300 python bench.py --no-cpu-baseline $BENCH_EXTRA ) > gpurun_out/${TAG}_${name}.log 2>&1
  grep "^{" gpurun_out/${TAG}_${name}.log | tail -1 > gpurun_out/${TAG}_${name}.json
  python - "$TAG" "$name" <<'PY'
import json, sys
try:
    d = json.load(open(f"gpurun_out/{sys.argv[1]}_{sys.argv[2]}.json"))
    r = d["roofline"]
    print(sys.argv[2], d["ms_per_step"], "steps", d["config"].get("block_steps_per_pass"), "fwd", r["avg_launch_ms"], r.get("device_ms_per_pass"), r.get("host_ms_per_pass"),
          "single", (d.get("single_stream") or {}).get("us_per_block"), "eq", (d["parity"].get("reference_equality_rank0") or {}).get("logs_equal_to_oracle_with_l2_hook"), "exact", d["parity"].get("p1_frames_bit_exact_vs_truth"))
except Exception as ex:
    print(sys.argv[2], "no json", ex); print(open(f"gpurun_out/{sys.argv[1]}_{sys.argv[2]}.log").read()[-2000:])
PY
}
run base X=0
i=0
for v in "$@"; do i=$((i+1)); run v$i $v; done
bash tools/gpu_trace.sh ${TAG}_trace > /dev/null 2>&1; head -22 gpurun_out/${TAG}_trace_summary.txt
