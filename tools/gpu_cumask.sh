#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
run() { ( timeout 300 python bench.py --no-cpu-baseline --steps 3 ) > gpurun_out/b.log 2>&1
python - <<PY
import json
l=[x for x in open("gpurun_out/b.log") if x.startswith("{")]
if l:
    j=json.loads(l[-1]); d=j["roofline"]["device_ms_per_pass"]; print("$1", j["value"], j["ms_per_step"], "vit", d["p1_viterbi"], "sync", d["sync"], "mix", d["mixfft"], j["parity"]["p1_frames_bit_exact_vs_truth"])
else:
    print("$1 failed", open("gpurun_out/b.log").read()[-300:])
PY
}
run base
NRSC5HIP_CU_DEC=64 run dec64
NRSC5HIP_CU_DEC=96 run dec96
NRSC5HIP_CU_DEC=128 run dec128
NRSC5HIP_CU_DEC=64 NRSC5HIP_CU_MAIN_ALL=1 run dec64_mainall
NRSC5HIP_CU_DEC=96 NRSC5HIP_CU_MAIN_ALL=1 run dec96_mainall
