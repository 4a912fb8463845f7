#!/bin/bash
# after the oscillator kernel's six-instruction step and the table-free large-argument reduction of ref_sincosf: the parity tests that pin them, the timeline of one pass
# (k_nco_exact, step 0's k_sync), the FM bench without the extra legs
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r06h}
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch256.py tests/test_gpu_fuzz.py -m gpu -q -x -k "oscillator or batch256 or fuzz_fm or golden or oracle_end_to_end" ) > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/${TAG}_tests.log | cut -c1-300
bash tools/gpu_trace.sh ${TAG}_trace > /dev/null 2>&1; grep -a "k_nco_exact\|k_sync  \|k_mixfft  \|pass:\|step chain" gpurun_out/${TAG}_trace_summary.txt | head -12; rm -rf gpurun_out/${TAG}_trace_raw
for i in 1 2; do
( timeout 420 python bench.py --workload fm --no-extra-legs --steps 10 --warmup 2 ) > gpurun_out/${TAG}_bench$i.log 2>gpurun_out/${TAG}_bench$i.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/${TAG}_bench$i.log") if l.startswith("{")][-1])
    r = d["parity"]["reference_equality_rank0"]
    print(d["ms_per_step"], d["ms_per_step_min_max"], "failures", d["parity_failures"], "strict", r["streams_equal_under_the_strict_rule"], "transient", r["streams_with_transient_loop_state_deviation"], d["roofline"]["device_ms_per_pass"])
except Exception as ex:
    print("no line", ex)
PY
done
