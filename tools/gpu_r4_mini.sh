#!/bin/bash
# quick loop: seam / viterbi tests, drop-in leg (no profiler), drop-in timeline.   gpurun --timeout 900 -- 'bash tools/gpu_r4_mini.sh TAG'
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-r04m}
( time timeout 600 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_parity.py -m gpu -x -q -k "dropin or deferred_seam or block_exact or push_size or small_fifo or golden_end_to_end or halfband or mode_switch or l2_feedback or viterbi or oracle_end_to_end or pids" ) > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/${TAG}_tests.log | cut -c1-300
python tools/gpu_dropin.py 3 2>&1 | grep "^{" | cut -c1-1300
bash tools/gpu_dropin_trace.sh ${TAG}_dtrace 2>&1 | tail -16 | cut -c1-200
