#!/bin/bash
# FM parity subset + the FM pass twice.   gpurun --timeout 900 -- 'bash tools/gpu_fm_ab.sh [tests]'
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
[ "$1" = "tests" ] && timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fft or golden or oracle_end or zero_copy or replay_equals or full_size or halfband" 2>&1 | tail -3
run() { timeout 200 python bench.py --no-extra-legs --no-cpu-baseline --steps 4 --warmup 1 --oracle-streams 8 "$@" 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; pe=d['parity'].get('reference_equality_rank0') or {}; print('$*', '->', d['ms_per_step'], 'ms', r.get('device_ms_per_pass'), d.get('parity_failures'), pe.get('lost_sync_streams_equal'), pe.get('other_streams_equal'), pe.get('streams_with_transient_loop_state_deviation'))"; }
run --workload fm
run --workload fm
run --workload fm
