#!/bin/bash
# AM pass under decode-stream count / queue priority / segment count, then the FM pass.   gpurun --timeout 900 -- 'bash tools/gpu_am_sweep.sh'
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
run() { timeout 200 python bench.py --no-extra-legs --no-cpu-baseline --steps 3 --warmup 1 --oracle-streams 0 "$@" 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$*', '->', d['ms_per_step'], 'ms', r.get('device_ms_per_pass'), d.get('parity_failures'))"; }
run --workload am-cs16
run --workload am-cs16 --tune am_decode_streams=1
run --workload am-cs16 --tune am_decode_streams=2
run --workload am-cs16 --tune am_decode_streams=3
run --workload fm
run --workload fm
