#!/bin/bash
# FM pass under decode-stream count / forward segments / queue priority.   gpurun --timeout 900 -- 'bash tools/gpu_fm_sweep.sh'
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
run() { timeout 200 python bench.py --no-extra-legs --no-cpu-baseline --steps 4 --warmup 1 --oracle-streams 0 "$@" 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$*', '->', d['ms_per_step'], 'ms', r.get('device_ms_per_pass'), d.get('parity_failures'))"; }
run
run --tune decode_streams=2
run --tune decode_streams=4
run --tune decode_streams=2 --tune fwd_segments=8
run --tune fwd_segments=2
run --tune fwd_segments=8
run --tune decode_priority=1
run
