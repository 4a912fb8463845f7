#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
[ "$1" = "tests" ] && timeout 600 python -m pytest tests -x -q -m gpu -k "mixed or am or dropin or mode_switch or interleaved" 2>&1 | tail -3
run() { timeout 200 python bench.py --no-extra-legs --no-cpu-baseline --steps 4 --warmup 1 --oracle-streams 4 "$@" 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$*', '->', d['ms_per_step'], 'ms', r.get('device_ms_per_pass'), d.get('parity_failures'))"; }
run --workload mixed
run --workload mixed
run --workload am-cs16
run --workload fm
