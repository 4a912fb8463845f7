#!/bin/bash
# A/B of engine knobs on the FM pass.   gpurun --timeout 900 -- 'bash tools/gpu_r4_ab.sh "mixfft_syms=1" "mixfft_syms=4" ...'
# each argument is one --tune list (comma separated) for one bench run; "-" = no tune
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
run() { timeout 300 python bench.py --no-extra-legs --no-cpu-baseline --steps 6 --warmup 2 $1 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '->', d['ms_per_step'], 'ms (median', d['ms_per_step_median'], ')', r.get('device_ms_per_pass'), 'dom', r['kernel'], r['avg_launch_ms'], d.get('parity_failures'), d['parity']['p1_frames_bit_exact_vs_truth'], '/', d['parity']['p1_frames_decoded'], 'tb', d.get('traceback_walk', {}).get('chunks_rewalked'), '/', d.get('traceback_walk', {}).get('chunk_boundaries_checked'))"; }
for t in "$@"; do
  if [ "$t" = "-" ]; then run ""; else a=""; for kv in ${t//,/ }; do a="$a --tune $kv"; done; run "$a"; fi
done
