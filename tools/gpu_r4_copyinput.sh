#!/bin/bash
# zero-copy batch (half-band inside k_mixfft) against the copying batch (K1 in chunks on a side stream, k_mixfft reads the Q15 FIFO): what the symbol
# kernel costs without its half-band
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
run() { timeout 300 python bench.py --no-extra-legs --no-cpu-baseline --steps 6 --warmup 2 $1 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', '->', d['ms_per_step'], 'ms (median', d['ms_per_step_median'], ')', r.get('device_ms_per_pass'), 'dom', r['kernel'], r['avg_launch_ms'], d.get('parity_failures'), d['parity']['p1_frames_bit_exact_vs_truth'], '/', d['parity']['p1_frames_decoded'])"; }
run ""
run "--copy-input"
run ""
run "--copy-input"
