#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
for V in fuse nofuse; do
  if [ $V = nofuse ]; then export NRSC5HIP_NO_FUSE=1; fi
  ( timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline ) > gpurun_out/bench_$V.log 2>&1
  echo "$V $(grep -o '"value": [0-9.]*' gpurun_out/bench_$V.log | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench_$V.log)"
  grep -o '"device_ms_per_pass": {[^}]*}' gpurun_out/bench_$V.log
done
unset NRSC5HIP_NO_FUSE
cd /tmp && ( timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_full -o full --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline ) > $R/gpurun_out/prof_full.log 2>&1; cd $R
rm -f gpurun_out/prof_full/*kernel_trace.csv
python - <<'PY'
import csv
rows = list(csv.DictReader(open('gpurun_out/prof_full/full_kernel_stats.csv')))
for r in rows:
    if 'nrsc5' in r['Name']: print('  ', r['Name'][:40].ljust(40), r['Calls'].rjust(6), f"{float(r['AverageNs'])/1e3:10.1f} us  max {float(r['MaxNs'])/1e3:10.1f}")
PY
