#!/bin/bash
# PMC passes (counters only; no trace domains), each its own run, then the kernel-trace stats of the same command.
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline"
cd /tmp
( time timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o f --output-format csv -- $CMD ) > $R/gpurun_out/pmc_fetch.log 2>&1; echo "fetch rc=$?"
( time timeout 600 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o w --output-format csv -- $CMD ) > $R/gpurun_out/pmc_write.log 2>&1; echo "write rc=$?"
( time timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_full -o full --output-format csv -- $CMD ) > $R/gpurun_out/prof_full.log 2>&1; echo "stats rc=$?"
cd $R
python profiles/collect_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write p1_viterbi gpurun_out/traffic_latest.json
rm -f gpurun_out/prof_full/*kernel_trace.csv gpurun_out/pmc_fetch/*/*kernel_trace* 2>/dev/null
ls -la gpurun_out/pmc_fetch gpurun_out/prof_full | head -20
( timeout 300 python bench.py --steps 3 --warmup 1 --traffic-json gpurun_out/traffic_latest.json ) > gpurun_out/bench_final.log 2>&1; echo "bench rc=$?"; tail -c 1500 gpurun_out/bench_final.log
