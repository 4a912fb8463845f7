#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace -o tr -- env GPU_MAX_HW_QUEUES=8 python $GRAFT_REPO_ROOT/tools/gpu_am_bench.py --streams 256 --frames 41 --fmt cs16 --steps 1 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/trace.log 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv, glob, gzip
f = glob.glob("gpurun_out/trace/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id", "Stream_Id"]
with gzip.open("gpurun_out/trace_am_compact.csv.gz", "wt") as g:
    w = csv.writer(g); w.writerow(keep)
    for r in rows:
        if "nrsc5::" in r["Kernel_Name"]:
            w.writerow([r[k].split("(")[0].replace("nrsc5::", "") if k == "Kernel_Name" else r[k] for k in keep])
PY
rm -rf gpurun_out/trace
