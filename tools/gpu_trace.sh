#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace -o tr -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --l2-feedback 0 > $GRAFT_REPO_ROOT/gpurun_out/trace.log 2>&1
cd $GRAFT_REPO_ROOT; ls -la gpurun_out/trace | head; python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
print(len(rows), rows[0].keys())
# keep a compact version
import gzip
keep = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id", "Stream_Id"] if "Stream_Id" in rows[0] else ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Queue_Id"]
with gzip.open("gpurun_out/trace_compact.csv.gz", "wt") as g:
    w = csv.writer(g); w.writerow(keep)
    for r in rows:
        w.writerow([r[k].split("(")[0].replace("nrsc5::", "") if k == "Kernel_Name" else r[k] for k in keep])
PY
rm -rf gpurun_out/trace
