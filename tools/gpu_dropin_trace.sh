#!/bin/bash
# kernel + copy timeline of the drop-in leg (one capture through nrsc5_pipe_samples_cu8): per block what runs when, durations and gaps
#   gpurun --timeout 600 -- 'bash tools/gpu_dropin_trace.sh TAG'
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out; TAG=${1:-dtrace}
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/${TAG}_raw -o tr -- python $R/tools/gpu_dropin.py 1 ) > gpurun_out/${TAG}.log 2>&1
grep "^{" gpurun_out/${TAG}.log | tail -1 | cut -c1-300
python - "$TAG" <<'PY' | tee gpurun_out/${TAG}_summary.txt
import csv, glob, sys, collections, statistics as st
tag = sys.argv[1]
ops = []
for f in glob.glob(f"gpurun_out/{tag}_raw/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "nrsc5::" in r["Kernel_Name"] or "k_" in r["Kernel_Name"]:
            ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("nrsc5::", "").replace("void ", "")))
for f in glob.glob(f"gpurun_out/{tag}_raw/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", r.get("Kind", "?")) + ":" + r.get("Size", "?")))
ops.sort()
syncs = [i for i, o in enumerate(ops) if o[2].startswith("k_sync")]
print(len(ops), "ops,", len(syncs), "k_sync launches")
# the drop-in session is the LAST burst of single-stream k_sync launches; take its steady state
if len(syncs) > 150:
    lo, hi = syncs[-120], syncs[-100]
    # start at a report kernel boundary
    while lo > 0 and not (ops[lo][2].startswith("k_stream_tail") or ops[lo][2].startswith("k_sync")): lo -= 1      # (round 6: most steps end with k_sync, which posts the report itself)
    t0 = ops[lo][1]
    prev_end = t0
    for s, e, n in ops[lo + 1:hi]:
        print(f"  +{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  {n[:60]}")
        prev_end = max(prev_end, e)
    seg = ops[syncs[-200]:syncs[-20]]
    by = collections.defaultdict(list); gap = collections.defaultdict(list)
    pe = seg[0][1]
    for s, e, n in seg[1:]:
        k = n.split(":")[0] + (":" + n.split(":")[1] if n.startswith("copy") else "")
        by[k].append((e - s) / 1e3); gap[k].append((s - pe) / 1e3); pe = max(pe, e)
    nb = sum(1 for o in seg if o[2].startswith("k_sync"))
    print(f"steady state over {nb} blocks: span {(seg[-1][1] - seg[0][0]) / 1e3 / nb:.1f} us per block")
    for k in sorted(by, key=lambda k: -sum(by[k])):
        print(f"  {k:28s} n/block {len(by[k]) / nb:5.2f}  dur median {st.median(by[k]):7.1f} mean {st.mean(by[k]):7.1f}  gap-before median {st.median(gap[k]):7.1f} mean {st.mean(gap[k]):7.1f}  -> {(sum(by[k]) + sum(gap[k])) / nb:7.1f} us per block")
PY
rm -rf gpurun_out/${TAG}_raw
