#!/bin/bash
# k_am_block launch durations inside one AM cs16 pass and what runs beside the long ones.   gpurun --timeout 600 -- 'bash tools/gpu_trace_am_block.sh [extra bench args]'
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp
rm -rf /tmp/tra; ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/tra -o tr -- python $R/bench.py --workload am-cs16 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-extra-legs "$@" ) > /tmp/tra.log 2>&1
python - <<'PY'
import csv, glob, collections, statistics as st
f = glob.glob("/tmp/tra/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "nrsc5::" in r["Kernel_Name"]]
for r in rows: r["s"]=int(r["Start_Timestamp"]); r["e"]=int(r["End_Timestamp"]); r["n"]=r["Kernel_Name"].split("(")[0].replace("void ","").split("<")[0].replace("nrsc5::","")
rows.sort(key=lambda r: r["s"])
blk=[r for r in rows if r["n"]=="k_am_block"]; blk=blk[len(blk)//2:]
t0=blk[0]["s"]; d=[(r["e"]-r["s"])/1e3 for r in blk]
print("k_am_block launches", len(d), "sum ms %.2f" % (sum(d)/1e3), "median %.1f" % st.median(d), "sorted top", sorted([round(x) for x in d], reverse=True)[:24])
hist=collections.Counter(int(x//25)*25 for x in d); print("histogram (us bucket: count):", sorted(hist.items()))
P=[r for r in rows if r["s"]>=t0]
for nm in ("k_am_interleave","k_am_decode_fwd","k_am_decode_fix","k_am_decode_tb","k_am_decode_finish","k_rollback_am"):
    x=[(r["e"]-r["s"])/1e3 for r in P if r["n"]==nm]
    if x: print(f"  {nm:20s} n={len(x):4d} median {st.median(x):8.1f} us mean {st.mean(x):8.1f} total {sum(x)/1e3:7.2f} ms")
longest=sorted(zip(d,blk), key=lambda t:-t[0])[:3]
for x,L in longest:
    print("during a %d us k_am_block:" % x, [(r["n"], round((r["s"]-L["s"])/1e3), round((r["e"]-r["s"])/1e3)) for r in rows if r["e"]>L["s"] and r["s"]<L["e"] and r is not L][:8])
print("pass span ms %.2f" % ((max(r["e"] for r in P)-t0)/1e6))
PY
