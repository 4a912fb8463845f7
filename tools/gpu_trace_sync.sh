cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/trs -o tr -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-l2-index --no-extra-legs "$@" ) > /tmp/trs.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/trs/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "nrsc5::" in r["Kernel_Name"]]
for r in rows: r["s"]=int(r["Start_Timestamp"]); r["e"]=int(r["End_Timestamp"]); r["n"]=r["Kernel_Name"].split("(")[0].replace("void ","").split("<")[0].replace("nrsc5::","")
rows.sort(key=lambda r: r["s"])
sync=[r for r in rows if r["n"]=="k_sync"]; sync=sync[len(sync)//2:]   # last pass
t0=sync[0]["s"]
d=[(r["e"]-r["s"])/1e3 for r in sync]
print("k_sync launches", len(d), "sum ms", sum(d)/1e3, "sorted top", sorted([round(x) for x in d], reverse=True)[:30])
print("long ones (index, start ms, us):", [(i, round((r["s"]-t0)/1e6,2), round(x)) for i,(r,x) in enumerate(zip(sync,d)) if x>60])
mix=[r for r in rows if r["n"]=="k_mixfft"]; mix=mix[len(mix)//2:]
dm=[(r["e"]-r["s"])/1e3 for r in mix]
print("k_mixfft sorted top", sorted([round(x) for x in dm], reverse=True)[:15], "median", sorted(dm)[len(dm)//2])
# what runs concurrently with the longest k_sync
L=max(zip(d,sync), key=lambda t:t[0])[1]
print("during the longest k_sync:", [(r["n"], round((r["s"]-L["s"])/1e3), round((r["e"]-r["s"])/1e3)) for r in rows if r["e"]>L["s"] and r["s"]<L["e"] and r is not L][:12])
PY
