#!/bin/bash
# kernel timeline of ONE AM bench pass: per hardware queue busy time / span, gaps on the step-chain queue, decode launches
# gpurun --timeout 600 -- 'bash tools/gpu_trace_am.sh TAG [extra bench args]'
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-trace_am}; shift
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_raw -o tr -- python $R/bench.py --workload am-cs16 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-l2-index --no-extra-legs "$@" ) > gpurun_out/${TAG}.log 2>&1
grep "^{" gpurun_out/${TAG}.log | tail -1 | cut -c1-200
python - "$TAG" <<'PY' | tee gpurun_out/${TAG}_summary.txt
import csv, glob, sys, collections, statistics as st
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/{tag}_raw/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "nrsc5::" in r["Kernel_Name"]]
for r in rows:
    r["n"] = r["Kernel_Name"].split("(")[0].replace("nrsc5::", ""); r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
blk = [r for r in rows if r["n"] == "k_am_block"]
# passes are separated by the host-side fetch / reset: the last gap > 3 ms between two k_am_block launches starts the last pass
cut = 0
for i, (a, b) in enumerate(zip(blk[:-1], blk[1:])):
    if b["s"] - a["e"] > 3_000_000: cut = i + 1
t0 = blk[cut]["s"]
P = [r for r in rows if r["s"] >= t0 - 3_000_000]
t0 = min(r["s"] for r in P); tend = max(r["e"] for r in P)
print(f"pass: {len(P)} kernels, {(tend - t0) / 1e6:.2f} ms from first to last kernel")
byq = collections.defaultdict(list)
for r in P: byq[r["Queue_Id"]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: kv[1][0]["s"]):
    busy = sum(r["e"] - r["s"] for r in rs)
    names = collections.Counter(r["n"] for r in rs).most_common(4)
    print(f" queue {q}: {len(rs):5d} kernels, span {(rs[0]['s'] - t0) / 1e6:7.2f} .. {(max(r['e'] for r in rs) - t0) / 1e6:7.2f} ms, busy {busy / 1e6:7.2f} ms | " + ", ".join(f"{n} x{c}" for n, c in names))
for nm in sorted(set(r["n"] for r in P)):
    d = [(r["e"] - r["s"]) / 1e3 for r in P if r["n"] == nm]
    print(f"  {nm:26s} n={len(d):4d} median {st.median(d):9.1f} us  mean {st.mean(d):9.1f} us  max {max(d):9.1f} us  total {sum(d) / 1e3:8.2f} ms")
pb = [r for r in P if r["n"] == "k_am_block"]
cq = pb[0]["Queue_Id"]; rs = byq[cq]
gaps = [(b["s"] - a["e"]) / 1e3 for a, b in zip(rs[:-1], rs[1:])]
big = [(g, a, b) for (a, b), g in zip(zip(rs[:-1], rs[1:]), gaps) if g > 100]
print(f"chain queue: {len(rs)} kernels, gaps median {st.median(gaps):.1f} us, mean {st.mean(gaps):.1f} us, total {sum(g for g in gaps if g > 0) / 1e3:.2f} ms; {len(big)} gaps > 100 us totalling {sum(g for g, _, _ in big) / 1e3:.2f} ms")
for g, a, b in big[:40]: print(f"   gap {g:8.1f} us after {a['n']} (ends {(a['e'] - t0) / 1e6:.2f} ms) before {b['n']}")
print("time per 8-step window along the chain (ms):")
for w in range(0, len(pb), 8):
    seg = pb[w:w + 8]
    nxt = pb[w + 8]["s"] if w + 8 < len(pb) else max(r["e"] for r in rs)
    print(f" {w // 8:2d}:{(nxt - seg[0]['s']) / 1e6:5.2f}", end="\n" if (w // 8) % 12 == 11 else "")
print()
print("decode launches (queue, start ms, duration us):")
for r in P:
    if r["n"] == "k_am_decode": print(f"   q{r['Queue_Id']} {(r['s'] - t0) / 1e6:9.3f} {(r['e'] - r['s']) / 1e3:9.1f}")
PY
rm -rf gpurun_out/${TAG}_raw
