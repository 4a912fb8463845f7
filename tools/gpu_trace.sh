#!/bin/bash
# kernel timeline of ONE bench pass: per hardware queue busy time / span, where the pass ends relative to the step chain
# gpurun --timeout 600 -- 'bash tools/gpu_trace.sh TAG [extra bench args]'
cd "${GRAFT_REPO_ROOT:-.}"; R=$PWD; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-trace}; shift
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_raw -o tr -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-l2-index --no-extra-legs "$@" ) > gpurun_out/${TAG}.log 2>&1
grep "^{" gpurun_out/${TAG}.log | tail -1 | cut -c1-200
python - "$TAG" <<'PY' | tee gpurun_out/${TAG}_summary.txt
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/{tag}_raw/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "nrsc5::" in r["Kernel_Name"]]
for r in rows:
    r["n"] = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].replace("nrsc5::", ""); r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# the last pass starts at the last k_decimate_fm_cu8 burst: find the last big gap before a decimate kernel
dec = [r for r in rows if r["n"] in ("k_decimate_fm_cu8", "k_attach_raw")]
starts = [dec[0]["s"]]
for a, b in zip(dec[:-1], dec[1:]):
    if b["s"] - a["e"] > 5_000_000: starts.append(b["s"])
t0 = starts[-1]
P = [r for r in rows if r["s"] >= t0 - 2_000_000]
t0 = min(r["s"] for r in P); tend = max(r["e"] for r in P)
print(f"pass: {len(P)} kernels, {(tend - t0) / 1e6:.2f} ms from first to last kernel")
byq = collections.defaultdict(list)
for r in P: byq[r["Queue_Id"]].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: kv[1][0]["s"]):
    busy = sum(r["e"] - r["s"] for r in rs)
    names = collections.Counter(r["n"] for r in rs).most_common(4)
    print(f" queue {q}: {len(rs):5d} kernels, span {(rs[0]['s'] - t0) / 1e6:7.2f} .. {(max(r['e'] for r in rs) - t0) / 1e6:7.2f} ms, busy {busy / 1e6:7.2f} ms | " + ", ".join(f"{n} x{c}" for n, c in names))
sync = [r for r in P if r["n"] == "k_sync"]
mix = [r for r in P if r["n"] == "k_mixfft"]
print(f"step chain: {len(sync)} steps, first k_mixfft at {(mix[0]['s'] - t0) / 1e6:.2f} ms, last k_sync ends at {(sync[-1]['e'] - t0) / 1e6:.2f} ms")
import statistics as st
for nm in ("k_attach_raw", "k_acq_decimate", "k_mixfft", "k_sync", "k_p1_forward", "k_p1_traceback", "k_p1_deint", "k_pids_decode", "k_rollback", "k_prepare", "k_acq_fir", "k_decimate_fm_cu8"):
    d = [(r["e"] - r["s"]) / 1e3 for r in P if r["n"] == nm]
    if d: print(f"  {nm:20s} n={len(d):4d} median {st.median(d):9.1f} us  mean {st.mean(d):9.1f} us  max {max(d):9.1f} us  total {sum(d) / 1e3:8.2f} ms")
# gaps on the chain queue between consecutive kernels
cq = sync[0]["Queue_Id"]; rs = byq[cq]
gaps = [(b["s"] - a["e"]) / 1e3 for a, b in zip(rs[:-1], rs[1:])]
for (a, b), g in zip(zip(rs[:-1], rs[1:]), gaps):
    if g > 100: print(f"   gap {g:8.1f} us after {a['n']} (ends {(a['e'] - t0) / 1e6:.2f} ms) before {b['n']}; chain kernel index {rs.index(a)} of {len(rs)}")
print(f"chain queue gaps: median {st.median(gaps):.1f} us, mean {st.mean(gaps):.1f} us, total {sum(g for g in gaps if g > 0) / 1e3:.2f} ms; gaps > 100 us: {[round(g) for g in gaps if g > 100][:20]}")
# the first 20 steps in detail: everything on the chain queue up to the 20th k_sync
n_sync = 0
print("first steps on the chain queue (start ms, duration us):")
for r in rs:
    print(f"   {(r['s'] - t0) / 1e6:8.3f}  {(r['e'] - r['s']) / 1e3:9.1f}  {r['n']}")
    if r["n"] == "k_sync":
        n_sync += 1
        if n_sync == 20: break
# step-by-step: time per 16-step window along the chain
for w in range(0, len(sync), 16):
    seg = sync[w:w + 16]
    print(f"   window {w // 16:2d}: steps {w:3d}..{w + len(seg) - 1:3d}  {(seg[-1]['e'] - (mix[w]['s'] if w < len(mix) else seg[0]['s'])) / 1e6:6.2f} ms", end="")
    if (w // 16) % 4 == 3: print()
print()
print("decode kernels (queue, start ms, duration us):")
for r in P:
    if r["n"] in ("k_p1_forward", "k_p1_traceback"):
        print(f"   q{r['Queue_Id']} {(r['s'] - t0) / 1e6:9.3f} {(r['e'] - r['s']) / 1e3:9.1f}  {r['n']}")
PY
rm -rf gpurun_out/${TAG}_raw
