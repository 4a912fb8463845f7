cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "k9" 2>&1 | tail -5
timeout 300 python tools/gpu_k9_bench.py 2>&1 | tee gpurun_out/r03_k9_bench.txt
timeout 600 python -m pytest tests -x -q -m gpu -k "am" 2>&1 | tail -5
timeout 300 python bench.py --workload am-cs16 --no-extra-legs > gpurun_out/r03am2_bench.log 2>&1; grep "^{" gpurun_out/r03am2_bench.log | tail -1 > gpurun_out/r03am2_bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03am2_bench.json")); r = d["roofline"]
print("ms_per_step", d["ms_per_step"], "x", d["x_realtime"], r.get("device_ms_per_pass"), r.get("host_ms_per_pass"), d.get("parity_failures"))
print(json.dumps(d["parity"].get("reference_equality_rank0"))[:600])
PY
