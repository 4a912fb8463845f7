"""GPU: one slice of the AM fuzz (tests/test_gpu_fuzz.py::test_gpu_fuzz_am_batch) by its seed counter, with the failing streams' first differences printed and their captures
dumped to gpurun_out/ for the CPU twin.   python tools/gpu_am_fuzz_repro.py COUNTER"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench

def main():
    counter = int(sys.argv[1])
    base = 50000 + 128 * counter
    args = bench.parse(["--workload", "am-cs16", "--streams", "128", "--stream-base", str(base), "--am-frames", "14", "--steps", "1", "--warmup", "0", "--no-extra-legs"])
    dev = torch.device("cuda", 0)
    my = bench.my_stream_ids(args, 1, 0)
    W = bench.make_workload("am-cs16", args, dev, 0, my)
    steps, (recs, counts, frames) = W.one_pass()
    out = bench.reference_equality(W, recs, counts, frames, W.to_log, am=True)
    print(json.dumps({k: out[k] for k in ("streams_compared", "streams_equal_under_the_strict_rule", "streams_failing_by_class", "first_diffs", "streams_with_lost_sync_this_pass")}, indent=1)[:3000])
    os.makedirs("gpurun_out", exist_ok=True)
    for fd in out["first_diffs"]:
        k = my.index(fd["stream"])
        np.save(f"gpurun_out/am_capture_stream{fd['stream']}.npy", W.stream_iq(k))
        print("dumped", fd["stream"], W.stream_iq(k).shape, W.stream_iq(k).dtype)
    if bench._POOL: bench._POOL.close()


if __name__ == "__main__":
    main()
