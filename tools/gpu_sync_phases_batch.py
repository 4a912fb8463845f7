"""Shader cycles per phase of k_sync for stream 0 INSIDE the 256-stream batch pass (decode streams running beside the chain).
gpurun -- 'python tools/gpu_sync_phases_batch.py'"""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
sys.argv = ["bench.py", "--no-cpu-baseline"]
args = bench.parse()
dev = torch.device("cuda", 0)
W = bench.Fm(args, dev, 0, list(range(256)))
from nrsc5_amd import engine as _eng
W.E.tune(_eng.TUNE_SYNC_PHASES, 1)
W.one_pass()
W.E.debug_sync_phases()                                        # read + reset? (accumulates: take the difference)
c0 = W.E.debug_sync_phases()
steps, _ = W.one_pass()
c1 = W.E.debug_sync_phases()
names = ["head: state burst + refs", "costas", "coarse / CFO search", "samperr/angle", "equalise+MER", "soft bits", "PIDS gather", "finish + checkpoint"]
tot = 0
for nm, a, b in zip(names, c0, c1):
    d = (b - a) / float(steps)
    tot += d
    print(f"{nm:16s} {d:9.0f} cycles/block")
print(f"{'total':16s} {tot:9.0f} cycles/block over {steps} steps")
