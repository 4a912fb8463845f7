"""Shader cycles per phase of k_am_block for stream 0 INSIDE the 256-stream AM cs16 batch pass (decode streams running beside the chain).
gpurun -- 'python tools/gpu_am_phases.py'"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
sys.argv = ["bench.py", "--no-cpu-baseline", "--workload", "am-cs16"]
args = bench.parse()
dev = torch.device("cuda", 0)
W = bench.Am(args, dev, 0, list(range(256)), "cs16")
from nrsc5_amd import engine as _eng
W.E.tune(_eng.TUNE_SYNC_PHASES, 1)
W.one_pass()
c0 = W.E.debug_sync_phases()
steps, _ = W.one_pass()
c1 = W.E.debug_sync_phases()
names = ["tables + coarse acquisition", "bookkeeping + NCO set-up", "pass-1 fold", "carrier / line fit / NCO correction", "pass-2 fold", "32 x FFT-256",
         "sideband combine + reference decode", "PIDS carriers + equaliser taps", "timing + equalise / slice", "PIDS gather + hand-off", "tail"]
nblk = 41 * 8
tot = 0
for nm, a, b in zip(names, c0, c1):
    d = (b - a) / nblk
    tot += d
    print(f"{nm:40s} {d:9.0f} cycles/block")
print(f"{'total':40s} {tot:9.0f} cycles/block ({nblk} blocks of stream 0) over {steps} steps")
