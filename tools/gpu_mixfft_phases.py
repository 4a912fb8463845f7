"""Where a k_mixfft workgroup's time goes INSIDE the 256-stream batch pass: shader cycles of wave 0 of stream 0's workgroups between marks
(diagnostic build: python -m nrsc5_amd.build --mixfft-phases  ->  nrsc5_amd/libnrsc5hip_mixphases.so).
gpurun -- 'python tools/gpu_mixfft_phases.py [--copy-input]'"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from nrsc5_amd import engine as _eng
_eng.DEFAULT_LIB = os.path.join(ROOT, "nrsc5_amd", "libnrsc5hip_mixphases.so")
import bench
extra = [a for a in sys.argv[1:]]
sys.argv = ["bench.py", "--no-cpu-baseline"] + extra
args = bench.parse()
dev = torch.device("cuda", 0)
W = bench.Fm(args, dev, 0, list(range(256)))
W.E.tune(_eng.TUNE_SYNC_PHASES, 1)
W.one_pass()
c0 = W.E.debug_sync_phases()
steps, _ = W.one_pass()
c1 = W.E.debug_sync_phases()
d = (c1 - c0)[8:].astype(float)
nwg = 32.0 * 224                                                # stream 0: 32 workgroups per block step, 224 blocks
names = ["entry -> parameters (state loads)", "set-up + capture loads arrive", "half-band + barrier", "NCO, mix, fold + barrier", "FFT (three barriers)", "stores, waited for"]
tot = 0.0
for nm, v in zip(names, d):
    tot += v / nwg
    print(f"{nm:36s} {v / nwg:9.0f} cycles per workgroup")
print(f"{'total':36s} {tot:9.0f} cycles per workgroup; {steps} steps; extra args {extra}")
