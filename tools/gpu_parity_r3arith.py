"""bench.py on the diagnostic library with the symbol kernel's round-3 float sequence (tools/build_diag_r3arith.py): every FM stream against the unmodified reference.
    gpurun -- 'python tools/gpu_parity_r3arith.py > gpurun_out/r04_parity_r3arith.log'"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nrsc5_amd import engine as _eng
_eng.DEFAULT_LIB = os.path.join(ROOT, "nrsc5_amd", "libnrsc5hip_r3arith.so")
import bench
sys.argv = ["bench.py", "--no-extra-legs", "--oracle-streams", "256", "--oracle-lost-max", "256", "--steps", "2", "--warmup", "1", "--cpu-baseline-seconds", "2"]
bench.main()
