"""Where the two kernels of a FINE block step of ONE stream spend their time on the fast streaming seam: shader cycles between the marks of k_sync (tid 0) and of
k_mixfft (wave 0 of each of the 32 symbol workgroups; diagnostic build: python -m nrsc5_amd.build --mixfft-phases), with the host-resident capture (the symbol kernel
reads pinned host memory across PCIe) and with the FIFO seam (it reads the decimated FIFO in HBM).
gpurun -- 'python tools/gpu_seam_phases.py'"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from nrsc5_amd import engine as eng, synth
from tests import common
lib = os.path.join(ROOT, "nrsc5_amd", "libnrsc5hip_mixphases.so")
cap = synth.fm_mp1_capture(1, seed=5, cfo_hz=120.0, offset=700, snr_db=25.0, n_blocks=80)
raw = cap.iq[:cap.iq.size - cap.iq.size % 4]
names_sync = ["head: state burst + refs", "costas", "coarse / CFO search", "samperr/angle", "equalise+MER", "soft bits", "PIDS gather (+ inline decode issue)", "record (NCO phase, bookkeeping)"]
names_mix = ["entry -> parameters (state loads, local prepare)", "set-up + capture loads arrive", "half-band + barrier", "NCO, mix, fold + barrier", "FFT (three barriers)", "stores, waited for"]
for hc in (1, 0):
    E = eng.Engine(max_streams=1, q15_capacity=400000, record_capacity=512, p1_slots=8, lib_path=lib)
    E.tune(eng.TUNE_HOST_CAPTURE, hc)
    E.tune(eng.TUNE_SYNC_PHASES, 1)
    half = raw.size // 2 // 4 * 4
    common.run_engine_streaming(E, 0, raw[:half], chunk=32768); n0 = len(E.drain(0)); c0 = E.debug_sync_phases()
    common.run_engine_streaming(E, 0, raw[half:], chunk=32768); n1 = len(E.drain(0)); c1 = E.debug_sync_phases()
    d = (c1 - c0).astype(float); nb = max(n1, 1)
    print(f"== host-resident capture {'ON' if hc else 'OFF (FIFO seam)'}: {nb} blocks (all FINE)")
    tot = 0.0
    for nm, v in zip(names_sync, d[:8]):
        tot += v / nb; print(f"  k_sync   {nm:52s} {v / nb:9.0f} cycles per block")
    print(f"  k_sync   {'inline PIDS decode (wave 1) + barrier':52s} {d[14] / nb:9.0f} cycles per block"); tot += d[14] / nb
    print(f"  k_sync   {'total':52s} {tot:9.0f}")
    tot = 0.0
    for nm, v in zip(names_mix, d[8:14]):
        tot += v / nb / 32; print(f"  k_mixfft {nm:52s} {v / nb / 32:9.0f} cycles per workgroup")
    print(f"  k_mixfft {'total':52s} {tot:9.0f}")
    E.close()
