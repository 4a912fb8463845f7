"""CPU only: ONE reference carrier's Costas loop through the first block of a capture, step by step, three ways -- a float32 model of the reference's adjust_ref (sync.c:101-113:
numpy float32 arithmetic + this host's glibc sincosf / atan2f through ctypes) run (a) on the oracle's FFT bins and (b) on the CPU twin's bins, next to the loop states the oracle and
the twin themselves end the block with.  The model reproducing BOTH end states exactly says both implementations compute the reference's loop on their own inputs; where (a) and (b)
part, the transforms' last bits decided.  Written to trace the one lock in ~3600 that still deviates on the MI355X (stream 103547 of the GPU fuzz, profiles/r06_residual_lock_traced.txt).
    python tools/cpu_trace_costas_loop.py capture.npy REF     (REF = 0 .. 29: reference carrier index, lower / upper interleaved from the band edges)"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

libm = ctypes.CDLL('libm.so.6')
libm.sincosf.argtypes = [ctypes.c_float, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]; libm.atan2f.restype = ctypes.c_float
f32 = np.float32
def sincosf(x):
    s = ctypes.c_float(); c = ctypes.c_float(); libm.sincosf(ctypes.c_float(float(x)), ctypes.byref(s), ctypes.byref(c)); return f32(s.value), f32(c.value)
def atan2f(y, x): return f32(libm.atan2f(ctypes.c_float(float(y)), ctypes.c_float(float(x))))
bw, damp = f32(0.05), f32(0.70710678)
den = f32(1) + (f32(2) * damp * bw) + (bw * bw)
alpha = (f32(4) * damp * bw) / den; beta = (f32(4) * bw * bw) / den
def run(zs, freq, phase, verbose=None):
    freq, phase = f32(freq), f32(phase); out = []
    for n in range(32):
        zx, zy = f32(zs[n].real), f32(zs[n].imag)
        s2, c2 = sincosf(f32(-2) * phase)
        wr = zx * zx - zy * zy; wi = zx * zy + zy * zx
        ur = wr * c2 - wi * s2; ui = wr * s2 + wi * c2
        err = atan2f(ui, ur) * f32(0.5)
        freq = freq + beta * err
        if freq > 0.5: freq = f32(0.5)
        if freq < -0.5: freq = f32(-0.5)
        phase = phase + ((freq + f32(0)) + (alpha * err))
        if float(phase) > np.pi: phase = f32(float(phase) - 2 * np.pi)
        if float(phase) < -np.pi: phase = f32(float(phase) + 2 * np.pi)
        out.append((float(err), float(freq), float(phase), float(ur), float(ui)))
    return out

from nrsc5_amd import engine as eng, build
from oracle import port
iq = np.load(sys.argv[1]); REF = int(sys.argv[2])
LB0, UB1, PW, LIVE_HALF, UB0 = 478, 1570, 19, 267, 1304
refbins = [LB0 + PW * (r >> 1) if (r & 1) == 0 else UB1 - PW * (r >> 1) for r in range(30)]
b = refbins[REF]; live = b - LB0 if b < 1024 else LIVE_HALF + (b - UB0)
O = port.Oracle(); L = O.lib
s = L.orc_open(); L.orc_set_taps(s, port.TAP_FFT, 1)
E = eng.Engine(max_streams=1, q15_capacity=400000, lib_path=build.EMU_LIB)
E.tune(eng.TUNE_NCO_EXACT, 3); E.tune(eng.TUNE_LOOP_EXACT, 1)
snap = port._Snapshot(); off = 0; chunk = 8192; recs = []
while off < iq.size and not len(recs):
    part = np.ascontiguousarray(iq[off:off + chunk]); off += chunk
    L.orc_push_cu8(s, part.ctypes.data, part.size); E.push_cu8(0, part)
    recs = E.drain(0)
L.orc_snapshot(s, ctypes.byref(snap)); f, p = E.debug_fetch_costas(0)
tbins = E.debug_fetch(0)[1][:, live].copy()
p_ = ctypes.c_void_p(); n = L.orc_buf(s, 2, ctypes.byref(p_)); fft = np.frombuffer(ctypes.string_at(p_, n), dtype=np.complex64).reshape(-1, 2048)
obins = fft[0:32, b]
samperr = int(recs[0]['samperr']); adj = 1080 - samperr
ph0 = np.float32(0.0 - (adj * (b - 1024)) * 2 * np.pi / 2048)
print('block 0: samperr', samperr, 'adj', adj, 'initial phase', ph0, 'rec cfo', recs[0]['cfo'], 'state_after', recs[0]['state_after'])
print('after block 0: oracle', snap.costas_freq[REF], snap.costas_phase[REF], ' twin', f[live], p[live])
print('bins identical', int((obins == tbins).sum()), 'of 32; max rel diff', np.max(np.abs(obins - tbins) / np.maximum(np.abs(obins), 1e-30)))
ro = run(obins, 0.0, ph0); rt = run(tbins, 0.0, ph0)
print('model(oracle bins) ends', ro[-1][1:3], ' model(twin bins) ends', rt[-1][1:3])
for n in range(32):
    a, c = ro[n], rt[n]
    print(n, 'err %.7f %.7f  freq %.8f %.8f  phase %.6f %.6f  u=(%.5g,%.5g) (%.5g,%.5g) |z|=%.4g %s' % (a[0], c[0], a[1], c[1], a[2], c[2], a[3], a[4], c[3], c[4], abs(obins[n]), '' if a[2] == c[2] else '<<'))
