"""After each of a capture's first blocks: the Costas loop state (frequency, phase) of the 30 reference carriers as the oracle holds it against the engine's -- the CPU twin
(default) or, with --gpu, the MI355X library -- and the block records' timing / angle fields.  Says WHICH carrier's loop parted and in which block when a stream of the GPU fuzz
deviates (profiles/r06_residual_lock_traced.txt; tools/cpu_trace_costas_loop.py then walks that loop step by step).
    python tools/scan_ref_loops.py capture.npy [blocks] [--gpu] [--dump-bins BLOCK REF]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nrsc5_amd import engine as eng, build
from oracle import port
args = [a for a in sys.argv[1:] if not a.startswith('--')]; gpu = '--gpu' in sys.argv
iq = np.load(args[0]); nblk = int(args[1]) if len(args) > 1 else 1
LB0, UB1, PW, LIVE_HALF, UB0 = 478, 1570, 19, 267, 1304
refbins = [LB0 + PW * (r >> 1) if (r & 1) == 0 else UB1 - PW * (r >> 1) for r in range(30)]

# a float32 model of the reference's adjust_ref (sync.c:101-113) on numpy float32 + this host's glibc sincosf / atan2f (as in tools/cpu_trace_costas_loop.py)
libm = ctypes.CDLL('libm.so.6')
libm.sincosf.argtypes = [ctypes.c_float, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
libm.atan2f.argtypes = [ctypes.c_float, ctypes.c_float]; libm.atan2f.restype = ctypes.c_float
f32 = np.float32
def sincosf(x):
    sn = ctypes.c_float(); cs = ctypes.c_float(); libm.sincosf(ctypes.c_float(float(x)), ctypes.byref(sn), ctypes.byref(cs)); return f32(sn.value), f32(cs.value)
def atan2f(y, x): return f32(libm.atan2f(ctypes.c_float(float(y)), ctypes.c_float(float(x))))
bw, damp = f32(0.05), f32(0.70710678)
den = f32(1) + (f32(2) * damp * bw) + (bw * bw)
alpha = (f32(4) * damp * bw) / den; beta = (f32(4) * bw * bw) / den
def model(zs, freq, phase):
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
O = port.Oracle(); L = O.lib
s = L.orc_open(); L.orc_set_taps(s, port.TAP_FFT, 1)
E = eng.Engine(max_streams=1, q15_capacity=400000, lib_path=None if gpu else build.EMU_LIB)
E.tune(eng.TUNE_HOST_CAPTURE, 0)      # (the batch path's arithmetic is what the fuzz ran: FIFO or capture, the symbol kernel computes the same bins)
E.tune(eng.TUNE_NCO_EXACT, 1); E.tune(eng.TUNE_LOOP_EXACT, 1)
snap = port._Snapshot(); off = 0; chunk = 8192; recs = []
while off < iq.size and len(recs) < nblk:
    part = np.ascontiguousarray(iq[off:off + chunk]); off += chunk
    L.orc_push_cu8(s, part.ctypes.data, part.size); E.push_cu8(0, part)
    r = E.drain(0)
    for x in r: recs.append(x)
    if len(r):
        L.orc_snapshot(s, ctypes.byref(snap)); f, p = E.debug_fetch_costas(0)
        x = recs[-1]
        print('block', len(recs) - 1, 'state', x['state_before'], '->', x['state_after'], 'samperr', x['samperr'], 'next_samperr', x['next_samperr'], 'cfo', x['cfo'], 'next_angle', x['next_angle'], 'prev_angle', x['prev_angle'])
        p_ = ctypes.c_void_p(); nb_ = L.orc_buf(s, 2, ctypes.byref(p_)); fft = np.frombuffer(ctypes.string_at(p_, nb_), dtype=np.complex64).reshape(-1, 2048)[-32:]
        tb = E.debug_fetch(0)[1]
        for R in range(30):
            b = refbins[R]; live = b - LB0 if b < 1024 else LIVE_HALF + (b - UB0)
            df = abs(snap.costas_freq[R] - f[live]); dp = abs(snap.costas_phase[R] - p[live])
            if df > 1e-5 or dp > 1e-4:
                ob, eb = fft[:, b], tb[:, live]
                k = int(np.argmin(np.abs(ob)))
                if len(recs) == 1 and int(x['cfo']) != 0:
                    # block 0 of a stream whose CFO search found an integer offset: this carrier's loop ran over the un-corrected spectrum from (0, initial phase) -- the model on both sets of bins
                    adj = 1080 - int(x['samperr']); ph0 = np.float32(0.0 - (adj * (b - 1024)) * 2 * np.pi / 2048)
                    mo, me = model(ob, 0.0, ph0), model(eb, 0.0, ph0)
                    part = next((n for n in range(32) if mo[n][2] != me[n][2]), None)
                    print('      model(oracle bins) ends', mo[-1][1:3], '| model(engine bins) ends', me[-1][1:3], '| the two part at symbol', part)
                    if part is not None:
                        print('      symbol %d: |z| = %.4g (block mean %.4g); u = z^2 e^{-2i phase}: oracle (%.5g, %.5g)  engine (%.5g, %.5g); error %.6f against %.6f' % (part, abs(ob[part]), np.mean(np.abs(ob)), mo[part][3], mo[part][4], me[part][3], me[part][4], mo[part][0], me[part][0]))
                print('   ref', R, 'bin', b, 'oracle', snap.costas_freq[R], snap.costas_phase[R], 'engine', f[live], p[live], '| bins max rel diff %.3g, weakest cell: symbol %d |z| = %.4g (block mean %.4g) oracle %s engine %s' % (np.max(np.abs(ob - eb) / np.maximum(np.abs(ob), 1e-30)), k, abs(ob[k]), np.mean(np.abs(ob)), ob[k], eb[k]))
