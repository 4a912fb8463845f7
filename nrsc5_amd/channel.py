"""Propagation / front-end impairments for the synthetic captures -- TEST / BENCH SIGNAL SOURCE ONLY.

The transmitter models (synth.py, synth_am.py, synth_torch.py) produce a clean complex baseband; round 1-3 put only a
carrier offset, an integer delay and white noise on it.  A real RTL-SDR capture is none of that: the ADC clock is off by
tens of ppm (so the OFDM symbol boundary drifts by several samples per 32-symbol block and the receiver's FINE-state
timing feedback -- sync.c:426-463 -> acquire.c:110-119,259 -> sync_adjust, sync.c:769-777 -- works on EVERY block), the
channel has echoes (adjust_data's per-partition equaliser sees unequal reference magnitudes), a hybrid FM station's analog
host sits 20 dB above the digital sidebands in the middle of the spectrum, the 8-bit ADC clips, and the level fades.

`Impairments` describes one such channel; `apply` (numpy, float64: byte-reproducible, used by goldens and tests) and
`apply_torch` (float32 on the GPU, used by bench.py to build hundreds of captures) put it on a clean signal BEFORE the
noise and the quantiser.  Nothing here is on the product path.
"""
from __future__ import annotations

import dataclasses
import numpy as np


@dataclasses.dataclass(frozen=True)
class Impairments:
    ppm: float = 0.0                 # receiver sample clock error: the receiver takes (1 + ppm * 1e-6) samples per transmit sample period
    paths: tuple = ()                # echoes: ((delay_s, gain_db, doppler_hz, phase0_rad), ...) added to the direct path
    host_db: float | None = None     # analog FM host: carrier power over the total digital power, dB (hybrid FM: ~ +20)
    host_dev_hz: float = 60e3        # peak deviation of the host's programme
    host_tones_hz: tuple = (1000.0, 6300.0, 13700.0)
    fade_db: float = 0.0             # slow flat fading: gain swings between 0 and -fade_db ...
    fade_period_s: float = 1.0       # ... with this period (a raised cosine: one block is 93 ms)
    clip_rms: float | None = None    # ADC full scale in units of the rms of the impaired signal (None: the quantiser's own range only)

    def active(self) -> bool:
        return bool(self.ppm or self.paths or self.host_db is not None or self.fade_db or self.clip_rms is not None)


HALF_TAPS = 12                       # windowed-sinc interpolator: 24 taps (signal band <= 0.27 fs)
_KAISER_BETA = 8.0
_PHASES = 8192                       # fractional delays are rounded to 1 / 8192 sample (interpolation error ~ -75 dB)
_TABLE = None


def kernel_table() -> np.ndarray:
    """[_PHASES + 1][2 * HALF_TAPS] Kaiser-windowed sinc: row p, column k + HALF_TAPS - 1 = h(k - p / _PHASES)."""
    global _TABLE
    if _TABLE is None:
        frac = np.arange(_PHASES + 1, dtype=np.float64)[:, None] / _PHASES
        k = np.arange(-HALF_TAPS + 1, HALF_TAPS + 1, dtype=np.float64)[None, :]
        x = k - frac
        w = np.i0(_KAISER_BETA * np.sqrt(np.clip(1.0 - (x / HALF_TAPS) ** 2, 0.0, None))) / np.i0(_KAISER_BETA)
        _TABLE = np.sinc(x) * w
    return _TABLE


def resample(sig: np.ndarray, ppm: float) -> np.ndarray:
    """y[n] = sig(n / (1 + ppm * 1e-6)): the receiver's clock runs (1 + ppm e-6) times the transmitter's, so it sees every
    transmit interval stretched to that many samples.  Windowed-sinc interpolation, zero outside the signal."""
    if not ppm:
        return sig
    ratio = 1.0 / (1.0 + ppm * 1e-6)
    n_in = sig.shape[0]
    n_out = int(np.floor((n_in - 1) / ratio))
    pad = np.concatenate([np.zeros(HALF_TAPS, sig.dtype), sig, np.zeros(HALF_TAPS + 2, sig.dtype)])
    tab = kernel_table()
    out = np.zeros(n_out, dtype=sig.dtype)
    step = 1 << 20                                   # bounded temporaries
    for a in range(0, n_out, step):
        b = min(n_out, a + step)
        pos = np.arange(a, b, dtype=np.float64) * ratio
        i0 = np.floor(pos).astype(np.int64)
        ph = np.rint((pos - i0) * _PHASES).astype(np.int64)
        acc = np.zeros(b - a, dtype=sig.dtype)
        for k in range(2 * HALF_TAPS):
            acc += pad[i0 + (k + 1)] * tab[ph, k]
        out[a:b] = acc
    return out


def host_carrier(n: int, fs: float, imp: Impairments) -> np.ndarray:
    """Unit-amplitude analog FM host: exp(j * 2 pi * dev * integral of a three-tone programme)."""
    t = np.arange(n, dtype=np.float64) / fs
    ph = np.zeros(n)
    for i, f in enumerate(imp.host_tones_hz):
        # integral of cos(2 pi f t) = sin(2 pi f t) / (2 pi f); each tone takes a third of the deviation
        ph += (imp.host_dev_hz / len(imp.host_tones_hz)) / f * np.sin(2 * np.pi * f * t + 0.7 * i)
    return np.exp(1j * ph)


def apply(sig: np.ndarray, fs: float, imp: Impairments | None) -> np.ndarray:
    """Clean unit-power baseband at the capture's sample rate -> what the receiver's ADC sees before noise and quantisation
    (still before the FM receiver-side conjugation).  Order: host added at the transmitter, echoes + fading in the channel,
    sample-clock error at the receiver, clipping last."""
    if imp is None or not imp.active():
        return sig
    n = sig.shape[0]
    out = sig.astype(np.complex128, copy=True)
    if imp.host_db is not None:
        p = np.mean(np.abs(sig) ** 2)
        out += np.sqrt(p * 10 ** (imp.host_db / 10)) * host_carrier(n, fs, imp)
    if imp.paths:
        direct = out.copy()
        t = np.arange(n, dtype=np.float64) / fs
        for delay_s, gain_db, doppler_hz, phase0 in imp.paths:
            d = int(round(delay_s * fs))
            g = 10 ** (gain_db / 20) * np.exp(1j * (phase0 + 2 * np.pi * doppler_hz * t[d:]))
            out[d:] += g * direct[:n - d]
    if imp.fade_db:
        t = np.arange(n, dtype=np.float64) / fs
        lo = 10 ** (-imp.fade_db / 20)
        out *= lo + (1 - lo) * 0.5 * (1 + np.cos(2 * np.pi * t / imp.fade_period_s))
    if imp.ppm:
        out = resample(out, imp.ppm)
    if imp.clip_rms is not None:
        lim = imp.clip_rms * np.sqrt(np.mean(np.abs(out) ** 2))
        out = np.clip(out.real, -lim, lim) + 1j * np.clip(out.imag, -lim, lim)
    return out


# ------------------------------------------------------------------------------------------------ torch twin (bench)
def apply_torch(sig, fs: float, imp: Impairments | None):
    """Same channel on a torch complex64 tensor (device-side, float32 arithmetic with float64 time bases).  Not bit-identical to
    `apply` and not meant to be: the bench compares the ENGINE with the REFERENCE on whatever bytes come out."""
    import torch
    if imp is None or not imp.active():
        return sig
    dev = sig.device
    n = sig.shape[0]
    out = sig.clone()
    if imp.host_db is not None:
        p = float(torch.mean(sig.real ** 2 + sig.imag ** 2))
        t = torch.arange(n, device=dev, dtype=torch.float64) / fs
        ph = torch.zeros(n, device=dev, dtype=torch.float64)
        for i, f in enumerate(imp.host_tones_hz):
            ph += (imp.host_dev_hz / len(imp.host_tones_hz)) / f * torch.sin(2 * np.pi * f * t + 0.7 * i)
        ph = torch.remainder(ph, 2 * np.pi).to(torch.float32)
        out += float(np.sqrt(p * 10 ** (imp.host_db / 10))) * torch.complex(torch.cos(ph), torch.sin(ph))
        del t, ph
    if imp.paths:
        direct = out.clone()
        for delay_s, gain_db, doppler_hz, phase0 in imp.paths:
            d = int(round(delay_s * fs))
            ph = phase0 + 2 * np.pi * doppler_hz / fs * torch.arange(d, n, device=dev, dtype=torch.float64)
            ph = torch.remainder(ph, 2 * np.pi).to(torch.float32)
            out[d:] += float(10 ** (gain_db / 20)) * torch.complex(torch.cos(ph), torch.sin(ph)) * direct[:n - d]
        del direct
    if imp.fade_db:
        t = torch.arange(n, device=dev, dtype=torch.float64) / fs
        lo = 10 ** (-imp.fade_db / 20)
        out *= (lo + (1 - lo) * 0.5 * (1 + torch.cos(2 * np.pi * t / imp.fade_period_s))).to(torch.float32)
    if imp.ppm:
        ratio = 1.0 / (1.0 + imp.ppm * 1e-6)
        n_out = int(np.floor((n - 1) / ratio))
        pad = torch.cat([torch.zeros(HALF_TAPS, dtype=out.dtype, device=dev), out, torch.zeros(HALF_TAPS + 2, dtype=out.dtype, device=dev)])
        tab = torch.from_numpy(kernel_table().astype(np.float32)).to(dev)
        res = torch.zeros(n_out, dtype=out.dtype, device=dev)
        step = 1 << 22
        for a in range(0, n_out, step):
            b = min(n_out, a + step)
            pos = torch.arange(a, b, device=dev, dtype=torch.float64) * ratio
            i0 = torch.floor(pos).to(torch.int64)
            ph = torch.round((pos - i0) * _PHASES).to(torch.int64)
            acc = torch.zeros(b - a, dtype=out.dtype, device=dev)
            for k in range(2 * HALF_TAPS):
                acc += pad[i0 + (k + 1)] * tab[ph, k]
            res[a:b] = acc
        out = res
    if imp.clip_rms is not None:
        lim = float(imp.clip_rms * torch.sqrt(torch.mean(out.real ** 2 + out.imag ** 2)))
        out = torch.complex(torch.clamp(out.real, -lim, lim), torch.clamp(out.imag, -lim, lim))
    return out
