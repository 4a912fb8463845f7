"""Bulk synthetic-capture generator on the GPU (torch) -- BENCH / TEST SIGNAL SOURCE ONLY.

Bit-level content (L2 PDUs, PIDS, scrambling, convolutional code, interleaving, reference
carriers) comes from nrsc5_amd/synth.py; only the OFDM modulation and the channel (CFO, timing
offset, AWGN, cu8 quantisation) run in torch so that hundreds of 20-second captures can be
produced in seconds.  torch is plumbing here: nothing in this file is on the product path."""
from __future__ import annotations

import numpy as np
import torch

from . import synth


def payload_stream(n_frames: int, seed: int):
    """Truth + coded-bit matrix of one transmission: (p1 [F,146176] u8, pids [16F,80] u8, M [16F,23040] u8)."""
    p1s, pidss, ms = [], [], []
    for f in range(n_frames):
        prng = np.random.default_rng(0xBEEF00 + 7919 * seed + f)
        pdu, _ = synth.make_audio_pdu(f, prng)
        p1 = synth.p1_frame_bits(pdu)
        pids = np.stack([synth.pids_frame_bits(prng) for _ in range(16)])
        ms.append(synth.encode_l1_frame(p1, pids).reshape(16, synth.PM_BLOCK))
        p1s.append(p1)
        pidss.append(pids)
    return np.stack(p1s), np.concatenate(pidss), np.concatenate(ms)


_REF_TABLE = None


def _ref_table():
    """[16 bc][22 refs][32 symbols] +-1 reference-carrier values for PSMI 1."""
    global _REF_TABLE
    if _REF_TABLE is None:
        t = np.zeros((16, 22, 32), dtype=np.float32)
        for bc in range(16):
            for r, ri in enumerate(synth.REF_INDEX_MP1):
                t[bc, r] = synth.reference_bits(bc, 1, int(ri)).astype(np.float32) * 2 - 1
        _REF_TABLE = t
    return _REF_TABLE


def modulate(m_blocks: np.ndarray, device) -> torch.Tensor:
    """Coded bits [nblocks, 23040] -> unit-power complex64 baseband at 2x rate (1488375 S/s),
    nblocks * 32 * 4320 samples, before the receiver-side conjugation."""
    nb = m_blocks.shape[0]
    n = 2 * synth.FFT
    out = torch.empty(nb * 32 * 2 * synth.SYM, dtype=torch.complex64, device=device)
    pulse = torch.from_numpy(synth._pulse(2).astype(np.float32)).to(device)
    data_idx = torch.from_numpy(((synth.DATA_BINS_MP1 - synth.FFT // 2) % n).astype(np.int64)).to(device)
    ref_idx = torch.from_numpy(((synth.REF_BINS_MP1 - synth.FFT // 2) % n).astype(np.int64)).to(device)
    reft = torch.from_numpy(_ref_table()).to(device)
    inv = torch.tensor(1.0 / (1 + 1j), dtype=torch.complex64, device=device)
    step = 64                                      # blocks per chunk
    for b0 in range(0, nb, step):
        mb = torch.from_numpy(m_blocks[b0:b0 + step].astype(np.float32)).to(device)
        k = mb.shape[0]
        sb = mb.reshape(k, 32, 360, 2) * 2 - 1
        spec = torch.zeros((k, 32, n), dtype=torch.complex64, device=device)
        spec[:, :, data_idx] = torch.complex(sb[..., 0], sb[..., 1]) * inv
        bc = (torch.arange(b0, b0 + k, device=device) % 16)
        spec[:, :, ref_idx] = reft[bc].permute(0, 2, 1).to(torch.complex64)
        t = torch.fft.ifft(spec, dim=2) * n
        ext = torch.cat([t, t[:, :, :2 * synth.CP]], dim=2) * pulse
        out[b0 * 32 * 4320:(b0 + k) * 32 * 4320] = ext.reshape(-1)
    out /= torch.sqrt(torch.mean(out.real ** 2 + out.imag ** 2))
    return out


def channel_cu8(sig: torch.Tensor, cfo_hz: float, offset: int, snr_db: float, seed: int,
                rms_lsb: float = 20.0, tail: int = 8640, out: torch.Tensor | None = None) -> torch.Tensor:
    """CFO -> timing offset -> AWGN -> conj -> cu8 (interleaved I,Q bytes)."""
    dev = sig.device
    n = sig.shape[0]
    total = offset + n + tail
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    sigma = 10 ** (-snr_db / 20) / np.sqrt(2)
    re = torch.randn(total, generator=g, device=dev, dtype=torch.float32) * sigma
    im = torch.randn(total, generator=g, device=dev, dtype=torch.float32) * sigma
    ph = torch.arange(n, device=dev, dtype=torch.float64) * (2 * np.pi * cfo_hz / synth.FS_CU8)
    ph = torch.remainder(ph, 2 * np.pi).to(torch.float32)
    rot = sig * torch.complex(torch.cos(ph), torch.sin(ph))
    re[offset:offset + n] += rot.real
    im[offset:offset + n] += rot.imag
    scale = rms_lsb / np.sqrt(2)
    if out is None:
        out = torch.empty(2 * total, dtype=torch.uint8, device=dev)
    v = out[:2 * total].view(total, 2)
    v[:, 0] = torch.clamp(torch.round(127 + scale * re), 0, 255).to(torch.uint8)
    v[:, 1] = torch.clamp(torch.round(127 - scale * im), 0, 255).to(torch.uint8)     # conj
    return out[:2 * total]


def stream_params(k: int):
    """Config-3 family of SURVEY.md 8d: seed 1000+k, CFO uniform +-300 Hz, offset [0,4320), SNR 15/20/25 dB.  Every stream with
    k % 4 == 1 (64 of 256) additionally sees an impaired channel (nrsc5_amd/channel.py): a receiver sample clock off by +-20 ...
    +-100 ppm -- the FINE-state timing feedback then works on every block (sync.c:455 -> acquire.c:112,259 -> sync_adjust) --
    and, by k % 16: 1 = the clock error alone, 5 = + two echoes inside the cyclic prefix with slow Doppler, 9 = + an analog FM host
    20 dB above the digital sidebands (digital level 9 LSB rms so that the host fits the 8-bit range), 13 = + 8 dB block-scale
    fading.  -> kwargs of channel_cu8 (+ "chan")."""
    rng = np.random.default_rng(1000 + k)
    prm = dict(cfo_hz=float(rng.uniform(-300, 300)), offset=int(rng.integers(0, 4320)),
               snr_db=(15.0, 20.0, 25.0)[k % 3], seed=1000 + k, chan=None, rms_lsb=20.0)
    if k % 4 == 1:
        from .channel import Impairments
        crng = np.random.default_rng(77000 + k)
        ppm = float(crng.uniform(20.0, 100.0) * (1 if crng.integers(0, 2) else -1))
        kind = k % 16
        if kind == 1:
            prm["chan"] = Impairments(ppm=ppm)
        elif kind == 5:
            prm["chan"] = Impairments(ppm=ppm, paths=((float(crng.uniform(5e-6, 40e-6)), float(crng.uniform(-10.0, -3.0)), float(crng.uniform(-2.0, 2.0)), float(crng.uniform(0, 6.28))),
                                                      (float(crng.uniform(5e-6, 40e-6)), float(crng.uniform(-12.0, -6.0)), float(crng.uniform(-2.0, 2.0)), float(crng.uniform(0, 6.28)))))
        elif kind == 9:
            prm["chan"] = Impairments(ppm=ppm, host_db=20.0)
            prm["rms_lsb"] = 9.0
        else:
            prm["chan"] = Impairments(ppm=ppm, fade_db=8.0, fade_period_s=float(crng.uniform(0.7, 3.0)))
    return prm


def receive_cu8(clean: torch.Tensor, prm: dict, tail: int = 8640, out: torch.Tensor | None = None) -> torch.Tensor:
    """One receiver's cu8 capture of a clean transmission: stream_params' channel (impairments first, then CFO / offset / AWGN / 8 bits)."""
    sig = clean
    if prm.get("chan") is not None:
        from . import channel
        sig = channel.apply_torch(clean, synth.FS_CU8, prm["chan"])
    return channel_cu8(sig, prm["cfo_hz"], prm["offset"], prm["snr_db"], prm["seed"], rms_lsb=prm.get("rms_lsb", 20.0), tail=tail, out=out)


STRIDE_SLACK = 4096          # samples: +100 ppm stretch a 20.8-s capture by 3100 samples


def channel_am(sig: torch.Tensor, cfo_hz: float, offset: int, noise: float, seed: int, fmt: str = "cs16", tail: int = 1080,
               burst: tuple | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
    """One receiver's view of a clean hybrid-AM transmission (synth_am.am_ma1_signal, complex64 on the device): CFO -> timing
    offset -> AWGN (+ an optional interference burst (first L1 frame, frames, sigma)) -> cs16 (100 LSB per grid unit) or cu8
    (0.8 LSB per grid unit, sample rate x 32) -- the channel of synth_am.am_ma1_capture with a per-stream torch generator."""
    from . import synth_am
    dev = sig.device
    over = 1 if fmt == "cs16" else 32
    fs = synth_am.FS_CS16 if fmt == "cs16" else synth_am.FS_CU8
    n = sig.shape[0]
    total = offset + n + tail * over
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    sigma = noise * np.sqrt(over) / np.sqrt(2)
    re = torch.randn(total, generator=g, device=dev, dtype=torch.float32) * sigma
    im = torch.randn(total, generator=g, device=dev, dtype=torch.float32) * sigma
    if burst is not None:
        f0, nf, bs = burst
        per = 8 * 32 * synth_am.SYM * over
        a = offset + int(f0 * per); b = min(total, a + int(nf * per))
        if b > a:
            re[a:b] += torch.randn(b - a, generator=g, device=dev, dtype=torch.float32) * (bs * np.sqrt(over) / np.sqrt(2))
            im[a:b] += torch.randn(b - a, generator=g, device=dev, dtype=torch.float32) * (bs * np.sqrt(over) / np.sqrt(2))
    ph = torch.arange(n, device=dev, dtype=torch.float64) * (2 * np.pi * cfo_hz / fs)
    ph = torch.remainder(ph, 2 * np.pi).to(torch.float32)
    rot = sig * torch.complex(torch.cos(ph), torch.sin(ph))
    re[offset:offset + n] += rot.real
    im[offset:offset + n] += rot.imag
    if fmt == "cs16":
        if out is None:
            out = torch.empty(2 * total, dtype=torch.int16, device=dev)
        v = out[:2 * total].view(total, 2)
        v[:, 0] = torch.clamp(torch.round(100.0 * re), -32768, 32767).to(torch.int16)
        v[:, 1] = torch.clamp(torch.round(100.0 * im), -32768, 32767).to(torch.int16)
    else:
        if out is None:
            out = torch.empty(2 * total, dtype=torch.uint8, device=dev)
        v = out[:2 * total].view(total, 2)
        v[:, 0] = torch.clamp(torch.round(127 + 0.8 * re), 0, 255).to(torch.uint8)
        v[:, 1] = torch.clamp(torch.round(127 + 0.8 * im), 0, 255).to(torch.uint8)
    return out[:2 * total]


def am_stream_params(k: int, n_frames: int):
    """configs[4] family: seed 5000+k, CFO uniform +-100 Hz, timing offset anywhere in an OFDM symbol (cs16 samples; x 32 for
    cu8) plus up to 8 symbols, noise 0.4 / 0.6 / 0.8 grid units; every 16th stream is hit by an interference burst that breaks
    the first L2 header of some P1 PDUs (the reference then drops to SYNC_STATE_NONE and re-acquires: frame.c:535-540)."""
    rng = np.random.default_rng(5000 + k)
    prm = dict(cfo_hz=float(rng.uniform(-100, 100)), offset=int(rng.integers(0, 9 * 270)), noise=(0.4, 0.6, 0.8)[k % 3], seed=5000 + k, burst=None, chan=None)
    if k % 16 == 5 and n_frames >= 12:
        prm["burst"] = (float(rng.uniform(5.0, n_frames - 6.0)), float(rng.uniform(0.2, 0.6)), 40.0)
    if k % 4 == 2:
        # a quarter of the receivers: sample clock off by +-5 ... +-20 ppm (the reference's AM receiver corrects timing in whole
        # samples only and loses frames beyond ~30 ppm: tests/common.py IMPAIRED_AM_CASES), every other one with an echo
        from .channel import Impairments
        crng = np.random.default_rng(78000 + k)
        ppm = float(crng.uniform(5.0, 20.0) * (1 if crng.integers(0, 2) else -1))
        prm["chan"] = Impairments(ppm=ppm, paths=((float(crng.uniform(40e-6, 150e-6)), float(crng.uniform(-12.0, -6.0)), float(crng.uniform(-0.5, 0.5)), float(crng.uniform(0, 6.28))),) if k % 8 == 2 else ())
    return prm
