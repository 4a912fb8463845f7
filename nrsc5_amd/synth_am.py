"""Synthetic NRSC-5 AM hybrid (service mode MA1) transmitter -- TEST / BENCH SIGNAL SOURCE ONLY.

Derived, like synth.py, by inverting the receiver stage by stage:

  L2 PDU + PCI        <- frame_push (3750 / 24000-bit frames)        (frame.c:645-714, 516-643)
  scrambler           <- descramble                                  (decode.c:279-294)
  K=9 encoders E1/E2  <- bit_errors re-encoder, codes, punctures     (decode.c:47-61, 234-277)
  P1/P3 bit mapping   <- interleaver_ma1 (+ 3-frame diversity delay) (decode.c:66-231)
  PIDS bit mapping    <- decode_process_pids_am                      (decode.c:474-505)
  constellations      <- gray4/gray8/qpsk/qam16/qam64, training cells (sync.c:37-88, 664-716)
  carrier mapping     <- sync_process_am (complementary sidebands)   (sync.c:612-636)
  reference carrier   <- find_block_am / find_ref_am                 (sync.c:209-252)
  OFDM symbol         <- acquire_process AM window (centre-referenced)(acquire.c:170-257)
  sample formats      <- input_push_cs16; cu8 via the 5-stage /32    (input.c:52-124)

Nothing here is on the product path.
"""
from __future__ import annotations

import dataclasses
import numpy as np

from . import synth

FFT = 256
CP = 14
SYM = FFT + CP                       # 270 samples @ 46511.71875 Hz
BLKSZ = 32
BLOCKS_PER_FRAME = 8
FS_CS16 = 1488375.0 / 32
FS_CU8 = 1488375.0
P1_BITS = 3750
P3_BITS = 24000
PIDS_BITS = 80
P1_PDU_LEN = (P1_BITS - 22) // 8     # 466
P3_PDU_LEN = (P3_BITS - 24) // 8     # 2997
GENS_E1 = (0o561, 0o657, 0o711)
GENS_E2 = (0o561, 0o753, 0o711)
PUNCT_E1 = np.array([1, 0, 1, 1, 0, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1], dtype=bool)    # decode.c:268
PUNCT_E2 = np.array([1, 0, 1, 1, 0, 0], dtype=bool)                                # decode.c:274
BL_DELAY, ML_DELAY = (2, 1, 5), (11, 6, 7)                                         # decode.c:26-29
BU_DELAY, MU_DELAY = (10, 8, 9), (4, 3, 0)
EL_DELAY, EU_DELAY = (0, 1), (2, 3, 5, 4)
PIDS_IL_DELAY = (0, 1, 12, 13, 6, 5, 18, 17, 11, 7, 23, 19)                        # decode.c:63-64
PIDS_IU_DELAY = (2, 4, 14, 16, 3, 8, 15, 20, 9, 10, 21, 22)

GRAY4_LEVEL = {0: -1.5, 2: -0.5, 3: 0.5, 1: 1.5}                                   # inverse of sync.c:37-47
GRAY8_LEVEL = {0: -3.5, 4: -2.5, 6: -1.5, 2: -0.5, 3: 0.5, 7: 1.5, 5: 2.5, 1: 3.5}  # inverse of sync.c:49-67
_G4 = np.array([GRAY4_LEVEL[c] for c in range(4)])
_G8 = np.array([GRAY8_LEVEL[c] for c in range(8)])
TRAIN_QAM64 = 5 | (4 << 3)           # 2.5 - 2.5j
TRAIN_QAM16 = 1 | (2 << 2)           # 1.5 - 0.5j
TRAIN_QPSK = 0 | (1 << 1)            # -0.5 + 0.5j

# relative carrier levels (the receiver normalises every carrier by its training cells)
LEVEL_CARRIER = 12.0
LEVEL_REF = 1.5
LEVEL_PRIMARY = 1.0                  # QAM64 grid unit
LEVEL_SECONDARY = 0.6                # QAM16 grid unit
LEVEL_TERTIARY = 1.0                 # QPSK grid unit (points at +-0.5)
LEVEL_PIDS = 0.6


def conv_encode_k9(info: np.ndarray, gens) -> np.ndarray:
    """Tail-biting K=9 rate-1/3 encoder (decode.c:234-259 with k=9) -> flat [3*len] coded bits."""
    out = np.zeros(info.shape + (3,), dtype=np.uint8)
    taps = [np.roll(info, k, axis=-1) for k in range(9)]
    for gi, g in enumerate(gens):
        acc = np.zeros_like(info)
        for k in range(9):
            if (g >> (8 - k)) & 1:
                acc ^= taps[k]
        out[..., gi] = acc
    return out.reshape(info.shape[:-1] + (-1,))


def _puncture(flat: np.ndarray, pattern: np.ndarray) -> np.ndarray:
    keep = np.resize(pattern, flat.shape[-1])
    return flat[..., keep]


def _rev_groups(logical: np.ndarray) -> np.ndarray:
    """bits[byte_start + byte_len - 1 - (i & 7)] = logical[i], last group may be short (frame.c:689-693)."""
    n = logical.shape[0]
    out = np.empty_like(logical)
    full = n // 8 * 8
    out[:full] = logical[:full].reshape(-1, 8)[:, ::-1].reshape(-1)
    out[full:] = logical[full:][::-1]
    return out


def frame_bits(pdu: bytes, nbits: int, start: int, step: int, pci_len: int, pci: int = synth.PCI_AUDIO) -> np.ndarray:
    """Bits as handed to frame_push for an AM logical channel (frame.c:645-714 inverted)."""
    logical = np.zeros(nbits, dtype=np.uint8)
    pos = start + step * np.arange(pci_len)
    mask = np.ones(nbits, dtype=bool)
    mask[pos] = False
    logical[pos] = [(pci >> (23 - h)) & 1 for h in range(pci_len)]
    logical[mask] = np.unpackbits(np.frombuffer(pdu, dtype=np.uint8))
    return _rev_groups(logical)


def _cell_index(b, k):
    """bit_map's cell for (block b, index k): decode.c:66-71."""
    col = (9 * k) % 25
    row = (11 * col + 16 * (k // 25) + 11 * (k // 50)) % 32
    return 25 * (b * 32 + row) + col


def _idx_tables():
    n = np.arange(18000)
    t = {}
    t["bl"] = (_cell_index(n // 2250, (n + n // 750 + 1) % 750), n % 3)
    t["ml"] = (_cell_index((3 * n + 3) % 8, (n + n // 3000 + 3) % 750), 3 + n % 3)
    t["bu"] = (_cell_index(n // 2250, (n + n // 750) % 750), n % 3)
    t["mu"] = (_cell_index((3 * n) % 8, (n + n // 3000 + 2) % 750), 3 + n % 3)
    n = np.arange(12000)
    t["el"] = (_cell_index((3 * n + n // 3000) % 8, (n + n // 6000) % 750), n % 2)
    n = np.arange(24000)
    t["eu"] = (_cell_index((3 * n + n // 3000 + 2 * (n // 12000)) % 8, (n + n // 6000) % 750), n % 4)
    # MA3 (decode.c:114-133): the tertiary / secondary partitions carry a second E1 code word like the primary pair
    n = np.arange(18000)
    t["ebl"] = (_cell_index((3 * n + 3) % 8, (n + n // 3000 + 3) % 750), n % 3)
    t["eml"] = (_cell_index((3 * n + 3) % 8, (n + n // 3000 + 3) % 750), 3 + n % 3)
    t["ebu"] = (_cell_index((3 * n) % 8, (n + n // 3000 + 2) % 750), n % 3)
    t["emu"] = (_cell_index((3 * n) % 8, (n + n // 3000 + 2) % 750), 3 + n % 3)
    return t


_IDX = _idx_tables()
P3_BITS_MA3 = 30000
P3_PDU_LEN_MA3 = (P3_BITS_MA3 - 24) // 8     # 3747
_SCR = synth.scrambler_sequence(P3_BITS_MA3)


def _set_bits(matrix: np.ndarray, key: str, bits: np.ndarray):
    cell, p = _IDX[key]
    np.bitwise_or.at(matrix, cell, (bits.astype(np.uint8) << p.astype(np.uint8)))


def _split_p1(coded72000: np.ndarray):
    c = coded72000.reshape(6000, 12)
    return (c[:, list(BL_DELAY)].reshape(-1), c[:, list(ML_DELAY)].reshape(-1),
            c[:, list(BU_DELAY)].reshape(-1), c[:, list(MU_DELAY)].reshape(-1))


def _split_p3(coded36000: np.ndarray):
    c = coded36000.reshape(6000, 6)
    return c[:, list(EL_DELAY)].reshape(-1), c[:, list(EU_DELAY)].reshape(-1)


def _train_rows(col):
    return (5 + 11 * col) % 32, (21 + 11 * col) % 32


def _with_training(matrix: np.ndarray, code: int) -> np.ndarray:
    m = matrix.reshape(8, 32, 25)
    for col in range(25):
        r1, r2 = _train_rows(col)
        assert not m[:, r1, col].any() and not m[:, r2, col].any()
        m[:, r1, col] = code
        m[:, r2, col] = code
    return m


def _qam64(code):
    return _G8[code & 7] + 1j * _G8[code >> 3]


def _qam16(code):
    return _G4[code & 3] + 1j * _G4[code >> 2]


def _qpsk(code):
    return ((code & 1) - 0.5) + 1j * ((code >> 1) - 0.5)


def block_spectrum_ma3(pl, pu, s, t, pids1, pids2, bc: int, rdbi: int = 0) -> np.ndarray:
    """All-digital MA3 layout (sync.c:612-767 with psmi == 2): nothing is complementary; primary = +-(2..26), secondary =
    +(28..52), tertiary = -(28..52), PIDS at -27 / +27, everything QAM64 except PIDS (QAM16)."""
    x = np.zeros((BLKSZ, FFT), dtype=np.complex128)
    c = FFT // 2
    col = np.arange(25)
    x[:, c - 2 - col] = -np.conj(LEVEL_PRIMARY * _qam64(pl))
    x[:, c + 2 + col] = LEVEL_PRIMARY * _qam64(pu)
    x[:, c + 28 + col] = LEVEL_PRIMARY * _qam64(s)
    x[:, c - 28 - col] = -np.conj(LEVEL_PRIMARY * _qam64(t))
    x[:, c - 27] = -np.conj(LEVEL_PIDS * _qam16(pids1))
    x[:, c + 27] = LEVEL_PIDS * _qam16(pids2)
    # The reference's acquisition filter (acquire.c:63-96) only passes carriers 53..81, which MA3 does not occupy: its
    # cyclic-prefix correlation would never find symbol timing.  Test-vector aid: fixed QPSK filler in +-(57..81), which
    # the MA3 demodulator ignores, gives the correlator something to lock to.
    filler = ((((col * 7 + 3) % 4) & 1) - 0.5) + 1j * ((((col * 7 + 3) % 4) >> 1) - 0.5)
    x[:, c + 57 + col] = 2.0 * filler
    x[:, c - 57 - col] = 2.0 * np.conj(filler)
    ref = 1j * LEVEL_REF * (reference_bits(bc, psmi=2, rdbi=rdbi).astype(np.float64) * 2 - 1)
    x[:, c + 1] = ref
    x[:, c - 1] = -np.conj(ref)
    x[:, c] = LEVEL_CARRIER
    return x


def reference_bits(bc: int, psmi: int = 1, rdbi: int = 0) -> np.ndarray:
    """32 BPSK bits of the AM reference carrier for block `bc` (find_block_am, sync.c:209-238)."""
    d = np.zeros(32, dtype=np.uint8)
    d[[1, 2, 5, 9, 21, 22]] = 1
    # d7 = pli = 0 (d8 = d7), d11 = hppi, d12 = aabi: all 0; d15 = rdbi (reduced digital bandwidth: the receiver skips the P3 frame,
    # decode.c:524)
    d[15] = rdbi & 1
    d[17], d[18], d[19] = (bc >> 2) & 1, (bc >> 1) & 1, bc & 1
    d[20] = d[15] ^ d[16] ^ d[17] ^ d[18] ^ d[19]
    for k in range(5):
        d[26 + k] = (psmi >> (4 - k)) & 1
    d[31] = d[23] ^ d[24] ^ d[25] ^ d[26] ^ d[27] ^ d[28] ^ d[29] ^ d[30]
    return d


def pids_symbols(pids_bits80: np.ndarray):
    """80 PIDS bits (as given to pids_frame_push) -> (pids1[32], pids2[32]) QAM16 codes (decode.c:474-505)."""
    c = conv_encode_k9(pids_bits80 ^ _SCR[:PIDS_BITS], GENS_E2).reshape(10, 24)
    il = c[:, list(PIDS_IL_DELAY)].reshape(-1)
    iu = c[:, list(PIDS_IU_DELAY)].reshape(-1)
    n = np.arange(120)
    p = n % 4
    s1 = np.zeros(32, dtype=np.uint8)
    s2 = np.zeros(32, dtype=np.uint8)
    k = (n + n // 60 + 11) % 30
    np.bitwise_or.at(s1, (11 * (k + k // 15) + 3) % 32, (il << p).astype(np.uint8))
    k = (n + n // 60) % 30
    np.bitwise_or.at(s2, (11 * (k + k // 15) + 3) % 32, (iu << p).astype(np.uint8))
    for row in (8, 24):
        assert s1[row] == 0 and s2[row] == 0
        s1[row] = s2[row] = TRAIN_QAM16
    return s1, s2


def block_spectrum(pl, pu, s, t, pids1, pids2, bc: int, rdbi: int = 0) -> np.ndarray:
    """One block of symbol codes [32][25] x4 + PIDS + reference -> X[32 symbols, 256 bins] (bin 128 = carrier)."""
    x = np.zeros((BLKSZ, FFT), dtype=np.complex128)
    c = FFT // 2
    col = np.arange(25)

    def comp(idx, v):                    # complementary pair: receiver forms U + (-conj(L))
        x[:, c + idx] = v
        x[:, c - idx] = -np.conj(v)

    x[:, c - 57 - col] = -np.conj(LEVEL_PRIMARY * _qam64(pl))      # receiver negates-conjugates the lower sideband
    x[:, c + 57 + col] = LEVEL_PRIMARY * _qam64(pu)
    comp(28 + col, LEVEL_SECONDARY * _qam16(s))
    comp(2 + col, LEVEL_TERTIARY * _qpsk(t))
    comp(27, LEVEL_PIDS * _qam16(pids1))
    comp(53, LEVEL_PIDS * _qam16(pids2))
    comp(1, 1j * LEVEL_REF * (reference_bits(bc, rdbi=rdbi).astype(np.float64) * 2 - 1))
    x[:, c] = LEVEL_CARRIER
    return x


def ofdm_modulate(x: np.ndarray, oversample: int = 1) -> np.ndarray:
    """X[nsym, 256] -> time samples; carrier phases are referenced to the symbol centre (sample 135), because
    the receiver rotates its FFT input by (FFT-CP)/2 = 121 samples (acquire.c:239-247)."""
    n = FFT * oversample
    spec = np.zeros((x.shape[0], n), dtype=np.complex128)
    spec[:, (np.arange(FFT) - FFT // 2) % n] = x
    period = np.fft.ifft(spec, axis=1) * n                     # s(t), t = m / oversample, period 256
    m = (np.arange(SYM * oversample) - (SYM // 2) * oversample) % n
    tt = np.arange(SYM * oversample) / oversample
    pulse = np.ones_like(tt)
    pulse[tt < CP] = np.sin(np.pi / 2 * tt[tt < CP] / CP)
    tail = tt >= FFT
    pulse[tail] = np.cos(np.pi / 2 * (tt[tail] - FFT) / CP)
    return (period[:, m] * pulse[None, :]).reshape(-1)


@dataclasses.dataclass
class AmCapture:
    iq: np.ndarray            # int16 interleaved I,Q (cs16 @46511.71875) or uint8 (cu8 @1488375)
    p1_frames: list           # per L1 frame: [8] x uint8[3750] bits as given to frame_push
    p3_frames: list           # per L1 frame: uint8[24000]
    pids_frames: list         # per block: uint8[80]
    cfo_hz: float
    offset: int
    seed: int


def am_ma1_signal(n_frames: int, seed: int = 1, fmt: str = "cs16", mode: str = "MA1", rdbi: int = 0):
    """The clean transmission of am_ma1_capture (no CFO, offset, noise): complex128 baseband at the capture's sample rate and
    the transmitted truth (P1 frames, P3 frames, PIDS frames).  bench.py puts many receivers' channels on one such signal
    (synth_torch.channel_am)."""
    oversample = 1 if fmt == "cs16" else 32
    coded_p1, coded_p3, p1_list, p3_list, pids_list, chunks = [], [], [], [], [], []
    ma3 = mode == "MA3"
    for f in range(n_frames):
        prng = np.random.default_rng(0xA11CE + 1000003 * seed + f)
        p1 = []
        for b in range(BLOCKS_PER_FRAME):
            pdu, _ = synth.make_audio_pdu(8 * f + b, prng, nop=4, pdu_len=P1_PDU_LEN, slack=20)
            p1.append(frame_bits(pdu, P1_BITS, 120, 160, 22))
        p1 = np.stack(p1)
        c1 = _puncture(conv_encode_k9(p1 ^ _SCR[None, :P1_BITS], GENS_E1), PUNCT_E1).reshape(-1)      # 72000
        coded_p1.append(_split_p1(c1))
        if not ma3:
            pdu3, _ = synth.make_audio_pdu(f, prng, nop=16, pdu_len=P3_PDU_LEN, stream_id=1)
            p3 = frame_bits(pdu3, P3_BITS, 120, 992, 24)
            el, eu = _split_p3(_puncture(conv_encode_k9(p3 ^ _SCR[:P3_BITS], GENS_E2), PUNCT_E2))
        else:
            pdu3, _ = synth.make_audio_pdu(f, prng, nop=16, pdu_len=P3_PDU_LEN_MA3, stream_id=1)
            p3 = frame_bits(pdu3, P3_BITS_MA3, 120, 1240, 24)
            coded_p3.append(_split_p1(_puncture(conv_encode_k9(p3 ^ _SCR, GENS_E1), PUNCT_E1)))
        pids = np.stack([synth.pids_frame_bits(prng) for _ in range(BLOCKS_PER_FRAME)])

        pl = np.zeros(8 * 32 * 25, dtype=np.uint8); pu = np.zeros_like(pl)
        s = np.zeros_like(pl); t = np.zeros_like(pl)
        bl, ml, bu, mu = coded_p1[f]
        _set_bits(pl, "ml", ml); _set_bits(pu, "mu", mu)                    # main: this frame
        if f >= 3:                                                          # backup: content of 3 frames ago
            _set_bits(pl, "bl", coded_p1[f - 3][0]); _set_bits(pu, "bu", coded_p1[f - 3][2])
        if not ma3:
            _set_bits(t, "el", el); _set_bits(s, "eu", eu)
        else:
            _, eml, _, emu = coded_p3[f]
            _set_bits(t, "eml", eml); _set_bits(s, "emu", emu)
            if f >= 3:
                _set_bits(t, "ebl", coded_p3[f - 3][0]); _set_bits(s, "ebu", coded_p3[f - 3][2])
        pl = _with_training(pl, TRAIN_QAM64); pu = _with_training(pu, TRAIN_QAM64)
        s = _with_training(s, TRAIN_QAM64 if ma3 else TRAIN_QAM16); t = _with_training(t, TRAIN_QAM64 if ma3 else TRAIN_QPSK)
        for bc in range(BLOCKS_PER_FRAME):
            s1, s2 = pids_symbols(pids[bc])
            spec = block_spectrum_ma3 if ma3 else block_spectrum
            chunks.append(ofdm_modulate(spec(pl[bc], pu[bc], s[bc], t[bc], s1, s2, bc, rdbi), oversample))
            pids_list.append(pids[bc])
        p1_list.append(p1); p3_list.append(p3)
    return np.concatenate(chunks), p1_list, p3_list, pids_list


def am_ma1_capture(n_frames: int, seed: int = 1, cfo_hz: float = 3.0, offset: int = 1000, noise: float = 0.5,
                   fmt: str = "cs16", tail_samples: int = 1080, unit_lsb: float | None = None, mode: str = "MA1",
                   burst: tuple | None = None, rdbi: int = 0, chan=None) -> AmCapture:
    """Hybrid-AM MA1 capture of n_frames L1 frames (8 blocks x 32 symbols each).  `noise` = per-sample complex
    noise sigma in primary QAM64 grid units; `unit_lsb` = LSBs per grid unit (default 100 for cs16, 0.8 for cu8)."""
    rng = np.random.default_rng(seed)
    oversample = 1 if fmt == "cs16" else 32
    fs = FS_CS16 if fmt == "cs16" else FS_CU8
    sig, p1_list, p3_list, pids_list = am_ma1_signal(n_frames, seed, fmt, mode, rdbi)
    if chan is not None and chan.active():                     # channel.Impairments: sample-clock error, echoes, fading
        from . import channel
        sig = channel.apply(sig, fs, chan)
    n = sig.shape[0]
    if cfo_hz:
        sig *= np.exp(2j * np.pi * cfo_hz / fs * np.arange(n))
    full = np.zeros(offset + n + tail_samples * oversample, dtype=np.complex128)
    full[offset:offset + n] = sig
    sigma = noise * np.sqrt(oversample)      # keep the in-band noise density independent of the sample rate
    full += sigma * (rng.standard_normal(full.shape[0]) + 1j * rng.standard_normal(full.shape[0])) / np.sqrt(2)
    if burst is not None:                                      # (first L1 frame, number of frames, sigma): an interference burst
        f0, nf, bs = burst
        a = offset + int(f0 * 8 * BLKSZ * SYM * oversample)
        b = min(full.shape[0], a + int(nf * 8 * BLKSZ * SYM * oversample))
        brng = np.random.default_rng(seed + 4242)
        full[a:b] += bs * np.sqrt(oversample) * (brng.standard_normal(b - a) + 1j * brng.standard_normal(b - a)) / np.sqrt(2)
    if fmt == "cs16":
        unit = 100.0 if unit_lsb is None else unit_lsb
        q = np.rint(unit * np.stack([full.real, full.imag], axis=1))
        iq = np.clip(q, -32768, 32767).astype(np.int16).reshape(-1)
    else:
        unit = 0.8 if unit_lsb is None else unit_lsb
        q = np.rint(127 + unit * np.stack([full.real, full.imag], axis=1))
        iq = np.clip(q, 0, 255).astype(np.uint8).reshape(-1)
    return AmCapture(iq, p1_list, p3_list, pids_list, cfo_hz, offset, seed)
