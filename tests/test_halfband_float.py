"""CPU-only: the float32 formulation of the half-band decimator used by the zero-copy batch path
(nrsc5_amd/csrc/halfband_raw.h) is EXACTLY the reference's integer code (firdecim_q15.c:137-165 via the oracle), and
the divider-free x / 32767.0f equals the IEEE quotient for every int16 x.  numpy float32 mirrors the device arithmetic
operation by operation (every intermediate is an exactly representable small integer or a 24-bit product)."""
import numpy as np


def _fma32(a, b, c):
    """float32 fma: a*b+c with ONE rounding (products of a 24-bit and a 15-bit number are exact in float64)"""
    return (a.astype(np.float64) * np.float64(b) + c.astype(np.float64)).astype(np.float32)


def test_q15_to_float_equals_ieee_division_for_every_int16():
    y = np.arange(-32768, 32768, dtype=np.int64).astype(np.float32)
    r = np.float32(1.0) / np.float32(32767.0)
    q0 = (y * r).astype(np.float32)
    e = _fma32(-q0, 32767.0, y)
    q = (q0.astype(np.float64) + e.astype(np.float64) * np.float64(r)).astype(np.float32)
    assert np.array_equal(q, y / np.float32(32767.0))
    assert (q0 != y / np.float32(32767.0)).sum() > 1000          # the correction step is not decoration


def test_float_halfband_equals_oracle(oracle):
    rng = np.random.default_rng(11)
    n = 60000
    iq = rng.integers(0, 256, size=4 * n, dtype=np.uint8)
    iq[:4000] = rng.choice(np.array([0, 255], dtype=np.uint8), size=4000)      # full-scale: the accumulator's extremes
    exp, _ = oracle.halfband_fm_cu8(iq)                                          # [n, 2] int16, zero history
    x = iq.astype(np.float32).reshape(-1, 2) - np.float32(127.0)                # raw complex samples as x' = byte - 127
    x = np.concatenate([np.zeros((14, 2), dtype=np.float32), x])               # history before the stream: Q15 zero
    taps = [np.float32(v) for v in (0.6062333583831787, -0.13481467962265015, 0.032919470220804214, -0.00410953676328063)]
    hbq = [np.float32(np.int16(t * np.float32(32767.0))) for t in taps][::-1]   # window order, as DevTables::hb_q15
    tf = [h * np.float32(1.0 / 512.0) for h in hbq]
    m = np.arange(n)
    acc = np.float32(64.0) * x[2 * m + 7]                                        # raw sample 2a - 7 (index shifted by the 14 zeros)
    for i in range(4):
        s = x[2 * m + 2 * i] + x[2 * m + 14 - 2 * i]
        acc = acc + np.floor((s * tf[i]).astype(np.float32))
    assert np.array_equal(acc.astype(np.int16), exp)
    assert np.abs(exp.astype(int)).max() < 32768 - 8192                          # the int16 accumulator has room: it never wraps


def test_round_down_fma_chain_equals_oracle(oracle):
    """k_mixfft's form (halfband_raw.h: hb_fma4 under round-toward-minus-infinity): acc <- RD(acc + s * t_i / 512), started at
    HB_BIAS + 64 (o - 127).  Modelled exactly: the fma's unrounded value is acc + s * T_i / 512 (rational), RD to the float32
    grid; while acc is an integer in [2^23, 2^24) the grid step is 1, so RD is floor.  Checks the identity on random and
    full-scale input and that the accumulator never leaves the binade (which is what makes the single rounding a floor)."""
    rng = np.random.default_rng(12)
    n = 40000
    iq = rng.integers(0, 256, size=4 * n, dtype=np.uint8)
    iq[:8000] = rng.choice(np.array([0, 255], dtype=np.uint8), size=8000)
    exp, _ = oracle.halfband_fm_cu8(iq)
    raw = np.concatenate([np.full((14, 2), 127, dtype=np.int64), iq.astype(np.int64).reshape(-1, 2)])   # bytes; history = 127
    taps = [np.float32(v) for v in (0.6062333583831787, -0.13481467962265015, 0.032919470220804214, -0.00410953676328063)]
    T = [int(np.int16(t * np.float32(32767.0))) for t in taps][::-1]             # integer Q15 taps, window order
    BIAS = 12582912                                                               # 1.5 * 2^23
    m = np.arange(n)
    acc = BIAS - 127 * 64 + 64 * raw[2 * m + 7]                                   # fma(o, 64, HB_BIAS - 127 * 64): exact
    lo, hi = acc.min(), acc.max()
    for i in range(4):
        # the kernel biases the even-indexed sample of each pair by -254 (= both samples' -127) before the pair sum
        s = (raw[2 * m + 2 * i] - 254) + raw[2 * m + 14 - 2 * i]
        acc = acc + np.floor_divide(s * T[i], 512)                                # RD(acc + s T_i / 512) on a grid of step 1
        lo, hi = min(lo, acc.min()), max(hi, acc.max())
    assert 2 ** 23 <= lo and hi < 2 ** 24                                         # one binade: ulp 1 throughout
    assert np.array_equal((acc - BIAS).astype(np.int16), exp)
    # and the products are exact inside the fma: |s| < 2^9, |T| < 2^15 -> 24 significant bits
    assert max(abs(t) for t in T) < 2 ** 15


def test_float_phase_wrap_equals_double_formulation():
    """k_sync's Costas loop wraps the phase as the reference does (sync.c: `if (phase > M_PI) phase -= 2 * M_PI`, evaluated in
    double and rounded to float) but in float32 arithmetic while |phase| < 12: same decision, same result, bit for bit."""
    rng = np.random.default_rng(5)
    edge = np.nextafter(np.float32(np.pi), np.float32(0))       # largest float below pi
    ph = np.concatenate([rng.uniform(-12.0, 12.0, 2_000_000), edge + np.arange(-50, 50) * 2.4e-7, -edge + np.arange(-50, 50) * 2.4e-7]).astype(np.float32)
    d = ph.astype(np.float64)
    ref = np.where(d > np.pi, d - 2 * np.pi, d)
    ref = np.where(ref < -np.pi, ref + 2 * np.pi, ref).astype(np.float32)   # the reference applies both tests in sequence
    hi, lo, below = np.float32(6.28318548202514648), np.float32(-1.74845553e-7), np.float32(3.14159250259399414)
    assert below == edge
    mine = np.where(ph > below, ((ph - hi).astype(np.float32) - lo).astype(np.float32), ph)
    mine = np.where(mine < -below, ((mine + hi).astype(np.float32) + lo).astype(np.float32), mine)
    assert np.array_equal(ref, mine)
