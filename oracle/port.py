"""TEST INFRASTRUCTURE: ctypes driver for oracle/liboracle.so (the C restatement,
oracle/nrsc5_oracle.c).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this; the product path never does."""
from __future__ import annotations

import ctypes
import os
import subprocess
import numpy as np

from . import ref as _ref

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

TAP_Q15, TAP_FFT, TAP_SOFT = 1, 2, 4


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("nrsc5_oracle.c", "nrsc5_oracle_am.c", "nrsc5_oracle_l2.c", "nrsc5_oracle.h", "cpu_fft.c", "cpu_fft.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class L2Pdu(ctypes.Structure):
    _fields_ = ([("start", ctypes.c_uint32), ("psd_off", ctypes.c_uint32), ("psd_len", ctypes.c_int32), ("audio_off", ctypes.c_uint32),
                 ("crc_bad_lo", ctypes.c_uint32), ("crc_bad_hi", ctypes.c_uint32), ("pdu_marker", ctypes.c_uint32),
                 ("hef_pdu_len", ctypes.c_uint16), ("loc", ctypes.c_uint16 * 64)] +
                [(n, ctypes.c_uint8) for n in ("codec_mode", "stream_id", "pdu_seq", "blend_control", "per_stream_delay", "common_delay",
                                               "latency", "pfirst", "plast", "seq", "nop", "hef", "la_location", "rs_corrections",
                                               "class_ind", "prog_num", "access", "prog_type", "applied_services", "elastic_seq",
                                               "align_offset", "skipped")])


class L2Frame(ctypes.Structure):
    _fields_ = [("pci", ctypes.c_uint32), ("nbytes", ctypes.c_uint32), ("n_pdu", ctypes.c_uint32), ("status", ctypes.c_uint32),
                ("end_offset", ctypes.c_uint32), ("lost_sync", ctypes.c_uint32), ("pdu", L2Pdu * 16)]


L2_STATUS = ("end", "no_audio", "fixed_data", "header_rs", "bad_locators", "too_many_pdus", "hef_overrun", "bad_stream", "bad_length", "audio_end")


def l2_frame_to_dict(fr) -> dict:
    """Works for any ctypes struct with the field names above (the oracle's and the product's)."""
    out = {k: int(getattr(fr, k)) for k in ("pci", "nbytes", "n_pdu", "status", "end_offset", "lost_sync")}
    out["pdus"] = []
    for i in range(min(fr.n_pdu, 16)):
        p = fr.pdu[i]
        d = {name: int(getattr(p, name)) for name, _ in p._fields_ if name != "loc"}
        d["loc"] = [int(x) for x in p.loc[:p.nop]]
        out["pdus"].append(d)
    return out


class _Snapshot(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("sync_state", "bc", "psmi", "cfo_wait", "samperr_next", "mer_cnt", "acq_cfo", "keep_extra")] + \
               [(n, ctypes.c_float) for n in ("angle_next", "prev_angle", "phase_re", "phase_im", "error_lb", "error_ub")] + \
               [("costas_freq", ctypes.c_float * 30), ("costas_phase", ctypes.c_float * 30)]


P1_HOOK = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_uint)


class Oracle:
    def __init__(self):
        self.lib = L = ctypes.CDLL(build())
        vp, sz = ctypes.c_void_p, ctypes.c_size_t
        L.orc_open.restype = vp
        L.orc_close.argtypes = [vp]
        L.orc_reset.argtypes = [vp]
        L.orc_set_taps.argtypes = [vp, ctypes.c_uint, ctypes.c_uint]
        L.orc_set_p1_hook.argtypes = [vp, P1_HOOK, vp]
        L.orc_push_cu8.argtypes = [vp, vp, ctypes.c_uint32]
        L.orc_push_cs16.argtypes = [vp, vp, ctypes.c_uint32]
        L.orc_force_resync.argtypes = [vp]
        L.orc_buf.argtypes = [vp, ctypes.c_int, ctypes.POINTER(vp)]
        L.orc_buf.restype = sz
        L.orc_clear_bufs.argtypes = [vp]
        L.orc_snapshot.argtypes = [vp, ctypes.POINTER(_Snapshot)]
        L.orc_halfband_fm_cu8.argtypes = [vp, vp, sz, vp]
        L.orc_halfband_fm_cu8.restype = sz
        L.orc_fir32_fm.argtypes = [vp, vp, sz, vp]
        L.orc_cp_correlate_fm.argtypes = [vp, ctypes.POINTER(ctypes.c_int), vp]
        L.orc_deinterleave_p1.argtypes = [vp, vp]
        L.orc_deinterleave_pids.argtypes = [vp, ctypes.c_uint, vp]
        L.orc_viterbi_k7.argtypes = [vp, ctypes.c_int, vp]
        L.orc_viterbi.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp, vp]
        L.orc_descramble.argtypes = [vp, ctypes.c_uint]
        L.orc_bit_errors_k7.argtypes = [vp, vp, ctypes.c_int]
        L.oracle_fft_forward.argtypes = [ctypes.c_int, vp, vp]
        L.orc_l2_first_header_ok.argtypes = [vp, ctypes.c_uint]
        L.orc_rs255_247_decode.argtypes = [vp]
        L.orc_pids_crc_ok.argtypes = [vp]
        L.orc_l2_index.argtypes = [vp, ctypes.c_uint, vp, vp]
        L.orc_crc8.argtypes = [vp, ctypes.c_uint]
        L.orc_crc8.restype = ctypes.c_uint8
        # AM
        L.orc_am_open.restype = vp
        L.orc_am_close.argtypes = [vp]
        L.orc_am_set_taps.argtypes = [vp, ctypes.c_uint, ctypes.c_uint]
        L.orc_am_set_p1_hook.argtypes = [vp, P1_HOOK, vp]
        L.orc_am_push_cu8.argtypes = [vp, vp, ctypes.c_uint32]
        L.orc_am_push_cs16.argtypes = [vp, vp, ctypes.c_uint32]
        L.orc_am_force_resync.argtypes = [vp]
        L.orc_am_buf.argtypes = [vp, ctypes.c_int, ctypes.POINTER(vp)]
        L.orc_am_buf.restype = sz
        L.orc_am_decim_new.restype = vp
        L.orc_am_decim_free.argtypes = [vp]
        L.orc_am_decimate_cu8.argtypes = [vp, vp, sz, vp]
        L.orc_am_decimate_cu8.restype = sz
        L.orc_am_deinterleave.argtypes = [ctypes.c_int] + [vp] * 10
        L.orc_am_deinterleave_pids.argtypes = [vp, ctypes.c_int, vp]
        L.orc_bit_errors.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_int]

    # ---- stage functions -------------------------------------------------------------
    def halfband_fm_cu8(self, iq: np.ndarray, hist: np.ndarray | None = None):
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        h = np.zeros((14, 2), dtype=np.int16) if hist is None else np.ascontiguousarray(hist, dtype=np.int16).copy()
        out = np.zeros((iq.size // 4, 2), dtype=np.int16)
        self.lib.orc_halfband_fm_cu8(h.ctypes.data, iq.ctypes.data, iq.size, out.ctypes.data)
        return out, h

    def fir32_fm(self, x: np.ndarray, hist: np.ndarray | None = None):
        x = np.ascontiguousarray(x, dtype=np.int16)
        h = np.zeros((31, 2), dtype=np.int16) if hist is None else np.ascontiguousarray(hist, dtype=np.int16).copy()
        out = np.zeros_like(x)
        self.lib.orc_fir32_fm(h.ctypes.data, x.ctypes.data, x.shape[0], out.ctypes.data)
        return out, h

    def cp_correlate_fm(self, filtered: np.ndarray):
        f = np.ascontiguousarray(filtered, dtype=np.int16)
        assert f.shape == (71280, 2)
        se = ctypes.c_int()
        pk = np.zeros(2, dtype=np.float32)
        self.lib.orc_cp_correlate_fm(f.ctypes.data, ctypes.byref(se), pk.ctypes.data)
        return se.value, complex(pk[0], pk[1])

    def deinterleave_p1(self, pm: np.ndarray) -> np.ndarray:
        pm = np.ascontiguousarray(pm, dtype=np.int8)
        out = np.zeros(438528, dtype=np.int8)
        self.lib.orc_deinterleave_p1(pm.ctypes.data, out.ctypes.data)
        return out

    def deinterleave_pids(self, pm: np.ndarray, bc: int) -> np.ndarray:
        pm = np.ascontiguousarray(pm, dtype=np.int8)
        out = np.zeros(240, dtype=np.int8)
        self.lib.orc_deinterleave_pids(pm.ctypes.data, bc, out.ctypes.data)
        return out

    def viterbi_k7(self, soft: np.ndarray) -> np.ndarray:
        soft = np.ascontiguousarray(soft, dtype=np.int8)
        out = np.zeros(soft.size // 3, dtype=np.uint8)
        self.lib.orc_viterbi_k7(soft.ctypes.data, soft.size // 3, out.ctypes.data)
        return out

    def viterbi(self, soft: np.ndarray, k: int, gens) -> np.ndarray:
        soft = np.ascontiguousarray(soft, dtype=np.int8)
        g = (ctypes.c_uint * 3)(*gens)
        out = np.zeros(soft.size // 3, dtype=np.uint8)
        self.lib.orc_viterbi(soft.ctypes.data, soft.size // 3, k, g, out.ctypes.data)
        return out

    def descramble(self, bits: np.ndarray) -> np.ndarray:
        b = np.ascontiguousarray(bits, dtype=np.uint8).copy()
        self.lib.orc_descramble(b.ctypes.data, b.size)
        return b

    def bit_errors_k7(self, coded: np.ndarray, decoded: np.ndarray) -> int:
        c = np.ascontiguousarray(coded, dtype=np.int8)
        d = np.ascontiguousarray(decoded, dtype=np.uint8)
        return self.lib.orc_bit_errors_k7(c.ctypes.data, d.ctypes.data, d.size)

    def fft(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.complex64)
        out = np.zeros_like(x)
        self.lib.oracle_fft_forward(x.size, x.ctypes.data, out.ctypes.data)
        return out

    def l2_first_header_ok(self, bits: np.ndarray) -> bool:
        """The L2 -> L1 feedback decision of frame_process for one P1 frame (bits as handed to frame_push)."""
        b = np.ascontiguousarray(bits, dtype=np.uint8)
        return bool(self.lib.orc_l2_first_header_ok(b.ctypes.data, b.size))

    def pids_crc_ok(self, bits80: np.ndarray) -> bool:
        b = np.ascontiguousarray(bits80, dtype=np.uint8)
        return bool(self.lib.orc_pids_crc_ok(b.ctypes.data))

    def l2_index(self, bits: np.ndarray):
        """frame_push + frame_process restated as an index: (dict, PDU bytes with RS-corrected headers)."""
        b = np.ascontiguousarray(bits, dtype=np.uint8)
        fr = L2Frame()
        by = np.zeros(b.size // 8 + 8, dtype=np.uint8)
        if self.lib.orc_l2_index(b.ctypes.data, b.size, ctypes.addressof(fr), by.ctypes.data) != 0:
            raise ValueError("unknown frame length %d" % b.size)
        return l2_frame_to_dict(fr), by[:fr.nbytes].copy()

    def l2_index_struct(self, bits: np.ndarray):
        """(ctypes struct with the nrsc5hip_l2_frame layout, PDU bytes) -- what a consumer of the C ABI receives."""
        b = np.ascontiguousarray(bits, dtype=np.uint8)
        fr = L2Frame()
        by = np.zeros(b.size // 8 + 8, dtype=np.uint8)
        if self.lib.orc_l2_index(b.ctypes.data, b.size, ctypes.addressof(fr), by.ctypes.data) != 0:
            raise ValueError("unknown frame length %d" % b.size)
        return fr, by[:fr.nbytes].copy()

    def l2_hook(self):
        """p1_hook for run(): drop to SYNC_NONE exactly when the reference's frame_process would."""
        return lambda bits: 0 if self.l2_first_header_ok(bits) else 1

    def rs_decode(self, word255: np.ndarray):
        w = np.ascontiguousarray(word255, dtype=np.uint8).copy()
        return self.lib.orc_rs255_247_decode(w.ctypes.data), w

    # ---- AM stage functions -------------------------------------------------------------
    def am_decimate_cu8(self, chunks) -> np.ndarray:
        """Feed a list of cu8 chunks through one decimator state; returns the concatenated Q15 output [n, 2]."""
        d = self.lib.orc_am_decim_new()
        outs = []
        try:
            for c in chunks:
                c = np.ascontiguousarray(c, dtype=np.uint8)
                out = np.zeros((c.size // 64 + 2, 2), dtype=np.int16)
                n = self.lib.orc_am_decimate_cu8(d, c.ctypes.data, c.size, out.ctypes.data)
                outs.append(out[:n])
        finally:
            self.lib.orc_am_decim_free(d)
        return np.concatenate(outs) if outs else np.zeros((0, 2), dtype=np.int16)

    def am_deinterleave(self, psmi, pl, pu, s, t, queues=None):
        """interleaver_ma1 for one L1 frame; queues = [ml, mu, eml, emu] delay lines (uint8[54000]), updated in place."""
        arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in (pl, pu, s, t)]
        if queues is None:
            queues = [np.zeros(54000, dtype=np.uint8) for _ in range(4)]
        v1 = np.zeros(90000, dtype=np.int8)
        v3 = np.zeros(90000, dtype=np.int8)
        self.lib.orc_am_deinterleave(psmi, *[a.ctypes.data for a in arrs], *[q.ctypes.data for q in queues], v1.ctypes.data, v3.ctypes.data)
        return v1, v3[:72000 if psmi != 2 else 90000], queues

    def am_deinterleave_pids(self, sym64, pids1_disabled=0) -> np.ndarray:
        s = np.ascontiguousarray(sym64, dtype=np.uint8)
        out = np.zeros(240, dtype=np.int8)
        self.lib.orc_am_deinterleave_pids(s.ctypes.data, pids1_disabled, out.ctypes.data)
        return out

    def bit_errors(self, coded, decoded, k, gens, puncture) -> int:
        c = np.ascontiguousarray(coded, dtype=np.int8)
        d = np.ascontiguousarray(decoded, dtype=np.uint8)
        g = (ctypes.c_uint * 3)(*gens)
        p = np.ascontiguousarray(puncture, dtype=np.uint8)
        return self.lib.orc_bit_errors(c.ctypes.data, d.ctypes.data, k, d.size, g, p.ctypes.data, p.size)

    # ---- whole path -------------------------------------------------------------------
    def run(self, iq: np.ndarray, taps: int = 0, chunk: int = 32768, fft_blocks: int = 4, p1_hook=None, mode: int = 0):
        iq = np.ascontiguousarray(iq)
        L = self.lib
        fn = {0: (L.orc_open, L.orc_close, L.orc_set_taps, L.orc_set_p1_hook, L.orc_push_cu8, L.orc_push_cs16, L.orc_buf),
              1: (L.orc_am_open, L.orc_am_close, L.orc_am_set_taps, L.orc_am_set_p1_hook, L.orc_am_push_cu8, L.orc_am_push_cs16, L.orc_am_buf)}[mode]
        f_open, f_close, f_taps, f_hook, f_cu8, f_cs16, f_buf = fn
        s = f_open()
        keep = None
        try:
            f_taps(s, taps, fft_blocks)
            if p1_hook is not None:
                keep = P1_HOOK(lambda user, bits, n: int(p1_hook(np.ctypeslib.as_array(bits, shape=(n,)))))
                f_hook(s, keep, None)
            step = chunk if iq.dtype == np.uint8 else chunk
            for off in range(0, iq.size, step):
                part = iq[off:off + step]
                if iq.dtype == np.uint8:
                    f_cu8(s, part.ctypes.data, part.size - part.size % 4)
                else:
                    f_cs16(s, part.ctypes.data, part.size - part.size % 2)
            bufs = []
            for which in range(3):
                p = ctypes.c_void_p()
                n = f_buf(s, which, ctypes.byref(p))
                bufs.append(ctypes.string_at(p, n) if n else b"")
        finally:
            f_close(s)
        return (_ref.parse_log(bufs[0]), np.frombuffer(bufs[1], dtype=np.int16).reshape(-1, 2),
                np.frombuffer(bufs[2], dtype=np.complex64))
