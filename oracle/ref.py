"""TEST INFRASTRUCTURE: ctypes driver for oracle/_ref/libnrsc5_ref*.so -- the UNMODIFIED
reference compiled by oracle/Makefile, instrumented by ref_shim/ref_harness.c.
Only tests/, golden-vector generation and bench.py's cpu_baseline leg may import this."""
from __future__ import annotations

import ctypes
import os
import struct
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

TAP_Q15, TAP_FFT, TAP_SOFT, TAP_VIT, TAP_HDC, TAP_L2 = 1, 2, 4, 8, 16, 32
REC_BLOCK, REC_STATE, REC_SOFT, REC_PIDS, REC_FRAME, REC_SYNC, REC_LOST_SYNC, REC_MER, REC_BER, REC_HDC, REC_VIT, REC_AMSYM, REC_PXSOFT, REC_STATION, REC_L2PKT, REC_L2ALIGN, REC_L2AAS, REC_L2SVC = range(1, 19)
MODE_FM, MODE_AM = 0, 1

BLOCK_FIELDS = ("state_before", "state_after", "samperr", "cfo", "keep", "bc", "psmi", "cfo_wait",
                "next_samperr", "prev_angle", "phase_re", "phase_im", "next_angle")


def lib_path(sse: bool = False) -> str:
    return os.path.join(_HERE, "_ref", "libnrsc5_ref_sse.so" if sse else "libnrsc5_ref.so")


def available(sse: bool = False) -> bool:
    return os.path.exists(lib_path(sse))


class RefLib:
    def __init__(self, sse: bool = False, path: str | None = None):
        # path: another build of the unmodified reference (oracle/_ref/libnrsc5_ref_sse_dp.so: the same translation units on a different FFT)
        self.lib = ctypes.CDLL(path or lib_path(sse))
        L = self.lib
        L.refh_open.argtypes = [ctypes.c_int, ctypes.c_uint, ctypes.c_uint]
        L.refh_run_cu8.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
        L.refh_run_cs16.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
        L.refh_buf.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        L.refh_buf.restype = ctypes.c_size_t
        L.refh_sizeof_session.restype = ctypes.c_size_t
        L.refh_frame_push.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        L.refh_frame_push_indexed.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        for name in ("nrsc5_conv_decode_p1", "nrsc5_conv_decode_pids"):
            getattr(L, name).argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.nrsc5_conv_decode_p3_p4.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.nrsc5_conv_decode_e1.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.nrsc5_conv_decode_e2_e3.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]

    def _buf(self, which: int) -> bytes:
        p = ctypes.c_void_p()
        n = self.lib.refh_buf(which, ctypes.byref(p))
        return ctypes.string_at(p, n) if n else b""

    def run(self, iq: np.ndarray, mode: int = MODE_FM, taps: int = 0, chunk: int = 32768,
            fft_blocks: int = 4):
        """Feed a whole capture the way src/main.c:1097-1120 does; returns parsed taps."""
        iq = np.ascontiguousarray(iq)
        if self.lib.refh_open(mode, taps, fft_blocks) != 0:
            raise RuntimeError("refh_open failed")
        try:
            if iq.dtype == np.uint8:
                self.lib.refh_run_cu8(iq.ctypes.data, iq.size, chunk)
            elif iq.dtype == np.int16:
                self.lib.refh_run_cs16(iq.ctypes.data, iq.size, chunk)
            else:
                raise TypeError(iq.dtype)
            log, q15, fft = self._buf(0), self._buf(1), self._buf(2)
        finally:
            self.lib.refh_close()
        return parse_log(log), np.frombuffer(q15, dtype=np.int16).reshape(-1, 2), np.frombuffer(fft, dtype=np.complex64)

    def run_with_mode_switch(self, iq_a: np.ndarray, iq_b: np.ndarray, mode: int = MODE_FM, taps: int = 0, chunk: int = 32768, mode_b: int | None = None):
        """One session: capture A in `mode`, then nrsc5_set_mode(mode_b, default: the same mode) on the live session (input_reset,
        input.c:126-138), then capture B.  Returns (log of A, log of B, q15 of B) -- what the reference does after a reset of a
        USED session, including firdecim_q15_reset leaving stale samples in its windows (firdecim_q15.c:53-56)."""
        L = self.lib
        L.refh_set_mode.restype = ctypes.c_size_t
        L.refh_q15_len.restype = ctypes.c_size_t
        if L.refh_open(mode, taps, 0) != 0:
            raise RuntimeError("refh_open failed")
        try:
            a = np.ascontiguousarray(iq_a); b = np.ascontiguousarray(iq_b)
            (L.refh_run_cu8 if a.dtype == np.uint8 else L.refh_run_cs16)(a.ctypes.data, a.size, chunk)
            q_split = L.refh_q15_len()
            split = L.refh_set_mode(mode if mode_b is None else mode_b)
            (L.refh_run_cu8 if b.dtype == np.uint8 else L.refh_run_cs16)(b.ctypes.data, b.size, chunk)
            log, q15 = self._buf(0), self._buf(1)
        finally:
            L.refh_close()
        return parse_log(log[:split]), parse_log(log[split:]), np.frombuffer(q15[q_split:], dtype=np.int16).reshape(-1, 2)

    def run_epochs(self, epochs, taps: int = 0, chunk: int = 32768):
        """One session, several captures: epochs = [(mode, iq), ...]; nrsc5_set_mode(mode) (input_reset, input.c:126-162) in front of every capture but the
        first, whose mode the session is opened in.  Returns [(log, q15), ...] per epoch -- the reference's used-session behaviour over any history."""
        L = self.lib
        L.refh_set_mode.restype = ctypes.c_size_t
        L.refh_q15_len.restype = ctypes.c_size_t
        if L.refh_open(epochs[0][0], taps, 0) != 0:
            raise RuntimeError("refh_open failed")
        cuts = []
        try:
            for k, (mode, iq) in enumerate(epochs):
                iq = np.ascontiguousarray(iq)
                cuts.append((L.refh_set_mode(mode) if k else 0, L.refh_q15_len()))
                (L.refh_run_cu8 if iq.dtype == np.uint8 else L.refh_run_cs16)(iq.ctypes.data, iq.size, chunk)
            log, q15 = self._buf(0), self._buf(1)
        finally:
            L.refh_close()
        out = []
        for k, (l0, q0) in enumerate(cuts):
            l1, q1 = cuts[k + 1] if k + 1 < len(cuts) else (len(log), len(q15))
            out.append((parse_log(log[l0:l1]), np.frombuffer(q15[q0:q1], dtype=np.int16).reshape(-1, 2)))
        return out

    def l2_frames(self, frames, mode: int = MODE_FM, lc: int = 0):
        """Hand logical frames (bit arrays as frame_push takes them) straight to the reference's L2 in one session;
        returns, per frame, the ordered taps: output_align / output_push calls, state changes, HDC events."""
        if self.lib.refh_open(mode, TAP_L2 | TAP_HDC, 0) != 0:
            raise RuntimeError("refh_open failed")
        out, seen = [], 0
        try:
            for bits in frames:
                b = np.ascontiguousarray(bits, dtype=np.uint8)
                self.lib.refh_frame_push(b.ctypes.data, b.size, lc)
                log = self._buf(0)
                out.append(parse_log(log[seen:]))
                seen = len(log)
        finally:
            self.lib.refh_close()
        return out

    def l2_frames_indexed(self, items, mode: int = MODE_FM, lc: int = 0):
        """Same taps as l2_frames, but every frame enters through frame_push_indexed (oracle/ref_shim/frame_indexed.c):
        items = (ctypes index struct with the nrsc5hip_l2_frame layout, PDU bytes) per frame, one session."""
        if self.lib.refh_open(mode, TAP_L2 | TAP_HDC, 0) != 0:
            raise RuntimeError("refh_open failed")
        out, seen = [], 0
        try:
            for ix, by in items:
                b = np.ascontiguousarray(by, dtype=np.uint8)
                self.lib.refh_frame_push_indexed(ctypes.addressof(ix), b.ctypes.data, lc)
                log = self._buf(0)
                out.append(parse_log(log[seen:]))
                seen = len(log)
        finally:
            self.lib.refh_close()
        return out

    def conv_decode(self, soft: np.ndarray, kind: str = "p1") -> np.ndarray:
        soft = np.ascontiguousarray(soft, dtype=np.int8)
        n = soft.size // 3
        out = np.zeros(n, dtype=np.uint8)
        if kind == "p1":
            assert n == 146176
            self.lib.nrsc5_conv_decode_p1(soft.ctypes.data, out.ctypes.data)
        elif kind == "pids":
            assert n == 80
            self.lib.nrsc5_conv_decode_pids(soft.ctypes.data, out.ctypes.data)
        elif kind == "p3":
            self.lib.nrsc5_conv_decode_p3_p4(soft.ctypes.data, out.ctypes.data, n)
        elif kind == "e1":
            self.lib.nrsc5_conv_decode_e1(soft.ctypes.data, out.ctypes.data, n)
        elif kind == "e2":
            self.lib.nrsc5_conv_decode_e2_e3(soft.ctypes.data, out.ctypes.data, n)
        else:
            raise ValueError(kind)
        return out


def parse_log(log: bytes):
    """Ordered list of (kind, payload-dict) records."""
    out, off = [], 0
    while off < len(log):
        kind, n = struct.unpack_from("<II", log, off)
        off += 8
        pl = log[off:off + n]
        off += (n + 3) & ~3
        if kind == REC_BLOCK:
            v = struct.unpack("<9i4f", pl)
            out.append(("block", dict(zip(BLOCK_FIELDS, v))))
        elif kind == REC_STATE:
            a, b = struct.unpack("<2i", pl)
            out.append(("state", {"old": a, "new": b}))
        elif kind == REC_SOFT:
            out.append(("soft", {"bc": struct.unpack_from("<I", pl)[0], "bits": np.frombuffer(pl, dtype=np.int8, offset=4)}))
        elif kind == REC_PIDS:
            out.append(("pids", {"bits": np.frombuffer(pl, dtype=np.uint8)}))
        elif kind == REC_FRAME:
            lc, ln = struct.unpack_from("<II", pl)
            out.append(("frame", {"lc": lc, "bits": np.frombuffer(pl, dtype=np.uint8, offset=8, count=ln)}))
        elif kind == REC_SYNC:
            v = struct.unpack("<f5i", pl)
            out.append(("sync", dict(zip(("freq_offset", "psmi", "pli", "hppi", "aabi", "rdbi"), v))))
        elif kind == REC_LOST_SYNC:
            out.append(("lost_sync", {}))
        elif kind == REC_MER:
            lo, up = struct.unpack("<2f", pl)
            out.append(("mer", {"lower": lo, "upper": up}))
        elif kind == REC_BER:
            out.append(("ber", {"cber": struct.unpack("<f", pl)[0]}))
        elif kind == REC_HDC:
            prog, cnt, flags = struct.unpack_from("<3I", pl)
            out.append(("hdc", {"program": prog, "count": cnt, "flags": flags, "data": pl[12:]}))
        elif kind == REC_AMSYM:
            sym = np.frombuffer(pl, dtype=np.uint8, offset=4).reshape(4, 800)
            out.append(("amsym", {"bc": struct.unpack_from("<I", pl)[0], "pl": sym[0], "pu": sym[1], "s": sym[2], "t": sym[3]}))
        elif kind == REC_PXSOFT:
            ch, bc, ln = struct.unpack_from("<3I", pl)
            out.append(("pxsoft", {"ch": ch, "bc": bc, "bits": np.frombuffer(pl, dtype=np.int8, offset=12, count=ln)}))
        elif kind == REC_L2PKT:
            prog, sid, seq, size, flags, shape = struct.unpack_from("<6I", pl)
            out.append(("l2pkt", {"program": prog, "stream_id": sid, "seq": seq, "size": size, "flags": flags, "shape": shape,
                                  "data": pl[24:24 + size + 1]}))
        elif kind == REC_L2ALIGN:
            prog, sid, align = struct.unpack("<3I", pl)
            out.append(("l2align", {"program": prog, "stream_id": sid, "offset": align}))
        elif kind == REC_L2AAS:
            out.append(("l2aas", {"data": bytes(pl)}))
        elif kind == REC_L2SVC:
            out.append(("l2svc", dict(zip(("program", "access", "type", "codec_mode", "blend_control", "digital_audio_gain", "common_delay", "latency"), struct.unpack("<8i", pl)))))
        elif kind == REC_STATION:
            fcc, cc = struct.unpack("<i4s", pl)
            out.append(("station", {"fcc": fcc, "country": cc.rstrip(b"\0").decode()}))
        elif kind == REC_VIT:
            ln = struct.unpack_from("<I", pl)[0]
            out.append(("vit", {"in": np.frombuffer(pl, dtype=np.int8, offset=4, count=3 * ln),
                                "out": np.frombuffer(pl, dtype=np.uint8, offset=4 + 3 * ln, count=ln)}))
    return out
