"""ctypes binding of libnrsc5hip.so (include/nrsc5hip.h) -- the same stub a Python caller of the
reference would use next to support/nrsc5.py:676-690.  No fallbacks: if the HIP library is missing
or a call fails, this raises."""
from __future__ import annotations

import ctypes
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libnrsc5hip.so")
# A/B measurements only (tools/gpu_r5_ab.sh): another build of the library -- e.g. one made from an earlier commit -- in place of the tree's own.  The
# freshness check is then skipped WITH a notice on stderr; nothing measured this way may be recorded as this tree's result.
_AB_LIB = os.environ.get("NRSC5HIP_AB_LIB")
if _AB_LIB:
    import sys as _sys
    print(f"nrsc5_amd.engine: NRSC5HIP_AB_LIB={_AB_LIB}: running a library that is NOT built from this tree (A/B measurement)", file=_sys.stderr)
    DEFAULT_LIB = _AB_LIB

SYNC_NONE, SYNC_COARSE, SYNC_FINE = 0, 1, 2
REC_PROCESSED, REC_TO_COARSE, REC_TO_FINE, REC_MER, REC_PIDS, REC_P1 = 1, 2, 4, 8, 16, 32
REC_P3, REC_P4 = 128, 256
REC_LOST_SYNC = 64
REC_PIDS_CRC = 512
REC_DISCARDED = 1024      # never delivered by drain / batch_fetch* (replay, k_replay.hip)
PX_WORDS = 144
MODE_FM, MODE_AM = 0, 1
AM_P1_BITS, AM_P1_WORDS, AM_P3_WORD0 = 3750, 118, 944
P1_BITS, P1_WORDS, PIDS_BITS = 146176, 4568, 80

RECORD_DTYPE = np.dtype([
    ("flags", "<u4"), ("state_before", "<i4"), ("state_after", "<i4"), ("samperr", "<i4"), ("cfo", "<i4"),
    ("keep", "<i4"), ("bc", "<i4"), ("psmi", "<i4"), ("cfo_wait", "<i4"), ("next_samperr", "<i4"),
    ("prev_angle", "<f4"), ("phase_re", "<f4"), ("phase_im", "<f4"), ("next_angle", "<f4"),
    ("freq_offset", "<f4"), ("mer_lb", "<f4"), ("mer_ub", "<f4"), ("ber", "<f4"),
    ("p1_slot", "<i4"), ("bc_decoded", "<i4"), ("pids", "<u4", (3,)), ("sis", "<u4")])
assert RECORD_DTYPE.itemsize == 96


class _Config(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int), ("max_streams", ctypes.c_int), ("q15_capacity", ctypes.c_longlong),
                ("record_capacity", ctypes.c_int), ("p1_slots", ctypes.c_int), ("p1_async", ctypes.c_int),
                ("l2_feedback", ctypes.c_int), ("am_enable", ctypes.c_int), ("batch_zero_copy", ctypes.c_int), ("l2_index", ctypes.c_int)]


class L2Pdu(ctypes.Structure):
    """nrsc5hip_l2_pdu (include/nrsc5hip.h)"""
    _fields_ = ([("start", ctypes.c_uint32), ("psd_off", ctypes.c_uint32), ("psd_len", ctypes.c_int32), ("audio_off", ctypes.c_uint32),
                 ("crc_bad_lo", ctypes.c_uint32), ("crc_bad_hi", ctypes.c_uint32), ("pdu_marker", ctypes.c_uint32),
                 ("hef_pdu_len", ctypes.c_uint16), ("loc", ctypes.c_uint16 * 64)] +
                [(n, ctypes.c_uint8) for n in ("codec_mode", "stream_id", "pdu_seq", "blend_control", "per_stream_delay", "common_delay",
                                               "latency", "pfirst", "plast", "seq", "nop", "hef", "la_location", "rs_corrections",
                                               "class_ind", "prog_num", "access", "prog_type", "applied_services", "elastic_seq",
                                               "align_offset", "skipped")])


class L2Frame(ctypes.Structure):
    """nrsc5hip_l2_frame"""
    _fields_ = [("pci", ctypes.c_uint32), ("nbytes", ctypes.c_uint32), ("n_pdu", ctypes.c_uint32), ("status", ctypes.c_uint32),
                ("end_offset", ctypes.c_uint32), ("lost_sync", ctypes.c_uint32), ("pdu", L2Pdu * 16)]


class L2Job(ctypes.Structure):
    """nrsc5hip_l2_job"""
    _fields_ = [("stream", ctypes.c_int32), ("slot", ctypes.c_int32), ("kind", ctypes.c_int32), ("which", ctypes.c_int32),
                ("nbits", ctypes.c_int32)]


L2_FM_P1, L2_FM_PX, L2_AM = 0, 1, 2
TUNE_DECODE_STREAMS, TUNE_AM_DECODE_STREAMS, TUNE_VERDICT_LAG, TUNE_SYNC_PHASES, TUNE_FWD_SEGMENTS, TUNE_FWD_WARM, TUNE_AM_SEGMENTS, TUNE_DECODE_CUS, TUNE_DECODE_PRIORITY, TUNE_AM_WARM, TUNE_MIXFFT_SYMS, TUNE_DEFER_WAIT, TUNE_TRACEBACK_WALK, TUNE_SYNC_LANES, TUNE_DIRECT_DECIMATE, TUNE_EARLY_FLUSH_KB, TUNE_SEAM_PREPARE, TUNE_NCO_EXACT, TUNE_FLOW_MIN, TUNE_LOOP_EXACT, TUNE_HOST_CAPTURE, TUNE_FOLD_REPORT = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21
L2_STATUS = ("end", "no_audio", "fixed_data", "header_rs", "bad_locators", "too_many_pdus", "hef_overrun", "bad_stream", "bad_length", "audio_end")


def l2_frame_to_dict(fr: L2Frame) -> dict:
    out = {k: int(getattr(fr, k)) for k in ("pci", "nbytes", "n_pdu", "status", "end_offset", "lost_sync")}
    out["pdus"] = []
    for i in range(min(fr.n_pdu, 16)):
        p = fr.pdu[i]
        d = {name: int(getattr(p, name)) for name, _ in p._fields_ if name != "loc"}
        d["loc"] = [int(x) for x in p.loc[:p.nop]]
        out["pdus"].append(d)
    return out


HDC_CB = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.POINTER(ctypes.c_uint8), ctypes.c_uint, ctypes.c_uint)


class Nrsc5HipError(RuntimeError):
    pass


def load_library(path: str | None = None) -> ctypes.CDLL:
    path = path or DEFAULT_LIB
    if not os.path.exists(path):
        raise Nrsc5HipError(f"{path} not found: build it with `python -m nrsc5_amd.build` (hipcc, gfx950). "
                            "There is no CPU fallback.")
    lib = ctypes.CDLL(path)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    lib.nrsc5hip_engine_create.argtypes = [ctypes.POINTER(_Config), ctypes.POINTER(vp)]
    lib.nrsc5hip_engine_destroy.argtypes = [vp]
    lib.nrsc5hip_engine_destroy.restype = None
    lib.nrsc5hip_last_error.restype = ctypes.c_char_p
    lib.nrsc5hip_source_sha.restype = ctypes.c_char_p
    lib.nrsc5hip_engine_hip_stream.argtypes = [vp]
    lib.nrsc5hip_engine_hip_stream.restype = vp
    lib.nrsc5hip_push_cu8.argtypes = [vp, ci, vp, ctypes.c_uint32]
    lib.nrsc5hip_push_cs16.argtypes = [vp, ci, vp, ctypes.c_uint32]
    lib.nrsc5hip_stream_reset.argtypes = [vp, ci]
    lib.nrsc5hip_stream_fresh.argtypes = [vp, ci]
    lib.nrsc5hip_force_resync.argtypes = [vp, ci]
    lib.nrsc5hip_bytes_to_next_block.argtypes = [vp, ci, ci]
    lib.nrsc5hip_bytes_to_next_block.restype = ctypes.c_longlong
    lib.nrsc5hip_px_frame_bits.argtypes = [vp, ci, ci, ci, ci, vp]
    lib.nrsc5hip_batch_fetch_px.argtypes = [vp, ci, vp, vp]
    lib.nrsc5hip_stream_set_mode.argtypes = [vp, ci, ci]
    lib.nrsc5hip_am_frame_bits.argtypes = [vp, ci, ci, ci, ci, vp]
    lib.nrsc5hip_stage_viterbi_k9.argtypes = [vp, vp, ci, ci, vp, vp]
    lib.nrsc5hip_batch_append_cu8.argtypes = [vp, ci, vp, vp, ctypes.c_longlong, vp]
    lib.nrsc5hip_batch_append_cs16.argtypes = [vp, ci, vp, vp, ctypes.c_longlong, vp]
    lib.nrsc5hip_batch_process.argtypes = [vp, ci, vp, ci, ctypes.POINTER(ci)]
    lib.nrsc5hip_drain.argtypes = [vp, ci, vp, ci, ctypes.POINTER(ci)]
    lib.nrsc5hip_p1_frame_packed.argtypes = [vp, ci, ci, vp]
    lib.nrsc5hip_p1_frame_bits.argtypes = [vp, ci, ci, vp]
    lib.nrsc5hip_batch_fetch.argtypes = [vp, ci, vp, vp, ci, vp, vp]
    lib.nrsc5hip_unpack_bits.argtypes = [vp, ci, vp]
    lib.nrsc5hip_unpack_bits.restype = None
    lib.nrsc5hip_stage_halfband_fm_cu8.argtypes = [vp, vp, ctypes.c_uint32, vp]
    lib.nrsc5hip_stage_fft2048.argtypes = [vp, vp, vp, ci]
    lib.nrsc5hip_stage_viterbi_k7.argtypes = [vp, vp, ci, ci, vp]
    lib.nrsc5hip_debug_fetch.argtypes = [vp, ci, vp, vp]
    lib.nrsc5hip_debug_fetch_costas.argtypes = [vp, ci, vp, vp]
    lib.nrsc5hip_stage_selftest.argtypes = [vp, ctypes.POINTER(ci)]
    lib.nrsc5hip_stage_viterbi_k7_debug.argtypes = [vp, vp, ci, vp, vp]
    lib.nrsc5hip_stage_viterbi_bench.argtypes = [vp, ci, ci, ci, ci, ctypes.POINTER(ctypes.c_float)]
    lib.nrsc5hip_debug_sync_phases.argtypes = [vp, vp]
    lib.nrsc5hip_debug_tune.argtypes = [vp, ci, ci]
    lib.nrsc5hip_debug_fwd_stats.argtypes = [vp, vp]
    lib.nrsc5hip_debug_flow_stats.argtypes = [vp, vp]
    lib.nrsc5hip_debug_host_capture_stats.argtypes = [vp, vp]
    lib.nrsc5hip_debug_k9_stats.argtypes = [vp, vp]
    lib.nrsc5hip_debug_tb_stats.argtypes = [vp, vp]
    lib.nrsc5hip_stage_first_header.argtypes = [vp, vp, ci, ci, ci, vp]
    lib.nrsc5hip_debug_poison_results.argtypes = [vp]
    lib.nrsc5hip_debug_seam_totals.argtypes = [vp, ci]
    lib.nrsc5hip_debug_seam_totals.restype = None
    lib.nrsc5hip_debug_seam_counts.argtypes = [vp, ci]
    lib.nrsc5hip_debug_seam_counts.restype = None
    lib.nrsc5hip_drain_ready.argtypes = [vp, ci, vp, ci, ctypes.POINTER(ci)]
    lib.nrsc5hip_stream_set_manual_step.argtypes = [vp, ci, ci]
    lib.nrsc5hip_stream_step.argtypes = [vp, ci]
    lib.nrsc5hip_stream_step_ahead.argtypes = [vp, ci, ctypes.POINTER(ci)]
    lib.nrsc5hip_batch_fetch_view.argtypes = [vp, ci, ctypes.POINTER(vp), vp, ctypes.POINTER(vp)]
    lib.nrsc5hip_reset_all.argtypes = [vp]
    lib.nrsc5hip_profile.argtypes = [vp, ci, vp, vp]
    lib.nrsc5hip_l2_index.argtypes = [vp, ci, vp, vp, vp, ctypes.c_longlong]
    lib.nrsc5hip_stage_l2_index.argtypes = [vp, vp, ci, ci, vp, vp, ctypes.c_longlong]
    lib.nrsc5hip_l2_frame_get.argtypes = [vp, ci, ci, vp]
    lib.nrsc5hip_batch_fetch_l2.argtypes = [vp, ci, vp, vp]
    lib.nrsc5hip_batch_fetch_l2_px.argtypes = [vp, ci, vp, vp]
    lib.nrsc5hip_batch_fetch_l2_am.argtypes = [vp, ci, vp, vp]
    lib.nrsc5hip_hdc_create.argtypes = [ci, ctypes.POINTER(vp)]
    lib.nrsc5hip_hdc_destroy.argtypes = [vp]
    lib.nrsc5hip_hdc_destroy.restype = None
    lib.nrsc5hip_hdc_reset.argtypes = [vp, ci]
    lib.nrsc5hip_hdc_push_frame.argtypes = [vp, ci, ci, vp, vp]
    lib.nrsc5hip_hdc_fixed_audio_end.argtypes = [vp, ci, ci, vp, ctypes.c_uint]
    lib.nrsc5hip_hdc_fixed_audio_end.restype = ctypes.c_uint
    lib.nrsc5hip_l2_apply_audio_end.argtypes = [vp, ctypes.c_uint]
    lib.nrsc5hip_hdc_frame_reset.argtypes = [vp, ci]
    lib.nrsc5hip_hdc_advance.argtypes = [vp, ci, ci, HDC_CB, vp]
    lib.nrsc5hip_hdc_adts.argtypes = [vp, ctypes.c_uint, vp]
    lib.nrsc5hip_hdc_adts.restype = ctypes.c_size_t
    lib.nrsc5hip_hdc_host_bytes.argtypes = [vp]
    lib.nrsc5hip_hdc_host_bytes.restype = ctypes.c_size_t
    return lib


EXPORTED_SYMBOLS = [
    "nrsc5hip_engine_create", "nrsc5hip_engine_destroy", "nrsc5hip_last_error", "nrsc5hip_source_sha", "nrsc5hip_engine_hip_stream",
    "nrsc5hip_push_cu8", "nrsc5hip_push_cs16", "nrsc5hip_stream_reset", "nrsc5hip_stream_fresh", "nrsc5hip_force_resync", "nrsc5hip_bytes_to_next_block",
    "nrsc5hip_batch_append_cu8", "nrsc5hip_batch_append_cs16", "nrsc5hip_batch_process", "nrsc5hip_drain",
    "nrsc5hip_p1_frame_packed", "nrsc5hip_p1_frame_bits", "nrsc5hip_batch_fetch", "nrsc5hip_unpack_bits",
    "nrsc5hip_stage_halfband_fm_cu8", "nrsc5hip_stage_fft2048", "nrsc5hip_stage_viterbi_k7", "nrsc5hip_debug_fetch", "nrsc5hip_debug_fetch_costas",
    "nrsc5hip_debug_fetch_q15", "nrsc5hip_debug_alloc_copy", "nrsc5hip_debug_free", "nrsc5hip_reset_all", "nrsc5hip_profile", "nrsc5hip_stage_selftest", "nrsc5hip_stage_viterbi_k7_debug", "nrsc5hip_stage_viterbi_bench", "nrsc5hip_debug_sync_phases", "nrsc5hip_debug_tune", "nrsc5hip_debug_fwd_stats", "nrsc5hip_debug_flow_stats", "nrsc5hip_debug_host_capture_stats", "nrsc5hip_abi_version", "nrsc5hip_debug_tb_stats", "nrsc5hip_debug_k9_stats", "nrsc5hip_stage_first_header", "nrsc5hip_debug_seam_totals", "nrsc5hip_debug_seam_counts", "nrsc5hip_drain_ready", "nrsc5hip_stream_set_manual_step", "nrsc5hip_stream_step", "nrsc5hip_stream_step_ahead", "nrsc5hip_debug_poison_results", "nrsc5hip_device_count", "nrsc5hip_device_upload", "nrsc5hip_device_free", "nrsc5hip_batch_fetch_view", "nrsc5hip_batch_fetch_l2_px", "nrsc5hip_batch_fetch_l2_am",
    "nrsc5hip_stream_set_mode", "nrsc5hip_am_frame_bits", "nrsc5hip_stage_viterbi_k9", "nrsc5hip_px_frame_bits",
    "nrsc5hip_batch_fetch_px", "nrsc5hip_debug_fetch_px", "nrsc5hip_stage_viterbi_k9_bench",
    "nrsc5hip_l2_index", "nrsc5hip_stage_l2_index", "nrsc5hip_l2_frame_get", "nrsc5hip_batch_fetch_l2",
    "nrsc5hip_hdc_create", "nrsc5hip_hdc_destroy", "nrsc5hip_hdc_reset", "nrsc5hip_hdc_push_frame", "nrsc5hip_hdc_advance",
    "nrsc5hip_hdc_adts", "nrsc5hip_hdc_host_bytes", "nrsc5hip_hdc_fixed_audio_end", "nrsc5hip_l2_apply_audio_end", "nrsc5hip_hdc_frame_reset"]


def library_sha(path: str | None = None) -> str:
    """Source fingerprint of the library FILE, read from its bytes -- not through dlopen: a library this process has already
    mapped keeps reporting the old build after the file was rebuilt (dlopen hands back the mapped handle)."""
    path = path or DEFAULT_LIB
    if not os.path.exists(path):
        raise Nrsc5HipError(f"{path} not found: build it with `python -m nrsc5_amd.build` (hipcc, gfx950). There is no CPU fallback.")
    data = open(path, "rb").read()
    marker = b"NRSC5HIP_SOURCE_SHA="
    k = data.find(marker)
    if k < 0:
        return "unmarked"
    return data[k + len(marker):k + len(marker) + 16].split(b"\0")[0].decode(errors="replace")


def check_fresh(path: str | None = None):
    """Raise unless the library was built from the device sources of THIS tree (a stale .so silently measures / tests old code).
    Looks at the file only, so that a caller can rebuild and check again in the same process."""
    from . import build
    if _AB_LIB and (path is None or path == _AB_LIB):
        return
    got = library_sha(path)
    want = build.source_sha()
    if got != want:
        raise Nrsc5HipError(f"{path or DEFAULT_LIB} was built from other sources (library {got}, tree {want}): run `python -m nrsc5_amd.build`")


def unpack_bits(words: np.ndarray, nbits: int) -> np.ndarray:
    w = np.ascontiguousarray(words, dtype="<u4")
    return np.unpackbits(w.view(np.uint8), bitorder="little")[:nbits]


class Engine:
    """One engine per GPU/process; `max_streams` independent IQ streams resident on the device."""

    def __init__(self, max_streams: int = 1, q15_capacity: int = 1 << 20, record_capacity: int = 256,
                 p1_slots: int = 4, p1_async: bool = False, device: int = 0, lib_path: str | None = None,
                 am_enable: bool = False, l2_feedback: bool = False, l2_index: bool = False, batch_zero_copy: bool = False):
        self.lib = load_library(lib_path)
        self.cfg = _Config(device, max_streams, q15_capacity, record_capacity, p1_slots, int(p1_async), int(l2_feedback), int(am_enable), int(batch_zero_copy), int(l2_index))
        self._h = ctypes.c_void_p()
        self._check(self.lib.nrsc5hip_engine_create(ctypes.byref(self.cfg), ctypes.byref(self._h)))
        self.max_streams, self.record_capacity, self.p1_slots = max_streams, record_capacity, p1_slots

    def _check(self, rc: int):
        if rc != 0:
            raise Nrsc5HipError(f"libnrsc5hip error {rc}: {self.lib.nrsc5hip_last_error().decode()}")

    def poison_results(self):
        """nrsc5hip_debug_poison_results: frame / record rings and their host mirrors get a pattern no decode produces"""
        self._check(self.lib.nrsc5hip_debug_poison_results(self._h))

    def tune(self, knob: int, value: int):
        """nrsc5hip_debug_tune: TUNE_DECODE_STREAMS / TUNE_AM_DECODE_STREAMS / TUNE_VERDICT_LAG (test hook) / TUNE_SYNC_PHASES"""
        self._check(self.lib.nrsc5hip_debug_tune(self._h, knob, value))

    def close(self):
        if self._h:
            self.lib.nrsc5hip_engine_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def hip_stream(self) -> int:
        return self.lib.nrsc5hip_engine_hip_stream(self._h) or 0

    # ---- streaming seam ---------------------------------------------------------------------
    def push_cu8(self, stream: int, iq: np.ndarray):
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        self._check(self.lib.nrsc5hip_push_cu8(self._h, stream, iq.ctypes.data, iq.size))

    def push_cs16(self, stream: int, iq: np.ndarray):
        iq = np.ascontiguousarray(iq, dtype=np.int16)
        self._check(self.lib.nrsc5hip_push_cs16(self._h, stream, iq.ctypes.data, iq.size))

    def reset(self, stream: int):
        """input_reset of a session that may have been used: the FIR windows are rewound, not cleared (include/nrsc5hip.h)"""
        self._check(self.lib.nrsc5hip_stream_reset(self._h, stream))

    def fresh(self, stream: int):
        """a new session on this slot (nrsc5_close + nrsc5_open_pipe)"""
        self._check(self.lib.nrsc5hip_stream_fresh(self._h, stream))

    def set_mode(self, stream: int, mode: int):
        """nrsc5_set_mode for one stream (MODE_FM / MODE_AM); resets it."""
        self._check(self.lib.nrsc5hip_stream_set_mode(self._h, stream, mode))

    def reset_all(self):
        self._check(self.lib.nrsc5hip_reset_all(self._h))

    PROF_CLASSES = ("decimate", "acquire", "prepare", "mixfft", "sync", "p1_deint", "p1_viterbi", "pids", "am", "am_decode", "p1_traceback", "flow")

    def profile(self, enable: int = -1):
        """Per-kernel-class {name: (total_ms, launches)} from HIP events; enable 1/0 starts/stops, a class name starts timing
        that class only."""
        if isinstance(enable, str):
            enable = 0x100 | self.PROF_CLASSES.index(enable)
        ms = np.zeros(len(self.PROF_CLASSES), dtype=np.float64)
        n = np.zeros(len(self.PROF_CLASSES), dtype=np.int64)
        self._check(self.lib.nrsc5hip_profile(self._h, enable, ms.ctypes.data, n.ctypes.data))
        return {k: (float(a), int(b)) for k, a, b in zip(self.PROF_CLASSES, ms, n)}

    def bytes_to_next_block(self, stream: int, cu8: bool = True) -> int:
        return int(self.lib.nrsc5hip_bytes_to_next_block(self._h, stream, int(cu8)))

    def force_resync(self, stream: int):
        self._check(self.lib.nrsc5hip_force_resync(self._h, stream))

    # ---- batch path (device pointers as ints) ---------------------------------------------------
    def batch_append_cu8(self, dev_ptr: int, stride_bytes: int, nbytes, stream_ids=None):
        nb = np.ascontiguousarray(nbytes, dtype=np.uint32)
        ids = None if stream_ids is None else np.ascontiguousarray(stream_ids, dtype=np.int32)
        self._check(self.lib.nrsc5hip_batch_append_cu8(self._h, nb.size, None if ids is None else ids.ctypes.data,
                                                       dev_ptr, stride_bytes, nb.ctypes.data))

    def batch_append_cs16(self, dev_ptr: int, stride_elems: int, nelems, stream_ids=None):
        ne = np.ascontiguousarray(nelems, dtype=np.uint32)
        ids = None if stream_ids is None else np.ascontiguousarray(stream_ids, dtype=np.int32)
        self._check(self.lib.nrsc5hip_batch_append_cs16(self._h, ne.size, None if ids is None else ids.ctypes.data,
                                                        dev_ptr, stride_elems, ne.ctypes.data))

    def batch_process(self, nstreams: int, stream_ids=None, max_steps: int = 0) -> int:
        ids = None if stream_ids is None else np.ascontiguousarray(stream_ids, dtype=np.int32)
        done = ctypes.c_int()
        self._check(self.lib.nrsc5hip_batch_process(self._h, nstreams, None if ids is None else ids.ctypes.data,
                                                    max_steps, ctypes.byref(done)))
        return done.value

    # ---- results ---------------------------------------------------------------------------------
    def drain(self, stream: int, max_records: int | None = None) -> np.ndarray:
        mx = max_records or self.record_capacity
        out = np.zeros(mx, dtype=RECORD_DTYPE)
        n = ctypes.c_int()
        self._check(self.lib.nrsc5hip_drain(self._h, stream, out.ctypes.data, mx, ctypes.byref(n)))
        return out[:n.value]

    def drain_ready(self, stream: int, max_records: int | None = None) -> np.ndarray:
        """nrsc5hip_drain_ready: the records reported so far; never waits for a block step that is still running"""
        mx = max_records or self.record_capacity
        out = np.zeros(mx, dtype=RECORD_DTYPE)
        n = ctypes.c_int()
        self._check(self.lib.nrsc5hip_drain_ready(self._h, stream, out.ctypes.data, mx, ctypes.byref(n)))
        return out[:n.value]

    def set_manual_step(self, stream: int, on: bool = True):
        self._check(self.lib.nrsc5hip_stream_set_manual_step(self._h, stream, int(on)))

    def stream_step(self, stream: int):
        self._check(self.lib.nrsc5hip_stream_step(self._h, stream))

    def stream_step_ahead(self, stream: int) -> bool:
        """nrsc5hip_stream_step_ahead: True if the next block's step was queued behind the one in flight"""
        done = ctypes.c_int()
        self._check(self.lib.nrsc5hip_stream_step_ahead(self._h, stream, ctypes.byref(done)))
        return bool(done.value)

    def seam_counts(self, reset: bool = False) -> dict:
        """deferred steps / mispredicted read positions / steps without P1 decode launches / late P1 decodes (calling thread)"""
        out = np.zeros(6, dtype=np.float64)
        self.lib.nrsc5hip_debug_seam_counts(out.ctypes.data, int(reset))
        return dict(zip(("deferred_steps", "mispredicted_rd", "steps_without_p1_launches", "late_p1_decodes", "steps_ahead", "host_capture_pushes"), (int(x) for x in out)))

    def p1_frame_bits(self, stream: int, slot: int) -> np.ndarray:
        bits = np.zeros(P1_BITS, dtype=np.uint8)
        self._check(self.lib.nrsc5hip_p1_frame_bits(self._h, stream, slot, bits.ctypes.data))
        return bits

    def px_frame_bits(self, stream: int, slot: int, channel: int, nbits: int) -> np.ndarray:
        """FM extended sidebands: channel 0 = P3, 1 = P4; nbits 2304 (MP2) or 4608 (MP3 / MP11)."""
        bits = np.zeros(nbits, dtype=np.uint8)
        self._check(self.lib.nrsc5hip_px_frame_bits(self._h, stream, slot, channel, nbits, bits.ctypes.data))
        return bits

    def batch_fetch_px(self, nstreams: int, stream_ids=None) -> np.ndarray:
        ids = None if stream_ids is None else np.ascontiguousarray(stream_ids, dtype=np.int32)
        out = np.zeros((nstreams, 8 * self.p1_slots, 2, PX_WORDS), dtype=np.uint32)
        self._check(self.lib.nrsc5hip_batch_fetch_px(self._h, nstreams, None if ids is None else ids.ctypes.data, out.ctypes.data))
        return out

    def am_frame_bits(self, stream: int, slot: int, which: int, nbits: int) -> np.ndarray:
        """AM: which = 0..7 -> P1 frame of that block (3750 bits), 8 -> the P3 frame (24000 / 30000 bits)."""
        bits = np.zeros(nbits, dtype=np.uint8)
        self._check(self.lib.nrsc5hip_am_frame_bits(self._h, stream, slot, which, nbits, bits.ctypes.data))
        return bits

    def p1_frame_packed(self, stream: int, slot: int) -> np.ndarray:
        w = np.zeros(P1_WORDS, dtype=np.uint32)
        self._check(self.lib.nrsc5hip_p1_frame_packed(self._h, stream, slot, w.ctypes.data))
        return w

    def batch_fetch(self, nstreams: int, stream_ids=None, with_frames: bool = True):
        ids = None if stream_ids is None else np.ascontiguousarray(stream_ids, dtype=np.int32)
        recs = np.zeros((nstreams, self.record_capacity), dtype=RECORD_DTYPE)
        counts = np.zeros(nstreams, dtype=np.int32)
        frames = np.zeros((nstreams, self.p1_slots, P1_WORDS), dtype=np.uint32) if with_frames else None
        self._check(self.lib.nrsc5hip_batch_fetch(self._h, nstreams, None if ids is None else ids.ctypes.data,
                                                  recs.ctypes.data, self.record_capacity, counts.ctypes.data,
                                                  None if frames is None else frames.ctypes.data))
        return recs, counts, frames

    def batch_fetch_view(self, nstreams: int, with_frames: bool = True):
        """Zero-copy views (numpy arrays over engine-owned pinned memory, valid until the next fetch/reset)."""
        rp, fp = ctypes.c_void_p(), ctypes.c_void_p()
        counts = np.zeros(nstreams, dtype=np.int32)
        self._check(self.lib.nrsc5hip_batch_fetch_view(self._h, nstreams, ctypes.byref(rp), counts.ctypes.data,
                                                       ctypes.byref(fp) if with_frames else None))
        rec_bytes = nstreams * self.record_capacity * RECORD_DTYPE.itemsize
        recs = np.frombuffer((ctypes.c_char * rec_bytes).from_address(rp.value), dtype=RECORD_DTYPE).reshape(nstreams, self.record_capacity)
        frames = None
        if with_frames:
            fr_bytes = nstreams * self.p1_slots * P1_WORDS * 4
            frames = np.frombuffer((ctypes.c_char * fr_bytes).from_address(fp.value), dtype=np.uint32).reshape(nstreams, self.p1_slots, P1_WORDS)
        return recs, counts, frames

    # ---- stage-level entry points (parity tests) ----------------------------------------------------
    def stage_halfband_fm_cu8(self, iq: np.ndarray) -> np.ndarray:
        iq = np.ascontiguousarray(iq, dtype=np.uint8)
        out = np.zeros((iq.size // 4, 2), dtype=np.int16)
        self._check(self.lib.nrsc5hip_stage_halfband_fm_cu8(self._h, iq.ctypes.data, iq.size, out.ctypes.data))
        return out

    def stage_fft2048(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.complex64).reshape(-1, 2048)
        out = np.zeros_like(x)
        self._check(self.lib.nrsc5hip_stage_fft2048(self._h, x.ctypes.data, out.ctypes.data, x.shape[0]))
        return out

    def stage_viterbi_k7(self, soft: np.ndarray, length: int) -> np.ndarray:
        soft = np.ascontiguousarray(soft, dtype=np.int8).reshape(-1, 3 * length)
        bits = np.zeros((soft.shape[0], length), dtype=np.uint8)
        self._check(self.lib.nrsc5hip_stage_viterbi_k7(self._h, soft.ctypes.data, length, soft.shape[0], bits.ctypes.data))
        return bits

    def stage_viterbi_k9(self, soft: np.ndarray, length: int, gens) -> np.ndarray:
        soft = np.ascontiguousarray(soft, dtype=np.int8).reshape(-1, 3 * length)
        bits = np.zeros((soft.shape[0], length), dtype=np.uint8)
        g = (ctypes.c_uint * 3)(*gens)
        self._check(self.lib.nrsc5hip_stage_viterbi_k9(self._h, soft.ctypes.data, length, soft.shape[0], g, bits.ctypes.data))
        return bits

    def l2_index(self, jobs, want_bytes: bool = True):
        """L2 audio transport index of decoded frames still on the device.  jobs: (stream, slot, kind, which, nbits)
        tuples; returns [(dict, PDU bytes or None)] in job order."""
        n = len(jobs)
        arr = (L2Job * n)(*[L2Job(*j) for j in jobs])
        out = (L2Frame * n)()
        stride = 18272
        by = np.zeros((n, stride), dtype=np.uint8) if want_bytes else None
        self._check(self.lib.nrsc5hip_l2_index(self._h, n, arr, out, by.ctypes.data if want_bytes else None, stride))
        return [(l2_frame_to_dict(out[k]), by[k, :out[k].nbytes].copy() if want_bytes else None) for k in range(n)]

    def l2_index_raw(self, jobs):
        """As l2_index, but returns the C structs themselves: (L2Frame ctypes array, PDU bytes [n, 18272]) -- what
        nrsc5hip_hdc_push_frame takes."""
        n = len(jobs)
        arr = (L2Job * n)(*[L2Job(*j) for j in jobs])
        out = (L2Frame * n)()
        by = np.zeros((n, 18272), dtype=np.uint8)
        self._check(self.lib.nrsc5hip_l2_index(self._h, n, arr, out, by.ctypes.data, 18272))
        return out, by

    def l2_frame(self, stream: int, slot: int) -> dict:
        """Engine option l2_index: the index computed in the pipeline for the P1 frame in `slot`."""
        fr = L2Frame()
        self._check(self.lib.nrsc5hip_l2_frame_get(self._h, stream, slot, ctypes.byref(fr)))
        return l2_frame_to_dict(fr)

    def batch_fetch_l2(self, nstreams: int):
        """[nstreams][p1_slots] L2Frame ctypes array for streams 0..nstreams-1."""
        out = ((L2Frame * self.p1_slots) * nstreams)()
        self._check(self.lib.nrsc5hip_batch_fetch_l2(self._h, nstreams, None, out))
        return out

    def batch_fetch_l2_px(self, nstreams: int):
        """[nstreams][8 * p1_slots][2] L2Frame array: pipeline index of the P3 ([slot][0]) / P4 ([slot][1]) frames."""
        out = (((L2Frame * 2) * (8 * self.p1_slots)) * nstreams)()
        self._check(self.lib.nrsc5hip_batch_fetch_l2_px(self._h, nstreams, None, out))
        return out

    def batch_fetch_l2_am(self, nstreams: int):
        """[nstreams][p1_slots][9] L2Frame array: pipeline index of the 8 P1 frames + the P3 frame of every AM L1 frame slot."""
        out = (((L2Frame * 9) * self.p1_slots) * nstreams)()
        self._check(self.lib.nrsc5hip_batch_fetch_l2_am(self._h, nstreams, None, out))
        return out

    def stage_l2_index(self, frames_bits: np.ndarray, want_bytes: bool = True):
        """Same kernel on logical frames given as frame_push takes them: frames_bits [nframes][nbits] of 0/1."""
        b = np.ascontiguousarray(frames_bits, dtype=np.uint8)
        if b.ndim == 1:
            b = b[None, :]
        n, nbits = b.shape
        out = (L2Frame * n)()
        stride = 18272
        by = np.zeros((n, stride), dtype=np.uint8) if want_bytes else None
        self._check(self.lib.nrsc5hip_stage_l2_index(self._h, b.ctypes.data, nbits, n, out, by.ctypes.data if want_bytes else None, stride))
        return [(l2_frame_to_dict(out[k]), by[k, :out[k].nbytes].copy() if want_bytes else None) for k in range(n)]

    def stage_l2_index_raw(self, frames_bits: np.ndarray):
        """As stage_l2_index, but returns the C structs themselves: (L2Frame ctypes array, PDU bytes [n, 18272])."""
        b = np.ascontiguousarray(frames_bits, dtype=np.uint8)
        if b.ndim == 1:
            b = b[None, :]
        n, nbits = b.shape
        out = (L2Frame * n)()
        by = np.zeros((n, 18272), dtype=np.uint8)
        self._check(self.lib.nrsc5hip_stage_l2_index(self._h, b.ctypes.data, nbits, n, out, by.ctypes.data, 18272))
        return out, by

    def stage_viterbi_k7_debug(self, soft: np.ndarray, length: int):
        soft = np.ascontiguousarray(soft, dtype=np.int8)
        bits = np.zeros(length, dtype=np.uint8)
        dec = np.zeros(length + 64, dtype=np.uint64)
        self._check(self.lib.nrsc5hip_stage_viterbi_k7_debug(self._h, soft.ctypes.data, length, bits.ctypes.data, dec.ctypes.data))
        return bits, dec

    def stage_viterbi_bench(self, length: int, nframes: int, phases: int = 3, reps: int = 3) -> float:
        ms = ctypes.c_float()
        self._check(self.lib.nrsc5hip_stage_viterbi_bench(self._h, length, nframes, phases, reps, ctypes.byref(ms)))
        return ms.value

    def stage_viterbi_k9_bench(self, length: int, nframes: int, phases: int = 3, reps: int = 3) -> float:
        ms = ctypes.c_float()
        self.lib.nrsc5hip_stage_viterbi_k9_bench.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
        self._check(self.lib.nrsc5hip_stage_viterbi_k9_bench(self._h, length, nframes, phases, reps, ctypes.byref(ms)))
        return ms.value

    def fwd_stats(self):
        """(segment boundaries checked, segments repaired) of the segmented forward trellis pass since the engine was created"""
        st = (ctypes.c_int * 2)()
        self._check(self.lib.nrsc5hip_debug_fwd_stats(self._h, st))
        return int(st[0]), int(st[1])

    def flow_stats(self):
        """dataflow bursts (TUNE_FLOW_MIN): (bursts, block steps) issued as k_flow launches since the engine was created"""
        st = (ctypes.c_longlong * 2)()
        self._check(self.lib.nrsc5hip_debug_flow_stats(self._h, st))
        return int(st[0]), int(st[1])

    def host_capture_stats(self):
        """host-resident capture of the fast seam (TUNE_HOST_CAPTURE): dict(attaches, detaches, rebases, stream)"""
        st = (ctypes.c_longlong * 5)()
        self._check(self.lib.nrsc5hip_debug_host_capture_stats(self._h, st))
        return dict(zip(("attaches", "detaches", "rebases", "stream", "reports_folded"), (int(x) for x in st)))

    def tb_stats(self):
        """single-path traceback: (chunk boundaries checked, chunks re-walked) since the engine was created"""
        st = (ctypes.c_int * 2)()
        self._check(self.lib.nrsc5hip_debug_tb_stats(self._h, st))
        return int(st[0]), int(st[1])

    def stage_first_header(self, bits: np.ndarray, threads: int = 64) -> np.ndarray:
        """First-header verdicts (1 = the reference stays synchronised) of descrambled P1 frames, bits[nframes][146176 or 3750]."""
        bits = np.ascontiguousarray(bits, dtype=np.uint8)
        ok = np.zeros(bits.shape[0], dtype=np.int32)
        self._check(self.lib.nrsc5hip_stage_first_header(self._h, bits.ctypes.data, bits.shape[1], bits.shape[0], threads, ok.ctypes.data))
        return ok

    def k9_stats(self):
        """K=9 decode in segment waves: (forward boundaries checked, segments re-run, traceback boundaries checked, segments re-walked)"""
        st = (ctypes.c_int * 4)()
        self._check(self.lib.nrsc5hip_debug_k9_stats(self._h, st))
        return tuple(int(v) for v in st)

    def debug_sync_phases(self) -> np.ndarray:
        c = np.zeros(16, dtype=np.int64)                        # [0..7] k_sync, [8..15] k_mixfft (diagnostic build only)
        self._check(self.lib.nrsc5hip_debug_sync_phases(self._h, c.ctypes.data))
        return c

    def stage_selftest(self) -> int:
        n = ctypes.c_int(-1)
        self._check(self.lib.nrsc5hip_stage_selftest(self._h, ctypes.byref(n)))
        return n.value

    def debug_fetch_costas(self, stream: int):
        f = np.zeros(534, dtype=np.float32); p = np.zeros(534, dtype=np.float32)
        self._check(self.lib.nrsc5hip_debug_fetch_costas(self._h, stream, f.ctypes.data, p.ctypes.data))
        return f, p

    def debug_fetch_px(self, stream: int) -> np.ndarray:
        out = np.zeros((2, 2, 4608), dtype=np.int8)
        self.lib.nrsc5hip_debug_fetch_px.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        self._check(self.lib.nrsc5hip_debug_fetch_px(self._h, stream, out.ctypes.data))
        return out

    def debug_fetch(self, stream: int):
        pm = np.zeros(16 * 23040, dtype=np.int8)
        bins = np.zeros((32, 534), dtype=np.complex64)
        self._check(self.lib.nrsc5hip_debug_fetch(self._h, stream, pm.ctypes.data, bins.ctypes.data))
        return pm, bins


class HdcConsumer:
    """Slim batch consumer of the L2 index (nrsc5hip_hdc_*): elastic buffers of `nstreams` streams in ~40 KB each instead of
    one nrsc5_t per stream; delivers the reference's NRSC5_EVENT_HDC sequence."""

    def __init__(self, nstreams: int, lib: ctypes.CDLL | None = None, lib_path: str | None = None):
        self.lib = lib or load_library(lib_path)
        self._h = ctypes.c_void_p()
        if self.lib.nrsc5hip_hdc_create(nstreams, ctypes.byref(self._h)) != 0:
            raise Nrsc5HipError("nrsc5hip_hdc_create failed")
        self.events = []
        self._cb = HDC_CB(self._on_packet)

    def _on_packet(self, opaque, stream, program, data, count, flags):
        self.events.append((int(stream), int(program), int(count), int(flags), bytes(ctypes.string_at(data, count)) if count else b""))

    def push_frame(self, stream: int, frame: L2Frame, pdu_bytes: np.ndarray, lc: int = 0):
        b = np.ascontiguousarray(pdu_bytes, dtype=np.uint8)
        if self.lib.nrsc5hip_hdc_push_frame(self._h, stream, lc, ctypes.byref(frame), b.ctypes.data) != 0:
            raise Nrsc5HipError("nrsc5hip_hdc_push_frame failed")

    def fixed_audio_end(self, stream: int, lc: int, pdu_bytes: np.ndarray) -> int:
        b = np.ascontiguousarray(pdu_bytes, dtype=np.uint8)
        return int(self.lib.nrsc5hip_hdc_fixed_audio_end(self._h, stream, lc, b.ctypes.data, b.size))

    def frame_reset(self, stream: int):
        self.lib.nrsc5hip_hdc_frame_reset(self._h, stream)

    def advance(self, stream: int, mode: int = MODE_FM) -> int:
        return self.lib.nrsc5hip_hdc_advance(self._h, stream, mode, self._cb, None)

    def reset(self, stream: int):
        self.lib.nrsc5hip_hdc_reset(self._h, stream)

    def host_bytes(self) -> int:
        return int(self.lib.nrsc5hip_hdc_host_bytes(self._h))

    def adts(self, data: bytes) -> bytes:
        out = (ctypes.c_uint8 * (len(data) + 7))()
        src = (ctypes.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b"\0")
        n = self.lib.nrsc5hip_hdc_adts(src, len(data), out)
        return bytes(out[:n])

    def close(self):
        if self._h:
            self.lib.nrsc5hip_hdc_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def feed_hdc(engine: Engine, consumer: HdcConsumer, stream: int, recs: np.ndarray, mode: int = MODE_FM, target_stream: int | None = None):
    """Replays one stream's block records into the consumer in the reference's order: output_advance at the top of every
    processed block (acquire.c:108), then the frames that block delivers (frame_push -> frame_process)."""
    t = stream if target_stream is None else target_stream
    jobs_all = []
    per_rec = []
    for r in recs:
        jobs = l2_jobs_from_records(stream, np.array([r], dtype=RECORD_DTYPE), mode)
        per_rec.append((len(jobs_all), len(jobs)))
        jobs_all += jobs
    frames, by = engine.l2_index_raw(jobs_all) if jobs_all else (None, None)
    for r, (first, n) in zip(recs, per_rec):
        if int(r["flags"]) & REC_PROCESSED:
            consumer.advance(t, mode)
        if int(r["flags"]) & REC_TO_FINE:
            consumer.frame_reset(t)                              # sync.c:405-409
        for k in range(first, first + n):
            kind, which = jobs_all[k][2], jobs_all[k][3]
            lc = 0 if kind == L2_FM_P1 or (kind == L2_AM and which < 8) else (1 + which if kind == L2_FM_PX else 1)
            consumer.push_frame(t, frames[k], by[k, :frames[k].nbytes], lc)


def l2_jobs_from_records(stream: int, recs: np.ndarray, mode: int = 0):
    """nrsc5hip_l2_job tuples for every logical frame the records announce, in the order frame_push would see them."""
    jobs = []
    for r in recs:
        fl = int(r["flags"])
        if mode == MODE_AM:
            if fl & REC_P1:
                jobs.append((stream, int(r["p1_slot"]), L2_AM, int(r["bc_decoded"]), AM_P1_BITS))
            if fl & REC_P3:
                jobs.append((stream, int(r["p1_slot"]), L2_AM, 8, 30000 if int(r["psmi"]) == 2 else 24000))
        else:
            if fl & REC_P1:
                jobs.append((stream, int(r["p1_slot"]), L2_FM_P1, 0, P1_BITS))
            for flag, ch in ((REC_P3, 0), (REC_P4, 1)):
                if fl & flag:
                    jobs.append((stream, int(r["sis"]), L2_FM_PX, ch, 2304 if int(r["psmi"]) == 2 else 4608))
    return jobs


_BLOCK_KEYS = ("state_before", "state_after", "samperr", "cfo", "keep", "bc", "psmi", "cfo_wait",
               "next_samperr", "prev_angle", "phase_re", "phase_im", "next_angle")


def am_records_to_log(engine: Engine, stream: int, recs: np.ndarray, frames: np.ndarray | None = None):
    """AM twin of records_to_log: the reference's order inside one acquire_process call is
    [state, sync], pids, P1 frame, [P3 frame, ber], block (sync.c:639-765, decode.c:507-554)."""
    out = []
    for r in recs:
        fl = int(r["flags"])
        if fl & REC_TO_COARSE:
            out.append(("state", {"old": int(r["state_before"]), "new": SYNC_COARSE}))
        if fl & REC_TO_FINE:
            sis = int(r["sis"])
            out.append(("state", {"old": SYNC_COARSE, "new": SYNC_FINE}))
            out.append(("sync", {"freq_offset": float(r["freq_offset"]), "psmi": int(r["psmi"]), "pli": sis & 1, "hppi": (sis >> 1) & 1,
                                 "aabi": (sis >> 2) & 1, "rdbi": (sis >> 3) & 1}))
        if fl & REC_PIDS:
            out.append(("pids", {"bits": unpack_bits(r["pids"], PIDS_BITS)}))
        slot, bc = int(r["p1_slot"]), int(r["bc_decoded"])
        if fl & REC_P1:
            if frames is not None:
                bits = unpack_bits(frames[slot][bc * AM_P1_WORDS:(bc + 1) * AM_P1_WORDS], AM_P1_BITS)
            else:
                bits = engine.am_frame_bits(stream, slot, bc, AM_P1_BITS)
            out.append(("frame", {"lc": 0, "bits": bits}))
            if fl & REC_LOST_SYNC:                              # frame_push(P1) -> frame_process -> input_set_sync_state(NONE)
                out.append(("state", {"old": SYNC_FINE, "new": SYNC_NONE}))
                out.append(("lost_sync", {}))
        if fl & REC_P3:
            n3 = 30000 if int(r["psmi"]) == 2 else 24000
            if frames is not None:
                bits = unpack_bits(frames[slot][AM_P3_WORD0:AM_P3_WORD0 + (n3 + 31) // 32], n3)
            else:
                bits = engine.am_frame_bits(stream, slot, 8, n3)
            out.append(("frame", {"lc": 1, "bits": bits}))
        if (fl & REC_P1) and bc == 7:
            out.append(("ber", {"cber": float(r["ber"])}))
        out.append(("block", {k: (float(r[k]) if RECORD_DTYPE[k].kind == "f" else int(r[k])) for k in _BLOCK_KEYS}))
    return out


def records_to_log(engine: Engine, stream: int, recs: np.ndarray, frames: np.ndarray | None = None, px_frames: np.ndarray | None = None):
    """Expand block records into the ordered event list used by the oracle/reference harness logs
    (oracle/ref.py: parse_log), i.e. the order in which the reference fires them inside one
    acquire_process call."""
    out = []
    for r in recs:
        fl = int(r["flags"])
        if fl & REC_TO_COARSE:
            out.append(("state", {"old": int(r["state_before"]), "new": SYNC_COARSE}))
        if fl & REC_TO_FINE:
            out.append(("state", {"old": SYNC_COARSE, "new": SYNC_FINE}))
            out.append(("sync", {"freq_offset": float(r["freq_offset"]), "psmi": int(r["psmi"]), "pli": -1, "hppi": -1, "aabi": -1, "rdbi": -1}))
        if fl & REC_MER:
            out.append(("mer", {"lower": float(r["mer_lb"]), "upper": float(r["mer_ub"])}))
        if fl & REC_PIDS:
            out.append(("pids", {"bits": unpack_bits(r["pids"], PIDS_BITS)}))
        if fl & REC_P1:
            out.append(("ber", {"cber": float(r["ber"])}))
            if frames is not None:
                bits = unpack_bits(frames[int(r["p1_slot"])], P1_BITS)
            else:
                bits = engine.p1_frame_bits(stream, int(r["p1_slot"]))
            out.append(("frame", {"lc": 0, "bits": bits}))
        for flag, ch in ((REC_P3, 0), (REC_P4, 1)):           # decode_push_px1 / px2 (decode.c:393-437)
            if fl & flag:
                nbits = 2304 if int(r["psmi"]) == 2 else 4608
                if px_frames is not None:
                    bits = unpack_bits(px_frames[int(r["sis"]), ch], nbits)
                else:
                    bits = engine.px_frame_bits(stream, int(r["sis"]), ch, nbits)
                out.append(("frame", {"lc": 1 + ch, "bits": bits}))
        if fl & REC_LOST_SYNC:
            out.append(("state", {"old": SYNC_FINE, "new": SYNC_NONE}))
            out.append(("lost_sync", {}))
        blk = {k: (float(r[k]) if RECORD_DTYPE[k].kind == "f" else int(r[k]))
               for k in ("state_before", "state_after", "samperr", "cfo", "keep", "bc", "psmi", "cfo_wait",
                         "next_samperr", "prev_angle", "phase_re", "phase_im", "next_angle")}
        out.append(("block", blk))
    return out
