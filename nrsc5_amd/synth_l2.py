"""Signal source for the L2 audio-transport index (tests only, never on the product path): builds logical frames
the way a transmitter would so that frame_push / frame_process (frame.c:516-714) have every branch to walk --
several audio PDUs per frame, 12- and 16-bit locators, header expansion fields, enhanced streams, half packets,
header byte errors inside and beyond the RS(255,247) correction radius, CRC-8 failures."""
from __future__ import annotations

import numpy as np

from . import synth
from .synth_am import frame_bits

# frame_push's switch (frame.c:651-686): nbits -> (start, step, pci_len)
LAYOUT = {146176: (146176 - 30000, 1248, 24), 4608: (120, 184, 24), 2304: (120, 88, 24),
          3750: (120, 160, 22), 24000: (120, 992, 24), 30000: (120, 1240, 24)}
PCI_AUDIO, PCI_AUDIO_OPP, PCI_AUDIO_FIXED, PCI_AUDIO_FIXED_OPP, PCI_FIXED = 0x38D8D3, 0xCE3634, 0xE3634C, 0x8D8D33, 0x3634CE


def pdu_bytes_of(nbits: int) -> int:
    return (nbits - LAYOUT[nbits][2]) // 8


def lc_bits(codec_mode: int, stream_id: int) -> int:
    """calc_lc_bits, frame.c:267-287"""
    if codec_mode in (1, 2, 3):
        return 12 if stream_id == 0 else 16
    return 12 if codec_mode in (10, 13) else 16


def hef_bytes(prog_num=None, class_ind=None, access=None, prog_type=None, pdu_len=None, marker=None) -> bytes:
    """Header expansion fields (parse_hef, frame.c:198-265)."""
    fields = []
    if class_ind is not None:
        fields.append([0x00 | (class_ind & 0xf)])
    if prog_num is not None:
        if pdu_len is None:
            fields.append([0x10 | ((prog_num & 7) << 1)])
        else:
            fields.append([0x10 | ((prog_num & 7) << 1) | 1, 0x80 | ((pdu_len >> 7) & 0x7f), pdu_len & 0x7f])
    if access is not None or prog_type is not None:
        a, t = access or 0, prog_type or 0
        fields.append([0x20 | (a << 3) | (t >> 7), 0x80 | (t & 0x7f)])
    if marker is not None:
        fields.append([0x48 | 0x5, 0x80 | ((marker >> 14) & 0x7f), 0x80 | ((marker >> 7) & 0x7f), marker & 0x7f])
    out = []
    for k, f in enumerate(fields):                # the do/while tests bit 7 of the LAST byte of each field
        f = [x & 0x7f for x in f]
        if k != len(fields) - 1:
            f[-1] |= 0x80
        out += f
    return bytes(out)


def make_pdu(rng: np.random.Generator, room: int, nop: int = 4, codec_mode: int = 0, stream_id: int = 0, pdu_seq: int = 0,
             seq: int = 0, pfirst: int = 0, plast: int = 0, latency: int = 0, blend: int = 0, psd_delay: int = 0,
             common_delay: int = 0, hef: bytes = b"", psd: bytes = b"\x7e\x7e", bad_crc=(), fill: bool = True) -> bytes:
    """One audio PDU of at most `room` bytes: RS-protected header, locators, optional HEF, PSD bytes, nop packets
    with CRC-8.  Packets share the room evenly (fill) or are short."""
    lcb = lc_bits(codec_mode, stream_id)
    loc_bytes = (lcb * nop + 4) // 8
    la_location = 14 + loc_bytes + len(hef) + len(psd) - 1
    assert la_location <= 255 and room > la_location + 1 + 2 * nop
    pdu = bytearray(room)
    pdu[8] = codec_mode | (stream_id << 4) | ((pdu_seq & 3) << 6)
    pdu[9] = (pdu_seq >> 2) | (blend << 1) | (psd_delay << 3)
    pdu[10] = common_delay | ((latency & 3) << 6)
    pdu[11] = (latency >> 2) | (pfirst << 1) | (plast << 2) | ((seq & 31) << 3)
    pdu[12] = (seq >> 5) | (nop << 1) | ((1 if hef else 0) << 7)
    pdu[13] = la_location
    pos = 14 + loc_bytes
    pdu[pos:pos + len(hef)] = hef
    pos += len(hef)
    pdu[pos:pos + len(psd)] = psd
    pos += len(psd)
    assert pos == la_location + 1
    avail = room - pos
    locs = []
    for j in range(nop):
        size = (avail // nop - 1) if fill else int(rng.integers(1, max(2, min(40, avail // nop - 1))))
        size = min(size, 4095 - pos - 1) if lcb == 12 else size
        if j == nop - 1 and pos + size + 1 < 100:
            size = 100 - pos                      # the RS code word spans 96 bytes: a PDU cannot be shorter
        payload = rng.integers(0, 256, size=max(size, 1), dtype=np.uint8).tobytes()
        pdu[pos:pos + len(payload)] = payload
        pos += len(payload)
        pdu[pos] = synth.crc8(payload) ^ (0x5a if j in bad_crc else 0)
        locs.append(pos)
        pos += 1
    for j, loc in enumerate(locs):
        assert loc < (1 << lcb)
        if lcb == 16:
            pdu[14 + 2 * j] = loc & 0xff
            pdu[15 + 2 * j] = loc >> 8
        elif j % 2 == 0:
            pdu[14 + j // 2 * 3] = loc & 0xff
            pdu[14 + j // 2 * 3 + 1] |= loc >> 8
        else:
            pdu[14 + j // 2 * 3 + 1] |= (loc & 0xf) << 4
            pdu[14 + j // 2 * 3 + 2] = loc >> 4
    data = [0] * 159 + [pdu[254 - k] for k in range(159, 247)]
    par = synth._GF.rs_parity(data)
    for k in range(8):
        pdu[7 - k] = par[k]
    return bytes(pdu[:pos])


def frame_from_bytes(body: bytes, nbits: int, pci: int = PCI_AUDIO, tail: bytes | None = None) -> np.ndarray:
    """Pad PDU bytes with zeros (or `tail` repeated) to the frame's byte count and return the bits frame_push takes."""
    n = pdu_bytes_of(nbits)
    assert len(body) <= n
    pad = n - len(body)
    body = bytes(body) + ((tail * (pad // len(tail) + 1))[:pad] if tail else bytes(pad))
    start, step, pci_len = LAYOUT[nbits]
    return frame_bits(body, nbits, start, step, pci_len, pci)


def corrupt(body: bytes, positions, rng: np.random.Generator) -> bytes:
    out = bytearray(body)
    for q in positions:
        out[q] ^= int(rng.integers(1, 256))
    return bytes(out)


def reparity(pdu: bytes) -> bytes:
    """Recompute the RS(255,247) parity of a PDU whose first 96 bytes were edited."""
    pdu = bytearray(pdu)
    data = [0] * 159 + [pdu[254 - k] for k in range(159, 247)]
    par = synth._GF.rs_parity(data)
    for k in range(8):
        pdu[7 - k] = par[k]
    return bytes(pdu)


def test_frames(nbits: int, seed: int = 0):
    """(name, frame bits, reference_safe) cases that walk every branch of frame_process for one frame length.
    reference_safe = False marks inputs on which the reference itself is undefined (parse_hdlc length wrap) or
    on which the index is deliberately bounded (more than 16 PDUs)."""
    rng = np.random.default_rng(1000 * seed + nbits)
    n = pdu_bytes_of(nbits)
    big = n > 2500
    cases = []

    def add(name, body, safe=True, **kw):
        cases.append((name, frame_from_bytes(body, nbits, **kw), safe))

    nop_full = 32 if big else 3
    full = make_pdu(rng, n, nop=nop_full, seq=5, pdu_seq=1)
    add("single", full)
    add("single_opp_pci", full, pci=PCI_AUDIO_OPP)
    add("no_audio_pci", full, pci=PCI_FIXED)
    add("fixed_pci", full, pci=PCI_AUDIO_FIXED)
    add("fixed_opp_pci", full, pci=PCI_AUDIO_FIXED_OPP)
    add("pci_low_bits", full, pci=PCI_AUDIO ^ 3)
    for k in (1, 4, 5, 9):
        add(f"hdr_err{k}", corrupt(full, rng.choice(96, size=k, replace=False), rng))
    add("garbage", rng.integers(0, 256, size=n, dtype=np.uint8).tobytes())
    add("zeros", bytes(n))
    add("ones", b"\xff" * n)
    if big:
        room = n // 5
        p = [make_pdu(rng, room, nop=9, seq=60, pdu_seq=3, pfirst=1, plast=1, latency=3, bad_crc=(0, 4, 8)),
             make_pdu(rng, room, nop=7, codec_mode=13, hef=hef_bytes(prog_num=1, class_ind=3, access=1, prog_type=0x85), latency=2, pdu_seq=5, seq=33),
             make_pdu(rng, room, nop=5, codec_mode=2, stream_id=1, hef=hef_bytes(prog_num=1), fill=False),
             make_pdu(rng, room, nop=3, codec_mode=1, hef=hef_bytes(prog_num=2, pdu_len=1234, marker=0x12345), psd=b"\x7e\x21abc\x7d\x5e\x7e", blend=2, psd_delay=17, common_delay=41),
             make_pdu(rng, room // 2, nop=63, codec_mode=10, stream_id=1, hef=hef_bytes(prog_num=7), fill=False)]
        body = b"".join(p)
        add("multi", body)
        add("multi_tail_ff", body, tail=b"\xff")
        add("multi_hdr2_err3", corrupt(body, [len(p[0]) + k for k in (1, 13, 40)], rng))
        add("multi_hdr2_err8", corrupt(body, [len(p[0]) + k for k in (1, 2, 3, 4, 5, 6, 7, 9)], rng))
        add("multi_rs_fixes_packet", corrupt(body, [len(p[0]) + len(p[1]) + 95], rng))     # byte 95 belongs to packet 0 of PDU 3
        add("stream2_skipped", make_pdu(rng, room, nop=3, stream_id=2) + make_pdu(rng, room, nop=3, stream_id=1) + make_pdu(rng, room, nop=2, stream_id=3))
        small = [make_pdu(rng, 400, nop=2, hef=hef_bytes(prog_num=k % 8), fill=False) for k in range(20)]
        add("too_many_pdus", b"".join(small), safe=False)
    else:
        half = n // 2
        p = [make_pdu(rng, half, nop=2, pfirst=1, bad_crc=(1,), seq=63), make_pdu(rng, n - half, nop=2, codec_mode=13, hef=hef_bytes(prog_num=3), plast=1)]
        body = b"".join(p)
        add("two", body)
        add("two_hdr2_err2", corrupt(body, [len(p[0]) + 8, len(p[0]) + 30], rng))
        add("two_hdr2_err7", corrupt(body, [len(p[0]) + k for k in (0, 9, 20, 31, 42, 53, 64)], rng))
    # locator faults: each of the returns of frame.c:547-556
    base = bytearray(make_pdu(rng, min(n, 3000), nop=4))
    f = bytearray(base); f[13] = 14; add("la_before_locators", reparity(bytes(f)))
    f = bytearray(base); f[14], f[15] = f[13], 0; add("loc0_not_after_la", reparity(bytes(f)))
    f = bytearray(base); f[16], f[17] = f[14], f[15]; add("loc_not_increasing", reparity(bytes(f)))
    f = bytearray(base); f[20], f[21] = 0xff, 0xff; add("loc_past_end", reparity(bytes(f)))
    f = bytearray(base); f[12] &= 0x81; add("nop0", reparity(bytes(f)))
    f = bytearray(make_pdu(rng, min(n, 3000), nop=2, stream_id=2)); f[12] &= 0x81; add("stream2_nop0", reparity(bytes(f)), safe=False)
    # header expansion that runs past la_location (the reference's PSD length wraps): index must say so, not guess
    f = bytearray(make_pdu(rng, min(n, 3000), nop=2, hef=b"\x80\x80\x80\x00", psd=b"")); f[13] -= 2; add("hef_overrun", reparity(bytes(f)), safe=False)
    f = bytearray(make_pdu(rng, min(n, 3000), nop=2, hef=b"\x91\x80", psd=b"")); add("hef_truncated_field", reparity(bytes(f)), safe=False)
    return cases


def random_frame(rng: np.random.Generator):
    """(nbits, frame bits): 1..6 PDUs with random codec modes (12- / 16-bit locators, unknown modes), packet counts,
    enhanced streams, HEF combinations, half packets, CRC failures, then maybe 1..6 corrupted bytes in one header."""
    nbits = int(rng.choice([146176, 146176, 24000, 30000, 4608, 3750, 2304]))
    n = pdu_bytes_of(nbits)
    pdus, used = [], 0
    for _ in range(int(rng.integers(1, 7))):
        room = n - used
        if room < 260:
            break
        codec = int(rng.choice([0, 0, 1, 2, 3, 10, 13, 5]))
        sid = int(rng.integers(0, 2))
        twelve = lc_bits(codec, sid) == 12
        room = int(min(room, rng.integers(260, 5000), 4000 if twelve else 1 << 30))
        hef = b"" if rng.random() < 0.4 else hef_bytes(
            prog_num=int(rng.integers(0, 8)), class_ind=int(rng.integers(0, 16)) if rng.random() < 0.5 else None,
            access=int(rng.integers(0, 2)) if rng.random() < 0.5 else None, prog_type=int(rng.integers(0, 256)),
            pdu_len=int(rng.integers(0, 1 << 14)) if rng.random() < 0.3 else None,
            marker=int(rng.integers(0, 1 << 21)) if rng.random() < 0.3 else None)
        psd = bytes(rng.integers(0, 256, size=int(rng.integers(0, 12)), dtype=np.uint8))
        max_nop = max(1, min(63, (room - 60 - len(hef) - len(psd)) // 4, (200 - len(hef) - len(psd)) // 2))
        nop = int(rng.integers(1, max_nop + 1))
        p = make_pdu(rng, room, nop=nop, codec_mode=codec, stream_id=sid, pdu_seq=int(rng.integers(0, 8)),
                     seq=int(rng.integers(0, 64)), pfirst=int(rng.integers(0, 2)), plast=int(rng.integers(0, 2)),
                     latency=int(rng.integers(0, 8)), blend=int(rng.integers(0, 4)), psd_delay=int(rng.integers(0, 32)),
                     common_delay=int(rng.integers(0, 64)), hef=hef, psd=psd,
                     bad_crc=tuple(int(x) for x in rng.choice(nop, size=int(rng.integers(0, 3)), replace=False)) if nop > 2 else (),
                     fill=bool(rng.integers(0, 2)))
        pdus.append(p)
        used += len(p)
    body = b"".join(pdus)
    if rng.random() < 0.5:
        which = int(rng.integers(0, len(pdus)))
        base = sum(len(p) for p in pdus[:which])
        body = corrupt(body, base + rng.choice(96, size=int(rng.integers(1, 7)), replace=False), rng)
    bits = frame_from_bytes(body, nbits, pci=int(rng.choice([PCI_AUDIO, PCI_AUDIO_OPP])), tail=None if rng.random() < 0.7 else b"\xa5")
    return nbits, bits


def fcs16(data: bytes) -> int:
    """PPP FCS-16 (RFC 1662; the table of frame.c:92-125): reflected 0x8408, initial value 0xFFFF."""
    crc = 0xFFFF
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ 0x8408 if crc & 1 else crc >> 1
    return crc


def hdlc_frame(payload: bytes) -> bytes:
    """One HDLC frame as parse_hdlc / aas_push expect it (frame.c:330-391): 0x7E, escaped payload + FCS, 0x7E."""
    body = payload + (fcs16(payload) ^ 0xFFFF).to_bytes(2, "little")
    out = bytearray([0x7E])
    for b in body:
        if b in (0x7E, 0x7D):
            out += bytes([0x7D, b ^ 0x20])
        else:
            out.append(b)
    out.append(0x7E)
    return bytes(out)


def psd_sequence(seed: int = 0, nbits: int = 146176, n_frames: int = 6):
    """Frames whose PDUs carry a PSD byte stream for two programs: AAS packets (protocol 0x21, good and bad FCS, escaped
    bytes) cut at arbitrary places so that HDLC frames span PDUs and logical frames, plus changing service parameters."""
    rng = np.random.default_rng(seed)
    streams = {}
    for prog in (0, 3):
        s = bytearray()
        for k in range(12):
            pay = bytes([0x21]) + rng.integers(0, 256, size=int(rng.integers(5, 60)), dtype=np.uint8).tobytes() + (b"\x7e\x7d" if k % 3 == 0 else b"")
            f = bytearray(hdlc_frame(pay))
            if k % 5 == 4:
                f[3] ^= 0x10                                   # broken FCS: dropped by aas_push
            s += f + b"\x7e" * int(rng.integers(0, 3))
        streams[prog] = bytes(s)
    pos = {0: 0, 3: 0}
    frames = []
    n = pdu_bytes_of(nbits)
    for fi in range(n_frames):
        pdus = []
        for prog in (0, 3):
            take = int(rng.integers(10, 70))
            psd = streams[prog][pos[prog]:pos[prog] + take]
            pos[prog] += take
            pdus.append(make_pdu(rng, min(n // 3, 3500), nop=4 + fi % 3, codec_mode=0 if prog == 0 else 13, seq=(7 * fi) % 64, pdu_seq=fi % 8,
                                 hef=hef_bytes(prog_num=prog, access=fi // 4 % 2, prog_type=10 + fi // 3) if prog else b"",
                                 psd=psd, latency=fi // 2 % 8, blend=fi // 3 % 4, common_delay=fi // 2 % 64))
        frames.append(frame_from_bytes(b"".join(pdus), nbits))
    return frames


def fixed_data_session(seed: int = 0, nbits: int = 146176, n_frames: int = 7, sub_len: int = 4000, sync_byte: int = 0x88, fixed_only=()):
    """Frames that carry fixed-data sub-channels next to audio (PCI_AUDIO_FIXED / _OPP, frame.c:138-151,458-514): the last
    byte is the sync byte (0x88: 16 CCC bytes per frame in front of it), the CCC bytes carry an HDLC message announcing one
    sub-channel of `sub_len` bytes, and five audio PDUs fill the frame -- so that audio_end is length - 1 for the first two
    frames (sync not yet confirmed), length - 17 while the CCC message is incomplete, and length - 17 - sub_len after it:
    the last PDUs then lie in the fixed-data region and the reference's walk stops in front of them.  Frames listed in
    `fixed_only` carry PCI_FIXED (no audio): frame_process still runs process_fixed_data on them before it returns."""
    rng = np.random.default_rng(7000 + seed)
    n = pdu_bytes_of(nbits)
    width = (sync_byte & 0xF) * 2 if sync_byte else 1
    msg = hdlc_frame(bytes([0x00]) + (0).to_bytes(2, "little") + sub_len.to_bytes(2, "little"))
    # the CCC byte stream, `width` bytes of it per frame; the first two frames only establish the sync byte (their CCC bytes are
    # not parsed), the message then straddles frames 3 and 4
    ccc = (b"\x7e" * (3 * width - 4) + msg + b"\x7e" * 128)
    frames, pos = [], 0
    for fi in range(n_frames):
        room = n // 5
        pdus = [make_pdu(rng, room, nop=6 + (fi + k) % 4, seq=(5 * fi + 11 * k) % 64, pdu_seq=(fi + k) % 8, codec_mode=0 if k % 2 == 0 else 13,
                         hef=hef_bytes(prog_num=k % 3) if k else b"") for k in range(5)]
        body = bytearray(b"".join(pdus))
        body += bytes(n - len(body))
        body[n - 1] = sync_byte
        body[n - 1 - width:n - 1] = ccc[pos:pos + width]
        pos += width
        frames.append(frame_from_bytes(bytes(body), nbits, pci=PCI_FIXED if fi in fixed_only else PCI_AUDIO_FIXED if fi % 2 == 0 else PCI_AUDIO_FIXED_OPP))
    return frames
