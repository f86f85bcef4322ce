"""Independent pure-Python parser of the .flo container (tests only).

Written from the byte layout in SURVEY.md §8a (a10, a21) / Appendix A so that framing produced by the
oracle and by the HIP library is checked by a third implementation.
"""
import struct
import zlib
from dataclasses import dataclass, field
from typing import List


@dataclass
class Channel:
    raw: bytes = b""              # whole channel payload as stored
    coeffs: List[int] = field(default_factory=list)
    shift_bits: int = 0
    encoding: int = 0
    rice_k: int = 0
    residuals: bytes = b""


@dataclass
class Frame:
    frame_type: int
    frame_samples: int
    flags: int
    channels: List[Channel]
    size: int


@dataclass
class FloFile:
    version: tuple
    flags: int
    sample_rate: int
    channels: int
    bit_depth: int
    total_samples: int
    level: int
    data_crc32: int
    header_size: int
    toc_size: int
    data_size: int
    extra_size: int
    meta_size: int
    toc: list
    frames: List[Frame]
    data: bytes
    meta: bytes

    @property
    def crc_valid(self):
        return (zlib.crc32(self.data) & 0xFFFFFFFF) == self.data_crc32

    @property
    def is_lossy(self):
        return bool(self.flags & 1)

    @property
    def lossy_quality(self):
        return (self.flags >> 8) & 0xFF


def parse(b: bytes) -> FloFile:
    assert b[:4] == b"FLO!", "bad magic"
    (vmaj, vmin, flags, sr, ch, bd, total, level) = struct.unpack_from("<BBHIBBQB", b, 4)
    assert b[23:26] == b"\0\0\0"
    crc, hsz, tsz, dsz, esz, msz = struct.unpack_from("<IQQQQQ", b, 26)
    assert hsz == 66
    pos = 70
    n = struct.unpack_from("<I", b, pos)[0]
    assert tsz == 4 + 20 * n
    toc = [struct.unpack_from("<IQII", b, pos + 4 + 20 * i) for i in range(n)]
    pos += tsz
    data = b[pos:pos + dsz]
    assert len(data) == dsz
    meta = b[pos + dsz + esz: pos + dsz + esz + msz]
    assert len(meta) == msz and pos + dsz + esz + msz == len(b)
    frames = []
    for (idx, off, size, ts) in toc:
        p = off
        ft, fs, fl = struct.unpack_from("<BIB", data, p)
        p += 6
        chans = []
        nch = 1 if ft == 253 else ch
        for _ in range(nch):
            clen = struct.unpack_from("<I", data, p)[0]
            p += 4
            raw = data[p:p + clen]
            c = Channel(raw=raw)
            if 1 <= ft <= 12:
                q = 0
                nco = raw[q]; q += 1
                c.coeffs = list(struct.unpack_from("<%di" % nco, raw, q)); q += 4 * nco
                c.shift_bits = raw[q]; c.encoding = raw[q + 1]; q += 2
                if c.encoding == 0:
                    c.rice_k = raw[q]; q += 1
                c.residuals = raw[q:]
            else:
                c.residuals = raw
            chans.append(c)
            p += clen
        assert p - off == size, (p - off, size)
        frames.append(Frame(ft, fs, fl, chans, size))
    return FloFile((vmaj, vmin), flags, sr, ch, bd, total, level, crc, hsz, tsz, dsz, esz, msz, toc, frames, data, meta)


def decode_varint(buf, pos):
    v = 0
    shift = 0
    while True:
        byte = buf[pos]
        pos += 1
        v |= (byte & 0x7F) << shift
        if not byte & 0x80:
            break
        shift += 7
    return v, pos


def parse_transform_blob(blob: bytes):
    """-> (n_ch, sf_words [ch][25], coefficient lists [ch][1024])"""
    import numpy as np
    assert blob[0] == 0
    nch = blob[1]
    p = 2
    sfw = np.frombuffer(blob[p:p + 50 * nch], dtype="<u2").reshape(nch, 25).copy()
    p += 50 * nch
    qs = np.zeros((nch, 1024), dtype=np.int16)
    for c in range(nch):
        ln = struct.unpack_from("<I", blob, p)[0]
        p += 4
        sp = blob[p:p + ln]
        p += ln
        i = 0
        k = 0
        while i < len(sp) and k < 1024:
            z, i = decode_varint(sp, i)
            k += z
            if i >= len(sp):
                break
            nz = sp[i]
            i += 1
            vals = np.frombuffer(sp[i:i + 2 * nz], dtype="<i2")
            qs[c, k:k + nz] = vals
            k += nz
            i += 2 * nz
    assert p == len(blob)
    return nch, sfw, qs


def build_lossless(sample_rate: int, channels: int, frames, version=(1, 2), bit_depth=16, level=5) -> bytes:
    """Writer for hand-made lossless files (tests only). `frames` = [(frame_type, frame_samples, flags, [channel, ...])]
    where a channel is either raw payload bytes (silence / raw frames) or a dict
    {coeffs: [...], shift: int, k: int, residuals: bytes} for an ALPC wrapper (encoding byte 0 = Rice)."""
    data = bytearray()
    toc = []
    ts = 0
    for i, (ft, fs, fl, chans) in enumerate(frames):
        off = len(data)
        data += struct.pack("<BIB", ft, fs, fl)
        for c in chans:
            if isinstance(c, dict):
                body = bytes([len(c["coeffs"])]) + struct.pack("<%di" % len(c["coeffs"]), *c["coeffs"])
                body += bytes([c["shift"] & 0xFF, 0, c["k"] & 0xFF]) + bytes(c["residuals"])
            else:
                body = bytes(c)
            data += struct.pack("<I", len(body)) + body
        toc.append((i, off, len(data) - off, ts))
        ts += fs * 1000 // max(sample_rate, 1)
    tocb = struct.pack("<I", len(toc)) + b"".join(struct.pack("<IQII", *t) for t in toc)
    total = sum(f[1] for f in frames)
    head = b"FLO!" + struct.pack("<BBHIBBQB", version[0], version[1], 0, sample_rate, channels, bit_depth, total, level) + b"\0\0\0"
    head += struct.pack("<IQQQQQ", zlib.crc32(bytes(data)) & 0xFFFFFFFF, 66, len(tocb), len(data), 0, 0)
    assert len(head) == 70
    return head + tocb + bytes(data)


def build_transform(sample_rate: int, channels: int, frames, quality_byte=140) -> bytes:
    """Writer for hand-made transform (lossy) files (tests only). `frames` = [[(sf_words[25], sparse_bytes), ... per
    channel], ...]; every frame is a Long block of 1024 sample-frames (writer.rs:236-254, encoder.rs:243-280)."""
    data = bytearray()
    toc = []
    for i, chans in enumerate(frames):
        blob = bytes([0, len(chans)])
        for sfw, _ in chans:
            blob += struct.pack("<25H", *[int(x) for x in sfw])
        for _, sp in chans:
            blob += struct.pack("<I", len(sp)) + bytes(sp)
        off = len(data)
        data += struct.pack("<BIB", 253, 1024, 0) + struct.pack("<I", len(blob)) + blob
        toc.append((i, off, len(data) - off, i * 1024 * 1000 // max(sample_rate, 1)))
    tocb = struct.pack("<I", len(toc)) + b"".join(struct.pack("<IQII", *t) for t in toc)
    total = 1024 * len(frames)
    flags = 0x01 | (quality_byte << 8)
    head = b"FLO!" + struct.pack("<BBHIBBQB", 1, 2, flags, sample_rate, channels, 16, total, 5) + b"\0\0\0"
    head += struct.pack("<IQQQQQ", zlib.crc32(bytes(data)) & 0xFFFFFFFF, 66, len(tocb), len(data), 0, 0)
    assert len(head) == 70
    return head + tocb + bytes(data)


def encode_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)
