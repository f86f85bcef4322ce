"""The META chunk the reference CLI writes (reflo/src/lib.rs:202-283): a MessagePack map of the FloMetadata fields that
are set, in declaration order (rmp_serde::to_vec_named; every Option field is skipped when None,
libflo/src/core/metadata.rs:328-665). For a file without tags the reference sets exactly
    length_ms, encoding_time, encoder_settings, flo_encoder_version, source_format
(metadata.rs:442,463,518,654,658). Only these - plus title / artist / album, which the CLI accepts - are written here;
the rest of the metadata model (pictures, lyrics, analysis data ...) is out of this repository's scope.
"""
import struct
import time

import numpy as np

# declaration order of the fields this module can write (metadata.rs line numbers)
_ORDER = ["title", "artist", "album", "length_ms", "encoding_time", "encoder_settings", "flo_encoder_version", "source_format"]
ENCODER_VERSION = "reflo 0.1.2"       # format!("reflo {}", CARGO_PKG_VERSION): what files made by the reference tool carry


def _str(s: str) -> bytes:
    b = s.encode("utf-8")
    if len(b) < 32:
        return bytes([0xA0 | len(b)]) + b
    if len(b) < 256:
        return b"\xd9" + bytes([len(b)]) + b
    if len(b) < 65536:
        return b"\xda" + struct.pack(">H", len(b)) + b
    return b"\xdb" + struct.pack(">I", len(b)) + b


def _uint(v: int) -> bytes:
    if v < 128:
        return bytes([v])
    if v < 256:
        return b"\xcc" + bytes([v])
    if v < 65536:
        return b"\xcd" + struct.pack(">H", v)
    if v < 1 << 32:
        return b"\xce" + struct.pack(">I", v)
    return b"\xcf" + struct.pack(">Q", v)


def pack_fields(fields: dict) -> bytes:
    items = [(k, fields[k]) for k in _ORDER if fields.get(k) is not None]
    assert len(items) < 16
    out = bytes([0x80 | len(items)])
    for k, v in items:
        out += _str(k) + (_uint(v) if isinstance(v, int) else _str(v))
    return out


def encoder_settings(lossy: bool, quality: float, bitrate, level: int) -> str:
    """reflo/src/lib.rs:262-271"""
    if lossy or bitrate is not None:
        if bitrate is not None:
            return f"Lossy, target {bitrate}kbps"
        return f"Lossy, quality {float(np.float32(quality) * np.float32(100.0)):.0f}%"
    return f"Lossless, level {level}"


def cli_metadata(n_interleaved: int, sample_rate: int, channels: int, source_format: str, lossy: bool, quality: float,
                 bitrate, level: int, title=None, artist=None, album=None, encoding_time=None) -> bytes:
    """META bytes of a file encoded by the CLI (encoding_time: "%Y-%m-%dT%H:%M:%SZ" in UTC, now if None)."""
    total = n_interleaved // channels
    length_ms = int(total / sample_rate * 1000.0)          # (total as f64 / sample_rate as f64 * 1000.0) as u64
    if encoding_time is None:
        encoding_time = time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime())
    return pack_fields(dict(title=title, artist=artist, album=album, length_ms=length_ms, encoding_time=encoding_time,
                            encoder_settings=encoder_settings(lossy, quality, bitrate, level),
                            flo_encoder_version=ENCODER_VERSION, source_format=source_format))


def unpack(data: bytes):
    """MessagePack -> Python (maps, arrays, strings, binaries, integers, floats, booleans, nil): enough to show the META
    chunk of any .flo file, whoever wrote it. Raises ValueError on malformed input."""
    pos = 0

    def need(n):
        nonlocal pos
        if pos + n > len(data):
            raise ValueError("truncated MessagePack data")
        b = data[pos:pos + n]
        pos += n
        return b

    def one():
        t = need(1)[0]
        if t < 0x80:
            return t
        if t >= 0xE0:
            return t - 256
        if 0x80 <= t <= 0x8F:
            return {one(): one() for _ in range(t & 15)}
        if 0x90 <= t <= 0x9F:
            return [one() for _ in range(t & 15)]
        if 0xA0 <= t <= 0xBF:
            return need(t & 31).decode("utf-8", "replace")
        if t == 0xC0:
            return None
        if t in (0xC2, 0xC3):
            return t == 0xC3
        if t in (0xC4, 0xC5, 0xC6):
            n = int.from_bytes(need(1 << (t - 0xC4)), "big")
            return bytes(need(n))
        if t == 0xCA:
            return struct.unpack(">f", need(4))[0]
        if t == 0xCB:
            return struct.unpack(">d", need(8))[0]
        if 0xCC <= t <= 0xCF:
            return int.from_bytes(need(1 << (t - 0xCC)), "big")
        if 0xD0 <= t <= 0xD3:
            return int.from_bytes(need(1 << (t - 0xD0)), "big", signed=True)
        if t in (0xD9, 0xDA, 0xDB):
            n = int.from_bytes(need(1 << (t - 0xD9)), "big")
            return need(n).decode("utf-8", "replace")
        if t in (0xDC, 0xDD):
            n = int.from_bytes(need(2 if t == 0xDC else 4), "big")
            return [one() for _ in range(n)]
        if t in (0xDE, 0xDF):
            n = int.from_bytes(need(2 if t == 0xDE else 4), "big")
            return {one(): one() for _ in range(n)}
        raise ValueError(f"unsupported MessagePack type 0x{t:02x}")

    v = one()
    return v
