"""WAV ingestion and output with the reference CLI's scaling rules.

Reading mirrors what `reflo` does with a WAV file (reflo/src/audio.rs:57-166 decodes with symphonia, then
`append_samples`, audio.rs:238-275, converts to interleaved f32):
    unsigned 8-bit  -> (x - 128) / 128
    signed 16-bit   -> x * (1 / 32768)
    signed 24-bit   -> decoded into 32-bit containers (left-justified), then the 32-bit rule
    signed 32-bit   -> x * (1 / 2147483648)
    float 32-bit    -> as is
Writing is `write_wav_to_bytes` (audio.rs:290-320): a 44-byte RIFF header, format tag 3 (IEEE float), 32 bits.
Other encodings (64-bit float, ADPCM, mu-law ...) are refused with an error, where the reference's fallback arm
silently yields no samples.
"""
import struct

import numpy as np


class WavError(ValueError):
    pass


def read_wav_bytes(data: bytes):
    """-> (interleaved float32 samples, sample_rate, channels)"""
    if len(data) < 12 or data[0:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise WavError("not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack_from("<I", data, pos + 4)[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            if len(body) < 16:
                raise WavError("short fmt chunk")
            tag, ch, sr, _rate, _align, bits = struct.unpack_from("<HHIIHH", body, 0)
            if tag == 0xFFFE and len(body) >= 26:           # WAVE_FORMAT_EXTENSIBLE: the sub-format's first two bytes
                tag = struct.unpack_from("<H", body, 24)[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise WavError("missing fmt or data chunk")
    tag, ch, sr, bits = fmt
    if ch == 0 or sr == 0:
        raise WavError("zero channels or sample rate")
    if tag == 3 and bits == 32:
        x = np.frombuffer(pcm[: len(pcm) // 4 * 4], dtype="<f4").astype(np.float32)
    elif tag == 1 and bits == 8:
        x = (np.frombuffer(pcm, dtype=np.uint8).astype(np.float32) - np.float32(128.0)) / np.float32(128.0)
    elif tag == 1 and bits == 16:
        x = np.frombuffer(pcm[: len(pcm) // 2 * 2], dtype="<i2").astype(np.float32) * np.float32(1.0 / 32768.0)
    elif tag == 1 and bits == 24:
        raw = np.frombuffer(pcm[: len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.uint32)
        v = ((raw[:, 0] << 8) | (raw[:, 1] << 16) | (raw[:, 2] << 24)).astype(np.uint32).view(np.int32)   # left-justified
        x = v.astype(np.float32) * np.float32(1.0 / 2147483648.0)
    elif tag == 1 and bits == 32:
        x = np.frombuffer(pcm[: len(pcm) // 4 * 4], dtype="<i4").astype(np.float32) * np.float32(1.0 / 2147483648.0)
    else:
        raise WavError(f"unsupported WAV encoding (format tag {tag}, {bits} bits)")
    x = x[: x.size // ch * ch]
    return np.ascontiguousarray(x, dtype=np.float32), int(sr), int(ch)


def write_wav_bytes(samples, sample_rate: int, channels: int) -> bytes:
    x = np.ascontiguousarray(samples, dtype="<f4").reshape(-1)
    data_size = x.size * 4
    head = b"RIFF" + struct.pack("<I", 36 + data_size) + b"WAVE" + b"fmt " + struct.pack(
        "<IHHIIHH", 16, 3, channels, sample_rate, sample_rate * channels * 4, channels * 4, 32) + b"data" + struct.pack("<I", data_size)
    return head + x.tobytes()
