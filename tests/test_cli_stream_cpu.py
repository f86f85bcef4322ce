"""CPU side of SURVEY 8f-4: WAV ingestion rules, the CLI's file inspection, and the oracle's streaming encoder."""
import struct

import numpy as np
import pytest

import flofile
import signals
from conftest import example_bytes
from flo_amd import cli
from flo_amd.wav import WavError, read_wav_bytes, write_wav_bytes
from oracle import oracle as O


def _wav(tag, bits, ch, sr, payload, extensible=False):
    if extensible:
        fmt = struct.pack("<HHIIHH", 0xFFFE, ch, sr, sr * ch * bits // 8, ch * bits // 8, bits) + struct.pack("<HHI", 22, bits, 3) + struct.pack("<H", tag) + b"\x00" * 14
    else:
        fmt = struct.pack("<HHIIHH", tag, ch, sr, sr * ch * bits // 8, ch * bits // 8, bits)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 3) + b"abc\x00" + b"data" + struct.pack("<I", len(payload)) + payload
    return b"RIFF" + struct.pack("<I", len(body)) + body


def test_wav_scaling_rules_of_the_reference_cli():
    # reflo/src/audio.rs:238-275
    s16 = np.array([0, 1, -1, 32767, -32768, 12345], np.int16)
    x, sr, ch = read_wav_bytes(_wav(1, 16, 2, 48000, s16.tobytes()))
    assert (sr, ch) == (48000, 2) and x.dtype == np.float32
    assert np.array_equal(x, s16.astype(np.float32) * np.float32(1 / 32768.0))
    u8 = np.array([0, 128, 255, 200], np.uint8)
    x, _, _ = read_wav_bytes(_wav(1, 8, 1, 8000, u8.tobytes()))
    assert np.array_equal(x, (u8.astype(np.float32) - 128) / 128)
    s32 = np.array([0, 1 << 30, -(1 << 31), 2147483647], np.int32)
    x, _, _ = read_wav_bytes(_wav(1, 32, 1, 44100, s32.tobytes()))
    assert np.array_equal(x, s32.astype(np.float32) * np.float32(1 / 2147483648.0))
    s24 = [0, 1, -1, 8388607, -8388608]
    raw = b"".join(int(v & 0xFFFFFF).to_bytes(3, "little") for v in s24)
    x, _, _ = read_wav_bytes(_wav(1, 24, 1, 44100, raw))
    assert np.array_equal(x, (np.array(s24, np.int64) << 8).astype(np.float32) * np.float32(1 / 2147483648.0))
    f32 = np.array([0.0, 0.5, -1.0, 1e-8], np.float32)
    x, _, _ = read_wav_bytes(_wav(3, 32, 2, 96000, f32.tobytes(), extensible=True))
    assert np.array_equal(x, f32)
    with pytest.raises(WavError):
        read_wav_bytes(_wav(3, 64, 1, 44100, np.zeros(4).tobytes()))
    with pytest.raises(WavError):
        read_wav_bytes(b"RIFF1234WAVX")


def test_wav_writer_is_the_reference_writer():
    # the reference's own 1-second example: 32-bit float, 44-byte header (reflo/src/audio.rs:290-320)
    ref = example_bytes("audio.wav")
    x, sr, ch = read_wav_bytes(ref)
    assert (sr, ch, x.size) == (44100, 2, 88200) and not x.any()
    assert write_wav_bytes(x, sr, ch) == ref
    y = signals.music_like(22050, 1000, 1, seed=2)
    back, sr2, ch2 = read_wav_bytes(write_wav_bytes(y, 22050, 1))
    assert sr2 == 22050 and ch2 == 1 and np.array_equal(back, y)


def test_cli_info_and_validate_need_no_device(tmp_path, capsys):
    p = tmp_path / "a.flo"
    p.write_bytes(example_bytes("lossy_chord_high.flo"))
    assert cli.main(["info", str(p)]) == 0
    out = capsys.readouterr().out
    assert "Sample rate: 44100 Hz" in out and "Encoding:    Lossy (High)" in out and "CRC valid:   yes" in out
    assert cli.main(["validate", str(p)]) == 0
    bad = bytearray(example_bytes("chord_cmajor_stereo.flo"))
    bad[200] ^= 0x55
    q = tmp_path / "b.flo"
    q.write_bytes(bytes(bad))
    assert cli.main(["validate", str(q)]) == 1
    q.write_bytes(b"not a flo file at all")
    assert cli.main(["validate", str(q)]) == 1
    info = cli.flo_info(example_bytes("hires_96khz.flo"))
    assert info["sample_rate"] == 96000 and info["channels"] == 1 and not info["is_lossy"] and info["crc_valid"]
    assert abs(info["duration_secs"] - 1.0) < 1e-9


def test_oracle_streaming_encoder_matches_its_own_one_shot_frames():
    # the streaming encoder re-serialises what Encoder::encode puts into a one-frame file (encoder.rs:215-257):
    # header fields, TOC, timestamps and the frame layout [type][samples][flags] + [len][k][coeffs][residuals]
    sr, ch = 8000, 2
    pcm = signals.music_like(sr, 3 * sr + 1234, ch, seed=5)
    st = O.StreamingEncoder(sr, ch, 16, 5)
    st.push_samples(pcm[: 5000 * ch])
    assert st.pending_frames() == 0 and st.pending_samples() == 5000
    st.push_samples(pcm[5000 * ch:])
    assert st.pending_frames() == 3 and st.pending_samples() == 1234
    f0 = st.next_frame()
    assert (f0["index"], f0["timestamp_ms"], f0["samples"]) == (0, 0, sr) and st.pending_frames() == 2
    # frame 0 against the one-shot encoder's first frame
    one = flofile.parse(O.encode_lossless(pcm[: sr * ch], sr, ch, 16, 5))
    fr = one.frames[0]
    d = f0["data"]
    assert d[0] == fr.frame_type and int.from_bytes(d[1:5], "little") == sr and d[5] == fr.flags
    pos = 6
    for c in fr.channels:
        n = int.from_bytes(d[pos:pos + 4], "little")
        body = d[pos + 4:pos + 4 + n]
        pos += 4 + n
        if fr.frame_type in (253, 254):
            assert body == c.residuals
        else:
            assert body[0] == c.rice_k
            k = len(c.coeffs)
            assert np.array_equal(np.frombuffer(body[1:1 + 4 * k], "<i4"), np.array(c.coeffs, np.int32))
            assert body[1 + 4 * k:] == c.residuals
    assert pos == len(d)
    # finalize: the two queued frames plus the flushed remainder (frame 0 was pulled and is not in the file)
    out = st.finalize(b"META")
    assert out[:4] == b"FLO!" and out[-4:] == b"META"
    toc_size, data_size = int.from_bytes(out[38:46], "little"), int.from_bytes(out[46:54], "little")
    assert int.from_bytes(out[70:74], "little") == 3 and toc_size == 4 + 3 * 20
    ent = [struct.unpack_from("<IQII", out, 74 + 20 * i) for i in range(3)]
    assert [e[0] for e in ent] == [1, 2, 3] and [e[3] for e in ent] == [1000, 2000, 3000]
    assert int.from_bytes(out[14:22], "little") == 2 * sr + 1234            # total_samples
    assert O.crc32(out[70 + toc_size:70 + toc_size + data_size]) == int.from_bytes(out[26:30], "little")
    assert st.pending_frames() == 0 and st.finalize() [70:74] == b"\x00\x00\x00\x00"


def test_meta_chunk_of_every_reference_made_file_is_reproduced():
    # all 18 Examples/*.flo were written by the reference CLI (reflo/src/lib.rs:202-283): five MessagePack fields.
    # flo_amd/meta.py rebuilds each of them byte for byte from (length, source format, settings, the file's own time)
    import glob
    import os
    from conftest import ROOT
    from flo_amd import meta
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "examples", "*.flo")))
    assert len(files) == 18
    for f in files:
        b = open(f, "rb").read()
        ms = int.from_bytes(b[62:70], "little")
        chunk = b[len(b) - ms:]
        md = meta.unpack(chunk)
        assert list(md) == ["length_ms", "encoding_time", "encoder_settings", "flo_encoder_version", "source_format"], f
        sr, ch = int.from_bytes(b[8:12], "little"), b[12]
        lossy = bool(int.from_bytes(b[6:8], "little") & 1)
        # the source length is not in the file; length_ms is: any sample count that yields it will do
        n_sf = -(-md["length_ms"] * sr // 1000)
        while int(n_sf / sr * 1000.0) < md["length_ms"]:
            n_sf += 1
        q = None
        if lossy:
            q = float(md["encoder_settings"].split("quality ")[1].rstrip("%")) / 100.0
        rebuilt = meta.cli_metadata(n_sf * ch, sr, ch, md["source_format"], lossy, q if q is not None else 0.6, None, b[22],
                                    encoding_time=md["encoding_time"])
        assert rebuilt == chunk, (f, rebuilt, chunk)
    assert meta.encoder_settings(True, 0.6, 192, 5) == "Lossy, target 192kbps"
    assert meta.unpack(meta.pack_fields(dict(title="T" * 40, artist="A", album=None, length_ms=70000))) == {"title": "T" * 40, "artist": "A", "length_ms": 70000}


def test_merge_analysis_follows_serde_on_malformed_caller_metadata():
    # lib.rs:228: from_slice::<FloMetadata>(metadata).unwrap_or_default() - one field of the wrong type and ALL of the
    # caller's metadata is gone; empty sequences are skipped on the way out; unknown keys are dropped
    from flo_amd import meta
    an = (b"\x84" + meta._str("length_ms") + b"\xcd\x03\xe8" + meta._str("waveform_data") + b"\x80"
          + meta._str("spectrum_fingerprint") + b"\xc4\x01\x07" + meta._str("loudness_profile") + b"\x91\x80")
    names = lambda b: list(meta.unpack(b))      # noqa: E731
    ok = b"\x82" + meta._str("title") + meta._str("Song") + meta._str("track_number") + b"\x03"
    assert names(meta.merge_analysis(ok, an)) == ["title", "track_number", "length_ms", "waveform_data", "spectrum_fingerprint", "loudness_profile"]
    # title as an integer: serde rejects the document
    bad = b"\x82" + meta._str("title") + b"\x05" + meta._str("album") + meta._str("LP")
    assert names(meta.merge_analysis(bad, an)) == ["length_ms", "waveform_data", "spectrum_fingerprint", "loudness_profile"]
    # a negative track number, a float where a string belongs, a string where a sequence belongs
    for raw in (meta._str("track_number") + b"\xff", meta._str("artist") + b"\xca\x00\x00\x00\x00", meta._str("comments") + meta._str("x")):
        assert names(meta.merge_analysis(b"\x82" + meta._str("album") + meta._str("LP") + raw, an)) == \
            ["length_ms", "waveform_data", "spectrum_fingerprint", "loudness_profile"]
    # empty Vec fields disappear, an empty custom map too; an unknown key is dropped; nil is "not set"
    doc = (b"\x85" + meta._str("album") + meta._str("LP") + meta._str("comments") + b"\x90" + meta._str("custom") + b"\x80"
           + meta._str("not_a_field") + b"\x01" + meta._str("genre") + b"\xc0")
    assert names(meta.merge_analysis(doc, an)) == ["album", "length_ms", "waveform_data", "spectrum_fingerprint", "loudness_profile"]
    # a key that is not a string (here an array): rejected, not a crash
    assert names(meta.merge_analysis(b"\x81\x91\x01\x02", an)) == ["length_ms", "waveform_data", "spectrum_fingerprint", "loudness_profile"]
    # the caller's own analysis fields win, except an empty loudness profile
    own = b"\x82" + meta._str("waveform_data") + b"\x81" + meta._str("x") + b"\x01" + meta._str("loudness_profile") + b"\x90"
    m = meta.unpack(meta.merge_analysis(own, an))
    assert m["waveform_data"] == {"x": 1} and m["loudness_profile"] == [{}]
