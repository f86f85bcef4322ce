"""Pin the oracle against bytes the real reference produced."""
import numpy as np
import pytest

import flofile
from conftest import example_bytes
from fixtures_util import LOSSLESS_EXAMPLES, LOSSY_EXAMPLES, dequantise, lossless_input_for, lossy_source_pcm
from oracle import oracle as O


def _wav_f32(b):
    assert b[:4] == b"RIFF" and b[8:12] == b"WAVE"
    pos = 12
    fmt = None
    while pos < len(b):
        cid, sz = b[pos:pos + 4], int.from_bytes(b[pos + 4:pos + 8], "little")
        if cid == b"fmt ":
            fmt = b[pos + 8:pos + 8 + sz]
        if cid == b"data":
            tag = int.from_bytes(fmt[:2], "little")
            assert tag == 3
            return np.frombuffer(b[pos + 8:pos + 8 + sz], dtype="<f4"), int.from_bytes(fmt[4:8], "little"), int.from_bytes(fmt[2:4], "little")
        pos += 8 + sz + (sz & 1)
    raise AssertionError("no data chunk")


def test_config1_audio_wav_lossless_exact_bytes():
    # BASELINE config 1; SURVEY §8c-2 gives the exact bytes
    pcm, sr, ch = _wav_f32(example_bytes("audio.wav"))
    assert sr == 44100 and ch == 2 and pcm.size == 88200 and not pcm.any()
    enc = O.encode_lossless(pcm, sr, ch, 16, 5)
    ref = example_bytes("audio_lossless.flo")
    body = len(ref) - 138
    assert enc[:62] == ref[:62]                 # header up to meta_size
    assert enc[70:] == ref[70:body]             # TOC + DATA
    hdr = bytes.fromhex("464c4f210102000044ac0000021044ac00000000000005000000637b73d1"
                        "420000000000000018000000000000000e000000000000000000000000000000")
    assert enc[:62] == hdr
    assert enc[94:] == bytes.fromhex("0044ac0000000000000000000000")


def test_config1_audio_wav_lossy_data_chunk():
    # 45 identical frames of 126 B; DATA CRC 0x00CA7202; header flags 0x0201 at q=0.6
    pcm, sr, ch = _wav_f32(example_bytes("audio.wav"))
    enc = O.encode_lossy(pcm, sr, ch, 0.6)
    f = flofile.parse(enc)
    ref = flofile.parse(example_bytes("audio_lossy.flo"))
    assert f.data == ref.data and f.data_crc32 == 0x00CA7202 == ref.data_crc32
    assert f.flags == ref.flags == 0x0201 and f.total_samples == 46080 and len(f.frames) == 45
    one = bytes.fromhex("fd0004000000" "74000000" "0002") + bytes.fromhex("0080") * 50 + bytes.fromhex("03000000800800") * 2
    assert f.data == one * 45
    # current writer timestamps (writer.rs:216), unlike the stale file
    assert [t[3] for t in f.toc[:3]] == [0, 23, 46]


@pytest.mark.parametrize("name", LOSSLESS_EXAMPLES)
def test_lossless_fixture_reencode_byte_identical(name):
    ref, f32, ints, sr, ch = lossless_input_for(name)
    info = O.info(ref)
    assert info.crc_computed == info.data_crc32
    enc = O.encode_lossless(f32, sr, ch, info.bit_depth, info.compression_level)
    body = len(ref) - info.meta_size
    assert enc[:62] == ref[:62]
    assert enc[70:] == ref[70:body]
    # and the oracle decoder inverts it
    back, _, _ = O.decode_lossless_i32(enc)
    assert (back == ints).all()


def test_lossless_fixture_structure_known_facts():
    # SURVEY §4 fixture notes: orders/k chosen by the reference encoder
    f = flofile.parse(example_bytes("chord_cmajor_stereo.flo"))
    assert [c.shift_bits for c in f.frames[0].channels] == [131, 131] and f.frames[0].channels[0].rice_k == 4
    assert f.frames[0].frame_type == 8 and f.frames[0].flags == 0
    g = flofile.parse(example_bytes("sweep_20_20k.flo"))
    assert g.frames[4].frame_type == 254 and len(g.frames[4].channels[0].raw) == 88200
    h = flofile.parse(example_bytes("hires_96khz.flo"))
    assert h.sample_rate == 96000 and h.channels == 1 and h.frames[0].frame_samples == 96000
    assert h.frames[0].channels[0].shift_bits == 132 and h.frames[0].channels[0].rice_k == 4
    assert len(h.frames[0].channels[0].residuals) == 60847
    s = flofile.parse(example_bytes("silence_1sec.flo"))
    assert s.frames[0].frame_type == 254 and s.frames[0].channels[0].raw == bytes(5513)


@pytest.mark.parametrize("name,q,src", LOSSY_EXAMPLES)
def test_lossy_near_goldens(name, q, src):
    pcm, sr, ch = lossy_source_pcm(src)
    ref = flofile.parse(example_bytes(name + ".flo"))
    assert len(ref.frames) == 88 and ref.total_samples == 90112 and ref.crc_valid
    RQ = np.zeros((88, ch, 1024), np.int16)
    RS = np.zeros((88, ch, 25), np.uint16)
    for i, fr in enumerate(ref.frames):
        assert fr.frame_type == 253 and fr.frame_samples == 1024
        n, sfw, qs = flofile.parse_transform_blob(fr.channels[0].raw)
        assert n == ch
        RQ[i], RS[i] = qs, sfw
    a = O.lossy_analyze(pcm, sr, ch, q)
    assert a["q"].shape == RQ.shape
    flips = int(((a["q"] != 0) != (RQ != 0)).sum())
    band = O.psy_tables(sr)[1]
    d_o, d_r = dequantise(a["q"], a["sf_words"], band), dequantise(RQ, RS, band)
    rel = float(np.sqrt(((d_o - d_r) ** 2).sum() / (d_r ** 2).sum()))
    assert rel <= 1e-5, rel
    if q <= 0.8:
        assert flips <= 1e-4 * RQ.size, flips      # survey transcription and this oracle both get 0
    enc = flofile.parse(O.encode_lossy(pcm, sr, ch, q))
    assert enc.flags == ref.flags and enc.total_samples == ref.total_samples and len(enc.frames) == 88
    assert enc.crc_valid
    if q <= 0.8:
        assert enc.data_size == ref.data_size
