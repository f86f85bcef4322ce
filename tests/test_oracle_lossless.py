"""Oracle lossless round trips and edge cases (lossless_decoder_tests.rs, edge_case_tests.rs, integration_tests.rs)."""
import numpy as np
import pytest

import flofile
import signals
from oracle import oracle as O


def _roundtrip(pcm, sr, ch, level=5, bit_depth=16):
    enc = O.encode_lossless(pcm, sr, ch, bit_depth, level)
    f = flofile.parse(enc)
    assert f.crc_valid and f.version == (1, 2) and f.bit_depth == bit_depth and f.level == min(level, 9)
    assert f.total_samples == pcm.size // ch
    dec, sr2, ch2 = O.decode(enc)
    assert (sr2, ch2) == (sr, ch) and dec.size == (pcm.size // ch) * ch
    return f, dec


@pytest.mark.parametrize("ch,n", [(1, 44100), (2, 44100), (6, 8000), (1, 1), (2, 3), (1, 44099), (1, 44101), (2, 88201)])
def test_roundtrip_precision_and_length(ch, n):
    # lossless_decoder_tests.rs:21-110: max error <= 1/32768 + 1e-6, exact length
    if ch <= 2:
        pcm = signals.music_like(44100, n, ch, seed=n)
    else:
        pcm = np.stack([signals.sine(200.0 * (c + 1), 44100, n, 0.3) for c in range(ch)], axis=1).reshape(-1)
    pcm = pcm[: n * ch]
    f, dec = _roundtrip(pcm, 44100, ch)
    assert np.abs(dec - pcm[: dec.size]).max() <= 1 / 32768 + 1e-6
    assert len(f.frames) == -(-n // 44100)


@pytest.mark.parametrize("sr", [8000, 22050, 48000, 96000, 192000])
def test_roundtrip_sample_rates(sr):
    pcm = signals.sine(440.0, sr, sr // 2 + 17, 0.7)
    f, dec = _roundtrip(pcm, sr, 1)
    assert f.sample_rate == sr and np.abs(dec - pcm).max() <= 1 / 32768 + 1e-6


@pytest.mark.parametrize("level", range(0, 10))
def test_levels_are_integer_exact(level):
    pcm = signals.music_like(44100, 30000, 2, seed=level)
    ints = np.array([O.f32_to_i32(x) for x in pcm[:4000]])
    enc = O.encode_lossless(pcm, 44100, 2, 16, level)
    f = flofile.parse(enc)
    if level == 0:
        # max order 0 => every channel has order_used == 0 => Raw label even when Rice won (reference quirk):
        # such files are not decodable by the reference decoder either; only the bytes are pinned.
        assert all(fr.frame_type == 254 for fr in f.frames)
        return
    back, _, _ = O.decode_lossless_i32(enc)
    assert (back[:4000] == ints).all()
    maxo = [0, 2, 4, 4, 6, 8, 8, 10, 12, 12][level]
    for fr in f.frames:
        assert fr.frame_type in (254, maxo if 1 <= maxo <= 12 else 8)
        for c in fr.channels:
            if fr.frame_type != 254:
                assert len(c.coeffs) <= maxo and (len(c.coeffs) == 0 or (level >= 3 and len(c.coeffs) >= 5))


def test_mid_side_is_used_and_inverts():
    # no fixture exercises flags=1: pin by round trip
    base = signals.music_like(44100, 20000, 1, seed=4)
    pcm = np.stack([base, base * np.float32(0.98)], axis=1).reshape(-1)
    enc = O.encode_lossless(pcm, 44100, 2)
    f = flofile.parse(enc)
    assert f.frames[0].flags == 1
    back, _, _ = O.decode_lossless_i32(enc)
    assert (back == np.array([O.f32_to_i32(x) for x in pcm])).all()


def test_tonal_stereo_compresses_2x():
    # lossless_encoder_tests.rs:113-138
    pcm = signals.sine(440.0, 44100, 44100, 0.5, channels=2)
    assert pcm.size * 2 / len(O.encode_lossless(pcm, 44100, 2)) > 2.0


def test_silence_and_dither_quirks():
    f = flofile.parse(O.encode_lossless(np.zeros(1000, np.float32), 44100, 2))
    assert f.frames[0].frame_type == 0 and f.data == bytes([0]) + (500).to_bytes(4, "little") + bytes(1 + 8)
    g = flofile.parse(O.encode_lossless(np.full(44100, 3e-5, np.float32), 44100, 1))
    assert g.frames[0].frame_type == 254 and g.frames[0].channels[0].raw == bytes(5513)   # Raw-labelled Rice


def test_raw_labelled_rice_quirk_is_reproduced_not_fixed():
    # when every channel's winner is fixed order 0 + Rice, order_used == 0 for all channels, the
    # frame is labelled Raw (254) and the writer emits the Rice bytes with no parameters (writer.rs:266-269).
    # Half-scale uniform noise hits this: Rice(k) is ~3 % smaller than 2 B/sample, so the file is NOT decodable
    # by the reference's own decoder. The encoders here must produce the same bytes, not a "fixed" stream.
    pcm = signals.fast_noise(8000, 9, 0.5)
    f = flofile.parse(O.encode_lossless(pcm, 44100, 1))
    ints = np.array([O.f32_to_i32(x) for x in pcm], dtype=np.int32)
    k = O.estimate_rice_parameter_i32(ints)
    assert f.frames[0].frame_type == 254 and f.frames[0].channels[0].raw == O.rice_encode_i32(ints, k)
    assert len(f.frames[0].channels[0].raw) < 16000


def test_noise_falls_back_to_raw_pcm():
    pcm = signals.fast_noise(44100, 3, 1.0)
    f = flofile.parse(O.encode_lossless(pcm, 44100, 1))
    assert f.frames[0].frame_type == 254 and len(f.frames[0].channels[0].raw) == 88200


def test_nan_inf_do_not_crash():
    x = signals.fast_noise(5000, 2)
    x[10], x[20], x[30] = np.nan, np.inf, -np.inf
    dec, _, _ = O.decode(O.encode_lossless(x, 44100, 1))
    assert dec[10] == 0.0 and dec[20] == 1.0 and dec[30] == np.float32(-32768 / 32767)


def test_empty_input():
    f = flofile.parse(O.encode_lossless(np.zeros(0, np.float32), 44100, 2))
    assert len(f.frames) == 0 and f.total_samples == 0 and f.data_size == 0 and f.data_crc32 == 0


def test_metadata_is_appended_verbatim():
    enc = O.encode_lossless(signals.fast_noise(100, 1), 44100, 1, meta=b"\x81\xa5title\xa3abc")
    f = flofile.parse(enc)
    assert f.meta == b"\x81\xa5title\xa3abc" and enc.endswith(f.meta)


def test_reader_rejects_garbage():
    # edge_case_tests.rs:209-336
    good = O.encode_lossless(signals.fast_noise(3000, 1), 44100, 1)
    for bad in (b"", b"FLO", b"NOPE" + good[4:], good[:40], good[:80]):
        with pytest.raises(RuntimeError):
            O.decode(bad)
