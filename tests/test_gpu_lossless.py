"""Bit-exact parity of the HIP lossless (ALPC + Rice) path against the oracle and against the reference's own
fixture files. All calls go through the C ABI. Needs an MI355X."""
import numpy as np
import pytest

import flofile
import signals
from conftest import example_bytes
from fixtures_util import LOSSLESS_EXAMPLES, lossless_input_for
from gpu_util import ctx  # noqa: F401
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _same(ctx, pcm, sr, ch, level=5, bit_depth=16, meta=b""):
    g = ctx.encode_lossless(pcm, sr, ch, bit_depth, level, meta)
    o = O.encode_lossless(pcm, sr, ch, bit_depth, level, meta)
    if g != o:
        fg, fo = flofile.parse(g), flofile.parse(o)
        for i, (a, b) in enumerate(zip(fg.frames, fo.frames)):
            assert (a.frame_type, a.frame_samples, a.flags) == (b.frame_type, b.frame_samples, b.flags), i
            for c, (x, y) in enumerate(zip(a.channels, b.channels)):
                assert (x.coeffs, x.shift_bits, x.encoding, x.rice_k, len(x.raw)) == (y.coeffs, y.shift_bits, y.encoding, y.rice_k, len(y.raw)), (i, c)
                assert x.raw == y.raw, (i, c)
    assert g == o
    return g


@pytest.mark.parametrize("name", LOSSLESS_EXAMPLES + ["audio_lossless"])
def test_reference_fixture_files_are_reproduced(ctx, name):
    # SURVEY §8c-4 (and §8c-2 for audio.wav): bytes the real reference produced
    if name == "audio_lossless":
        ref = example_bytes("audio_lossless.flo")
        enc = ctx.encode_lossless(np.zeros(88200, np.float32), 44100, 2, 16, 5)
        assert enc[:62] == ref[:62] and enc[70:] == ref[70:len(ref) - 138]
        return
    ref, f32, ints, sr, ch = lossless_input_for(name)
    info = O.info(ref)
    enc = ctx.encode_lossless(f32, sr, ch, info.bit_depth, info.compression_level)
    body = len(ref) - info.meta_size
    assert enc[:62] == ref[:62]
    assert enc[70:] == ref[70:body]


@pytest.mark.parametrize("ch,n", [(1, 44100), (2, 44100), (2, 100000), (6, 9000), (1, 1), (2, 3), (1, 44099), (1, 44101), (2, 88201), (3, 50001)])
def test_byte_identical_to_oracle(ctx, ch, n):
    if ch <= 2:
        pcm = signals.music_like(44100, n, ch, seed=n)[: n * ch]
    else:
        pcm = np.stack([signals.sine(150.0 * (c + 1), 44100, n, 0.25) + signals.fast_noise(n, c, 0.01) for c in range(ch)], axis=1).reshape(-1)
    g = _same(ctx, pcm, 44100, ch)
    back, _, _ = O.decode_lossless_i32(g)
    want = np.array([O.f32_to_i32(x) for x in pcm[: min(pcm.size, 3000)]])
    assert (back[: want.size] == want).all()


@pytest.mark.parametrize("level", range(10))
def test_all_compression_levels(ctx, level):
    pcm = signals.music_like(44100, 30000, 2, seed=100 + level)
    _same(ctx, pcm, 44100, 2, level)


@pytest.mark.parametrize("sr", [8000, 22050, 48000, 96000, 192000])
def test_sample_rates(ctx, sr):
    pcm = signals.sine(440.0, sr, sr + sr // 3, 0.6) + signals.fast_noise(sr + sr // 3, 1, 0.003)
    _same(ctx, pcm.astype(np.float32), sr, 1)


def test_config5_hires_96k_stereo(ctx):
    # BASELINE config 5 ("96 kHz hi-res stereo ... bit-exact vs CPU"): the fixture is mono, so add synthetic stereo
    n = 96000 * 3 + 777
    pcm = O.synth_clip(n, 2, clip_id=5)
    g = _same(ctx, pcm, 96000, 2)
    f = flofile.parse(g)
    assert len(f.frames) == 4 and f.frames[-1].frame_samples == 777


def test_mid_side_path(ctx):
    base = signals.music_like(44100, 50000, 1, seed=4)
    pcm = np.stack([base, base * np.float32(0.97)], axis=1).reshape(-1)
    f = flofile.parse(_same(ctx, pcm, 44100, 2))
    assert f.frames[0].flags == 1


def test_quirks_are_reproduced(ctx):
    _same(ctx, np.zeros(5000, np.float32), 44100, 2)                          # Silence frame
    _same(ctx, np.full(44100, 3e-5, np.float32), 44100, 1)                    # Raw-labelled Rice of zeros
    _same(ctx, signals.fast_noise(8000, 9, 0.5), 44100, 1)                    # Raw-labelled Rice (undecodable quirk)
    _same(ctx, signals.fast_noise(44100, 3, 1.0), 44100, 1)                   # true raw PCM fallback
    _same(ctx, np.zeros(0, np.float32), 44100, 2)                             # no frames at all
    x = signals.fast_noise(5000, 2)
    x[10], x[20], x[30] = np.nan, np.inf, -np.inf
    _same(ctx, x, 44100, 1)
    # mixed: one channel noise (raw inside an ALPC frame), one tonal
    n = 30000
    pcm = np.stack([signals.fast_noise(n, 1, 1.0), signals.sine(300.0, 44100, n, 0.4)], axis=1).reshape(-1)
    f = flofile.parse(_same(ctx, pcm, 44100, 2))
    assert f.frames[0].frame_type == 8 and f.frames[0].channels[0].encoding == 2


def test_large_amplitude_mid_channel_wraps_like_reference(ctx):
    # mid = L + R can exceed i16; `s as i16` wraps when raw wins (encoder.rs:223)
    n = 20000
    noise = signals.fast_noise(n, 5, 1.0)
    pcm = np.stack([noise, noise], axis=1).reshape(-1)     # side = 0 -> mid/side chosen, mid = 2*noise
    _same(ctx, pcm, 44100, 2)


def test_metadata_and_bit_depth_echo(ctx):
    pcm = signals.music_like(44100, 3000, 1, seed=1)
    g = _same(ctx, pcm, 44100, 1, level=7, bit_depth=24, meta=b"\x81\xa1k\xa1v")
    f = flofile.parse(g)
    assert f.bit_depth == 24 and f.level == 7 and f.meta == b"\x81\xa1k\xa1v"


def test_ragged_batch(ctx):
    import flo_amd
    lens = [0, 1, 44100 * 2, 12345 * 2, 100001, 88200 * 2 + 2]
    clips = [signals.music_like(44100, (n + 1) // 2, 2, seed=n)[:n] for n in lens]
    outs = ctx.encode_batch(flo_amd.MODE_LOSSLESS, clips, 44100, 2, 5)
    for c, o in zip(clips, outs):
        assert o == O.encode_lossless(c, 44100, 2, 16, 5)


def test_api_mirror_classes(ctx):
    import flo_amd
    pcm = signals.music_like(44100, 20000, 2, seed=8)
    enc = flo_amd.Encoder(44100, 2, 16, ctx=ctx).with_compression(5)
    assert enc.encode(pcm, b"m") == O.encode_lossless(pcm, 44100, 2, 16, 5, b"m")
    assert flo_amd.Encoder(44100, 2, 16, ctx=ctx).with_compression(99).compression_level == 9
    lo = flo_amd.LossyEncoder(44100, 2, 0.55, ctx=ctx).encode_to_flo(pcm)
    assert flofile.parse(lo).is_lossy


def test_lossless_roundtrip_at_scale(ctx):
    # size-independent property on a larger device-generated batch: decode(encode(x)) == f32_to_i32(x)
    import flo_amd
    n_clips, n_sf = 24, 5 * 44100 + 123
    b = flo_amd.Batch(ctx, flo_amd.MODE_LOSSLESS, [n_sf * 2] * n_clips, 44100, 2, 5)
    b.fill_synthetic(seed=0xF10A0D10, clip_id0=7)
    b.encode()
    b.sync()
    for i in (0, 11, 23):
        enc = b.fetch(i)
        pcm = O.synth_clip(n_sf, 2, 0xF10A0D10, 7 + i)
        back, _, _ = O.decode_lossless_i32(enc)
        want = np.trunc(np.clip(pcm * np.float32(32767.0), -32768, 32767)).astype(np.int32)
        assert (back == want).all()
        assert enc == O.encode_lossless(pcm, 44100, 2, 16, 5)
    b.close()


def test_finished_files_in_hbm_equal_the_oracle_files(ctx):
    # header, TOC and CRC32 are made on the device: the packed files must be the oracle's files byte for byte
    import torch
    import flo_amd
    sr = 44100
    clips = [signals.music_like(sr, n, 2, seed=n) for n in (0, 1, 30000, 44100, 100000)]
    b = flo_amd.Batch(ctx, flo_amd.MODE_LOSSLESS, [c.size for c in clips], sr, 2, 5)
    for i, c in enumerate(clips):
        b.upload(i, c)
    b.encode(0)
    b.sync()
    buf = torch.empty(b.data_bytes() + len(clips) * 256 + 1024, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    offs = b.pack_files(buf.data_ptr(), buf.numel())
    b.sync()
    host = buf.cpu().numpy()
    for i, c in enumerate(clips):
        want = O.encode_lossless(c, sr, 2, 16, 5)
        assert host[offs[i]:offs[i] + len(want)].tobytes() == want
        assert b.fetch(i, b"meta!") == O.encode_lossless(c, sr, 2, 16, 5, b"meta!")
    b.close()


@pytest.mark.parametrize("n_clips,seconds,sr", [(1250, 10, 44100), (64, 10, 96000), (3, 1, 8000)])
def test_full_size_batch_decodes_back_on_the_device(ctx, n_clips, seconds, sr):
    """encode -> flo_batch_decode without the payload leaving HBM: every sample of every clip comes back as
    f32_to_i32(x) / 32767 (audio_constants.rs:17-26), compared on the device; the first clip also against the oracle."""
    import torch
    import flo_amd
    ch = 2
    n_sf = seconds * sr + 17
    b = flo_amd.Batch(ctx, flo_amd.MODE_LOSSLESS, [n_sf * ch] * n_clips, sr, ch, 5)
    b.fill_synthetic(seed=0xF10A0D10, clip_id0=3)
    b.encode()
    b.sync()
    out = torch.full((n_clips * n_sf * ch,), 7.0, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    offs = b.decode_to(out.data_ptr(), out.numel())
    assert offs == [i * n_sf * ch for i in range(n_clips)]
    dec = out.view(n_clips, n_sf * ch)
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    step = 128
    for i0 in range(0, n_clips, step):
        k = min(step, n_clips - i0)
        src = torch.empty(k, n_sf * ch, dtype=torch.float32, device="cuda:0")
        for j in range(k):
            assert hip.hipMemcpy(src[j].data_ptr(), b.clip_device_ptr(i0 + j), n_sf * ch * 4, 3) == 0
        want = torch.trunc(torch.clamp(src * 32767.0, -32768.0, 32767.0)) * float(np.float32(1.0) / np.float32(32767.0))
        assert torch.equal(dec[i0:i0 + k], want), i0
    o_dec, _, _ = O.decode(b.fetch(0))
    assert np.array_equal(dec[0].cpu().numpy(), o_dec)
    b.close()


@pytest.mark.parametrize("level", [0, 2, 5, 8])
def test_batch_decode_equals_the_decode_of_each_fetched_file(ctx, level):
    """flo_batch_decode builds its work from the encoder's own records instead of parsing the files: for every kind of
    frame the encoder can emit (silence, raw, fixed, LPC, mid/side, a partial last frame, an empty clip, the level-0
    quirk) the result must be, bit for bit, what flo_decode makes of the fetched file."""
    import torch
    import flo_amd
    sr, ch = 44100, 2
    rng = np.random.default_rng(level)
    t = np.arange(sr) / sr
    tone = (0.4 * np.sin(2 * np.pi * 330 * t)).astype(np.float32)
    clips = [
        np.zeros(0, np.float32),
        np.zeros(sr * 2 * ch, np.float32),                                               # silence frames
        rng.uniform(-1, 1, sr * ch + 2 * 777).astype(np.float32),                         # noise: raw wins; partial frame
        np.stack([tone, tone * 0.98], 1).reshape(-1),                                    # mid/side
        signals.music_like(sr, 3 * sr + 1234, ch, seed=level + 1),
        np.concatenate([np.zeros(sr * ch, np.float32), signals.music_like(sr, 5000, ch, seed=9)]),
        np.zeros(0, np.float32),
    ]
    b = flo_amd.Batch(ctx, flo_amd.MODE_LOSSLESS, [c.size for c in clips], sr, ch, level)
    for i, c in enumerate(clips):
        if c.size:
            b.upload(i, c)
    b.encode()
    b.sync()
    total = sum((c.size // ch) * ch for c in clips)
    out = torch.full((total + 8,), 3.0, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    offs = b.decode_to(out.data_ptr(), out.numel())
    host = out.cpu().numpy()
    for i, c in enumerate(clips):
        want = ctx.decode(b.fetch(i))
        got = host[offs[i]:offs[i] + want.size]
        assert np.array_equal(got, want), (level, i)
    assert host[total] == 3.0       # nothing written behind the last clip
    b.close()
