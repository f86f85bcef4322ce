"""SURVEY 8f-3 on the device: the analysis metadata of libflo::encode* (lib.rs:219-283) against the oracle's
restatement, field by field and as META bytes (the order-bound sums keep the reference's order within a segment of
65 536 frames - clips up to that length are bit-equal - and agree to ~1e-15 beyond; the META bytes are equal throughout)."""
import numpy as np
import pytest

import flo_amd
import signals
from conftest import example_bytes
from flo_amd import meta
from flo_amd.wav import read_wav_bytes
from gpu_util import ctx  # noqa: F401
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(3)
    yield "stereo 3 s", signals.music_like(44100, 3 * 44100 + 17, 2, seed=1), 44100, 2
    yield "mono 8 kHz", signals.music_like(8000, 20000, 1, seed=2), 8000, 1
    yield "three channels", signals.music_like(22050, 30001, 3, seed=3), 22050, 3
    yield "six channels 96 kHz", signals.music_like(96000, 50000, 6, seed=4), 96000, 6
    yield "silence", np.zeros(88200, np.float32), 44100, 2
    yield "loud noise", (rng.uniform(-4, 4, 2 * 30000)).astype(np.float32), 48000, 2      # rms > 1: avg_loudness below 60
    yield "shorter than one FFT", signals.music_like(44100, 200, 2, seed=5), 44100, 2
    yield "one sample-frame", np.array([0.25, -0.5], np.float32), 44100, 2
    for n in (253, 254, 255, 509, 510, 511, 1021, 1022):      # 9 + 4 n around the 1 KiB chunk boundaries of the hash
        yield f"mono {n}", signals.music_like(16000, n, 1, seed=n), 16000, 1
    yield "nan and inf", np.array([0.1, np.nan, -np.inf, 0.2] * 3000, np.float32), 44100, 2


@pytest.mark.parametrize("name,pcm,sr,ch", list(_cases()), ids=[c[0] for c in _cases()])
def test_analysis_equals_the_oracle(ctx, name, pcm, sr, ch):
    a = ctx.analyze(pcm, sr, ch, 50)
    fp = O.spectral_fingerprint(pcm, ch, sr)
    assert np.array_equal(a["peaks"].view(np.uint32), O.waveform_peaks(pcm, ch, sr, 50).view(np.uint32))
    assert a["hash"] == fp["hash"]
    for k in ("duration_ms", "frequency_peaks", "energy_profile", "avg_loudness"):
        assert a[k] == fp[k], k
    lo = O.integrated_lufs(pcm, ch, sr)
    if pcm.size // ch <= 65536:     # one segment: the reference's own order of accumulation, bit for bit
        assert a["integrated_lufs"] == lo or (np.isnan(a["integrated_lufs"]) and np.isnan(lo))
    else:                           # segments with a filter warm-up (analysis_kernels.hip): equal to ~1e-15
        assert abs(a["integrated_lufs"] - lo) <= 1e-12 * abs(lo) or (np.isnan(a["integrated_lufs"]) and np.isnan(lo))
    assert ctx.analysis_metadata(pcm, sr, ch, 50) == O.analysis_metadata(pcm, sr, ch, 50)


def test_other_peak_rates_and_empty_input(ctx):
    pcm = signals.music_like(44100, 50000, 2, seed=9)
    for pps in (1, 10, 50, 200, 44100):
        assert ctx.analysis_metadata(pcm, 44100, 2, pps) == O.analysis_metadata(pcm, 44100, 2, pps)
    e = np.zeros(0, np.float32)
    assert ctx.analysis_metadata(e, 44100, 2, 50) == O.analysis_metadata(e, 44100, 2, 50)


def test_free_functions_equal_the_reference_pipeline(ctx):
    # libflo::encode / encode_lossy / encode_with_bitrate (lib.rs:97-206): analysis metadata, then the encoder
    pcm, sr, ch = read_wav_bytes(example_bytes("audio.wav"))
    m = O.analysis_metadata(pcm, sr, ch, 50)
    assert flo_amd.encode(pcm, sr, ch, 16) == O.encode_lossless(pcm, sr, ch, 16, 5, meta=m)
    pcm = signals.music_like(44100, 40000, 2, seed=11)
    m = O.analysis_metadata(pcm, 44100, 2, 50)
    assert flo_amd.encode(pcm, 44100, 2, 24) == O.encode_lossless(pcm, 44100, 2, 24, 5, meta=m)
    got = flo_amd.encode_lossy(pcm, 44100, 2, 16, 2)
    ref = O.encode_lossy(pcm, 44100, 2, 0.55, meta=m)
    assert got[-len(m):] == m and len(got) == len(ref)
    user = meta.pack_fields(dict(title="Song", album="LP"))
    got = flo_amd.encode_with_bitrate(pcm, 44100, 2, 16, 192, metadata=user)
    md = meta.unpack(got[len(got) - int.from_bytes(got[62:70], "little"):])
    assert list(md) == ["title", "album", "length_ms", "waveform_data", "spectrum_fingerprint", "loudness_profile"]


# ---- the reference's own property checks (tests/analysis_cases.py), applied to flo_analyze ----------------------------
import analysis_cases as AC  # noqa: E402


@pytest.mark.parametrize("name,pcm,ch,sr,check", list(AC.loudness_cases()), ids=[c[0] for c in AC.loudness_cases()])
def test_reference_loudness_bars_hold_on_the_device(ctx, name, pcm, ch, sr, check):
    a = ctx.analyze(pcm, sr, ch, 50)
    m = {k: a[k] for k in ("integrated_lufs", "loudness_range_lu", "true_peak_dbtp", "sample_peak_dbfs")}
    check(m)
    o = O.loudness_metrics(pcm, ch, sr)
    one_segment = pcm.size // max(ch, 1) <= 65536
    for k in m:
        if one_segment:
            assert m[k] == o[k], (k, m[k], o[k])     # one segment: the reference's own order of accumulation, bit for bit
        else:
            assert abs(m[k] - o[k]) <= 1e-9 * max(1.0, abs(o[k])), (k, m[k], o[k])


def test_reference_waveform_and_fingerprint_bars_hold_on_the_device(ctx):
    AC.check_waveform_peaks(lambda s, ch, sr, pps: ctx.analyze(s, sr, ch, pps)["peaks"])
    AC.check_fingerprint(lambda s, ch, sr: {k: v for k, v in ctx.analyze(s, sr, ch, 50).items() if k != "peaks"})


def test_long_clips_in_segments_agree_with_the_sequential_oracle(ctx):
    # beyond 65 536 frames the order-bound scans run in segments (filter warm-up, partial block sums): the loudness
    # agrees to ~1e-12, the META chunk (f32 loudness, u8 levels) byte for byte
    for sr, ch, secs in ((44100, 2, 12.3), (96000, 1, 5.0), (8000, 2, 40.0)):
        pcm = signals.music_like(sr, int(sr * secs), ch, seed=int(secs * 10))
        a = ctx.analyze(pcm, sr, ch, 50)
        o = O.loudness_metrics(pcm, ch, sr)
        for k in ("integrated_lufs", "loudness_range_lu", "true_peak_dbtp", "sample_peak_dbfs"):
            assert abs(a[k] - o[k]) <= 1e-9 * max(1.0, abs(o[k])), (sr, k, a[k], o[k])
        assert ctx.analysis_metadata(pcm, sr, ch, 50) == O.analysis_metadata(pcm, sr, ch, 50)


def _seq_sum_squares(x):
    """analysis.rs:338 literally: one f32 accumulator, sample after sample (numpy's cumsum is that recurrence)"""
    x = np.asarray(x, np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        return np.cumsum(x * x, dtype=np.float32)[-1] if x.size else np.float32(0)


def _sumsq_cases():
    rng = np.random.default_rng(11)
    yield "music 60 s stereo", signals.music_like(44100, 60 * 44100, 2, seed=5)
    yield "noise 2.7 M samples, odd length", (rng.uniform(-1, 1, 2_700_001) * 0.3).astype(np.float32)
    yield "constant 0.5: every addition a tie for a while", np.full(1_500_000, 0.5, np.float32)
    yield "constant 2^-7: exact powers of two", np.full(900_000, 2.0 ** -7, np.float32)
    q = np.concatenate([(rng.uniform(-1, 1, 1_000_000) * 1e-4), rng.uniform(-1, 1, 300_000) * 0.9, rng.uniform(-1, 1, 700_000) * 1e-3]).astype(np.float32)
    yield "quiet, loud, quiet", q
    yield "loud first", q[::-1].copy()
    t = (rng.uniform(-1, 1, 400_000) * 0.2).astype(np.float32)
    t[123_457] = np.inf
    yield "an infinite sample", t
    t2 = t.copy(); t2[123_457] = np.nan
    yield "a NaN sample", t2
    yield "denormal-small samples", (rng.uniform(-1, 1, 300_000) * 1e-20).astype(np.float32)
    big = (rng.uniform(-1, 1, 200_000) * 3e18).astype(np.float32)
    yield "squares that overflow the accumulator", big
    z = np.zeros(500_000, np.float32); z[400_000:] = 0.25
    yield "zeros, then a step", z
    g = (rng.standard_normal(1_200_000) * np.exp(rng.uniform(-12, 0, 1_200_000))).astype(np.float32)
    yield "terms across sixteen binades", g


@pytest.mark.parametrize("name,x", list(_sumsq_cases()), ids=[c[0] for c in _sumsq_cases()])
def test_sum_of_squares_is_the_sequential_f32_sum_at_any_length(ctx, name, x):
    # beyond one segment the device chains chunks of 1024 samples (integer increments per binade) instead of adding partial
    # sums: the accumulator must come out bit for bit as the reference's loop leaves it, and avg_loudness with it
    a = ctx.analyze(x, 44100, 2 if x.size % 2 == 0 else 1, 50)
    want = _seq_sum_squares(x)
    got = np.float32(a["sum_squares"])
    assert got.view(np.uint32) == want.view(np.uint32) or (np.isnan(got) and np.isnan(want)), (name, got, want)
    fp = O.spectral_fingerprint(x, 2 if x.size % 2 == 0 else 1, 44100)
    assert a["avg_loudness"] == fp["avg_loudness"]


def test_long_clip_loudness_two_pass_state_hand_over(ctx):
    # beyond 65 536 frames the K-weighting runs as two passes over 2048-frame segments with the filter state handed over
    # through the 4 x 4 transition matrix; 3 minutes, ragged length, several rates and channel counts
    for sr, ch, secs in ((44100, 2, 180.0), (48000, 1, 61.7), (8000, 3, 33.3), (192000, 2, 4.1)):
        pcm = signals.music_like(sr, int(sr * secs) + 13, ch, seed=int(secs))
        a = ctx.analyze(pcm, sr, ch, 50)
        o = O.loudness_metrics(pcm, ch, sr)
        for k in ("integrated_lufs", "loudness_range_lu", "true_peak_dbtp", "sample_peak_dbfs"):
            assert abs(a[k] - o[k]) <= 1e-9 * max(1.0, abs(o[k])), (sr, ch, k, a[k], o[k])
        assert ctx.analysis_metadata(pcm, sr, ch, 50) == O.analysis_metadata(pcm, sr, ch, 50)
