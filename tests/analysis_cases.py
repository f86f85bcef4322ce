"""The reference's own property checks for the analysis functions, restated once and applied twice: to the oracle (CPU
tests) and to flo_analyze on the device (GPU tests). Sources: libflo/tests/rust/loudness_tests.rs (all 16 tests),
analysis_tests.rs (the extract_waveform_peaks ones; extract_waveform_rms is not on this repository's path),
spectral_analysis_tests.rs (the extract_spectral_fingerprint ones; extract_dominant_frequencies and spectral_similarity
are not on the path). Signals are built as the Rust tests build them (f32 arithmetic where they use f32)."""
import numpy as np

F32 = np.float32


def sine(sr, n, freq, amp, ch=1):
    i = np.arange(n, dtype=np.float32)
    s = (F32(amp) * np.sin(F32(2.0) * F32(np.pi) * F32(freq) * i / F32(sr), dtype=np.float32)).astype(np.float32)
    return np.repeat(s, ch) if ch > 1 else s


def lcg_noise(n, amp):
    i = np.arange(n, dtype=np.uint64)
    seed = (i * np.uint64(1103515245) + np.uint64(12345)) & np.uint64(0x7FFFFFFF)
    r = seed.astype(np.float64) / float(2 ** 31 - 1)
    return (float(amp) * (r - 0.5) * 2.0).astype(np.float32)


def dynamic(sr):
    n = sr * 5
    t = np.arange(n, dtype=np.float32) / F32(sr)
    amp = np.where(t < 1, 0.1, np.where(t < 2, 0.3, np.where(t < 3, 0.7, np.where(t < 4, 0.2, 0.5)))).astype(np.float32)
    i = np.arange(n, dtype=np.float32)
    return (amp * np.sin(F32(2.0) * F32(np.pi) * F32(440.0) * i / F32(sr), dtype=np.float32)).astype(np.float32)


def loudness_cases():
    """(name, samples, channels, sample_rate, check) - check(m) asserts the reference test's bars on the metrics dict"""
    def eq_defaults(m):
        assert m["integrated_lufs"] == -23.0 and m["loudness_range_lu"] == 0.0
        assert m["true_peak_dbtp"] == -150.0 and m["sample_peak_dbfs"] == -150.0
    yield "empty", np.zeros(0, np.float32), 2, 44100, eq_defaults                       # loudness_tests.rs:4-12
    yield "silence", np.zeros(1000, np.float32), 2, 44100, eq_defaults                  # :15-23 (vec![0.0; 1000], stereo)

    def mono_sine(m):                                                                   # :26-40
        assert -25.0 < m["integrated_lufs"] < -5.0 and m["loudness_range_lu"] >= 0.0
        assert -7.0 < m["true_peak_dbtp"] < -5.0 and -7.0 < m["sample_peak_dbfs"] < -5.0
    yield "mono sine", sine(44100, 44100, 440.0, 0.5), 1, 44100, mono_sine

    def stereo_sine(m):                                                                 # :43-59
        assert -25.0 < m["integrated_lufs"] < -5.0 and -7.0 < m["true_peak_dbtp"] < -5.0
    yield "stereo sine", sine(44100, 88200, 440.0, 0.5, ch=2), 2, 44100, stereo_sine

    def noise(m):                                                                       # :62-80
        assert -40.0 < m["integrated_lufs"] < -10.0 and m["loudness_range_lu"] >= 0.0
        assert m["true_peak_dbtp"] <= 0.0 and m["sample_peak_dbfs"] <= 0.0
    yield "white noise", lcg_noise(88200, 0.1), 1, 44100, noise

    def rates(m):                                                                       # :83-97
        assert -25.0 < m["integrated_lufs"] < -5.0 and -15.0 < m["true_peak_dbtp"] < 0.0
    for sr in (22050, 44100, 48000, 96000):
        yield f"sine at {sr} Hz", sine(sr, sr, 440.0, 0.5), 1, sr, rates

    def chans(m):                                                                       # :100-117
        assert -35.0 < m["integrated_lufs"] < -5.0 and -15.0 < m["true_peak_dbtp"] < 0.0
    for ch in (1, 2, 4, 6):
        yield f"{ch} channels", sine(44100, 44100, 440.0, 0.3, ch=ch), ch, 44100, chans

    for amp_db in (-30.0, -20.0, -12.0, -6.0, -3.0, 0.0):                               # :120-138
        def amps(m, e=amp_db):
            assert abs(m["true_peak_dbtp"] - e) < 1.0 and e - 30.0 < m["integrated_lufs"] < e + 10.0
        amp = float(np.float32(10.0) ** np.float32(amp_db / 20.0))
        yield f"amplitude {amp_db:g} dB", sine(44100, 44100, 440.0, amp), 1, 44100, amps

    def dyn(m):                                                                         # :141-164
        assert -25.0 < m["integrated_lufs"] < -10.0 and m["loudness_range_lu"] > 2.0 and m["true_peak_dbtp"] < -3.0
    yield "dynamic content", dynamic(44100), 1, 44100, dyn

    yield "consistency", np.array([0.5, -0.3, 0.8, -0.2, 0.1, -0.9, 0.4, -0.6], np.float32), 1, 44100, lambda m: None   # :167-177

    def short(m):                                                                       # :180-187
        assert -150.0 < m["integrated_lufs"] < 0.0 and m["true_peak_dbtp"] <= 0.0 and m["sample_peak_dbfs"] <= 0.0
    yield "four samples", np.array([0.5, -0.3, 0.8, -0.2], np.float32), 1, 44100, short

    def accuracy(m):                                                                    # :190-212
        assert abs(m["sample_peak_dbfs"]) < 0.1 and m["true_peak_dbtp"] > -3.0
    t = np.arange(44100, dtype=np.float32) / F32(44100)
    yield "peak accuracy", np.sin(F32(2.0) * F32(np.pi) * F32(1000.0) * t, dtype=np.float32).astype(np.float32), 1, 44100, accuracy

    def gating(m):                                                                      # :215-227
        assert m["integrated_lufs"] <= -23.0 and m["loudness_range_lu"] == 0.0
    yield "below the absolute gate", sine(44100, 44100, 440.0, float(np.float32(10.0) ** np.float32(-4.0))), 1, 44100, gating


def check_waveform_peaks(peaks_fn):
    """analysis_tests.rs:4-45, :62-72; peaks_fn(samples, channels, sample_rate, peaks_per_second) -> f32 array"""
    s = np.array([0.5, -0.3, 0.8, -0.2, 0.1, -0.9], np.float32)
    for ch in (1, 2):
        p = peaks_fn(s, ch, 44100, 10)
        assert p.size > 0 and np.all((p >= 0.0) & (p <= 1.0))
        assert np.array_equal(p, peaks_fn(s, ch, 44100, 10))
    assert peaks_fn(np.zeros(0, np.float32), 1, 44100, 10).size == 0


def check_fingerprint(fp_fn):
    """spectral_analysis_tests.rs:6-76, :157-174, :228-248; fp_fn(samples, channels, sample_rate) -> dict"""
    i = np.arange(4410, dtype=np.float32)
    tone = (np.sin(i * F32(0.1), dtype=np.float32) * F32(0.5)).astype(np.float32)
    for ch in (1, 2):
        fp = fp_fn(np.repeat(tone, ch) if ch == 2 else tone, ch, 44100)
        assert fp["channels"] == ch and fp["sample_rate"] == 44100 and len(fp["hash"]) == 32
        assert len(fp["frequency_peaks"]) == 8 and len(fp["energy_profile"]) == 16 and 0 < fp["duration_ms"] < 1000
    e = fp_fn(np.zeros(0, np.float32), 1, 44100)
    assert e["duration_ms"] == 0 and e["hash"] == bytes(32) and list(e["frequency_peaks"]) == [0] * 8
    assert list(e["energy_profile"]) == [0] * 16 and e["avg_loudness"] == 0
    p2 = fp_fn((np.sin(np.arange(1024, dtype=np.float32) * F32(0.05), dtype=np.float32)).astype(np.float32), 1, 44100)
    assert p2["duration_ms"] > 0 and p2["hash"] != bytes(32)
    a, b = fp_fn(tone, 1, 44100), fp_fn(tone, 1, 44100)
    assert a == b
    other = fp_fn((np.sin(i * F32(0.3), dtype=np.float32) * F32(0.5)).astype(np.float32), 1, 44100)
    assert other["hash"] != a["hash"]
    for secs, tol in ((1.0, 50), (0.5, 25), (2.0, 50)):
        n = int(44100 * secs)
        fp = fp_fn((np.sin(np.arange(n, dtype=np.float32) * F32(0.01), dtype=np.float32)).astype(np.float32), 1, 44100)
        assert abs(int(fp["duration_ms"]) - int(secs * 1000)) < tol
