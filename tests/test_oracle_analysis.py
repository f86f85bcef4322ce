"""Oracle restatement of the analysis metadata (oracle/analysis.c): what can be pinned without the reference binary.
BLAKE3 against the published known answers; the MessagePack framing against the format specification and the field
order of core/metadata.rs; waveform peaks / R128 / fingerprint against independent numpy evaluations of the formulas.
No reference-made file carries analysis metadata, so the numbers themselves stay "parity unpinned" (oracle/analysis.c)."""
import numpy as np

import signals
from flo_amd import meta
from oracle import oracle as O


def test_blake3_known_answers():
    # BLAKE3 specification test vectors: "", "abc", and inputs of bytes i % 251 that cross chunk boundaries
    assert O.blake3(b"").hex() == "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"
    assert O.blake3(b"abc").hex() == "6437b3ac38465133ffb63b75273a8db548c558465d79db03fd359c6cd5bd9d85"
    vec = lambda n: bytes(i % 251 for i in range(n))      # noqa: E731
    assert O.blake3(vec(1024)).hex() == "42214739f095a406f3fc83deb889744ac00df831c10daa55189b5d121c855af7"
    assert O.blake3(vec(1025)).hex() == "d00278ae47eb27b34faecf67b4fe263f82d5412916c1ffd97c8cb7fb814b8444"
    assert O.blake3(vec(2049)).hex() == "5f4d72f40d7a5f82b15ca2b2e44b1de3c2ef86c426c95c1af0b6879522563030"


def test_waveform_peaks_formula():
    # analysis.rs:38-115 evaluated independently in numpy
    for ch, n_sf, sr in ((1, 5000, 8000), (2, 44100 + 333, 44100), (3, 7001, 22050)):
        pcm = signals.music_like(sr, n_sf, ch, seed=ch)
        got = O.waveform_peaks(pcm, ch, sr, 50)
        spp = sr / 50.0
        total = int(np.ceil(pcm.size / (spp * ch)))
        exp = []
        for i in range(total):
            a, b = int(i * spp) * ch, min(int((i + 1) * spp) * ch, pcm.size)
            if a >= pcm.size:
                break
            w = pcm[a:b]
            if ch == 1:
                exp.append(np.abs(w).max(initial=0.0))
            elif ch == 2:
                w = w[: w.size // 2 * 2].reshape(-1, 2)
                exp.append((np.float32(np.abs(w[:, 0]).max(initial=0.0)) + np.float32(np.abs(w[:, 1]).max(initial=0.0))) / np.float32(2.0))
            else:
                rows = [w[k:k + ch] for k in range(0, w.size, ch)]
                v = np.float32(0.0)
                for r in rows:
                    s = np.float32(0.0)
                    for x in r:
                        s = np.float32(s + x)
                    v = max(v, np.float32(s / np.float32(len(r))))
                exp.append(v)
        exp = np.array(exp, np.float32)
        exp = exp / exp.max() if exp.max() > 0 else exp
        assert got.shape == exp.shape and np.array_equal(got, exp.astype(np.float32)), ch
    assert O.waveform_peaks(np.zeros(0, np.float32), 2, 44100).size == 0


def test_integrated_loudness_against_a_numpy_r128():
    # ebu_r128.rs:182-318 (K-weighting, 400 ms blocks, -70 LUFS and -10 LU gates) evaluated with scipy's lfilter
    from scipy.signal import lfilter
    sr = 48000
    t = np.arange(3 * sr) / sr
    pcm = np.stack([0.25 * np.sin(2 * np.pi * 997 * t), 0.1 * np.sin(2 * np.pi * 220 * t)], axis=1).astype(np.float32)
    pcm[sr:2 * sr] *= 0.001          # a quiet second that the relative gate removes
    got = O.integrated_lufs(pcm, 2, sr)
    import ctypes as C
    L = O.lib()
    sh, hp = (C.c_double * 5)(), (C.c_double * 5)()
    L.flo_o_kweighting_coeffs.argtypes = [C.c_double, C.c_void_p, C.c_void_p]
    L.flo_o_kweighting_coeffs(float(sr), sh, hp)
    kw = []
    for c in range(2):
        y = lfilter(list(sh[:3]), [1.0, sh[3], sh[4]], pcm[:, c].astype(np.float64))
        kw.append(lfilter(list(hp[:3]), [1.0, hp[3], hp[4]], y))
    hop, frames = round(sr * 0.1), pcm.shape[0]
    en, start = [], 0
    while start < frames:
        end = min(start + 4 * hop, frames)
        en.append(sum(np.mean(k[start:end] ** 2) for k in kw))
        if end == frames:
            break
        start += hop
    en = np.array(en)
    g = en[en >= 10 ** ((-70 + 0.691) / 10)]
    ung = -0.691 + 10 * np.log10(g.mean())
    f = g[g >= 10 ** ((ung - 10 + 0.691) / 10)]
    exp = -0.691 + 10 * np.log10(f.mean())
    assert abs(got - exp) < 1e-6 and len(f) < len(g)
    assert O.integrated_lufs(np.zeros(96000, np.float32), 2, 48000) == -23.0        # nothing passes the absolute gate
    # a 997 Hz sine at -20 dBFS in one channel measures about -23 LUFS (K-weighting is ~ +0.7 dB there, RMS -3 dB)
    s = (0.1 * np.sin(2 * np.pi * 997 * t)).astype(np.float32)
    assert abs(O.integrated_lufs(s, 1, sr) - (-23.0)) < 0.2


def test_fingerprint_fields():
    sr, ch = 44100, 2
    t = np.arange(2 * sr) / sr
    tone = (0.5 * np.sin(2 * np.pi * 5000.0 * t)).astype(np.float32)          # 5 kHz: FFT bin 29 of 256 -> band 3 of 16, 1 of 8
    pcm = np.stack([tone, tone], axis=1).reshape(-1)
    fp = O.spectral_fingerprint(pcm, ch, sr)
    assert fp["duration_ms"] == 2000 and fp["sample_rate"] == sr and fp["channels"] == ch
    assert int(np.argmax(fp["energy_profile"])) == 3 and fp["energy_profile"][3] == 255
    assert fp["frequency_peaks"][1] == int(np.float32(29) / np.float32(256) * np.float32(255))
    assert fp["avg_loudness"] == 60                                                 # rms <= 1: the clamp leaves 0 + 60
    head = bytes([ch]) + sr.to_bytes(4, "little") + (pcm.size & 0xFFFFFFFF).to_bytes(4, "little")
    assert fp["hash"] == O.blake3(head + pcm.tobytes())
    z = O.spectral_fingerprint(np.zeros(0, np.float32), 2, 44100)
    assert z["hash"] == bytes(32) and z["duration_ms"] == 0


def test_meta_layout_and_merge():
    pcm = signals.music_like(22050, 30000, 2, seed=6)
    m = O.analysis_metadata(pcm, 22050, 2, 50)
    d = meta.unpack(m)
    assert list(d) == ["length_ms", "waveform_data", "spectrum_fingerprint", "loudness_profile"]      # metadata.rs:442,589,594,607
    assert d["length_ms"] == int(30000 / 22050 * 1000.0)
    assert list(d["waveform_data"]) == ["peaks_per_second", "peaks", "channels"] and d["waveform_data"]["peaks_per_second"] == 50
    assert np.array_equal(np.array(d["waveform_data"]["peaks"], np.float32), O.waveform_peaks(pcm, 2, 22050, 50))
    fp = meta.unpack(d["spectrum_fingerprint"])
    assert list(fp) == ["hash", "duration_ms", "sample_rate", "channels", "frequency_peaks", "energy_profile", "avg_loudness"]
    assert bytes(fp["hash"]) == O.spectral_fingerprint(pcm, 2, 22050)["hash"] and len(fp["energy_profile"]) == 16
    lp = d["loudness_profile"]
    assert len(lp) == 1 and lp[0]["timestamp_ms"] == 0 and lp[0]["lufs"] == np.float32(O.integrated_lufs(pcm, 2, 22050))
    # a caller's own metadata: its fields stay, in declaration order; analysis fields fill the gaps; length_ms is set
    user = meta.pack_fields(dict(title="T", artist="A", length_ms=5))
    merged = meta.unpack(meta.merge_analysis(user, m))
    assert list(merged) == ["title", "artist", "length_ms", "waveform_data", "spectrum_fingerprint", "loudness_profile"]
    assert merged["title"] == "T" and merged["length_ms"] == d["length_ms"]
    assert meta.merge_analysis(b"", m) == m and meta.merge_analysis(b"\\xff garbage", m) == m


# ---- the reference's own property checks (tests/analysis_cases.py), applied to the oracle ------------------------------
import pytest  # noqa: E402
import analysis_cases as AC  # noqa: E402


@pytest.mark.parametrize("name,pcm,ch,sr,check", list(AC.loudness_cases()), ids=[c[0] for c in AC.loudness_cases()])
def test_reference_loudness_bars_hold_for_the_oracle(name, pcm, ch, sr, check):
    m = O.loudness_metrics(pcm, ch, sr)
    check(m)
    assert m == O.loudness_metrics(pcm, ch, sr)                       # loudness_tests.rs:167-177
    assert m["integrated_lufs"] == O.integrated_lufs(pcm, ch, sr)     # the META path's own function agrees


def test_reference_waveform_and_fingerprint_bars_hold_for_the_oracle():
    AC.check_waveform_peaks(lambda s, ch, sr, pps: O.waveform_peaks(s, ch, sr, pps))

    def fp(s, ch, sr):
        d = dict(O.spectral_fingerprint(s, ch, sr))
        d.setdefault("channels", ch)
        d.setdefault("sample_rate", sr)
        return d
    AC.check_fingerprint(fp)
