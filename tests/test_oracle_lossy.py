"""Oracle lossy-path checks mirroring libflo/tests/rust/lossy_*_tests.rs."""
import numpy as np

import flofile
import signals
from oracle import oracle as O


def test_windows():
    # lossy_mdct_tests.rs:6-26,49-58
    for kind in ("vorbis", "sine"):
        w = O.window(2048, kind)
        assert np.allclose(w, w[::-1], atol=1e-6) and w.min() >= 0 and w.max() <= 1
    w = O.window(2048, "sine").astype(np.float64)
    assert np.allclose(w[:1024] ** 2 + w[1024:] ** 2, 1.0, atol=1e-6)     # Princen-Bradley
    v = O.window(2048, "vorbis").astype(np.float64)
    assert np.allclose(v[:1024] ** 2 + v[1024:] ** 2, 1.0, atol=1e-6)


def test_mdct_matches_direct_definition():
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, 2048).astype(np.float32)
    for kind in ("vorbis", "sine"):
        fast = O.mdct_forward(x, kind).astype(np.float64)
        direct = O.mdct_forward_direct_f64(x, kind)
        assert np.abs(fast - direct).max() <= 2e-5 * np.abs(direct).max()


def test_mdct_magnitudes():
    # 0.5-amp 440 Hz sine peaks ~263; ones -> ~719 at DC (no 1/N scaling)
    s = signals.sine(440.0, 44100, 2048, 0.5)
    assert 200 < np.abs(O.mdct_forward(s)).max() < 330
    assert 600 < abs(O.mdct_forward(np.ones(2048, np.float32))[0]) < 800


def test_fft_mdct_tdac_perfect_reconstruction():
    # lossy_mdct_tests.rs:188-231 (sine window, MSE < 1e-10)
    i = np.arange(3072, dtype=np.float32)
    sig = np.sin(np.float32(2 * np.pi) * i / np.float32(64.0)).astype(np.float32)
    for kind in ("sine", "vorbis"):
        r1 = O.mdct_inverse(O.mdct_forward(sig[:2048], kind), kind)
        r2 = O.mdct_inverse(O.mdct_forward(sig[1024:3072], kind), kind)
        rec = r1[1024:] + r2[:1024]
        assert float(np.mean((rec - sig[1024:2048]) ** 2)) < 1e-10


def test_short_block_256():
    rng = np.random.default_rng(2)
    x = rng.uniform(-1, 1, 256).astype(np.float32)
    assert np.abs(O.mdct_forward(x).astype(np.float64) - O.mdct_forward_direct_f64(x)).max() < 1e-4


def test_psychoacoustic_tables():
    # lossy_psychoacoustic_tests.rs:5-46
    L = O.lib()
    assert L.flo_o_ath(1000.0) < L.flo_o_ath(100.0) and L.flo_o_ath(1000.0) < L.flo_o_ath(15000.0)
    assert L.flo_o_ath(10.0) == 96.0 and L.flo_o_ath(20001.0) == 96.0
    assert abs(L.flo_o_freq_to_bark(500.0) - 5.0) < 1.0 and abs(L.flo_o_freq_to_bark(1000.0) - 8.5) < 1.0
    ath, band, spreading = O.psy_tables(44100)
    assert band[0] == 0 and band[-1] >= 20 and (np.diff(band.astype(int)) >= 0).all()
    counts = np.bincount(band, minlength=25).tolist()
    assert counts == [5, 4, 5, 5, 5, 5, 7, 7, 7, 9, 10, 11, 13, 15, 17, 21, 26, 32, 42, 51, 61, 83, 116, 163, 304]
    assert np.bincount(O.psy_tables(96000)[1], minlength=25).tolist()[-2:] == [75, 693]
    assert (spreading[np.tril_indices(25)] == 1.0).all()          # lower bands get full masking (quirk)
    assert np.isclose(spreading[0, 1], 10 ** -2.5, rtol=1e-5) and spreading[0, 24] == 0.0


def test_masking_threshold_near_masker():
    # lossy_psychoacoustic_tests.rs:48-59 — through the analysis hook: a pure tone in one bin
    a = O.lossy_analyze(signals.sine(1000.0, 44100, 8192, 0.5), 44100, 1, 0.55)
    smr, c = a["smr"][3, 0], a["coeffs"][3, 0]
    thr = np.where(np.abs(c) > 1e-10, 20 * np.log10(np.maximum(np.abs(c), 1e-30)), -100.0) - smr
    tone_bin = int(round(1000.0 / (44100 / 2048)))
    assert thr[tone_bin + 1] > thr[tone_bin + 100]


def test_threshold_floor_quirk_prev_energy_zero():
    # prev_energy starts at 0.0 => band thresholds never go below 0 dB => thr >= -10
    a = O.lossy_analyze(signals.fast_noise(8192, 3, 1e-4), 44100, 1, 0.55)
    c, smr = a["coeffs"], a["smr"]
    sig = np.where(np.abs(c) > 1e-10, 20 * np.log10(np.maximum(np.abs(c), 1e-30)), -100.0)
    assert ((sig - smr) >= -10.0 - 1e-4).all()


def test_smr_threshold_values():
    # SURVEY §8a a8
    assert abs(O.lib().flo_o_smr_threshold(0.35) - -11.626) < 2e-3
    assert abs(O.lib().flo_o_smr_threshold(0.55) - -19.751) < 2e-3
    assert abs(O.lib().flo_o_smr_threshold(0.6) - -22.053) < 2e-3
    assert O.lib().flo_o_smr_threshold(0.99) == -100.0 and O.lib().flo_o_smr_threshold(1.0) == -100.0


def test_sparse_known_answers():
    # lossy_decoder_tests.rs:6-25, lossy_encoder_tests.rs:5-13
    q = np.array([0, 0, 0, 100, 0, 0, 0, 0, -50, 25, 0, 0], np.int16)
    enc = O.serialize_sparse(q)
    assert enc == bytes([3, 1, 100, 0, 4, 2, 0xCE, 0xFF, 25, 0, 2, 0]) and len(enc) < 20
    assert (O.deserialize_sparse(enc, 12) == q).all()
    z = O.serialize_sparse(np.zeros(1024, np.int16))
    assert z == bytes([0x80, 0x08, 0x00])
    # 255-cap on non-zero runs: 600 non-zeros -> records (0,255) (0,255) (0,90)
    d = np.arange(1, 601, dtype=np.int16)
    e = O.serialize_sparse(d)
    assert e[0] == 0 and e[1] == 255 and e[2 + 510] == 0 and e[3 + 510] == 255 and len(e) == 3 * 2 + 1200
    assert (O.deserialize_sparse(e, 600) == d).all()
    rng = np.random.default_rng(5)
    for dens in (0.01, 0.2, 0.9, 1.0):
        x = (rng.integers(-3000, 3000, 1024) * (rng.uniform(size=1024) < dens)).astype(np.int16)
        assert (O.deserialize_sparse(O.serialize_sparse(x)) == x).all()


def test_scale_factor_word():
    L = O.lib()
    assert L.flo_o_scale_factor_word(1.0) == 32768 and L.flo_o_scale_factor_word(0.0) == 0
    assert L.flo_o_scale_factor_word(2.0) == 32768 + 256 and L.flo_o_scale_factor_word(1e-11) == 0
    assert L.flo_o_scale_factor_word(3e38) == 65535 or L.flo_o_scale_factor_word(3e38) == 32768 + int(np.log2(3e38) * 256)


def test_lossy_header_and_quality_levels():
    # seeking_integration_tests.rs:69-92 / lossy_quality_tests.rs
    pcm = signals.sine(440.0, 44100, 4410, 0.5, channels=2)
    for level, q in enumerate([0.0, 0.35, 0.55, 0.75, 1.0]):
        f = flofile.parse(O.encode_lossy(pcm, 44100, 2, q))
        expect = min(int(np.floor(q * 4 + 0.5)), 4)
        assert f.is_lossy and f.lossy_quality == expect and f.bit_depth == 16 and f.level == 5 and f.crc_valid
        assert len(f.frames) == -(-(4410 + 1024) // 1024) and f.total_samples == 1024 * len(f.frames)
        assert [t[3] for t in f.toc] == [i * 1024 * 1000 // 44100 for i in range(len(f.frames))]


def test_sine_decode_quality_snr():
    # lossy_transform_tests.rs:117-185 (q=0.75, SNR > 10 dB); decoded length = (hops-1)*1024 >= n
    orig = signals.sine(440.0, 44100, 44100, 0.5)
    dec, sr, ch = O.decode(O.encode_lossy(orig, 44100, 1, 0.75))
    assert sr == 44100 and ch == 1 and dec.size == (O.lib().flo_o_lossy_num_hops(44100, 1) - 1) * 1024 >= 44100
    d = dec[:44100].astype(np.float64)
    snr = 10 * np.log10((orig.astype(np.float64) ** 2).sum() / ((orig - d) ** 2).sum())
    assert snr > 10.0
    # at q = 1.0 everything is kept: reconstruction is much closer and stays within [-1, 1]
    d1, _, _ = O.decode(O.encode_lossy(orig, 44100, 1, 1.0))
    snr1 = 10 * np.log10((orig.astype(np.float64) ** 2).sum() / ((orig - d1[:44100]) ** 2).sum())
    assert snr1 > 40.0 and np.abs(d1).max() <= 1.0


def test_silence_decodes_to_silence():
    dec, _, _ = O.decode(O.encode_lossy(np.zeros(20000, np.float32), 44100, 2, 0.55))
    assert np.sqrt(np.mean(dec.astype(np.float64) ** 2)) < 0.01


def test_lossy_edge_inputs():
    for n in (0, 1, 2, 1023, 1024, 1025, 2048):
        f = flofile.parse(O.encode_lossy(signals.fast_noise(n, 1), 44100, 1, 0.55))
        assert len(f.frames) == (n + 1024 + 1023) // 1024 and f.crc_valid
    # NaN / inf must not crash (edge_case_tests.rs:425-459)
    x = signals.fast_noise(4096, 2)
    x[100], x[200], x[300] = np.nan, np.inf, -np.inf
    assert flofile.parse(O.encode_lossy(x, 44100, 2, 0.55)).crc_valid


def test_hand_made_transform_files_parse_and_decode():
    """tests/flofile.build_transform (the writer of the GPU decode tests' hand-made record chains) makes files the
    independent parser and the oracle decoder both accept, with the coefficients that were put in."""
    import flofile
    sfw = [32768 + 256 * 3] * 25
    vals = np.array([100, -200, 300], dtype="<i2")
    blob = flofile.encode_varint(5) + bytes([3]) + vals.tobytes() + flofile.encode_varint(200) + bytes([1]) + np.array([7], "<i2").tobytes()
    flo = flofile.build_transform(44100, 2, [[(sfw, blob), (sfw, b"")]] * 3)
    f = flofile.parse(flo)
    assert f.crc_valid and f.is_lossy and len(f.frames) == 3 and f.total_samples == 3 * 1024
    d0 = 70 + f.toc_size
    nch, sfw_back, qs = flofile.parse_transform_blob(flo[d0 + 10:d0 + f.toc[0][2]])
    assert nch == 2 and list(sfw_back[0]) == sfw
    assert list(qs[0, 5:8]) == [100, -200, 300] and qs[0, 208] == 7 and int(np.count_nonzero(qs)) == 4
    pcm, sr, ch = O.decode(flo)
    assert sr == 44100 and ch == 2 and pcm.size == 2 * 2048 and float(np.max(np.abs(pcm[0::2]))) > 0.01
    assert not np.any(pcm[1::2])
