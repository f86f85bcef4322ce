"""Known-answer tests of the oracle's core pieces (mirrors libflo/tests/rust/core_*_tests.rs, lossless_lpc_tests.rs)."""
import zlib

import numpy as np

from oracle import oracle as O


def test_crc32_known_answers():
    # libflo/tests/rust/core_crc32_tests.rs:4-14
    assert O.crc32(b"123456789") == 0xCBF43926
    assert O.crc32(b"") == 0
    rng = np.random.default_rng(0)
    for n in (1, 7, 256, 4097):
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert O.crc32(d) == zlib.crc32(d) & 0xFFFFFFFF


def test_f32_to_i32_truncates_and_saturates():
    # core/audio_constants.rs:18-20
    assert O.f32_to_i32(0.5) == 16383            # 16383.5 truncates
    assert O.f32_to_i32(-0.5) == -16383
    assert O.f32_to_i32(1.0) == 32767
    assert O.f32_to_i32(-1.0) == -32767
    assert O.f32_to_i32(2.0) == 32767
    assert O.f32_to_i32(-2.0) == -32768
    assert O.f32_to_i32(float("nan")) == 0
    assert O.f32_to_i32(float("inf")) == 32767
    assert O.f32_to_i32(float("-inf")) == -32768
    assert O.f32_to_i32(3e-5) == 0


def test_rice_roundtrip_i32():
    # core_rice_tests.rs:18-26
    res = np.array([100, -200, 50, -10, 0, 150, -300], dtype=np.int32)
    k = O.estimate_rice_parameter_i32(res)
    enc = O.rice_encode_i32(res, k)
    assert (O.rice_decode_i32(enc, k, len(res)) == res).all()


def test_rice_bit_layout_msb_first():
    # one sample 5 -> zigzag 10; k=2 -> quotient 2 ("110"), remainder 2 ("10") => 11010 000
    assert O.rice_encode_i32([5], 2) == bytes([0b11010000])
    # -1 -> zigzag 1; k=0 -> "10" ; 0 -> "0"
    assert O.rice_encode_i32([-1, 0, 0], 0) == bytes([0b10000000])
    # 44100 zeros at k=0 -> one zero bit each -> ceil(44100/8) = 5513 bytes (silence_1sec.flo quirk)
    assert O.rice_encode_i32(np.zeros(44100, np.int32), 0) == bytes(5513)


def test_rice_parameter_rules():
    # rice.rs:29-69 and SURVEY Appendix A
    assert O.estimate_rice_parameter_i32([]) == 4
    assert O.estimate_rice_parameter_i32([0, 0, 0]) == 0
    assert O.estimate_rice_parameter_i32([1, 1, 1]) == 1          # mean 1 -> bitlen 1
    assert O.estimate_rice_parameter_i32([127]) == 7              # 2*127 = 254 <= 255; mean 127 -> 7 bits
    assert O.estimate_rice_parameter_i32([128] + [0] * 1000) == 1  # 256 -> 9 bits - 8
    assert O.estimate_rice_parameter_i32([1 << 22] + [0] * 10) == 15  # clamp
    small = O.estimate_rice_parameter_i32([0, 1, -1, 2, -2, 1, 0, -1])
    large = O.estimate_rice_parameter_i32([1000, -2000, 1500, -1800, 2200])
    assert large > small      # lossless_lpc_tests.rs:122-133


def test_rice_random_roundtrip_all_k():
    rng = np.random.default_rng(1)
    for scale in (1, 30, 1000, 30000):
        res = rng.integers(-scale, scale + 1, 3000).astype(np.int32)
        k = O.estimate_rice_parameter_i32(res)
        enc = O.rice_encode_i32(res, k)
        assert (O.rice_decode_i32(enc, k, len(res)) == res).all()
        nbits = sum(min(int(((int(v) << 1) ^ (int(v) >> 31)) & 0xFFFFFFFF) >> k, 255) + 1 + k for v in res)
        assert len(enc) == (nbits + 7) // 8


def test_fixed_predictors_known():
    # lossless_lpc_tests.rs:102-120
    s = [100, 200, 300, 400, 500]
    assert O.fixed_predictor_residuals(s, 0).tolist() == s
    assert O.fixed_predictor_residuals(s, 1).tolist() == [100, 100, 100, 100, 100]
    assert O.fixed_predictor_residuals(s, 2).tolist() == [100, 100, 0, 0, 0]
    assert O.fixed_predictor_residuals(s, 3).tolist() == [100, 100, 0, 0, 0]
    assert O.fixed_predictor_residuals(s, 4).tolist() == [100, 100, 0, 0, 0]
    t = [3, -1, 4, 1, -5, 9, 2, -6]
    r4 = O.fixed_predictor_residuals(t, 4).tolist()
    assert r4[:4] == [3, -4, 4 + 2 + 3, 1 - 12 - 3 - 3]
    assert r4[4] == -5 - 4 * 1 + 6 * 4 - 4 * -1 + 3


def test_autocorr_int():
    # lossless_lpc_tests.rs:90-99
    s = np.array([(i * 100) % 32767 for i in range(100)], dtype=np.int32)
    ac = O.autocorr_int(s, 4)
    assert len(ac) == 5
    assert all(ac[0] >= abs(ac[i]) for i in range(1, 5))
    for lag in range(5):
        assert ac[lag] == int((s[lag:].astype(np.int64) * s[:len(s) - lag].astype(np.int64)).sum())


def test_levinson_matches_telephone_fixture_coefficients():
    # telephone_8khz.flo carries LPC order 5, shift 15, coeffs (51426,-41731,8660,-3278,1065)
    from conftest import example_bytes
    import flofile
    b = example_bytes("telephone_8khz.flo")
    f = flofile.parse(b)
    ch = f.frames[0].channels[0]
    assert ch.coeffs == [51426, -41731, 8660, -3278, 1065] and ch.shift_bits == 15 and ch.rice_k == 8
    pcm, sr, nch = O.decode_lossless_i32(b)
    got = O.levinson_durbin_int(O.autocorr_int(pcm, 5), 5)
    assert got is not None
    assert got[0].tolist() == ch.coeffs and got[1] == 15
    res = O.calc_residuals_int(pcm, got[0], 15, 5)
    assert O.estimate_rice_parameter_i32(res) == 8
    assert O.rice_encode_i32(res, 8) == ch.residuals


def test_levinson_rejects():
    assert O.levinson_durbin_int([0, 0, 0], 2) is None               # R[0] == 0
    assert O.levinson_durbin_int([10, 10, 10], 2) is None            # |gamma| >= 1
    assert O.levinson_durbin_int([100, 0, 0, 0], 3) is None          # all-zero coefficients


def test_lpc_residual_arithmetic_shift_floor():
    # prediction >>= shift floors toward -inf (lpc.rs:293)
    s = np.array([1, -3, 0], dtype=np.int32)
    # coeffs [1<<14] at shift 15 => pred = (s[i-1]*16384)>>15 : for -3 -> -49152>>15 = -2 (floor of -1.5)
    r = O.calc_residuals_int(s, [1 << 14], 15, 1)
    assert r.tolist() == [1, -3 - 0, 0 - (-2)]
