"""The parallel lossless decoder (lldec_kernels.hip) on hand-made files: every predictor kind, every Rice parameter it
takes, stream lengths around its tile size, and the cases it hands to the serial kernel. The checker is the oracle
decoder (decoder.rs:92-273, rice.rs:123-159 restated); integers must match bit for bit. All calls go through the C
ABI. Needs an MI355X."""
import numpy as np
import pytest

import flofile
from gpu_util import ctx  # noqa: F401
from oracle import oracle as O

pytestmark = pytest.mark.gpu

SR = 44100


def wrapper(res, k, coeffs=(), shift=0, cut=None, extra=b""):
    enc = O.rice_encode_i32(np.asarray(res, np.int32), k)
    if cut is not None:
        enc = enc[:cut]
    return dict(coeffs=[int(c) for c in coeffs], shift=shift, k=k, residuals=enc + extra)


def one_frame(n, chans, flags=0, ft=8):
    return flofile.build_lossless(SR, len(chans), [(ft, n, flags, chans)])


def same(ctx, flo):
    oi, _, _ = O.decode_lossless_i32(flo)
    di = ctx.decode_lossless_i32(flo)
    assert di.shape == oi.shape
    bad = np.nonzero(di != oi)[0]
    assert bad.size == 0, (bad[:8], di[bad[:8]], oi[bad[:8]])
    return oi


def residuals(rng, n, k):
    # Laplacian-ish magnitudes around 2^k with a few large values (quotients up to ~200)
    r = np.rint(rng.laplace(0, max(0.6, 0.7 * 2.0 ** k), n)).astype(np.int64)
    big = rng.random(n) < 0.002
    r[big] = rng.integers(-(100 << k), 100 << k, int(big.sum()), endpoint=True)
    return np.clip(r, -(127 << k), 127 << k).astype(np.int32)


@pytest.mark.parametrize("k", list(range(0, 15)))
def test_every_rice_parameter_of_the_parallel_form(ctx, k):
    rng = np.random.default_rng(100 + k)
    n = 20011
    r = residuals(rng, n, k)
    # order 0 (the residuals are the samples) isolates the Rice stages
    oi = same(ctx, one_frame(n, [wrapper(r, k, shift=128)]))
    assert np.array_equal(oi, r)


@pytest.mark.parametrize("order", [0, 1, 2, 3, 4, 5, 9])
@pytest.mark.parametrize("n", [1, 3, 4, 5, 64, 257, 5000])
def test_fixed_predictors_are_prefix_sums(ctx, order, n):
    rng = np.random.default_rng(order * 100 + n)
    r = residuals(rng, n, 4)
    same(ctx, one_frame(n, [wrapper(r, 4, shift=128 + order)]))


def test_fixed_predictor_wraps_like_i32(ctx):
    rng = np.random.default_rng(2)
    n = 30000
    r = rng.integers(-(1 << 20), 1 << 20, n).astype(np.int32)   # order-4 sums of these leave i32 quickly
    for order in (1, 2, 3, 4):
        same(ctx, one_frame(n, [wrapper(r, 14, shift=128 + order)]))


@pytest.mark.parametrize("order", list(range(1, 13)))
def test_lpc_recurrence_every_order(ctx, order):
    rng = np.random.default_rng(order)
    for n, shift in ((order, 12), (order + 1, 12), (63, 15), (64, 15), (65, 9), (129, 0), (7777, 14), (44100, 15)):
        # a stable predictor: decaying taps, quantised with `shift` bits
        taps = 0.9 * rng.uniform(-1, 1, order) * (0.6 ** np.arange(order))
        coeffs = np.rint(taps * (1 << shift)).astype(np.int64)
        r = residuals(rng, n, 6)
        same(ctx, one_frame(n, [wrapper(r, 6, coeffs, shift)]))


def test_lpc_shift_is_taken_modulo_64(ctx):
    rng = np.random.default_rng(7)
    r = residuals(rng, 3000, 5)
    for shift in (0, 1, 31, 32, 40, 63, 64 + 12, 127):
        same(ctx, one_frame(3000, [wrapper(r, 5, [3000, -1200, 77], shift)]))


def test_lpc_cases_left_to_the_serial_kernel(ctx):
    rng = np.random.default_rng(8)
    n = 6000
    r = residuals(rng, n, 7)
    # coefficient sum just under and at 2^21, shift at and over 20 (the f64 recurrence is exact below those)
    same(ctx, one_frame(n, [wrapper(r, 7, [(1 << 21) - 5, 4], 20)]))
    same(ctx, one_frame(n, [wrapper(r, 7, [(1 << 21) - 4, 4], 20)]))
    same(ctx, one_frame(n, [wrapper(r, 7, [(1 << 21) - 5, 4], 21)]))
    # the largest numbers the parallel form takes: an integrator climbing to just under 2^31 (it stays in i32, so this
    # one is not handed over), then the same ramp a little longer (it wraps, so it is)
    same(ctx, one_frame(2040, [wrapper(np.full(2040, 1 << 20, np.int32), 14, [1 << 20], 20)]))
    same(ctx, one_frame(2060, [wrapper(np.full(2060, 1 << 20, np.int32), 14, [1 << 20], 20)]))
    same(ctx, one_frame(2040, [wrapper(np.full(2040, -(1 << 20), np.int32), 14, [(1 << 20), -5, 5], 20)]))
    same(ctx, one_frame(n, [wrapper(r, 7, [1 << 30, -(1 << 29), 12345], 30)]))
    same(ctx, one_frame(n, [wrapper(r, 7, [-(1 << 31), (1 << 31) - 1], 31)]))
    # an unstable predictor: samples grow past i32 and wrap (decoder.rs:179 `as i32` + wrapping add)
    same(ctx, one_frame(n, [wrapper(r, 7, [2 << 10, 1 << 8], 10)]))
    same(ctx, one_frame(n, [wrapper(np.full(n, 1000, np.int32), 11, [1 << 12], 12)]))   # a ramp that reaches 2^31 late
    # Rice parameters the tile tables have no room for
    for k in (15, 16, 20, 31):
        rr = rng.integers(-(1 << 14), 1 << 14, 500).astype(np.int32)
        same(ctx, one_frame(500, [wrapper(rr, k, [1000, -300], 11)]))


def test_sample_at_the_edge_of_i32(ctx):
    # s = floor(pred) + r lands exactly on i32::MAX / i32::MIN / one past them
    n = 200
    for target in (2**31 - 1, -(2**31), 2**31, -(2**31) - 1):
        r = np.zeros(n, np.int32)
        r[0] = 1 << 20
        # s1 = ((c * s0) >> 0) + r1 with c chosen so that c * s0 + r1 == target
        c = target // (1 << 20)
        r[1] = target - c * (1 << 20)
        flo = one_frame(n, [wrapper(r[:2].tolist() + [0] * (n - 2), 14, [c], 0)])
        same(ctx, flo)


@pytest.mark.parametrize("k", [0, 3, 9])
def test_stream_lengths_around_the_tile_size(ctx, k):
    rng = np.random.default_rng(40 + k)
    n = 9000
    r = residuals(rng, n, k)
    full = len(O.rice_encode_i32(r, k))
    cuts = sorted({0, 1, 2, 3, 4, 5, 255, 256, 257, 511, 512, 513, 1024, 4096, full - 1, full} & set(range(full + 1)))
    for cut in cuts:   # the stream runs out: the rest decodes as zeros
        same(ctx, one_frame(n, [wrapper(r, k, [1500, -400], 11, cut=cut)]))
    for pad in (1, 3, 255, 256, 300):   # more bytes than samples: the rest of the stream is ignored
        same(ctx, one_frame(n, [wrapper(r, k, [1500, -400], 11, extra=bytes(rng.integers(0, 256, pad, dtype=np.uint8)))]))
    same(ctx, one_frame(10, [wrapper(r, k, [1500, -400], 11)]))   # far fewer samples than codes


def test_unary_runs_across_tile_and_staging_boundaries(ctx):
    # quotients of 200..255 placed so that the runs straddle byte 256 (a tile), byte 16384 (64 tiles: one wavefront of
    # the residual stage) and the very end of the stream
    k = 2
    for q in (200, 254, 255):
        r = np.zeros(70000, np.int32)
        r[:] = 1                      # 2 -> quotient 0, 4 bits per code
        big = (q << k) >> 1           # zigzag of a positive value v is 2 v
        for pos_bits in (256 * 8 - 100, 16384 * 8 - 37, 16384 * 8 * 2 - 250):
            r[pos_bits // 4] = big
        r[-1] = big
        same(ctx, one_frame(r.size, [wrapper(r, k, shift=128 + 1)]))
        same(ctx, one_frame(r.size, [wrapper(r, k, shift=128 + 1, cut=len(O.rice_encode_i32(r, k)) - 20)]))


def test_the_256_ones_escape_anywhere_in_the_stream(ctx):
    rng = np.random.default_rng(3)
    k = 5
    r = residuals(rng, 30000, k)
    enc = bytearray(O.rice_encode_i32(r, k))
    for at in (0, 250, 16380, len(enc) - 40):
        bad = bytearray(enc)
        bad[at:at + 33] = b"\xff" * 33   # at least 256 ones in a row
        flo = one_frame(r.size, [dict(coeffs=[900, -100], shift=10, k=k, residuals=bytes(bad))])
        same(ctx, flo)
    bad = bytearray(enc)
    bad[1000:1031] = b"\xff" * 31        # 248 ones plus whatever surrounds them: may or may not reach 256
    same(ctx, one_frame(r.size, [dict(coeffs=[900, -100], shift=10, k=k, residuals=bytes(bad))]))


def test_random_bytes_as_a_rice_stream(ctx):
    rng = np.random.default_rng(11)
    for k in (0, 1, 4, 8, 14):
        junk = bytes(rng.integers(0, 256, 20000, dtype=np.uint8))
        same(ctx, one_frame(12000, [dict(coeffs=[1200, -500, 60], shift=11, k=k, residuals=junk)]))
        same(ctx, one_frame(12000, [dict(coeffs=[], shift=128 + 3, k=k, residuals=junk)]))


def test_mixed_frames_of_one_file(ctx):
    rng = np.random.default_rng(21)
    frames = []
    for i, (n, kind) in enumerate([(SR, "lpc"), (SR, "fixed"), (1000, "silence"), (SR, "raw"), (5000, "lpc"), (77, "fixed")]):
        chans = []
        for c in range(2):
            if kind == "lpc":
                chans.append(wrapper(residuals(rng, n, 8), 8, [28000, -9000, 1200, -77], 14))
            elif kind == "fixed":
                chans.append(wrapper(residuals(rng, n, 3), 3, shift=128 + 2))
            elif kind == "raw":
                chans.append(dict(coeffs=[], shift=0, k=0, residuals=rng.integers(-2000, 2000, n).astype("<i2").tobytes()))
            else:
                chans.append(dict(coeffs=[], shift=0, k=0, residuals=b""))
        frames.append((8, n, i & 1, chans))   # odd frames carry the mid/side flag
    flo = flofile.build_lossless(SR, 2, frames)
    same(ctx, flo)
    of, _, _ = O.decode(flo)
    assert np.array_equal(ctx.decode(flo), of)


def test_many_wrappers_in_one_call(ctx):
    rng = np.random.default_rng(31)
    frames = []
    for i in range(300):
        n = int(rng.integers(1, 3000))
        k = int(rng.integers(0, 15))
        order = int(rng.integers(0, 13))
        if order:
            taps = 0.8 * rng.uniform(-1, 1, order) * (0.5 ** np.arange(order))
            ch = [wrapper(residuals(rng, n, k), k, np.rint(taps * 4096).astype(int), 12) for _ in range(2)]
        else:
            ch = [wrapper(residuals(rng, n, k), k, shift=128 + int(rng.integers(0, 5))) for _ in range(2)]
        frames.append((8, n, int(rng.integers(0, 2)), ch))
    same(ctx, flofile.build_lossless(SR, 2, frames))


@pytest.mark.parametrize("orders", [(8, 8, 8, 8), (12, 3, 9, 5), (6, 0, 7, 0), (0, 11, 0, 0), (1, 2, 12, 8)])
def test_long_wrappers_of_unequal_length_share_a_wavefront(ctx, orders):
    """ll_predict runs four consecutive LPC wrappers in one wavefront, a row of sixteen lanes each: inside every wrapper of
    the group a super-block's loads and stores are unpredicated, near a wrapper's end they are tested one by one; rows
    whose wrapper is a fixed predictor (order 0 here) run along idle. Lengths that end inside different super-blocks,
    orders on both sides of the eight-tap form, and fixed predictors in between must all decode bit for bit."""
    rng = np.random.default_rng(sum(orders) + 5)
    lens = [20000, 20001, 9000, 44100]
    frames = []
    for rep in range(3):
        for n in lens:
            chans = []
            for c in range(2):
                order = orders[(2 * len(frames) + c) % 4]
                k = int(rng.integers(2, 11))
                if order:
                    taps = 0.9 * rng.uniform(-1, 1, order) * (0.6 ** np.arange(order))
                    chans.append(wrapper(residuals(rng, n, k), k, np.rint(taps * 4096).astype(int), 12))
                else:
                    chans.append(wrapper(residuals(rng, n, k), k, shift=128 + int(rng.integers(1, 5))))
            frames.append((8, n, 0, chans))
    same(ctx, flofile.build_lossless(SR, 2, frames))
