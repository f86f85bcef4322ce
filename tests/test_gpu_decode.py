"""Device decode (flo_decode) against the oracle decoder and the reference's own fixture files.

Lossless files: the decoded integers are compared bit for bit, the floats bit for bit as well (one multiply by the
f32 constant 1/32767). Transform files: the inverse MDCT runs a different FFT than the oracle's, so PCM is compared
within 2e-6 absolute on full-scale audio (the encoder-side coefficient tolerance is 1e-5 relative RMS) and the file
geometry exactly. All calls go through the C ABI. Needs an MI355X."""
import numpy as np
import pytest

import flofile
import signals
from conftest import example_bytes
from fixtures_util import LOSSLESS_EXAMPLES, LOSSY_EXAMPLES
from gpu_util import ctx, snr_db  # noqa: F401
from oracle import oracle as O

import flo_amd

pytestmark = pytest.mark.gpu

LOSSY_TOL = 2e-6


@pytest.mark.parametrize("name", LOSSLESS_EXAMPLES + ["audio_lossless"])
def test_lossless_fixture_files_decode_bit_exactly(ctx, name):
    b = example_bytes(name + ".flo")
    oi, sr, ch = O.decode_lossless_i32(b)
    gi, gsr, gch = ctx.decode_lossless_i32(b, with_info=True)
    assert (gsr, gch) == (sr, ch)
    assert np.array_equal(gi, oi)
    of, _, _ = O.decode(b)
    gf = ctx.decode(b)
    assert gf.dtype == np.float32 and np.array_equal(gf.view(np.uint32), of.view(np.uint32))


@pytest.mark.parametrize("name,q,src", LOSSY_EXAMPLES + [("audio_lossy", 0.6, None)])
def test_reference_made_transform_files_decode_like_the_oracle(ctx, name, q, src):
    b = example_bytes(name + ".flo")
    of, sr, ch = O.decode(b)
    gf, gsr, gch = ctx.decode(b, with_info=True)
    assert (gsr, gch) == (sr, ch) and gf.shape == of.shape
    n_frames = len(flofile.parse(b).frames)
    assert gf.size == (n_frames - 1) * 1024 * ch          # the first frame is dropped (lib.rs:338-341)
    assert np.max(np.abs(gf - of), initial=0.0) <= LOSSY_TOL


@pytest.mark.parametrize("sr,ch,level", [(44100, 2, 5), (44100, 1, 5), (96000, 2, 5), (8000, 1, 9), (44100, 2, 2), (48000, 2, 8)])
def test_lossless_round_trip_through_the_device_both_ways(ctx, sr, ch, level):
    # encode on the device, decode on the device: exactly the integers the reference's f32_to_i32 makes of the input
    pcm = signals.music_like(sr, int(2.3 * sr), ch, seed=sr + level)
    flo = ctx.encode_lossless(pcm, sr, ch, 16, level)
    want = np.trunc(np.clip(pcm.astype(np.float32) * np.float32(32767.0), -32768.0, 32767.0)).astype(np.int32)
    got = ctx.decode_lossless_i32(flo)
    assert np.array_equal(got, want[: got.size]) and got.size == want.size
    assert np.array_equal(ctx.decode(flo).view(np.uint32), O.decode(flo)[0].view(np.uint32))


def test_lossless_special_frames(ctx):
    sr = 44100
    # silence frame, raw frame (noise at full scale: nothing beats raw), mid/side frame, a partial last frame
    rng = np.random.default_rng(5)
    sil = np.zeros(sr * 2, np.float32)
    noise = rng.uniform(-1, 1, sr * 2).astype(np.float32)
    t = np.arange(sr) / sr
    l = (0.4 * np.sin(2 * np.pi * 330 * t)).astype(np.float32)
    ms = np.stack([l, l * 0.98], axis=1).reshape(-1)
    tail = signals.music_like(sr, 1234, 2, seed=9)
    pcm = np.concatenate([sil, noise, ms, tail])
    flo = ctx.encode_lossless(pcm, sr, 2, 16, 5)
    f = flofile.parse(flo)
    assert f.frames[0].frame_type == 0 and any(fr.flags & 1 for fr in f.frames)
    oi, _, _ = O.decode_lossless_i32(flo)
    assert np.array_equal(ctx.decode_lossless_i32(flo), oi)


def test_rice_stream_that_runs_out_decodes_like_the_reference(ctx):
    # corrupt files: a larger Rice parameter makes the decoder consume the payload too fast (it runs out of bits and
    # pads with zeros, rice.rs:129-133), a smaller one makes it stop early with bits to spare; long unary runs hit
    # the 256 cap. Whatever the reference decoder makes of such a file, the device decoder must make the same.
    pcm = signals.music_like(44100, 50000, 2, seed=12)
    good = ctx.encode_lossless(pcm, 44100, 2, 16, 5)
    f = flofile.parse(good)
    data0 = 70 + f.toc_size
    # first channel of the first frame: [u32 size][u8 n_coef][coefs][u8 shift][u8 enc][u8 k]...
    ch0 = data0 + 6
    ncoef = good[ch0 + 4]
    kpos = ch0 + 4 + 1 + 4 * ncoef + 2
    assert good[kpos - 1] == 0 and good[kpos] == f.frames[0].channels[0].rice_k
    for newk in (0, 1, 15, 31, 40):
        bad = bytearray(good)
        bad[kpos] = newk
        oi, _, _ = O.decode_lossless_i32(bytes(bad))
        assert np.array_equal(ctx.decode_lossless_i32(bytes(bad)), oi), newk
    # an all-ones payload: every value is the 256-ones escape
    bad = bytearray(good)
    size0 = int.from_bytes(good[ch0:ch0 + 4], "little")
    res0 = kpos + 1
    bad[res0:ch0 + 4 + size0] = b"\xff" * (ch0 + 4 + size0 - res0)
    oi, _, _ = O.decode_lossless_i32(bytes(bad))
    assert np.array_equal(ctx.decode_lossless_i32(bytes(bad)), oi)


def test_level0_files_decode_like_the_reference_not_like_the_input(ctx):
    # the Raw-labelled Rice quirk: the reference cannot decode its own level-0 output; neither may we
    pcm = signals.music_like(44100, 30000, 2, seed=3)
    flo = O.encode_lossless(pcm, 44100, 2, 16, 0)
    oi, _, _ = O.decode_lossless_i32(flo)
    assert np.array_equal(ctx.decode_lossless_i32(flo), oi)


@pytest.mark.parametrize("q", [0.0, 0.35, 0.55, 1.0])
@pytest.mark.parametrize("ch", [1, 2])
def test_transform_round_trip_on_the_device(ctx, q, ch):
    sr = 44100
    pcm = signals.music_like(sr, 3 * sr + 777, ch, seed=17 + ch)
    flo = ctx.encode_lossy(pcm, sr, ch, q)
    g = ctx.decode(flo)
    o, _, _ = O.decode(flo)
    assert g.shape == o.shape and np.max(np.abs(g - o)) <= LOSSY_TOL
    # the reference's own acceptance bar for the codec (lossy tests: SNR well above 10 dB on tonal material)
    n = min(g.size, pcm.size)
    assert snr_db(pcm[:n], g[:n]) > (10.0 if q < 0.9 else 40.0)


def test_long_zero_runs_and_dense_frames_decode(ctx):
    # q = 1.0 on noise gives 255-capped records and dense frames; near-silence gives 3-byte varints and empty frames
    sr = 44100
    a = signals.fast_noise(40000 * 2, 11, 0.9)
    b = np.zeros(30000 * 2, np.float32)
    b[12345] = 0.5
    for pcm, q in ((a, 1.0), (b, 0.55), (np.concatenate([a, b]), 0.8)):
        flo = ctx.encode_lossy(pcm, sr, 2, q)
        g = ctx.decode(flo)
        o, _, _ = O.decode(flo)
        assert g.shape == o.shape and np.max(np.abs(g - o), initial=0.0) <= LOSSY_TOL * max(1.0, float(np.max(np.abs(o), initial=0.0)))


def _sparse_blob(records):
    """[(zero_run, [values...]) ...] -> sparse bytes (encoder.rs:284-314 layout: varint zeros, count, i16 values)"""
    out = bytearray()
    for z, vals in records:
        out += flofile.encode_varint(z) + bytes([len(vals)]) + np.asarray(vals, dtype="<i2").tobytes()
    return bytes(out)


@pytest.mark.parametrize("n_rec", [0, 1, 2, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 200, 340])
def test_hand_made_record_chains_decode_like_the_oracle(ctx, n_rec):
    # the device resolves the record chain by pointer doubling: lane i reaches records i and i + 64, more than 128 records
    # or 1024 bytes take the one-lane walk. Chains of every length around those borders, one- and two-byte zero runs,
    # a record that overruns position 1023 and is cut, records behind it that must be ignored.
    rng = np.random.default_rng(100 + n_rec)
    sfw = [32768 + 256 * 3] * 25    # scale factor 2^3: a few thousand in a coefficient decodes to about 1.0
    frames = []
    for f in range(3):
        chans = []
        for c in range(2):
            recs, pos = [], 0
            for r in range(n_rec):
                z = int(rng.integers(0, 4)) if r % 7 else int(rng.integers(0, 2)) * 130   # now and then a two-byte zero run
                cnt = int(rng.integers(1, 3))
                if n_rec <= 129 and pos + z + cnt > 1024 - 2 * (n_rec - r):   # keep short chains inside the frame
                    z = 0
                    cnt = 1
                recs.append((z, [int(v) for v in rng.integers(-3000, 3000, cnt)]))
                pos += z + cnt
            chans.append((sfw, _sparse_blob(recs)))
        frames.append(chans)
    flo = flofile.build_transform(44100, 2, frames)
    want = O.decode(flo)[0]
    got = ctx.decode(flo)
    assert got.shape == want.shape and want.size == 2 * 2048
    assert np.max(np.abs(got - want)) <= LOSSY_TOL * max(1.0, float(np.max(np.abs(want))))
    if n_rec:
        assert float(np.max(np.abs(want))) > 0


def test_hand_made_odd_record_headers_decode_like_the_oracle(ctx):
    # non-canonical and oversized varints, counts that run past the blob, a blob that ends inside a header, 255-long records,
    # blobs just under and over 1024 bytes: whatever the oracle decoder makes of them, the device makes too
    sfw = [32768 + 256 * 3] * 25
    v = lambda n, s=1: [((7 * i * s) % 4001) - 2000 for i in range(n)]
    blobs = [
        b"\x85\x80\x00" + bytes([2]) + np.asarray([5, -5], "<i2").tobytes(),            # 5 as a three-byte varint
        b"\x80\x80\x80\x80\x00" + bytes([1]) + np.asarray([9], "<i2").tobytes(),      # 0 as a five-byte varint
        b"\xff\xff\xff\xff\x7f" + bytes([1]) + np.asarray([9], "<i2").tobytes(),      # a huge zero run: nothing decodes
        _sparse_blob([(3, v(2))]) + b"\x02\x09" + b"\x11",                              # count 9, one byte of values left
        _sparse_blob([(3, v(2))]) + b"\x81",                                             # ends inside a varint
        _sparse_blob([(0, v(255)), (1, v(255, 3)), (2, v(255, 5)), (0, v(200, 7))]),      # 255-capped records, ~1.9 KB
        _sparse_blob([(0, v(250)), (2, v(250, 3))]) + _sparse_blob([(1, v(1))] * 4),      # 1018 bytes
        _sparse_blob([(0, v(250)), (2, v(250, 3))]) + _sparse_blob([(1, v(1))] * 6),      # 1026 bytes
        _sparse_blob([(1000, v(30))]),                                                    # cut at position 1023
        _sparse_blob([(1020, v(2)), (1, v(3)), (0, v(5)), (4, v(1))]),                    # the third record is cut, the fourth ignored
        b"",
    ]
    for i, blob in enumerate(blobs):
        frames = [[(sfw, blob), (sfw, blobs[(i + 3) % len(blobs)])] for _ in range(3)]
        flo = flofile.build_transform(48000, 2, frames)
        want = O.decode(flo)[0]
        got = ctx.decode(flo)
        assert got.shape == want.shape, i
        assert np.max(np.abs(got - want), initial=0.0) <= LOSSY_TOL * max(1.0, float(np.max(np.abs(want), initial=0.0))), i


def test_empty_and_tiny_files(ctx):
    for n in (0, 1, 1024, 1025):
        pcm = signals.fast_noise(n * 2, 4, 0.3)
        for flo in (ctx.encode_lossy(pcm, 44100, 2, 0.55), ctx.encode_lossless(pcm, 44100, 2, 16, 5)):
            g = ctx.decode(flo)
            o, _, _ = O.decode(flo)
            assert g.shape == o.shape
            assert np.max(np.abs(g - o), initial=0.0) <= LOSSY_TOL


def test_malformed_input_is_an_error_not_a_crash(ctx):
    good = ctx.encode_lossy(signals.music_like(44100, 5000, 2, seed=1), 44100, 2, 0.55)
    with pytest.raises(flo_amd.FloError, match="bad magic"):
        ctx.decode(b"RIFF" + good[4:])
    with pytest.raises(flo_amd.FloError, match="Unexpected end of file"):
        ctx.decode(good[:40])
    with pytest.raises(flo_amd.FloError):
        ctx.decode(good[: len(good) // 2])
    # a transform blob that claims more channels than the header has
    f = flofile.parse(good)
    bad = bytearray(good)
    blob0 = 70 + f.toc_size + 10 + 1       # header, TOC, frame header + channel size, then [block_size][channels]
    bad[blob0] = 7
    with pytest.raises(flo_amd.FloError, match="deserialize"):
        ctx.decode(bytes(bad))


def test_module_level_decode_and_decoder_class(ctx):
    pcm = signals.music_like(44100, 20000, 2, seed=8)
    flo = ctx.encode_lossless(pcm, 44100, 2, 16, 5)
    a = flo_amd.Decoder(ctx).decode(flo)
    assert np.array_equal(a.view(np.uint32), O.decode(flo)[0].view(np.uint32))


# ----------------------------------------------------------------------------------------------- BASELINE sizes
def _d2d(dst_ptr, src_ptr, nbytes):
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")       # the runtime torch already loaded (same SONAME)
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert hip.hipMemcpy(dst_ptr, src_ptr, nbytes, 3) == 0


@pytest.mark.parametrize("n_clips,seconds,q", [(10000, 10, 0.55), (1250, 10, 0.55), (1024, 10, 0.35), (1, 180, 0.55)])
def test_full_size_configs_round_trip_on_the_device(ctx, n_clips, seconds, q):
    """BASELINE configs[3] whole (the 10 000-clip corpus on one GPU: 35 GB of PCM) and at its per-GPU shard, configs[2]
    and configs[1] at full size, through size-independent
    properties: every kernel form gives the same bytes (a checksum of the whole batch), encoding is idempotent, and
    encode -> device decode reproduces the input (per-clip SNR, input never leaves HBM)."""
    import torch
    sr, ch = 44100, 2
    n_sf = seconds * sr
    b = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [n_sf * ch] * n_clips, sr, ch, q)
    b.fill_synthetic(seed=0xF10A0D10, clip_id0=0)
    hops = (n_sf + 1024 + 1023) // 1024

    def packed(form):
        b.encode(form)
        b.sync()
        nbytes = b.data_bytes()
        buf = torch.zeros(nbytes + 16 * n_clips + 64, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()                   # the library works on its own non-blocking stream
        offs = b.pack_streams(buf.data_ptr(), buf.numel())
        b.sync()
        return buf[: offs[-1]], nbytes

    first, nbytes = packed(0)
    for form in (1, 2, 5, 0, 5, 1):    # one-wave-per-channel chain, frame-parallel, lock-step stereo chain, and again (idempotence)
        other, nb = packed(form)
        assert nb == nbytes and torch.equal(first, other), form
    assert 0.02 * n_sf * ch * n_clips < nbytes < 4.0 * n_sf * ch * n_clips * 0.5   # 2 % .. 50 % of the f32 input

    out = torch.empty(n_clips * (hops - 1) * 1024 * ch, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    offs = b.decode_to(out.data_ptr(), out.numel())
    assert offs == [i * (hops - 1) * 1024 * ch for i in range(n_clips)]
    dec = out.view(n_clips, (hops - 1) * 1024 * ch)[:, : n_sf * ch]
    worst = float("inf")
    step = max(1, min(n_clips, 128))
    for i0 in range(0, n_clips, step):
        k = min(step, n_clips - i0)
        src = torch.empty(k, n_sf * ch, dtype=torch.float32, device="cuda:0")
        for j in range(k):
            _d2d(src[j].data_ptr(), b.clip_device_ptr(i0 + j), n_sf * ch * 4)
        err = (dec[i0:i0 + k].double() - src.double()).pow(2).sum(dim=1)
        sig = src.double().pow(2).sum(dim=1)
        snr = 10 * torch.log10(sig / err.clamp_min(1e-30))
        worst = min(worst, float(snr.min()))
    # sanity bound for the codec at this quality (tones + a little noise), then the real tie: the first clip's
    # round-trip SNR equals the oracle's own encode -> decode of the same integer-exact synthetic clip
    assert worst > (12.0 if q >= 0.5 else 6.0), worst
    pcm0 = O.synth_clip(n_sf, ch, 0xF10A0D10, 0)
    o_dec, _, _ = O.decode(O.encode_lossy(pcm0, sr, ch, q))
    g_dec = dec[0].cpu().numpy()
    assert abs(snr_db(pcm0, o_dec[: pcm0.size]) - snr_db(pcm0, g_dec)) < 0.05
    b.close()


def test_damaged_transform_payloads_decode_like_the_oracle_or_fail_like_it(ctx):
    # single-byte damage anywhere in the DATA chunk of a lossy file: record headers, varints, counts, scale words,
    # blob lengths. Whatever the reference decoder does with such a file - error, or garbage - the device does too.
    pcm = signals.music_like(44100, 20000, 2, seed=31)
    good = ctx.encode_lossy(pcm, 44100, 2, 0.8)
    f = flofile.parse(good)
    d0 = 70 + f.toc_size
    rng = np.random.default_rng(7)
    spots = [int(x) for x in rng.integers(d0, d0 + f.data_size, 120)] + list(range(d0, d0 + 140))
    errors = same = 0
    for pos in spots:
        bad = bytearray(good)
        bad[pos] ^= int(rng.integers(1, 256))
        bad = bytes(bad)
        try:
            o, _, _ = O.decode(bad)
        except RuntimeError:
            with pytest.raises(flo_amd.FloError):
                ctx.decode(bad)
            errors += 1
            continue
        g = ctx.decode(bad)
        assert g.shape == o.shape, pos
        fin = np.isfinite(o) & np.isfinite(g)
        assert np.array_equal(np.isfinite(o), np.isfinite(g)) or (~fin).sum() < 4096, pos
        scale = max(1.0, float(np.max(np.abs(o[fin]), initial=0.0)))
        assert np.max(np.abs(g[fin] - o[fin]), initial=0.0) <= 4e-6 * scale, pos
        same += 1
    assert errors > 5 and same > 50


def test_random_damage_decodes_like_the_oracle_or_fails_like_it(ctx):
    """A short differential fuzz (diag/dec_fuzz.py is the long one): random byte damage inside DATA of lossless and
    lossy files. Whatever the oracle decoder makes of the file - an error, or PCM - the device decoder must make the
    same: lossless integers exactly; lossy within 2e-6 of the file's own magnitude (a damaged scale word can blow a
    frame up far beyond full scale)."""
    rng = np.random.default_rng(2027)
    pcm = signals.music_like(44100, 40000, 2, seed=4)
    goods = [ctx.encode_lossless(pcm, 44100, 2, 16, 5), ctx.encode_lossless(pcm[:20001], 44100, 1, 16, 8),
             ctx.encode_lossy(pcm, 44100, 2, 0.55), ctx.encode_lossy(pcm, 44100, 2, 1.0)]
    rejected = 0
    for it in range(80):
        g = goods[it % len(goods)]
        f = flofile.parse(g)
        d0 = 70 + f.toc_size
        b = bytearray(g)
        for _ in range(int(rng.integers(1, 6))):
            kind = int(rng.integers(0, 3))
            at = int(rng.integers(d0, d0 + f.data_size))
            if kind == 0:
                b[at] = int(rng.integers(0, 256))
            elif kind == 1:
                b[at:at + 4] = bytes(rng.integers(0, 256, 4, dtype=np.uint8))
            else:
                b[at:at + 40] = b"\xff" * min(40, len(b) - at)
        b = bytes(b)
        try:
            want = O.decode(b)[0]
        except Exception:  # noqa: BLE001 - the oracle reports malformed input through exceptions
            with pytest.raises(flo_amd.FloError):
                ctx.decode(b)
            rejected += 1
            continue
        got = ctx.decode(b)
        assert got.shape == want.shape, it
        if not want.size:
            continue
        with np.errstate(invalid="ignore", over="ignore"):
            fin = np.isfinite(want) & np.isfinite(got)
            assert np.array_equal(np.isnan(want), np.isnan(got)) and np.array_equal(np.isinf(want), np.isinf(got)), it
            diff = np.abs(want[fin].astype(np.float64) - got[fin].astype(np.float64))
            tol = 2e-6 * max(1.0, float(np.abs(want[fin]).max()) if fin.any() else 1.0) if f.is_lossy else 0.0
        assert (float(diff.max()) if diff.size else 0.0) <= tol, (it, f.is_lossy)


@pytest.mark.gpu
@pytest.mark.parametrize("ch,lens", [(2, [3, 50000, 1024, 441000, 2047, 90000, 12345]), (1, [70000, 5, 2048, 33333, 1, 100000, 7, 4096, 65536]),
                                     (2, [1000] * 17), (1, [44100] * 9)])
def test_ragged_batches_decode_like_their_files(ctx, ch, lens):
    """The batch decoder deals (clip, run, channel) triples to workgroups in groups of eight per channel (the channels
    of a run share an XCD): clip counts that are no multiple of eight, clips of one frame next to clips of hundreds, mono
    and stereo - every clip of the batch must come out exactly as the single-file decoder returns its file."""
    import torch
    sr = 44100
    b = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [n * ch for n in lens], sr, ch, 0.55)
    b.fill_synthetic(seed=0xF10A0D10, clip_id0=77)
    b.encode(0)
    b.sync()
    hops = [(n + 1024 + 1023) // 1024 for n in lens]
    total = sum(max(h - 1, 0) * 1024 * ch for h in hops)
    out = torch.full((total + 8,), float("nan"), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    offs = b.decode_to(out.data_ptr(), total)
    host = out.cpu().numpy()
    assert np.isnan(host[total:]).all()                      # nothing behind the batch is touched
    assert not np.isnan(host[:total]).any()                  # every sample of every block is written
    for i, h in enumerate(hops):
        want = ctx.decode(b.fetch(i))
        n_out = max(h - 1, 0) * 1024 * ch
        assert want.size == n_out, (i, want.size, n_out)
        np.testing.assert_array_equal(host[offs[i]: offs[i] + n_out], want, err_msg=f"clip {i}")
    b.close()
