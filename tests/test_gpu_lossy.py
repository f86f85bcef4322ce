"""Parity of the HIP lossy path against the oracle (all calls go through the C ABI). Needs an MI355X."""
import numpy as np
import pytest

import flofile
import signals
from conftest import example_bytes
from fixtures_util import LOSSY_EXAMPLES, dequantise, lossy_source_pcm
from gpu_util import compare_lossy_stage, ctx, same_structure, snr_db  # noqa: F401
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def test_mdct_forward_parity(ctx):
    rng = np.random.default_rng(0)
    frames = rng.uniform(-1, 1, (64, 2048)).astype(np.float32)
    frames[1] = signals.sine(440.0, 44100, 2048, 0.5)
    frames[2] = 1.0
    frames[3] = 0.0
    frames[4, :] = 0.0
    frames[4, 1000] = 1.0
    got = ctx.mdct_forward(frames)
    for i in range(frames.shape[0]):
        ref = O.mdct_forward(frames[i]).astype(np.float64)
        scale = max(np.abs(ref).max(), 1e-20)
        assert np.abs(got[i] - ref).max() <= 4e-6 * scale, i
    # accuracy yardstick: both against the f64 definition, the device must not be worse than 2x the oracle
    for i in (0, 1, 5):
        truth = O.mdct_forward_direct_f64(frames[i])
        e_o = np.abs(O.mdct_forward(frames[i]) - truth).max()
        e_g = np.abs(got[i] - truth).max()
        assert e_g <= max(2 * e_o, 1e-6 * np.abs(truth).max())
    assert not got[3].any()


def _patterns():
    rng = np.random.default_rng(1)
    pats = [np.zeros(1024, np.int16), np.ones(1024, np.int16), np.arange(1024, dtype=np.int16) - 300]
    for n in (1, 254, 255, 256, 257, 509, 510, 511, 765, 766, 1020, 1021, 1023):
        a = np.zeros(1024, np.int16); a[:n] = 7; pats.append(a)                  # run from the start
        b = np.zeros(1024, np.int16); b[1024 - n:] = -3; pats.append(b)          # run to the end
        c = np.zeros(1024, np.int16); c[3:3 + min(n, 1000)] = 9; pats.append(c)  # run after a short zero run
    for z in (126, 127, 128, 129, 255, 256, 1000, 1023):
        a = np.full(1024, 5, np.int16); a[5:5 + min(z, 1019)] = 0; pats.append(a)
        b = np.zeros(1024, np.int16); b[min(z, 1023)] = 1; pats.append(b)
    a = np.zeros(1024, np.int16); a[::2] = 1; pats.append(a)
    a = np.zeros(1024, np.int16); a[1::2] = -1; pats.append(a)
    a = np.zeros(1024, np.int16); a[15::16] = 2; pats.append(a)
    a = np.zeros(1024, np.int16); a[16::16] = 2; pats.append(a)
    a = np.zeros(1024, np.int16); a[0] = -32768; a[1023] = 32767; pats.append(a)
    for dens in (0.002, 0.02, 0.1, 0.5, 0.9, 0.99):
        for _ in range(6):
            pats.append((rng.integers(-32768, 32768, 1024) * (rng.uniform(size=1024) < dens)).astype(np.int16))
    return np.array(pats)


@pytest.mark.parametrize("form", [0, 1, 2], ids=["encoder", "general", "block"])
def test_sparse_pack_bit_exact(ctx, form):
    # form 0 = the packer as every encode runs it (item form up to 128 non-zeros, block form behind it, the general form
    # behind that for dense vectors); form 2 starts at the block form
    pats = _patterns()
    got = ctx.sparse_pack(pats, form)
    for i, p in enumerate(pats):
        assert got[i] == O.serialize_sparse(p), i


def test_sparse_pack_block_form_on_structured_vectors(ctx):
    # what the block form has to get right: zero runs of 127 / 128 / 129 and longer in front of a word's first run
    # (two-byte varints shift everything behind them by one byte), runs that cross 64-bit word boundaries, runs of
    # exactly 255 and 256, up to 126 and 127 runs (run table capacity), trailing zero runs of every varint size
    rng = np.random.default_rng(5)
    pats = []
    for gap in (1, 63, 64, 65, 126, 127, 128, 129, 191, 192, 193, 255, 256, 300, 511, 640, 1000):
        for first in (0, 1, 5, 63, 64, 100):
            a = np.zeros(1024, np.int16)
            pos, k = first, 0
            while pos < 1024:
                n = int(rng.integers(1, 5))
                a[pos:pos + n] = rng.integers(1, 30000, min(n, 1024 - pos)) * rng.choice([-1, 1])
                pos += n + gap + (k % 3 == 0)
                k += 1
            pats.append(a)
    for run in (62, 63, 64, 65, 127, 128, 129, 190, 191, 192, 193, 254, 255, 256, 257, 300):
        for start in (0, 1, 31, 63, 64, 65, 700):
            a = np.zeros(1024, np.int16)
            a[start:start + run] = -7
            a[min(1023, start + run + 130)] = 9
            pats.append(a)
    for runs in (1, 2, 63, 64, 65, 125, 126, 127, 128, 200, 512):
        a = np.zeros(1024, np.int16)
        a[0:2 * runs:2] = 3
        pats.append(a)
        b = np.zeros(1024, np.int16)
        b[1:2 * runs:2][:runs] = -3
        pats.append(b)
    for tail in (0, 1, 2, 127, 128, 129, 500, 1023):
        a = np.full(1024, 11, np.int16)
        a[::7] = 0
        if tail:
            a[1024 - tail:] = 0
        pats.append(a)
    for dens in (0.01, 0.03, 0.06, 0.12, 0.25):
        for _ in range(40):
            keep = rng.uniform(size=1024) < dens * np.exp(-np.arange(1024) / rng.uniform(100, 900))   # low-pass like a spectrum
            pats.append((rng.integers(-32768, 32768, 1024) * keep).astype(np.int16))
    # run-table limits (126 / 127 / 128 runs), runs of 255 / 256 / 257 behind wide gaps, records that straddle the
    # 128-position blocks and the 64-run header passes
    for n in (63, 64, 65, 127, 128, 129, 255, 256, 257):
        a = np.zeros(1024, np.int16); a[:2 * n:2][:n] = 5; pats.append(a)                    # n runs of one
        b = np.zeros(1024, np.int16); b[100:100 + n] = -9; b[900] = 1; pats.append(b)        # one run of n, a wide gap
        c = np.zeros(1024, np.int16); c[np.arange(n) * 3 % 1024] = 7; pats.append(c)
    for k in (64, 128, 192):
        a = np.zeros(1024, np.int16); a[:k] = 3; a[k + 200:k + 203] = 4; pats.append(a)      # wide record starts item k
        b = np.zeros(1024, np.int16); b[:k - 1] = 3; b[k + 200:k + 203] = 4; pats.append(b)
    # what the item form has to get right: 0 / 1 / 63 / 64 / 65 / 127 / 128 / 129 non-zeros (one and two items per lane, the
    # hand-over to the block form), records that start or end on item 63 / 64, zero runs of 127 / 128 / 129 in front of
    # the first item, between items and behind the last, a first item at position 0, a last one at 1023
    for n in (0, 1, 2, 63, 64, 65, 66, 126, 127, 128, 129, 130):
        for stride in (1, 2, 3, 7):
            a = np.zeros(1024, np.int16); a[np.arange(n) * stride] = rng.integers(1, 32767, n) * rng.choice([-1, 1], n); pats.append(a)
            b = np.zeros(1024, np.int16); b[1023 - np.arange(n) * stride] = -5; pats.append(b)
        for lead in (1, 126, 127, 128, 129, 130, 255, 256, 700):
            a = np.zeros(1024, np.int16); a[lead:lead + n] = 6; pats.append(a)                      # one run behind a lead of zeros
            b = np.zeros(1024, np.int16); b[:n // 2] = 6; b[n // 2 + lead:n // 2 + lead + (n - n // 2)] = -6; pats.append(b)
    for first_run in (62, 63, 64, 65):
        for gap in (1, 2, 127, 128, 129):
            for second in (1, 2, 30, 63, 64):
                a = np.zeros(1024, np.int16); a[3:3 + first_run] = 9; a[3 + first_run + gap:3 + first_run + gap + second] = -9; pats.append(a)
    for _ in range(200):
        n = int(rng.integers(0, 140))
        a = np.zeros(1024, np.int16)
        a[rng.choice(1024, n, replace=False)] = rng.integers(-32768, 32768, n)
        pats.append(a)
    pats = np.array(pats)
    got0, got1, got2 = ctx.sparse_pack(pats, 0), ctx.sparse_pack(pats, 1), ctx.sparse_pack(pats, 2)
    for i, p in enumerate(pats):
        ref = O.serialize_sparse(p)
        assert got0[i] == ref, i
        assert got1[i] == ref, i
        assert got2[i] == ref, i


@pytest.mark.parametrize("exact", [False, True], ids=["shipped", "exact"])
@pytest.mark.parametrize("q", [0.0, 0.35, 0.55, 0.75, 1.0])
def test_quantiser_fed_oracle_spectra(ctx, q, exact):
    # isolates psychoacoustics + quantiser + scale words from the transform: identical coefficients in,
    # identical integers out (up to libm-vs-device log rounding on a vanishing fraction).
    # "shipped" is the instantiation every encode entry point runs (amplitude-domain keep test, quantise<., EXACT=false>);
    # "exact" adds the reference's dB-domain re-check next to the threshold. Both meet the same bound.
    pcm = signals.music_like(44100, 30000, 2, seed=3)
    o = O.lossy_analyze(pcm, 44100, 2, q)
    for path in ((0,) if exact else (0, 5)):   # 5 = the benchmarked form: the quantiser runs in the packer wave, natural layout
        ctx.force_path(path)
        g = ctx.lossy_quantize(o["coeffs"], 44100, q, exact=exact)
        ctx.force_path(0)
        flips = ((g["q"] != 0) != (o["q"] != 0)).mean()
        mism = (g["q"] != o["q"]).mean()
        print(f"q={q} exact={exact} path={path}: keep/drop flip rate {flips:.2e}, integer mismatch rate {mism:.2e}")
        assert flips <= 1e-4, flips    # SURVEY 8c(ii) allows 5e-4; the band-energy summation order is the only difference
        assert mism <= 1e-4, mism
        assert np.abs(g["sf_words"].astype(int) - o["sf_words"].astype(int)).max() <= 1
        assert (g["sf_words"] != o["sf_words"]).mean() <= 1e-3


def test_shipped_quantiser_on_a_long_clip_reports_its_flip_rate(ctx):
    # the same isolation on ten seconds of the bench's own synthetic signal (864 frame-channels, 885 k coefficients),
    # production instantiation only: the measured keep/drop flip rate against the oracle's decisions
    pcm = O.synth_clip(441000, 2, 0xF10A0D10, 7)
    o = O.lossy_analyze(pcm, 44100, 2, 0.55)
    ctx.force_path(5)          # the benchmarked form
    g = ctx.lossy_quantize(o["coeffs"], 44100, 0.55, exact=False)
    ctx.force_path(0)
    flips = int(((g["q"] != 0) != (o["q"] != 0)).sum())
    print(f"shipped quantiser, 10 s synthetic clip: {flips} keep/drop flips in {o['q'].size} coefficients ({flips / o['q'].size:.2e})")
    assert flips <= 1e-4 * o["q"].size
    assert (g["q"] != o["q"]).mean() <= 1e-4


@pytest.mark.parametrize("q", [0.35, 0.55, 0.75, 1.0])
def test_kept_integers_are_no_further_from_an_exact_transform_than_the_oracle(ctx, q):
    """SURVEY 8c(ii) asked for |dq| <= 1 among coefficients kept by both; an f32 FFT cannot deliver that in quiet
    bands (absolute noise ~1e-7 of the frame maximum times a band gain of up to 30000 / band_max). The measurement
    that justifies the relaxed bound of compare_lossy_stage: run the SAME psychoacoustic model and quantiser behind
    a transform evaluated in double precision (oracle, f64_mdct=True) and require, band by band, that the device's
    integers are no further from that truth than the pinned oracle's own f32 FFT puts them, plus one rounding step:
        max_band |q_dev - q_f64|  <=  max_band |q_oracle - q_f64| + 1."""
    band = O.psy_tables(44100)[1].astype(np.int64)
    worst_dev = worst_orc = 0
    for seed, maker in ((3, lambda: signals.music_like(44100, 30000, 2, seed=3)), (0, lambda: O.synth_clip(40000, 2, 0xF10A0D10, 11))):
        pcm = maker()
        f = O.lossy_analyze(pcm, 44100, 2, q, f64_mdct=True)
        o = O.lossy_analyze(pcm, 44100, 2, q)
        g = ctx.lossy_analyze(pcm, 44100, 2, q)
        qf, qo, qg = (x["q"].astype(np.int64) for x in (f, o, g))
        kept = (qf != 0) & (qo != 0) & (qg != 0)
        d_dev = np.where(kept, np.abs(qg - qf), 0)
        d_orc = np.where(kept, np.abs(qo - qf), 0)
        for b in range(25):
            sel = band == b
            if not sel.any():
                continue
            md, mo = d_dev[..., sel].max(axis=-1), d_orc[..., sel].max(axis=-1)      # [hops][ch]
            assert (md <= mo + 1).all(), (q, seed, b, int((md - mo).max()))
        worst_dev, worst_orc = max(worst_dev, int(d_dev.max())), max(worst_orc, int(d_orc.max()))
        # and the device transform itself is at least as close to the exact one as the oracle's FFT
        e_dev = np.sqrt(((g["coeffs"].astype(np.float64) - f["coeffs"]) ** 2).sum())
        e_orc = np.sqrt(((o["coeffs"].astype(np.float64) - f["coeffs"]) ** 2).sum())
        assert e_dev <= 1.5 * e_orc + 1e-12, (e_dev, e_orc)
    print(f"q={q}: worst |q - q_f64| among kept coefficients: device {worst_dev}, oracle {worst_orc}")


@pytest.mark.parametrize("ch,q", [(1, 0.55), (2, 0.35), (2, 0.55), (2, 1.0), (1, 0.0)])
def test_analyze_parity(ctx, ch, q):
    pcm = signals.music_like(44100, 40000, ch, seed=10 + ch)
    o = O.lossy_analyze(pcm, 44100, ch, q)
    first = None
    for path in (1, 2, 5):      # 5 is the stereo chain form (mono falls back to 1)
        ctx.force_path(path)
        g = ctx.lossy_analyze(pcm, 44100, ch, q)
        compare_lossy_stage(g, o, 44100, tag=f"path{path}")
        # every form computes the same coefficients, integers and scale words, bit for bit
        if first is None:
            first = g
        else:
            for key in ("coeffs", "q", "sf_words"):
                assert np.array_equal(first[key].view(np.uint8), g[key].view(np.uint8)), (path, key)
    ctx.force_path(0)


@pytest.mark.parametrize("sr", [8000, 22050, 48000, 96000, 128000, 176400, 192000, 384000])
def test_other_sample_rates(ctx, sr):
    pcm = signals.music_like(sr, 20000, 2, seed=sr)
    o = O.lossy_analyze(pcm, sr, 2, 0.55)
    compare_lossy_stage(ctx.lossy_analyze(pcm, sr, 2, 0.55), o, sr)
    ctx.force_path(5)          # the lock-step stereo form has its own band-statistics code and quantises in the packer wave (natural layout): other band tables too
    compare_lossy_stage(ctx.lossy_analyze(pcm, sr, 2, 0.55), o, sr, tag="path5")
    ctx.force_path(0)
    # the whole drop-in call and the decode at this rate (from 128 kHz up band 24 spans more than 48 lane segments)
    flo = ctx.encode_lossy(pcm, sr, 2, 0.55)
    fo = O.encode_lossy(pcm, sr, 2, 0.55)
    same_structure(flo, fo)
    dec = ctx.decode(fo)
    odec = O.decode(fo)[0]
    assert dec.shape == odec.shape and float(np.max(np.abs(dec - odec))) <= 4e-6


def test_chain_and_frame_parallel_forms_give_identical_files(ctx):
    pcm = signals.music_like(44100, 50000, 2, seed=21)
    ctx.force_path(1)
    a = ctx.encode_lossy(pcm, 44100, 2, 0.55)
    ctx.force_path(2)
    b = ctx.encode_lossy(pcm, 44100, 2, 0.55)
    ctx.force_path(5)
    c = ctx.encode_lossy(pcm, 44100, 2, 0.55)
    ctx.force_path(0)
    assert a == b
    assert a == c


@pytest.mark.parametrize("q", [0.0, 0.55, 1.0])
def test_lock_step_pipeline_matches_chain_on_a_ragged_batch(ctx, q):
    # clips of different lengths (and one empty) in one launch: the transform wave and the packer wave of every
    # clip run their own frame counts; bytes must equal the one-wave-per-channel chain form
    lens = [0, 1, 1023, 1024, 5000, 44100, 70001, 3 * 1024]
    clips = [signals.music_like(44100, n, 2, seed=40 + i) for i, n in enumerate(lens)]
    ctx.force_path(1)
    a = ctx.encode_batch(1, clips, 44100, 2, q)
    ctx.force_path(5)     # lock-step stereo transform wave + quantiser-and-packer wave, persistent workgroups dealing the clips dynamically
    b = ctx.encode_batch(1, clips, 44100, 2, q)
    ctx.force_path(2)
    c = ctx.encode_batch(1, clips, 44100, 2, q)
    ctx.force_path(0)
    assert a == b
    assert a == c


@pytest.mark.parametrize("ch", [1, 2])
def test_encode_file_matches_oracle_structure_and_decodes(ctx, ch):
    pcm = signals.music_like(44100, 44100, ch, seed=5)
    g = ctx.encode_lossy(pcm, 44100, ch, 0.55, b"meta!")
    o = O.encode_lossy(pcm, 44100, ch, 0.55, b"meta!")
    fg, fo = same_structure(g, o)
    assert abs(fg.data_size - fo.data_size) <= 0.005 * fo.data_size + 8
    dg, _, _ = O.decode(g)
    do, _, _ = O.decode(o)
    assert dg.size == do.size and snr_db(do, dg) >= 70.0
    assert snr_db(pcm, dg[: pcm.size]) > 10.0


def test_config1_silence_is_byte_identical(ctx):
    # BASELINE config 1 input (all-zero stereo second): DATA must equal the reference's audio_lossy.flo DATA
    ref = flofile.parse(example_bytes("audio_lossy.flo"))
    got = flofile.parse(ctx.encode_lossy(np.zeros(88200, np.float32), 44100, 2, 0.6))
    assert got.data == ref.data and got.data_crc32 == 0x00CA7202 and got.flags == ref.flags == 0x0201
    assert got.total_samples == 46080


@pytest.mark.parametrize("name,q,src", LOSSY_EXAMPLES)
def test_near_goldens_reference_made_files(ctx, name, q, src):
    pcm, sr, ch = lossy_source_pcm(src)
    ref = flofile.parse(example_bytes(name + ".flo"))
    RQ = np.zeros((88, ch, 1024), np.int16)
    RS = np.zeros((88, ch, 25), np.uint16)
    for i, fr in enumerate(ref.frames):
        _, RS[i], RQ[i] = flofile.parse_transform_blob(fr.channels[0].raw)
    g = ctx.lossy_analyze(pcm, sr, ch, q)
    band = O.psy_tables(sr)[1]
    d_g, d_r = dequantise(g["q"], g["sf_words"], band), dequantise(RQ, RS, band)
    rel = float(np.sqrt(((d_g - d_r) ** 2).sum() / (d_r ** 2).sum()))
    assert rel <= 1e-5, rel
    if q <= 0.8:
        flips = int(((g["q"] != 0) != (RQ != 0)).sum())
        assert flips <= 1e-4 * RQ.size, flips
    enc = flofile.parse(ctx.encode_lossy(pcm, sr, ch, q))
    assert enc.flags == ref.flags and enc.total_samples == ref.total_samples and len(enc.frames) == 88 and enc.crc_valid


@pytest.mark.parametrize("n", [0, 1, 2, 1023, 1024, 1025, 2047, 2048, 4097])
@pytest.mark.parametrize("ch", [1, 2])
def test_edge_lengths(ctx, n, ch):
    pcm = signals.fast_noise(n * ch, 7, 0.4)
    o = O.encode_lossy(pcm, 44100, ch, 0.55)
    for path in (0, 1, 2, 5):     # 0 = what an encode call picks by itself, 5 = the benchmarked lock-step form
        ctx.force_path(path)
        g = ctx.encode_lossy(pcm, 44100, ch, 0.55)
        fg, _ = same_structure(g, o)
        assert len(fg.frames) == (n + 1024 + 1023) // 1024
    ctx.force_path(0)


def test_trailing_partial_sample_frame_is_dropped(ctx):
    pcm = signals.fast_noise(2001, 3)     # stereo with an odd sample count
    same_structure(ctx.encode_lossy(pcm, 44100, 2, 0.55), O.encode_lossy(pcm, 44100, 2, 0.55))


def test_nan_inf_do_not_crash(ctx):
    x = signals.fast_noise(8192, 2)
    x[100], x[2000], x[3001] = np.nan, np.inf, -np.inf
    f = flofile.parse(ctx.encode_lossy(x, 44100, 2, 0.55))
    assert f.crc_valid and len(f.frames) == 5


def test_quality_is_clamped_and_header_level(ctx):
    pcm = signals.sine(440.0, 44100, 8000, 0.5)
    for q, lvl in [(-1.0, 0), (0.0, 0), (0.35, 1), (0.55, 2), (0.75, 3), (1.0, 4), (7.0, 4)]:
        f = flofile.parse(ctx.encode_lossy(pcm, 44100, 1, q))
        assert f.is_lossy and f.lossy_quality == lvl and f.bit_depth == 16 and f.level == 5


def test_ragged_batch_equals_single_encodes(ctx):
    import flo_amd
    lens = [0, 1500, 44100, 10000, 1024, 33333]
    clips = [signals.music_like(44100, n, 2, seed=n) for n in lens]
    for which in (1, 2):
        b = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [c.size for c in clips], 44100, 2, 0.55)
        for i, c in enumerate(clips):
            b.upload(i, c)
        b.encode(which)
        b.sync()
        outs = [b.fetch(i) for i in range(len(clips))]
        assert b.data_bytes() == sum(flofile.parse(o).data_size for o in outs)
        b.close()
        ctx.force_path(which)
        for c, o in zip(clips, outs):
            assert o == ctx.encode_lossy(c, 44100, 2, 0.55)
    ctx.force_path(0)
    many = ctx.encode_batch(flo_amd.MODE_LOSSY, clips, 44100, 2, 0.55)
    assert [flofile.parse(m).total_samples for m in many] == [flofile.parse(o).total_samples for o in outs]


def test_sine_snr_reference_bar(ctx):
    # libflo/tests/rust/lossy_transform_tests.rs:117-185
    orig = signals.sine(440.0, 44100, 44100, 0.5)
    dec, _, _ = O.decode(ctx.encode_lossy(orig, 44100, 1, 0.75))
    assert snr_db(orig, dec[:44100]) > 10.0


def test_synthetic_corpus_properties_at_scale(ctx):
    # size-independent checks on a larger device-generated batch: every file parses, CRC-valid, frame counts right,
    # and a sample of clips agrees with the oracle run on the same integer-exact input
    import flo_amd
    n_clips, n_sf = 96, 3 * 44100
    b = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [n_sf * 2] * n_clips, 44100, 2, 0.55)
    b.fill_synthetic(seed=0xF10A0D10, clip_id0=1000)
    b.encode(1)
    b.sync()
    hops = (n_sf + 1024 + 1023) // 1024
    sizes = []
    for i in range(n_clips):
        f = flofile.parse(b.fetch(i))
        assert f.crc_valid and len(f.frames) == hops and f.total_samples == hops * 1024
        sizes.append(f.data_size)
    assert len(set(sizes)) > n_clips // 2           # clips differ
    for i in (0, 57):
        pcm = O.synth_clip(n_sf, 2, 0xF10A0D10, 1000 + i)
        fo = flofile.parse(O.encode_lossy(pcm, 44100, 2, 0.55))
        fg = flofile.parse(b.fetch(i))
        assert abs(fg.data_size - fo.data_size) <= 0.005 * fo.data_size + 8
        ctx.force_path(1)
        compare_lossy_stage(ctx.lossy_analyze(pcm, 44100, 2, 0.55), O.lossy_analyze(pcm, 44100, 2, 0.55), 44100, f"clip{i}")
        ctx.force_path(0)
    b.close()


def test_pack_streams_and_single_rank_rccl_gather(ctx):
    # the exchange step of the multi-GPU path, exercised with a 1-rank "nccl" (= RCCL) group on the one GPU here
    import os
    import torch
    import torch.distributed as dist
    import flo_amd
    from dist_ref import gather_payloads
    clips = [signals.music_like(44100, n, 2, seed=n) for n in (5000, 44100, 12345)]
    b = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [c.size for c in clips], 44100, 2, 0.55)
    for i, c in enumerate(clips):
        b.upload(i, c)
    b.encode(1)
    b.sync()
    buf = torch.empty(b.data_bytes() + 16 * 3 + 64, dtype=torch.uint8, device="cuda:0")
    offs = b.pack_streams(buf.data_ptr(), buf.numel())
    b.sync()
    host = buf.cpu().numpy()
    for i in range(3):
        f = flofile.parse(b.fetch(i))
        assert host[offs[i]:offs[i] + f.data_size].tobytes() == f.data and offs[i] % 16 == 0
    # the finished files (header, TOC, CRC made on the device) pack the same way and equal what fetch returns
    fbuf = torch.empty(b.data_bytes() + 3 * (74 + 20 * 64 + 16) + 64, dtype=torch.uint8, device="cuda:0")
    foffs = b.pack_files(fbuf.data_ptr(), fbuf.numel())
    b.sync()
    fhost = fbuf.cpu().numpy()
    for i in range(3):
        whole = b.fetch(i)
        assert fhost[foffs[i]:foffs[i] + len(whole)].tobytes() == whole and foffs[i] % 16 == 0
        assert flofile.parse(whole).crc_valid
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        got, sizes = gather_payloads(dist, buf[: offs[-1]], 0, 1, 0)
        assert sizes == [offs[-1]] and got[0].data_ptr() == buf.data_ptr()
        # the per-step object bench.py uses for N > 1: packs finished files, double-buffered, pipelined
        from dist_ref import BitstreamGather
        g = BitstreamGather(ctx, b, dist, 0, 1, 0)
        for _ in range(3):
            b.encode(0)
            b.sync()
            foffs2 = g.run()
            assert foffs2 == foffs
        outs = g.flush()
        assert g.last_total == foffs[-1]
        newest = [o for o in outs if o is not None][-1][0].cpu().numpy()
        assert newest[foffs[1]:foffs[1] + len(b.fetch(1))].tobytes() == b.fetch(1)
    finally:
        dist.destroy_process_group()
    b.close()


def test_blocked_scan_matches_sequential_chain_on_long_decays(ctx):
    # frame-parallel form uses a blocked temporal scan (64-frame warm-up); the chain kernel the true recurrence.
    # Loud burst, then long near-silence (levels decay through hundreds of frames), then another burst.
    sr, n = 44100, 44100 * 12
    x = signals.fast_noise(n * 2, 4, 1e-4)
    x[: sr] += signals.music_like(sr, sr // 2, 2, seed=2)[: sr] * 2.0
    x[8 * sr * 2: 8 * sr * 2 + sr] += signals.music_like(sr, sr // 2, 2, seed=3)[: sr]
    x = np.clip(x, -1, 1).astype(np.float32)
    ctx.force_path(1)
    a = ctx.encode_lossy(x, sr, 2, 0.55)
    ctx.force_path(2)
    b = ctx.encode_lossy(x, sr, 2, 0.55)
    ctx.force_path(0)
    assert a == b and len(flofile.parse(a).frames) == (n + 1024 + 1023) // 1024


@pytest.mark.parametrize("ch", [3, 6, 8])
def test_more_than_two_channels(ctx, ch):
    # beyond stereo the generic frame-parallel kernels walk the channels of a frame one after the other
    sr, n = 44100, 30000
    pcm = signals.music_like(sr, n, ch, seed=70 + ch)
    for q in (0.35, 1.0):
        compare_lossy_stage(ctx.lossy_analyze(pcm, sr, ch, q), O.lossy_analyze(pcm, sr, ch, q), sr, f"ch{ch} q{q}")
        g = ctx.encode_lossy(pcm, sr, ch, q)
        o = O.encode_lossy(pcm, sr, ch, q)
        fg, fo = same_structure(g, o)
        assert fg.channels == ch and len(fg.frames) == (n + 1024 + 1023) // 1024
        dg, do = ctx.decode(g), O.decode(g)[0]
        assert dg.shape == do.shape and np.max(np.abs(dg - do)) <= 2e-6
    with pytest.raises(Exception):
        ctx.encode_lossy(signals.music_like(sr, 2000, 9, seed=1), sr, 9, 0.55)


def test_api_misuse_is_reported_not_crashed(ctx):
    import torch
    import flo_amd
    b = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [2048 * 2, 4096 * 2], 44100, 2, 0.55)
    with pytest.raises(flo_amd.FloError, match="flo_batch_encode"):
        b.fetch(0)                                      # nothing encoded yet
    with pytest.raises(flo_amd.FloError, match="flo_batch_encode"):
        b.pack_files(0, 0)
    b.fill_synthetic()
    b.encode(0)
    with pytest.raises(flo_amd.FloError, match="flo_batch_encode"):
        b.fetch(0)                                      # encoded but not synced
    b.sync()
    small = torch.empty(64, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(flo_amd.FloError, match="too small"):
        b.pack_files(small.data_ptr(), small.numel())
    with pytest.raises(flo_amd.FloError, match="too small"):
        b.decode_to(small.data_ptr(), 1)
    with pytest.raises(flo_amd.FloError):
        b.encode(7)                                     # unknown kernel form
    assert flofile.parse(b.fetch(1)).crc_valid          # the batch is still usable
    b.close()
    with pytest.raises(flo_amd.FloError):
        flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [100], 0, 2, 0.55)       # sample_rate 0
    with pytest.raises(flo_amd.FloError):
        flo_amd.Batch(ctx, 5, [100], 44100, 2, 0.55)                    # unknown mode
    lb = flo_amd.Batch(ctx, flo_amd.MODE_LOSSLESS, [1000], 44100, 2, 5)
    lb.fill_synthetic()
    lb.encode(0)
    lb.sync()
    with pytest.raises(flo_amd.FloError, match="too small"):
        lb.decode_to(small.data_ptr(), 1)               # lossless batches decode too; the capacity is checked alike
    lb.close()


@pytest.mark.parametrize("amp", [1.0, 3000.0, 1e12])
def test_kernel_forms_agree_on_loud_and_lopsided_stereo(ctx, amp):
    """The lock-step forms put channel 1's bands on lanes 32..56 of the masking pass; far beyond full scale that pass
    walks all 24 band distances (the rare branch of spread_threshold), and a silent or much quieter channel next to a
    loud one is where a leak between the halves would show. Every form must give the same files."""
    import flo_amd
    sr, ch = 44100, 2
    clips = []
    for i in range(60):
        x = signals.music_like(sr, 12000 + 37 * i, ch, seed=i) * amp
        if i % 3 == 0:
            x[1::2] *= 1e-3
        if i % 5 == 0:
            x[::2] = 0.0
        clips.append(x.astype(np.float32))
    outs = {}
    for form in (5, 1, 2):
        b = flo_amd.Batch(ctx, flo_amd.MODE_LOSSY, [c.size for c in clips], sr, ch, 0.55)
        for i, c in enumerate(clips):
            b.upload(i, c)
        b.encode(form)
        b.sync()
        outs[form] = [b.fetch(i) for i in range(len(clips))]
        b.close()
    for form in (1, 2):
        assert outs[form] == outs[5], form


def test_transform_encoder_frame_methods(ctx):
    """TransformEncoder::{encode_frame, quantize_coefficients, reset} of the mirror class (lossy/encoder.rs:63,109,157), driven the
    way encode_to_flo drives them (encoder.rs:174-225: 1024 zeros of pre-roll, blocks of 2048 at a hop of 1024): frame by
    frame the integers and scale words must be what the whole-clip device pass and the oracle produce - also beyond the 65
    frames of history the mirror keeps for the temporal masking state."""
    import flo_amd
    sr, ch, q = 44100, 2, 0.55
    n = 80 * 1024 + 300
    pcm = signals.music_like(sr, n, ch, seed=91)
    o = O.lossy_analyze(pcm, sr, ch, q)
    g = ctx.lossy_analyze(pcm, sr, ch, q)
    hops = o["q"].shape[0]
    padded = np.zeros(((hops + 1) * 1024, ch), np.float32)
    padded[1024:1024 + n] = pcm.reshape(-1, ch)
    enc = flo_amd.TransformEncoder(sr, ch, q, ctx)
    band = O.psy_tables(sr)[1]
    for h in range(hops):
        fr = enc.encode_frame(padded[h * 1024:h * 1024 + 2048].reshape(-1))
        assert fr.block_size == 2048 and fr.num_samples == 1024 and len(fr.coefficients) == ch
        for c in range(ch):
            assert np.array_equal(fr.coefficients[c], g["q"][h, c]), (h, c)          # the device's own whole-clip pass, bit for bit
            assert np.array_equal(fr.scale_words[c], g["sf_words"][h, c]), (h, c)
            bm = np.array([np.abs(g["coeffs"][h, c][band == b]).max(initial=0.0) for b in range(25)], np.float32)
            want = np.where(bm > 1e-10, np.float32(30000.0) / np.where(bm > 1e-10, bm, 1).astype(np.float32), np.float32(1.0)).astype(np.float32)
            assert np.array_equal(fr.scale_factors[c], want), (h, c)                  # 30000 / band_max, IEEE f32
    stage = dict(coeffs=g["coeffs"], q=g["q"], sf_words=g["sf_words"])
    compare_lossy_stage(stage, o, sr, tag="encode_frame")                             # and the oracle, within the stage tolerances
    # reset(): the temporal state is gone - the next frame is encoded like a clip's first
    enc.reset()
    first = enc.encode_frame(padded[40 * 1024:40 * 1024 + 2048].reshape(-1))
    fresh = flo_amd.TransformEncoder(sr, ch, q, ctx).encode_frame(padded[40 * 1024:40 * 1024 + 2048].reshape(-1))
    assert all(np.array_equal(a, b) for a, b in zip(first.coefficients, fresh.coefficients))
    # quantize_coefficients with the caller's own signal-to-mask ratios (encoder.rs:109-154)
    rng = np.random.default_rng(3)
    coeffs = g["coeffs"][7, 0].copy()
    coeffs[100] = np.nan
    smr = rng.uniform(-80, 20, 1024).astype(np.float32)
    for quality in (0.0, 0.55, 1.0):
        enc.set_quality(quality)
        qv, sf = enc.quantize_coefficients(coeffs, smr)
        t = max(1.0 - quality, 0.001)
        thr = np.float32(-100.0) if quality >= 0.99 else np.float32(-60.0) * (np.float32(1.0) - np.float32(t) ** np.float32(0.5))
        bm = np.array([np.nanmax(np.abs(coeffs[band == b]), initial=0.0) for b in range(25)], np.float32)
        want_sf = np.where(bm > 1e-10, np.float32(30000.0) / np.where(bm > 1e-10, bm, 1).astype(np.float32), np.float32(1.0)).astype(np.float32)
        assert np.array_equal(sf, want_sf)
        scaled = (coeffs * want_sf[band]).astype(np.float32)
        tr = np.trunc(scaled)
        dd = (scaled - tr).astype(np.float32)                                          # exact
        r = np.where(np.isnan(scaled), 0, tr + np.trunc(dd + dd))                      # f32::round: half away from zero
        want_q = np.where(smr > thr, np.clip(r, -32768, 32767), 0).astype(np.int16)
        assert np.array_equal(qv, want_q), quality
