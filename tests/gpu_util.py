import numpy as np
import pytest

import flofile
from fixtures_util import dequantise
from oracle import oracle as O


@pytest.fixture(scope="session")
def ctx():
    import flo_amd
    c = flo_amd.Context(0)
    yield c
    c.close()


def compare_lossy_stage(g, o, sr, tag=""):
    """SURVEY §8c tolerances (i)-(iv) for device-vs-oracle on identical input. g/o: dicts with coeffs,q,sf_words."""
    co, cg = o["coeffs"].astype(np.float64), g["coeffs"].astype(np.float64)
    rel = np.sqrt(((cg - co) ** 2).sum() / max((co ** 2).sum(), 1e-30))
    assert rel <= 1e-5, (tag, "coefficient relative RMS", rel)
    qo, qg = o["q"].astype(np.int32), g["q"].astype(np.int32)
    flips = int(((qo != 0) != (qg != 0)).sum())
    assert flips <= 5e-4 * qo.size, (tag, "keep/drop flips", flips, qo.size)
    # integers kept by both: an f32 FFT carries ABSOLUTE rounding noise of ~1e-7 x the frame's largest coefficient
    # (the oracle's FFT does too); in quantised units that noise is multiplied by the band's gain 30000/band_max.
    # So |dq| <= 1 + noise*gain everywhere, and where noise*gain is small (strong bands) mismatches stay under 1 %.
    both = (qo != 0) & (qg != 0)
    if both.any():
        band_g = O.psy_tables(sr)[1]
        gain = np.where(o["sf"] > 0, o["sf"], 0.0)[..., band_g] if "sf" in o else None
        d = np.abs(qo - qg)
        if gain is not None:
            fmax = np.abs(co).max(axis=2, keepdims=True)
            noise_q = 2e-6 * fmax * gain
            assert (d[both] <= 1 + noise_q[both]).all(), (tag, "kept mismatches beyond FFT noise", d[both].max())
            strong = both & (noise_q < 0.05)
            if strong.any():
                assert (d[strong] != 0).mean() <= 0.01, (tag, "mismatch rate in strong bands", (d[strong] != 0).mean())
        else:
            assert d[both].max() <= 1
    # scale words (256 steps per octave of 30000 / band_max): identical or +-1, except where the band's largest
    # coefficient is itself at the FFT's noise floor (a band of one or two bins far below the frame's peak, as the
    # lowest bands are at 128 kHz and up): the same absolute noise of ~1e-7 x frame maximum then moves the word by
    # 256 log2(1 + noise / band_max)
    sw = np.abs(o["sf_words"].astype(np.int32) - g["sf_words"].astype(np.int32))
    if "sf" in o:
        fmax = np.abs(co).max(axis=2, keepdims=True)                      # [hops][ch][1]
        bmax = np.where(o["sf"] > 0, 30000.0 / np.maximum(o["sf"], 1e-30), np.inf)   # [hops][ch][25]
        allow = 1 + 256 * np.log2(1 + 2e-6 * fmax / bmax)
        assert (sw <= allow + 1e-9).all(), (tag, "scale words", sw.max(), float((sw - allow).max()))
        assert (sw > 1).mean() <= 0.01, (tag, "scale words off by more than one", float((sw > 1).mean()))
    else:
        assert sw.max() <= 1, (tag, "scale words", sw.max())
    band = O.psy_tables(sr)[1]
    do, dg = dequantise(qo, o["sf_words"], band), dequantise(qg, g["sf_words"], band)
    den = np.sqrt((do ** 2).mean())
    if den > 0:
        db = 20 * np.log10(max(np.sqrt(((do - dg) ** 2).mean()), 1e-30) / den)
        assert db <= -80.0, (tag, "spectral RMS dB", db)
    return dict(rel=rel, flips=flips)


def same_structure(a: bytes, b: bytes):
    fa, fb = flofile.parse(a), flofile.parse(b)
    assert fa.crc_valid and fb.crc_valid
    for k in ("version", "flags", "sample_rate", "channels", "bit_depth", "total_samples", "level", "toc_size", "meta"):
        assert getattr(fa, k) == getattr(fb, k), k
    assert len(fa.frames) == len(fb.frames)
    assert [(f.frame_type, f.frame_samples, f.flags) for f in fa.frames] == [(f.frame_type, f.frame_samples, f.flags) for f in fb.frames]
    assert [t[3] for t in fa.toc] == [t[3] for t in fb.toc]
    return fa, fb


def snr_db(ref, x):
    ref, x = ref.astype(np.float64), x.astype(np.float64)
    n = min(ref.size, x.size)
    noise = ((ref[:n] - x[:n]) ** 2).sum()
    return 10 * np.log10(max((ref[:n] ** 2).sum(), 1e-30) / max(noise, 1e-30))
