"""The product's container reader (host C++, the front half of flo_decode) against the oracle's restatement of
Reader::read on the reference-made files and on thousands of damaged variants of them: both must accept or reject
the same inputs, with the same message, and agree on every header field and on the frame census. Runs without a GPU."""
import numpy as np
import pytest

import flo_amd
from conftest import example_bytes
from fixtures_util import LOSSLESS_EXAMPLES, LOSSY_EXAMPLES
from oracle import oracle as O

FILES = LOSSLESS_EXAMPLES + [n for n, _, _ in LOSSY_EXAMPLES] + ["audio_lossless", "audio_lossy"]


def _both(b):
    try:
        g = flo_amd.probe_container(b)
        gerr = None
    except flo_amd.FloError as e:
        g, gerr = None, str(e)
    try:
        o = O.info(b)
        oerr = None
    except RuntimeError as e:
        o, oerr = None, str(e)
    return g, gerr, o, oerr


def _same(b, tag):
    g, gerr, o, oerr = _both(b)
    assert (gerr is None) == (oerr is None), (tag, gerr, oerr)
    if gerr is not None:
        assert gerr == oerr, (tag, gerr, oerr)
        return False
    for f in ("version_major", "version_minor", "flags", "sample_rate", "channels", "bit_depth", "total_samples",
              "compression_level", "data_crc32", "data_size"):
        assert getattr(g, f) == getattr(o, f), (tag, f)
    assert g.n_frames == o.num_frames, tag
    return True


@pytest.mark.parametrize("name", FILES)
def test_reference_made_files(name):
    b = example_bytes(name + ".flo")
    assert _same(b, name)
    g = flo_amd.probe_container(b)
    assert g.is_transform == (1 if name.startswith("lossy") or name == "audio_lossy" else 0)
    assert g.frame_samples_sum == g.total_samples


@pytest.mark.parametrize("name", ["chord_cmajor_stereo", "lossy_chord_high", "telephone_8khz", "audio_lossless"])
def test_damaged_files_are_judged_like_the_reference_reader_judges_them(name):
    good = example_bytes(name + ".flo")
    rng = np.random.default_rng(len(good))
    accepted = rejected = 0
    # truncations everywhere in the header and TOC, and a sample of them further in
    cuts = list(range(0, 130)) + [int(x) for x in rng.integers(130, len(good), 150)]
    for n in cuts:
        if _same(good[:n], (name, "cut", n)):
            accepted += 1
        else:
            rejected += 1
    # single-byte damage: every header byte, then random positions (TOC entries, frame headers, payloads)
    spots = list(range(0, 70)) + [int(x) for x in rng.integers(70, min(len(good), 4000), 400)]
    for pos in spots:
        for val in (0x00, 0xFF, int(rng.integers(0, 256))):
            bad = bytearray(good)
            bad[pos] = val
            if _same(bytes(bad), (name, "byte", pos, val)):
                accepted += 1
            else:
                rejected += 1
    assert accepted > 100 and rejected > 100          # the corpus exercises both verdicts


def test_argument_errors_and_empty_input():
    with pytest.raises(flo_amd.FloError, match="Unexpected end of file"):
        flo_amd.probe_container(b"")
    with pytest.raises(flo_amd.FloError, match="bad magic"):
        flo_amd.probe_container(b"RIFFxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxx")
