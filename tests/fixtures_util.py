"""Helpers shared by the oracle and GPU fixture tests: regenerate encoder inputs from reference-made files."""
import numpy as np

from conftest import example_bytes
from oracle import oracle as O

LOSSLESS_EXAMPLES = ["chord_cmajor_stereo", "multitone_stereo", "sine_440hz_mono", "sweep_20_20k", "hires_96khz",
                     "telephone_8khz", "click_track_120bpm", "dtmf_tones", "silence_1sec", "white_noise"]
# (file, CLI quality as f32 — reflo/src/main.rs:236-242, source lossless sibling)
LOSSY_EXAMPLES = [("lossy_chord_low", 0.2, "chord_cmajor_stereo"), ("lossy_chord_medium", 0.4, "chord_cmajor_stereo"),
                  ("lossy_chord_high", 0.6, "chord_cmajor_stereo"), ("lossy_chord_veryhigh", 0.8, "chord_cmajor_stereo"),
                  ("lossy_chord_transparent", 1.0, "chord_cmajor_stereo"), ("lossy_music_pattern", 0.6, "multitone_stereo")]


def lossless_input_for(name):
    """f32 PCM that maps back to the fixture's integers under f32_to_i32 (truncation): (v + 0.5 sign v)/32767.
    silence_1sec / white_noise hold all-zero integers but were NOT Silence frames (dither-level floats):
    feed 3e-5 which truncates to 0 yet fails the |s| < 1e-7 silence test."""
    b = example_bytes(name + ".flo")
    ints, sr, ch = O.decode_lossless_i32(b)
    v = ints.astype(np.float64)
    f = ((v + 0.5 * np.sign(v)) / 32767.0).astype(np.float32)
    if name in ("silence_1sec", "white_noise"):
        f = np.full_like(f, 3e-5)
    return b, f, ints, sr, ch


def lossy_source_pcm(src_name):
    """The 16-bit source v is recoverable from the lossless integer i = trunc(v*32767/32768): v = i + sign(i);
    the lossy encoder saw v/32768 (reflo/src/audio.rs:247-253). Ambiguous only for |v| <= 1."""
    ints, sr, ch = O.decode_lossless_i32(example_bytes(src_name + ".flo"))
    v = ints + np.sign(ints)
    return (v / 32768.0).astype(np.float32), sr, ch


def dequantise(q, sf_words, band):
    """lossy/decoder.rs:96-99 + :42-44 — decoder-side spectrum from integers and log-scale words."""
    sf = np.where(sf_words > 0, np.exp2((sf_words.astype(np.float32) - 32768.0) / 256.0), 0.0).astype(np.float32)
    s = sf[..., band]
    return np.where(s > 0, q / np.where(s > 0, s, 1), 0.0)
