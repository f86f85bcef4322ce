"""SURVEY 8f-4 on the device: the reflo-shaped CLI end to end, and the streaming encoder against the oracle's."""
import numpy as np
import pytest

import flo_amd
import signals
from conftest import example_bytes
from flo_amd import cli
from flo_amd.wav import read_wav_bytes, write_wav_bytes
from gpu_util import ctx, same_structure  # noqa: F401
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def test_cli_encode_decode_of_the_reference_example(tmp_path, capsys, ctx):
    # BASELINE configs[0]: Examples/audio.wav -> lossless .flo; header + TOC + DATA equal the reference-made file
    wav = tmp_path / "audio.wav"
    wav.write_bytes(example_bytes("audio.wav"))
    out = tmp_path / "audio.flo"
    assert cli.main(["encode", str(wav), str(out)]) == 0
    ref = example_bytes("audio_lossless.flo")
    got = out.read_bytes()
    # the whole file, META included, is the file the reference tool wrote - the 20 characters of its encoding time aside
    assert len(got) == len(ref)
    t0 = ref.index(b"\xadencoding_time\xb4") + 15
    assert got[:t0] == ref[:t0] and got[t0 + 20:] == ref[t0 + 20:]
    assert cli.encode_from_audio(wav.read_bytes(), encoding_time="2026-03-09T20:46:05Z") == ref
    text = capsys.readouterr().out
    assert "Sample rate: 44100 Hz" in text and "Encoding to flo (lossless)..." in text
    back = tmp_path / "back.wav"
    assert cli.main(["decode", str(out), str(back)]) == 0
    x, sr, ch = read_wav_bytes(back.read_bytes())
    assert (sr, ch, x.size) == (44100, 2, 88200) and not x.any()
    # lossy with the CLI's quality names (high = 0.6, main.rs:236-242): DATA equals the reference-made lossy example
    lossy = tmp_path / "lossy.flo"
    assert cli.main(["encode", str(wav), str(lossy), "--lossy", "--quality", "high"]) == 0
    rl, gl = example_bytes("audio_lossy.flo"), lossy.read_bytes()
    toc, data = int.from_bytes(gl[38:46], "little"), int.from_bytes(gl[46:54], "little")
    assert gl[:38] == rl[:38] and gl[70 + toc:70 + toc + data] == rl[70 + toc:70 + toc + data]
    assert cli.get_metadata(gl)["encoder_settings"] == "Lossy, quality 60%" and len(gl) == len(rl)
    assert cli.main(["encode", str(wav), str(lossy), "--lossy", "--quality", "bogus"]) == 1
    assert cli.main(["info", str(lossy)]) == 0 and "Lossy (High)" in capsys.readouterr().out


def test_cli_on_16_bit_input_matches_the_oracle(tmp_path, ctx):
    import struct
    rng = np.random.default_rng(9)
    s16 = (signals.music_like(22050, 30000, 2, seed=4) * 30000).astype(np.int16)
    fmt = struct.pack("<HHIIHH", 1, 2, 22050, 22050 * 4, 4, 16)
    body = b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt + b"data" + struct.pack("<I", s16.nbytes) + s16.tobytes()
    wav = tmp_path / "in.wav"
    wav.write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)
    pcm = s16.astype(np.float32) * np.float32(1 / 32768.0)          # the reference's S16 rule (audio.rs:247-253)
    out = tmp_path / "o.flo"
    assert cli.main(["encode", str(wav), str(out), "--level", "7"]) == 0
    got = out.read_bytes()
    md = cli.get_metadata(got)
    assert md["source_format"] == "WAV" and md["encoder_settings"] == "Lossless, level 7" and md["length_ms"] == 30000 * 1000 // 22050
    assert got[: len(got) - int.from_bytes(got[62:70], "little")] == O.encode_lossless(pcm, 22050, 2, 16, 7, meta=b"x" * int.from_bytes(got[62:70], "little"))[: -int.from_bytes(got[62:70], "little")]
    assert cli.main(["encode", str(wav), str(out), "--bitrate", "128"]) == 0
    q = flo_amd.QualityPreset.from_bitrate(128, 22050, 2).as_f32()
    got = out.read_bytes()
    mb = got[len(got) - int.from_bytes(got[62:70], "little"):]
    assert cli.get_metadata(got)["encoder_settings"] == "Lossy, target 128kbps"
    same_structure(got, O.encode_lossy(pcm, 22050, 2, q, meta=mb))
    del rng


@pytest.mark.parametrize("sr,ch,level", [(8000, 2, 5), (44100, 1, 3), (22050, 2, 0), (48000, 6, 8)])
def test_streaming_encoder_equals_the_oracles(ctx, sr, ch, level):
    # libflo/src/streaming/encoder.rs: pushed in uneven pieces; every frame, every pull, flush and finalize byte-identical
    pcm = signals.music_like(sr, 3 * sr + 777, ch, seed=sr + ch)
    pcm[sr * ch:2 * sr * ch] = 0.0                       # one silent second (a Silence frame)
    g = flo_amd.StreamingEncoder(sr, ch, 16, ctx).with_compression(level)
    o = O.StreamingEncoder(sr, ch, 16, level)
    cuts = [0, 1000 * ch, (sr + 5) * ch, (2 * sr + sr // 2) * ch, pcm.size]
    for a, b in zip(cuts[:-1], cuts[1:]):
        g.push_samples(pcm[a:b])
        o.push_samples(pcm[a:b])
        assert g.pending_frames() == o.pending_frames() and g.pending_samples() == o.pending_samples()
    fg, fo = g.next_frame(), o.next_frame()
    assert (fg.index, fg.timestamp_ms, fg.samples, fg.data) == (fo["index"], fo["timestamp_ms"], fo["samples"], fo["data"])
    assert g.finalize(b"xyz") == o.finalize(b"xyz")
    assert g.next_frame() is None and g.flush() is None
    # flush returns the partial frame without queueing it
    g.push_samples(pcm[: 1234 * ch])
    o.push_samples(pcm[: 1234 * ch])
    fg, fo = g.flush(), o.flush()
    assert (fg.index, fg.timestamp_ms, fg.samples, fg.data) == (fo["index"], fo["timestamp_ms"], fo["samples"], fo["data"])
    assert g.pending_frames() == 0 and g.finalize() == o.finalize()
    g.close()


def test_cli_analysis_command(ctx, tmp_path, capsys):
    # reflo's `analysis` sub-command (reflo/src/main.rs:619-800) on a reference-made file: the numbers are flo_analyze's
    # (device) on the decoded samples and equal the oracle's restatement of compute_ebu_r128_loudness
    import json
    from flo_amd import cli
    from oracle import oracle as O
    src = example_bytes("chord_cmajor_stereo.flo")
    p = tmp_path / "c.flo"
    p.write_bytes(src)
    assert cli.main(["analysis", str(p), "--waveform", "--spectrum", "--json"]) == 0
    rep = json.loads(capsys.readouterr().out)
    pcm, sr, ch = O.decode(src)
    o = O.loudness_metrics(pcm, ch, sr)
    one_segment = pcm.size // ch <= 65536
    for k, v in o.items():
        assert rep["loudness"][k] == v if one_segment else abs(rep["loudness"][k] - v) <= 1e-9 * max(1.0, abs(v)), k
    assert rep["file_info"]["sample_rate"] == sr and rep["file_info"]["channels"] == ch
    fp = O.spectral_fingerprint(pcm, ch, sr)
    assert rep["spectral"]["spectral_hash_hex"] == fp["hash"][:8].hex() and rep["spectral"]["energy_profile"] == list(fp["energy_profile"])
    assert rep["waveform"]["total_peaks"] == O.waveform_peaks(pcm, ch, sr, 60).size and rep["waveform"]["peaks_per_second"] == 60
    assert cli.main(["analysis", str(p)]) == 0
    text = capsys.readouterr().out
    assert "Loudness Metrics (EBU R128)" in text and "Integrated loudness:" in text and "True peak:" in text
