"""A `reflo`-shaped command line on top of the HIP library (python -m flo_amd.cli ...).

Mirrors the reference CLI (reflo/src/main.rs:19-93, 218-420): the same sub-commands, options, quality names and
printed fields for the parts that sit on this repository's path -
    encode  <in.wav> <out.flo> [--level N] [--lossy | --transform] [--quality low|medium|high|veryhigh|transparent]
                               [--bitrate KBPS]
    decode  <in.flo> <out.wav>
    info    <in.flo>
    validate <in.flo>
    metadata <in.flo> [--json]
    analysis <in.flo> [--waveform] [--spectrum] [--json]
Ingestion is WAV only (flo_amd/wav.py; the reference demuxes MP3/FLAC/OGG/AAC through symphonia, reflo/src/audio.rs:57-166).
`encode` writes the META chunk the reference CLI writes for an untagged file (reflo/src/lib.rs:202-283, flo_amd/meta.py):
length_ms, encoding_time, encoder_settings, flo_encoder_version, source_format (+ --title / --artist / --album) - the
reference-made Examples/*.flo are reproduced including META, the encoding time aside. Tags inside the source file
(RIFF INFO) are not carried over. `analysis` (reflo/src/main.rs:619-800) decodes the file and prints what flo_analyze
computes on the device: EBU R128 loudness, range, true peak and sample peak (core/ebu_r128.rs), and on request the
waveform peaks at 60 per second and the spectral fingerprint (core/analysis.rs).
The quality names map as in the reference CLI (main.rs:236-242): low 0.2, medium 0.4, high 0.6, veryhigh 0.8,
transparent 1.0 - NOT the QualityPreset values the library API uses (lossy/mod.rs:39-47).
"""
import argparse
import sys

import json
import struct

from . import api, meta
from .wav import WavError, read_wav_bytes, write_wav_bytes

QUALITY = {"low": 0.2, "medium": 0.4, "med": 0.4, "high": 0.6, "veryhigh": 0.8, "vh": 0.8, "transparent": 1.0, "trans": 1.0}
QUALITY_NAMES = ["Low", "Medium", "High", "VeryHigh", "Transparent"]


def _wav_source_format(audio_bytes: bytes) -> str:
    """reflo/src/audio.rs:106-119 (bytes input carries no extension): the codec decides - 16 / 24 / 32-bit integer PCM
    is "WAV", everything else "UNKNOWN" (float and 8-bit PCM are not in the reference's list)."""
    pos = 12
    while pos + 8 <= len(audio_bytes):
        cid, size = audio_bytes[pos:pos + 4], struct.unpack_from("<I", audio_bytes, pos + 4)[0]
        if cid == b"fmt " and size >= 16:
            tag, bits = struct.unpack_from("<H", audio_bytes, pos + 8)[0], struct.unpack_from("<H", audio_bytes, pos + 22)[0]
            if tag == 0xFFFE and size >= 26:
                tag = struct.unpack_from("<H", audio_bytes, pos + 32)[0]
            return "WAV" if tag == 1 and bits in (16, 24, 32) else "UNKNOWN"
        pos += 8 + size + (size & 1)
    return "UNKNOWN"


def encode_from_audio(audio_bytes: bytes, level=5, lossy=False, quality=0.6, bitrate=None, ctx=None, title=None, artist=None,
                      album=None, encoding_time=None) -> bytes:
    """reflo::encode_from_audio (reflo/src/lib.rs:183-306) for WAV input."""
    samples, sr, ch = read_wav_bytes(audio_bytes)
    c = ctx or api.default_context()
    level = min(int(level), 9)
    is_lossy = bool(lossy or bitrate is not None)
    quality = min(max(float(quality), 0.0), 1.0)
    mb = meta.cli_metadata(samples.size, sr, ch, _wav_source_format(audio_bytes), is_lossy, quality, bitrate, level,
                           title, artist, album, encoding_time)
    if is_lossy:
        q = api.QualityPreset.from_bitrate(bitrate, sr, ch).as_f32() if bitrate is not None else quality
        return api.TransformEncoder(sr, ch, q, c).encode_to_flo(samples, mb)
    return api.Encoder(sr, ch, 16, c).with_compression(level).encode(samples, mb)


def get_metadata(flo_bytes: bytes):
    """reflo::get_metadata: the decoded META chunk (a dict), or None when the file has none."""
    i = api.probe_container(flo_bytes)
    meta_size = int.from_bytes(flo_bytes[62:70], "little")
    start = i.data_start + i.data_size + int.from_bytes(flo_bytes[54:62], "little")
    if meta_size == 0 or start + meta_size > len(flo_bytes):
        return None
    return meta.unpack(flo_bytes[start:start + meta_size])


def decode_to_wav(flo_bytes: bytes, ctx=None) -> bytes:
    """reflo::decode_to_wav: libflo::decode, then a 32-bit float WAV (reflo/src/audio.rs:290-320)."""
    c = ctx or api.default_context()
    pcm, sr, ch = c.decode(flo_bytes, with_info=True)
    return write_wav_bytes(pcm, sr, ch)


def flo_info(flo_bytes: bytes) -> dict:
    """reflo::get_flo_info (reflo/src/lib.rs:40-93): header fields, duration, compression ratio, CRC check."""
    import zlib
    i = api.probe_container(flo_bytes)
    data = flo_bytes[i.data_start:i.data_start + i.data_size]
    duration = i.total_samples / i.sample_rate if i.sample_rate else 0.0
    raw = i.total_samples * i.channels * (i.bit_depth // 8)
    return dict(version=f"{i.version_major}.{i.version_minor}", sample_rate=i.sample_rate, channels=i.channels,
                bit_depth=i.bit_depth, total_samples=i.total_samples, duration_secs=duration, file_size=len(flo_bytes),
                compression_ratio=(raw / len(flo_bytes)) if len(flo_bytes) else 0.0,
                crc_valid=(zlib.crc32(data) & 0xFFFFFFFF) == i.data_crc32, is_lossy=bool(i.flags & 1),
                lossy_quality=(i.flags >> 8) & 0x0F, compression_level=i.compression_level)


def analysis_report(flo_bytes: bytes, waveform=False, spectrum=False, ctx=None) -> dict:
    """What reflo's `analysis` command gathers (reflo/src/main.rs:619-735): file info, compute_ebu_r128_loudness on the
    decoded samples, optionally extract_waveform_peaks at 60 peaks per second and extract_spectral_fingerprint."""
    import numpy as np
    c = ctx or api.default_context()
    info = flo_info(flo_bytes)
    pcm, sr, ch = c.decode(flo_bytes, with_info=True)
    a = c.analyze(pcm, info["sample_rate"], info["channels"], 60)
    out = {"file_info": {"sample_rate": info["sample_rate"], "channels": info["channels"], "bit_depth": info["bit_depth"],
                         "duration_secs": info["duration_secs"], "total_samples": info["total_samples"]},
           "loudness": {"integrated_lufs": a["integrated_lufs"], "loudness_range_lu": a["loudness_range_lu"],
                        "true_peak_dbtp": a["true_peak_dbtp"], "sample_peak_dbfs": a["sample_peak_dbfs"]},
           "waveform": None, "spectral": None}
    if waveform:
        pk = a["peaks"]
        stats = None
        if pk.size:
            # (the reference averages with a sequential f32 sum: main.rs:662)
            acc = np.float32(0.0)
            for v in pk:
                acc = np.float32(acc + v)
            stats = {"min": float(pk.min()), "max": float(pk.max()), "average": float(acc / np.float32(pk.size))}
        out["waveform"] = {"peaks_per_second": 60, "total_peaks": int(pk.size), "channels": info["channels"], "peak_statistics": stats}
    if spectrum:
        out["spectral"] = {"duration_ms": a["duration_ms"], "sample_rate": a["sample_rate"], "channels": a["channels"],
                           "peak_frequency_bands": list(a["frequency_peaks"]), "energy_profile": list(a["energy_profile"]),
                           "average_loudness": a["avg_loudness"], "spectral_hash_hex": a["hash"][:8].hex()}
    return out


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="flo", description="flo audio format converter (MI355X-native encode / decode)")
    sub = ap.add_subparsers(dest="command", required=True)
    e = sub.add_parser("encode", help="Encode a WAV file to flo format")
    e.add_argument("input")
    e.add_argument("output")
    e.add_argument("-l", "--level", type=int, default=5, help="Compression level (0-9, default 5)")
    e.add_argument("--lossy", action="store_true", help="Enable lossy compression mode")
    e.add_argument("--transform", action="store_true", help="Use transform-based lossy")
    e.add_argument("--quality", default="high", help="Lossy quality level (low, medium, high, veryhigh, transparent)")
    e.add_argument("--bitrate", type=int, default=None, help="Target bitrate in kbps (alternative to quality)")
    e.add_argument("--title", default=None, help="Title metadata")
    e.add_argument("--artist", default=None, help="Artist metadata")
    e.add_argument("--album", default=None, help="Album metadata")
    d = sub.add_parser("decode", help="Decode a flo file to WAV")
    d.add_argument("input")
    d.add_argument("output")
    i = sub.add_parser("info", help="Show information about a flo file")
    i.add_argument("input")
    m = sub.add_parser("metadata", help="Display metadata from a flo file")
    m.add_argument("input")
    m.add_argument("--json", action="store_true", help="Output as JSON")
    v = sub.add_parser("validate", help="Validate a flo file")
    v.add_argument("input")
    an = sub.add_parser("analysis", help="Analyze audio with waveform peaks and spectral fingerprinting")
    an.add_argument("input")
    an.add_argument("-w", "--waveform", action="store_true", help="Show waveform peaks")
    an.add_argument("-s", "--spectrum", action="store_true", help="Show spectral fingerprint")
    an.add_argument("--json", action="store_true", help="Output as JSON")
    a = ap.parse_args(argv)
    try:
        if a.command == "encode":
            print(f"Reading {a.input}...")
            audio = open(a.input, "rb").read()
            samples, sr, ch = read_wav_bytes(audio)
            print(f"  Sample rate: {sr} Hz")
            print(f"  Channels: {ch}")
            print(f"  Duration: {samples.size / ch / sr:.2f}s")
            lossy = a.lossy or a.transform
            if lossy or a.bitrate is not None:
                if a.bitrate is not None:
                    print(f"Encoding to flo (lossy, ~{a.bitrate} kbps)...")
                    q = None
                else:
                    if a.quality.lower() not in QUALITY:
                        print(f"Invalid quality level: {a.quality}. Use: low, medium, high, veryhigh, transparent", file=sys.stderr)
                        return 1
                    q = QUALITY[a.quality.lower()]
                    print(f"Encoding to flo (lossy, {a.quality} quality)...")
                flo = encode_from_audio(audio, a.level, True, q if q is not None else 0.6, a.bitrate, None, a.title, a.artist, a.album)
            else:
                print("Encoding to flo (lossless)...")
                flo = encode_from_audio(audio, a.level, title=a.title, artist=a.artist, album=a.album)
            open(a.output, "wb").write(flo)
            original = int(samples.size * 4)
            print("Done!")
            print(f"  Output: {a.output}")
            print(f"  Size: {len(flo)} bytes ({original / max(len(flo), 1):.1f}x compression)")
        elif a.command == "decode":
            print(f"Reading {a.input}...")
            flo = open(a.input, "rb").read()
            info = flo_info(flo)
            print(f"  Sample rate: {info['sample_rate']} Hz")
            print(f"  Channels: {info['channels']}")
            print(f"  Duration: {info['duration_secs']:.2f}s")
            print("Decoding...")
            wav = decode_to_wav(flo)
            print("Writing WAV...")
            open(a.output, "wb").write(wav)
            print("Done!")
            print(f"  Output: {a.output}")
        elif a.command == "info":
            info = flo_info(open(a.input, "rb").read())
            print("flo Audio File")
            print("-" * 31)
            print(f"  Version:     {info['version']}")
            print(f"  Sample rate: {info['sample_rate']} Hz")
            print(f"  Channels:    {info['channels']}")
            print(f"  Bit depth:   {info['bit_depth']}")
            print(f"  Duration:    {info['duration_secs']:.2f}s")
            print(f"  Total sample-frames: {info['total_samples']}")
            print(f"  File size:   {info['file_size']} bytes")
            print(f"  Compression: {info['compression_ratio']:.1f}x")
            print(f"  CRC valid:   {'yes' if info['crc_valid'] else 'no'}")
            if info["is_lossy"]:
                ql = info["lossy_quality"]
                print(f"  Encoding:    Lossy ({QUALITY_NAMES[ql] if ql < len(QUALITY_NAMES) else 'Unknown'})")
            else:
                print("  Encoding:    Lossless")
        elif a.command == "metadata":
            md = get_metadata(open(a.input, "rb").read())
            if md is None:
                print("null" if a.json else "No metadata present")
            elif a.json:
                print(json.dumps(md, indent=2, default=lambda b: f"<{len(b)} bytes>"))
            else:
                for k, v in md.items():
                    shown = f"<{len(v)} bytes>" if isinstance(v, (bytes, list)) and len(v) > 16 else v
                    print(f"  {k}: {shown}")
        elif a.command == "analysis":
            rep = analysis_report(open(a.input, "rb").read(), a.waveform, a.spectrum)
            if a.json:
                print(json.dumps(rep, indent=2))
            else:
                fi, lo = rep["file_info"], rep["loudness"]
                print(f"Analyzing {a.input}...")
                print()
                print("File Information")
                print("\u2500" * 16)
                print(f"  Sample rate: {fi['sample_rate']} Hz")
                print(f"  Channels:    {fi['channels']}")
                print(f"  Bit depth:   {fi['bit_depth']} bits")
                print(f"  Duration:    {fi['duration_secs']:.2f}s")
                print(f"  Total samples: {fi['total_samples']}")
                print()
                print("Loudness Metrics (EBU R128)")
                print("\u2500" * 28)
                print(f"  Integrated loudness: {lo['integrated_lufs']:.2f} LUFS")
                print(f"  Loudness range:      {lo['loudness_range_lu']:.2f} LU")
                print(f"  True peak:           {lo['true_peak_dbtp']:.2f} dBTP")
                print(f"  Sample peak:         {lo['sample_peak_dbfs']:.2f} dBFS")
                print()
                if rep["waveform"]:
                    wf = rep["waveform"]
                    print("Waveform Analysis")
                    print("\u2500" * 17)
                    print(f"  Peaks per second:    {wf['peaks_per_second']}")
                    print(f"  Total peaks:         {wf['total_peaks']}")
                    print(f"  Channels:            {wf['channels']}")
                    if wf["peak_statistics"]:
                        st = wf["peak_statistics"]
                        print("  Peak statistics:")
                        print(f"    Min:               {st['min']:.6f}")
                        print(f"    Max:               {st['max']:.6f}")
                        print(f"    Average:           {st['average']:.6f}")
                    print()
                if rep["spectral"]:
                    sp = rep["spectral"]
                    print("Spectral Analysis")
                    print("\u2500" * 17)
                    print(f"  Duration:            {sp['duration_ms']} ms")
                    print(f"  Sample rate:         {sp['sample_rate']} Hz")
                    print(f"  Channels:            {sp['channels']}")
                    print(f"  Peak frequency bands: {sp['peak_frequency_bands']}")
                    print(f"  Energy profile (16 bands): {sp['energy_profile']}")
                    print(f"  Average loudness:    {sp['average_loudness']}")
                    print(f"  Spectral hash (first 8 bytes):   {sp['spectral_hash_hex']}")
                    print()
        elif a.command == "validate":
            try:
                ok = flo_info(open(a.input, "rb").read())["crc_valid"]
            except api.FloError:
                ok = False
            print("Valid flo file" if ok else "Invalid flo file")
            return 0 if ok else 1
    except (OSError, WavError, ValueError, api.FloError) as ex:
        print(f"Error: {ex}", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
