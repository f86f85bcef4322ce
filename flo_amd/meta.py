"""The META chunk the reference CLI writes (reflo/src/lib.rs:202-283): a MessagePack map of the FloMetadata fields that
are set, in declaration order (rmp_serde::to_vec_named; every Option field is skipped when None,
libflo/src/core/metadata.rs:328-665). For a file without tags the reference sets exactly
    length_ms, encoding_time, encoder_settings, flo_encoder_version, source_format
(metadata.rs:442,463,518,654,658). Only these - plus title / artist / album, which the CLI accepts - are written here;
the rest of the metadata model (pictures, lyrics, analysis data ...) is out of this repository's scope.
"""
import struct
import time

import numpy as np

# declaration order of the fields this module can write (metadata.rs line numbers)
_ORDER = ["title", "artist", "album", "length_ms", "encoding_time", "encoder_settings", "flo_encoder_version", "source_format"]
ENCODER_VERSION = "reflo 0.1.2"       # format!("reflo {}", CARGO_PKG_VERSION): what files made by the reference tool carry


def _str(s: str) -> bytes:
    b = s.encode("utf-8")
    if len(b) < 32:
        return bytes([0xA0 | len(b)]) + b
    if len(b) < 256:
        return b"\xd9" + bytes([len(b)]) + b
    if len(b) < 65536:
        return b"\xda" + struct.pack(">H", len(b)) + b
    return b"\xdb" + struct.pack(">I", len(b)) + b


def _uint(v: int) -> bytes:
    if v < 128:
        return bytes([v])
    if v < 256:
        return b"\xcc" + bytes([v])
    if v < 65536:
        return b"\xcd" + struct.pack(">H", v)
    if v < 1 << 32:
        return b"\xce" + struct.pack(">I", v)
    return b"\xcf" + struct.pack(">Q", v)


def pack_fields(fields: dict) -> bytes:
    items = [(k, fields[k]) for k in _ORDER if fields.get(k) is not None]
    assert len(items) < 16
    out = bytes([0x80 | len(items)])
    for k, v in items:
        out += _str(k) + (_uint(v) if isinstance(v, int) else _str(v))
    return out


def encoder_settings(lossy: bool, quality: float, bitrate, level: int) -> str:
    """reflo/src/lib.rs:262-271"""
    if lossy or bitrate is not None:
        if bitrate is not None:
            return f"Lossy, target {bitrate}kbps"
        return f"Lossy, quality {float(np.float32(quality) * np.float32(100.0)):.0f}%"
    return f"Lossless, level {level}"


def cli_metadata(n_interleaved: int, sample_rate: int, channels: int, source_format: str, lossy: bool, quality: float,
                 bitrate, level: int, title=None, artist=None, album=None, encoding_time=None) -> bytes:
    """META bytes of a file encoded by the CLI (encoding_time: "%Y-%m-%dT%H:%M:%SZ" in UTC, now if None)."""
    total = n_interleaved // channels
    length_ms = int(total / sample_rate * 1000.0)          # (total as f64 / sample_rate as f64 * 1000.0) as u64
    if encoding_time is None:
        encoding_time = time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime())
    return pack_fields(dict(title=title, artist=artist, album=album, length_ms=length_ms, encoding_time=encoding_time,
                            encoder_settings=encoder_settings(lossy, quality, bitrate, level),
                            flo_encoder_version=ENCODER_VERSION, source_format=source_format))


def unpack(data: bytes):
    """MessagePack -> Python (maps, arrays, strings, binaries, integers, floats, booleans, nil): enough to show the META
    chunk of any .flo file, whoever wrote it. Raises ValueError on malformed input."""
    pos = 0

    def need(n):
        nonlocal pos
        if pos + n > len(data):
            raise ValueError("truncated MessagePack data")
        b = data[pos:pos + n]
        pos += n
        return b

    def one():
        t = need(1)[0]
        if t < 0x80:
            return t
        if t >= 0xE0:
            return t - 256
        if 0x80 <= t <= 0x8F:
            return {one(): one() for _ in range(t & 15)}
        if 0x90 <= t <= 0x9F:
            return [one() for _ in range(t & 15)]
        if 0xA0 <= t <= 0xBF:
            return need(t & 31).decode("utf-8", "replace")
        if t == 0xC0:
            return None
        if t in (0xC2, 0xC3):
            return t == 0xC3
        if t in (0xC4, 0xC5, 0xC6):
            n = int.from_bytes(need(1 << (t - 0xC4)), "big")
            return bytes(need(n))
        if t == 0xCA:
            return struct.unpack(">f", need(4))[0]
        if t == 0xCB:
            return struct.unpack(">d", need(8))[0]
        if 0xCC <= t <= 0xCF:
            return int.from_bytes(need(1 << (t - 0xCC)), "big")
        if 0xD0 <= t <= 0xD3:
            return int.from_bytes(need(1 << (t - 0xD0)), "big", signed=True)
        if t in (0xD9, 0xDA, 0xDB):
            n = int.from_bytes(need(1 << (t - 0xD9)), "big")
            return need(n).decode("utf-8", "replace")
        if t in (0xDC, 0xDD):
            n = int.from_bytes(need(2 if t == 0xDC else 4), "big")
            return [one() for _ in range(n)]
        if t in (0xDE, 0xDF):
            n = int.from_bytes(need(2 if t == 0xDE else 4), "big")
            return {one(): one() for _ in range(n)}
        raise ValueError(f"unsupported MessagePack type 0x{t:02x}")

    v = one()
    return v


# FloMetadata fields in declaration order (libflo/src/core/metadata.rs:328-665): the order rmp_serde::to_vec_named
# writes the ones that are set
FIELD_ORDER = """title subtitle content_group album original_album set_subtitle track_number track_total disc_number disc_total
isrc artist album_artist conductor remixer original_artist composer lyricist original_lyricist encoded_by involved_people
musician_credits genre mood bpm key language length_ms year recording_time release_time original_release_time encoding_time
tagging_time copyright produced_notice publisher file_owner radio_station radio_station_owner album_sort artist_sort title_sort
original_filename playlist_delay encoder_settings url_commercial url_copyright url_audio_file url_artist url_audio_source
url_radio_station url_payment url_publisher user_urls comments lyrics synced_lyrics pictures user_text play_count popularimeter
waveform_data spectrum_fingerprint bpm_map key_changes loudness_profile integrated_loudness_lufs loudness_range_lu
true_peak_dbtp section_markers creator_notes collaboration_credits remix_chain animated_cover cover_variants artist_signature
flo_encoder_version source_format custom""".split()


def _skip(data: bytes, pos: int) -> int:
    """end offset of the MessagePack value that starts at pos"""
    t = data[pos]
    if t < 0x80 or t >= 0xE0 or t in (0xC0, 0xC2, 0xC3):
        return pos + 1
    if 0xA0 <= t <= 0xBF:
        return pos + 1 + (t & 31)
    if 0x80 <= t <= 0x8F or 0x90 <= t <= 0x9F or t in (0xDC, 0xDD, 0xDE, 0xDF):
        if t <= 0x9F:
            n, p = t & 15, pos + 1
        else:
            w = 2 if t in (0xDC, 0xDE) else 4
            n, p = int.from_bytes(data[pos + 1:pos + 1 + w], "big"), pos + 1 + w
        for _ in range(n * (2 if (0x80 <= t <= 0x8F or t in (0xDE, 0xDF)) else 1)):
            p = _skip(data, p)
        return p
    if t in (0xC4, 0xC5, 0xC6, 0xD9, 0xDA, 0xDB):
        w = 1 << (t - (0xC4 if t <= 0xC6 else 0xD9))
        return pos + 1 + w + int.from_bytes(data[pos + 1:pos + 1 + w], "big")
    if t == 0xCA:
        return pos + 5
    if t == 0xCB:
        return pos + 9
    if 0xCC <= t <= 0xCF:
        return pos + 1 + (1 << (t - 0xCC))
    if 0xD0 <= t <= 0xD3:
        return pos + 1 + (1 << (t - 0xD0))
    raise ValueError(f"unsupported MessagePack type 0x{t:02x}")


def _top_level(data: bytes) -> dict:
    """top-level map as {key: raw bytes of the value}"""
    t = data[0]
    if 0x80 <= t <= 0x8F:
        n, pos = t & 15, 1
    elif t == 0xDE:
        n, pos = int.from_bytes(data[1:3], "big"), 3
    elif t == 0xDF:
        n, pos = int.from_bytes(data[1:5], "big"), 5
    else:
        raise ValueError("META is not a MessagePack map")
    out = {}
    for _ in range(n):
        kend = _skip(data, pos)
        key = unpack(data[pos:kend])
        vend = _skip(data, kend)
        out[key] = data[kend:vend]
        pos = vend
    return out


# What serde accepts for each FloMetadata field (core/metadata.rs:328-665): the MessagePack type families. A value of
# another family makes `from_slice::<FloMetadata>` fail, and lib.rs:228's `unwrap_or_default()` then drops ALL of the
# caller's metadata; empty sequences and maps are skipped on re-serialisation (`skip_serializing_if = "...::is_empty"`).
_STR = set("""title subtitle content_group album original_album set_subtitle isrc artist album_artist conductor remixer
original_artist composer lyricist original_lyricist encoded_by genre mood key language recording_time release_time
original_release_time encoding_time tagging_time copyright produced_notice publisher file_owner radio_station
radio_station_owner album_sort artist_sort title_sort original_filename encoder_settings url_commercial url_copyright
url_audio_file url_artist url_audio_source url_radio_station url_payment url_publisher flo_encoder_version
source_format""".split())
_U32 = set("track_number track_total disc_number disc_total bpm year playlist_delay".split())
_U64 = set("length_ms play_count".split())
_F32 = set("integrated_loudness_lufs loudness_range_lu true_peak_dbtp".split())
_SEQ = set("""involved_people musician_credits user_urls comments lyrics synced_lyrics pictures user_text bpm_map key_changes
loudness_profile section_markers creator_notes collaboration_credits remix_chain cover_variants""".split())
_SEQ_SKIP_EMPTY = _SEQ - {"involved_people", "musician_credits"}      # those two are Option<Vec<..>>: Some(vec![]) is kept
_STRUCT = set("popularimeter waveform_data animated_cover artist_signature".split())    # a map (named) or a sequence (compact)


def _family(raw: bytes) -> str:
    t = raw[0]
    if t <= 0x7F or 0xCC <= t <= 0xCF:
        return "uint"
    if t >= 0xE0 or 0xD0 <= t <= 0xD3:
        return "int"
    if 0xA0 <= t <= 0xBF or t in (0xD9, 0xDA, 0xDB):
        return "str"
    if 0x90 <= t <= 0x9F or t in (0xDC, 0xDD):
        return "seq"
    if 0x80 <= t <= 0x8F or t in (0xDE, 0xDF):
        return "map"
    if t in (0xC4, 0xC5, 0xC6):
        return "bin"
    if t in (0xCA, 0xCB):
        return "float"
    if t in (0xC2, 0xC3):
        return "bool"
    return "nil" if t == 0xC0 else "other"


def _field_ok(key: str, raw: bytes) -> bool:
    fam = _family(raw)
    if fam == "nil":
        return True          # every field is an Option or has a default
    if key in _STR:
        return fam == "str"
    if key in _U32 or key in _U64:
        if fam == "int":
            v = unpack(raw)
            return isinstance(v, int) and v >= 0
        if fam != "uint":
            return False
        return unpack(raw) < (1 << (32 if key in _U32 else 64))
    if key in _F32:
        return fam in ("float", "uint", "int")
    if key in _SEQ:
        return fam == "seq"
    if key in _STRUCT:
        return fam in ("map", "seq")
    if key == "spectrum_fingerprint":
        return fam in ("bin", "seq", "str")      # serde_bytes
    if key == "custom":
        return fam == "map"
    return True


def _is_empty(raw: bytes) -> bool:
    return raw in (b"\x90", b"\x80", b"\xdc\x00\x00", b"\xdd\x00\x00\x00\x00", b"\xde\x00\x00", b"\xdf\x00\x00\x00\x00")


def merge_analysis(user_meta: bytes, analysis_meta: bytes) -> bytes:
    """add_analysis_data_if_missing (lib.rs:219-283) for a caller who passes metadata of their own: the caller's fields
    stay (values re-emitted as they came), waveform_data / spectrum_fingerprint / loudness_profile are added only where
    missing, length_ms is always set; fields come out in FloMetadata's declaration order, unknown keys are dropped as
    serde drops them, empty sequences / maps are skipped as `skip_serializing_if` skips them, and input serde would reject
    (undecodable, or a known field of the wrong MessagePack type) counts as empty (`unwrap_or_default`). Values that
    serde would re-encode in another form (an f64 where the field is f32, a non-minimal integer) are passed through as
    they came."""
    fields = {}
    if user_meta:
        try:
            top = _top_level(user_meta)
            if not all(isinstance(k, str) for k in top):
                raise ValueError("a field name is not a string")
            known = {k: v for k, v in top.items() if k in FIELD_ORDER}
            if not all(_field_ok(k, v) for k, v in known.items()):
                raise ValueError("a field has the wrong type: serde rejects the document, unwrap_or_default() takes over")
            fields = {k: v for k, v in known.items()
                      if v != b"\xc0" and not ((k in _SEQ_SKIP_EMPTY or k == "custom") and _is_empty(v))}
        except (ValueError, IndexError, TypeError, struct.error):
            fields = {}
    an = _top_level(analysis_meta)
    if "waveform_data" not in fields:
        fields["waveform_data"] = an["waveform_data"]
    if "spectrum_fingerprint" not in fields:
        fields["spectrum_fingerprint"] = an["spectrum_fingerprint"]
    if "loudness_profile" not in fields or fields["loudness_profile"] == b"\x90":
        fields["loudness_profile"] = an["loudness_profile"]
    fields["length_ms"] = an["length_ms"]
    keys = [k for k in FIELD_ORDER if k in fields]
    head = bytes([0x80 | len(keys)]) if len(keys) < 16 else b"\xde" + struct.pack(">H", len(keys))
    return head + b"".join(_str(k) + fields[k] for k in keys)
