"""ctypes binding of the CPU parity oracle (oracle/libflo_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg. Nothing under flo_amd/ may import this module — the product path must run on the HIP library alone.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_SO_OVERRIDE = None   # tests/test_sanitizers.py points this at the ASan/UBSan build before the first call


def build(force=False):
    so = os.path.join(_HERE, "libflo_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "libflo_oracle.so"])
    return so


class Info(C.Structure):
    _fields_ = [
        ("version_major", C.c_uint8), ("version_minor", C.c_uint8), ("flags", C.c_uint16),
        ("sample_rate", C.c_uint32), ("channels", C.c_uint8), ("bit_depth", C.c_uint8),
        ("total_samples", C.c_uint64), ("compression_level", C.c_uint8), ("data_crc32", C.c_uint32),
        ("header_size", C.c_uint64), ("toc_size", C.c_uint64), ("data_size", C.c_uint64),
        ("extra_size", C.c_uint64), ("meta_size", C.c_uint64), ("num_frames", C.c_uint32),
        ("crc_computed", C.c_uint32),
    ]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(_SO_OVERRIDE or build())
        u8p, f32p, i32p, i16p, i64p = (C.POINTER(t) for t in (C.c_uint8, C.c_float, C.c_int32, C.c_int16, C.c_int64))
        L.flo_o_free.argtypes = [C.c_void_p]
        L.flo_o_crc32.restype = C.c_uint32
        L.flo_o_crc32.argtypes = [C.c_char_p, C.c_size_t]
        L.flo_o_f32_to_i32.restype = C.c_int32
        L.flo_o_f32_to_i32.argtypes = [C.c_float]
        L.flo_o_i32_to_f32.restype = C.c_float
        L.flo_o_i32_to_f32.argtypes = [C.c_int32]
        L.flo_o_estimate_rice_parameter_i32.restype = C.c_uint8
        L.flo_o_estimate_rice_parameter_i32.argtypes = [C.c_void_p, C.c_size_t]
        L.flo_o_rice_encode_i32.argtypes = [C.c_void_p, C.c_size_t, C.c_uint8, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.flo_o_rice_decode_i32.argtypes = [C.c_char_p, C.c_size_t, C.c_uint8, C.c_size_t, C.c_void_p]
        L.flo_o_autocorr_int.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
        L.flo_o_levinson_durbin_int.restype = C.c_int
        L.flo_o_levinson_durbin_int.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.POINTER(C.c_uint8)]
        L.flo_o_calc_residuals_int.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint8, C.c_size_t, C.c_void_p]
        L.flo_o_fixed_predictor_residuals.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
        L.flo_o_vorbis_window.argtypes = [C.c_size_t, C.c_void_p]
        L.flo_o_sine_window.argtypes = [C.c_size_t, C.c_void_p]
        L.flo_o_mdct_forward.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.flo_o_mdct_inverse.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.flo_o_mdct_forward_direct_f64.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.flo_o_ath.restype = C.c_float
        L.flo_o_ath.argtypes = [C.c_float]
        L.flo_o_freq_to_bark.restype = C.c_float
        L.flo_o_freq_to_bark.argtypes = [C.c_float]
        L.flo_o_freq_to_bark_band.restype = C.c_size_t
        L.flo_o_freq_to_bark_band.argtypes = [C.c_float]
        L.flo_o_psy_tables.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.flo_o_serialize_sparse.restype = C.c_size_t
        L.flo_o_serialize_sparse.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.flo_o_deserialize_sparse.argtypes = [C.c_char_p, C.c_size_t, C.c_size_t, C.c_void_p]
        L.flo_o_smr_threshold.restype = C.c_float
        L.flo_o_smr_threshold.argtypes = [C.c_float]
        L.flo_o_scale_factor_word.restype = C.c_uint16
        L.flo_o_scale_factor_word.argtypes = [C.c_float]
        L.flo_o_lossy_num_hops.restype = C.c_size_t
        L.flo_o_lossy_num_hops.argtypes = [C.c_size_t, C.c_uint8]
        L.flo_o_lossy_analyze.restype = C.c_size_t
        L.flo_o_lossy_analyze_f64mdct.restype = C.c_size_t
        L.flo_o_lossy_analyze.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint8, C.c_float,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.flo_o_lossy_analyze_f64mdct.argtypes = L.flo_o_lossy_analyze.argtypes
        L.flo_o_encode_lossless.restype = C.c_int
        L.flo_o_encode_lossless.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint8, C.c_uint8, C.c_uint8,
                                            C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.flo_o_encode_lossy.restype = C.c_int
        L.flo_o_encode_lossy.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint8, C.c_float,
                                         C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.flo_o_decode.restype = C.c_int
        L.flo_o_decode.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                   C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)]
        L.flo_o_decode_lossless_i32.restype = C.c_int
        L.flo_o_decode_lossless_i32.argtypes = L.flo_o_decode.argtypes
        L.flo_o_last_error.restype = C.c_char_p
        L.flo_o_info_read.restype = C.c_int
        L.flo_o_info_read.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Info)]
        L.flo_o_synth_fill.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, C.c_uint32, C.c_uint64]
        _LIB = L
    return _LIB


def _take(ptr, n, dtype):
    """Copy a malloc'ed C buffer into numpy and free it."""
    if n == 0 or not ptr:
        if ptr:
            lib().flo_o_free(ptr)
        return np.zeros(0, dtype=dtype)
    nbytes = n * np.dtype(dtype).itemsize
    arr = np.frombuffer(C.string_at(ptr, nbytes), dtype=dtype).copy()
    lib().flo_o_free(ptr)
    return arr


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def crc32(data: bytes) -> int:
    return lib().flo_o_crc32(bytes(data), len(data))


def f32_to_i32(x) -> int:
    return lib().flo_o_f32_to_i32(float(np.float32(x)))


def estimate_rice_parameter_i32(res) -> int:
    r = _i32(res)
    return lib().flo_o_estimate_rice_parameter_i32(r.ctypes.data, r.size)


def rice_encode_i32(res, k) -> bytes:
    r = _i32(res)
    out, n = C.c_void_p(), C.c_size_t()
    lib().flo_o_rice_encode_i32(r.ctypes.data, r.size, k, C.byref(out), C.byref(n))
    return _take(out.value, n.value, np.uint8).tobytes()


def rice_decode_i32(enc: bytes, k, target_len):
    out = np.zeros(target_len, dtype=np.int32)
    lib().flo_o_rice_decode_i32(bytes(enc), len(enc), k, target_len, out.ctypes.data)
    return out


def autocorr_int(s, order):
    s = _i32(s)
    out = np.zeros(order + 1, dtype=np.int64)
    lib().flo_o_autocorr_int(s.ctypes.data, s.size, order, out.ctypes.data)
    return out


def levinson_durbin_int(autocorr, order):
    a = np.ascontiguousarray(autocorr, dtype=np.int64)
    coeffs = np.zeros(max(order, 1), dtype=np.int32)
    shift = C.c_uint8()
    ok = lib().flo_o_levinson_durbin_int(a.ctypes.data, a.size, order, coeffs.ctypes.data, C.byref(shift))
    return (coeffs[:order].copy(), shift.value) if ok else None


def calc_residuals_int(s, coeffs, shift, order):
    s, c = _i32(s), _i32(coeffs)
    out = np.zeros(s.size, dtype=np.int32)
    lib().flo_o_calc_residuals_int(s.ctypes.data, s.size, c.ctypes.data, c.size, shift, order, out.ctypes.data)
    return out


def fixed_predictor_residuals(s, order):
    s = _i32(s)
    out = np.zeros(s.size, dtype=np.int32)
    lib().flo_o_fixed_predictor_residuals(s.ctypes.data, s.size, order, out.ctypes.data)
    return out


def window(n, kind="vorbis"):
    out = np.zeros(n, dtype=np.float32)
    (lib().flo_o_vorbis_window if kind == "vorbis" else lib().flo_o_sine_window)(n, out.ctypes.data)
    return out


_WT = {"sine": 0, "vorbis": 2}


def mdct_forward(samples, window_type="vorbis"):
    s = _f32(samples)
    out = np.zeros(s.size // 2, dtype=np.float32)
    lib().flo_o_mdct_forward(s.ctypes.data, s.size, _WT[window_type], out.ctypes.data)
    return out


def mdct_inverse(spec, window_type="vorbis"):
    s = _f32(spec)
    out = np.zeros(s.size * 2, dtype=np.float32)
    lib().flo_o_mdct_inverse(s.ctypes.data, s.size * 2, _WT[window_type], out.ctypes.data)
    return out


def mdct_forward_direct_f64(samples, window_type="vorbis"):
    s = _f32(samples)
    out = np.zeros(s.size // 2, dtype=np.float64)
    lib().flo_o_mdct_forward_direct_f64(s.ctypes.data, s.size, _WT[window_type], out.ctypes.data)
    return out


def psy_tables(sample_rate):
    ath = np.zeros(1024, np.float32)
    band = np.zeros(1024, np.uint8)
    spreading = np.zeros((25, 25), np.float32)
    lib().flo_o_psy_tables(sample_rate, ath.ctypes.data, band.ctypes.data, spreading.ctypes.data)
    return ath, band, spreading


def serialize_sparse(q) -> bytes:
    q = np.ascontiguousarray(q, dtype=np.int16)
    out = np.zeros(q.size * 5 + 16, dtype=np.uint8)
    n = lib().flo_o_serialize_sparse(q.ctypes.data, q.size, out.ctypes.data, out.size)
    return out[:n].tobytes()


def deserialize_sparse(data: bytes, num_coeffs=1024):
    out = np.zeros(num_coeffs, dtype=np.int16)
    lib().flo_o_deserialize_sparse(bytes(data), len(data), num_coeffs, out.ctypes.data)
    return out


def lossy_analyze(pcm, sample_rate, channels, quality, f64_mdct=False):
    """Returns dict of per-frame intermediates of the reference lossy encoder, [hops][ch][...].
    f64_mdct: evaluate the transform in double precision from its definition (accuracy yardstick for the f32 FFTs)."""
    p = _f32(pcm)
    nh = lib().flo_o_lossy_num_hops(p.size, channels)
    coeffs = np.zeros((nh, channels, 1024), np.float32)
    smr = np.zeros((nh, channels, 1024), np.float32)
    q = np.zeros((nh, channels, 1024), np.int16)
    sf = np.zeros((nh, channels, 25), np.float32)
    sfw = np.zeros((nh, channels, 25), np.uint16)
    fn = lib().flo_o_lossy_analyze_f64mdct if f64_mdct else lib().flo_o_lossy_analyze
    got = fn(p.ctypes.data, p.size, sample_rate, channels, quality, coeffs.ctypes.data,
             smr.ctypes.data, q.ctypes.data, sf.ctypes.data, sfw.ctypes.data)
    assert got == nh
    return dict(coeffs=coeffs, smr=smr, q=q, sf=sf, sf_words=sfw)


def encode_lossless(pcm, sample_rate, channels, bit_depth=16, level=5, meta=b"") -> bytes:
    p = _f32(pcm)
    out, n = C.c_void_p(), C.c_size_t()
    rc = lib().flo_o_encode_lossless(p.ctypes.data, p.size, sample_rate, channels, bit_depth, level, meta, len(meta),
                                     C.byref(out), C.byref(n))
    if rc != 0:
        raise RuntimeError(lib().flo_o_last_error().decode())
    return _take(out.value, n.value, np.uint8).tobytes()


def encode_lossy(pcm, sample_rate, channels, quality, meta=b"") -> bytes:
    p = _f32(pcm)
    out, n = C.c_void_p(), C.c_size_t()
    rc = lib().flo_o_encode_lossy(p.ctypes.data, p.size, sample_rate, channels, quality, meta, len(meta),
                                  C.byref(out), C.byref(n))
    if rc != 0:
        raise RuntimeError(lib().flo_o_last_error().decode())
    return _take(out.value, n.value, np.uint8).tobytes()


def decode(flo: bytes):
    """-> (pcm f32 interleaved, sample_rate, channels)"""
    out, n, sr, ch = C.c_void_p(), C.c_size_t(), C.c_uint32(), C.c_uint8()
    rc = lib().flo_o_decode(bytes(flo), len(flo), C.byref(out), C.byref(n), C.byref(sr), C.byref(ch))
    if rc != 0:
        raise RuntimeError(lib().flo_o_last_error().decode())
    return _take(out.value, n.value, np.float32), sr.value, ch.value


def decode_lossless_i32(flo: bytes):
    out, n, sr, ch = C.c_void_p(), C.c_size_t(), C.c_uint32(), C.c_uint8()
    rc = lib().flo_o_decode_lossless_i32(bytes(flo), len(flo), C.byref(out), C.byref(n), C.byref(sr), C.byref(ch))
    if rc != 0:
        raise RuntimeError(lib().flo_o_last_error().decode())
    return _take(out.value, n.value, np.int32), sr.value, ch.value


def info(flo: bytes) -> Info:
    i = Info()
    if lib().flo_o_info_read(bytes(flo), len(flo), C.byref(i)) != 0:
        raise RuntimeError(lib().flo_o_last_error().decode())
    return i


class _Fingerprint(C.Structure):
    _fields_ = [("hash", C.c_uint8 * 32), ("duration_ms", C.c_uint32), ("sample_rate", C.c_uint32), ("channels", C.c_uint8),
                ("frequency_peaks", C.c_uint8 * 8), ("energy_profile", C.c_uint8 * 16), ("avg_loudness", C.c_uint8)]


def blake3(data: bytes) -> bytes:
    L = lib()
    L.flo_o_blake3.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
    o = C.create_string_buffer(32)
    L.flo_o_blake3(bytes(data), len(data), o)
    return o.raw


def waveform_peaks(pcm, channels, sample_rate, peaks_per_second=50):
    """core/analysis.rs:38-115 -> float32 array of normalised peaks"""
    L = lib()
    L.flo_o_waveform_peaks.restype = C.c_size_t
    L.flo_o_waveform_peaks.argtypes = [C.c_void_p, C.c_size_t, C.c_uint8, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]
    p = _f32(pcm)
    out = np.zeros(p.size // max(channels, 1) + 16, np.float32)
    n = L.flo_o_waveform_peaks(p.ctypes.data, p.size, channels, sample_rate, peaks_per_second, out.ctypes.data, out.size)
    return out[:n].copy()


def spectral_fingerprint(pcm, channels, sample_rate):
    """core/analysis.rs:223-357 -> dict"""
    L = lib()
    L.flo_o_spectral_fingerprint.argtypes = [C.c_void_p, C.c_size_t, C.c_uint8, C.c_uint32, C.POINTER(_Fingerprint)]
    p = _f32(pcm)
    fp = _Fingerprint()
    L.flo_o_spectral_fingerprint(p.ctypes.data, p.size, channels, sample_rate, C.byref(fp))
    return dict(hash=bytes(fp.hash), duration_ms=fp.duration_ms, sample_rate=fp.sample_rate, channels=fp.channels,
                frequency_peaks=list(fp.frequency_peaks), energy_profile=list(fp.energy_profile), avg_loudness=fp.avg_loudness)


def integrated_lufs(pcm, channels, sample_rate) -> float:
    """core/ebu_r128.rs:182-318 (integrated loudness only)"""
    L = lib()
    L.flo_o_integrated_lufs.restype = C.c_double
    L.flo_o_integrated_lufs.argtypes = [C.c_void_p, C.c_size_t, C.c_uint8, C.c_uint32]
    p = _f32(pcm)
    return L.flo_o_integrated_lufs(p.ctypes.data, p.size, channels, sample_rate)


def loudness_metrics(pcm, channels, sample_rate) -> dict:
    """compute_ebu_r128_loudness (ebu_r128.rs:182-355)"""
    L = lib()
    L.flo_o_loudness_metrics.restype = None
    L.flo_o_loudness_metrics.argtypes = [C.c_void_p, C.c_size_t, C.c_uint8, C.c_uint32, C.POINTER(C.c_double)]
    p = np.ascontiguousarray(pcm, np.float32).ravel()
    out = (C.c_double * 4)()
    L.flo_o_loudness_metrics(p.ctypes.data, p.size, channels, sample_rate, out)
    return dict(integrated_lufs=out[0], loudness_range_lu=out[1], true_peak_dbtp=out[2], sample_peak_dbfs=out[3])


def analysis_metadata(pcm, sample_rate, channels, peaks_per_second=50) -> bytes:
    """add_analysis_data_if_missing(&[], ...) (lib.rs:219-283): the META bytes libflo::encode* build"""
    L = lib()
    L.flo_o_analysis_metadata.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint8, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    p = _f32(pcm)
    out, n = C.c_void_p(), C.c_size_t()
    L.flo_o_analysis_metadata(p.ctypes.data, p.size, sample_rate, channels, peaks_per_second, C.byref(out), C.byref(n))
    b = C.string_at(out.value, n.value)
    L.flo_o_free(out)
    return b


class StreamingEncoder:
    """oracle restatement of streaming::StreamingEncoder (libflo/src/streaming/encoder.rs)"""

    def __init__(self, sample_rate, channels, bit_depth, level=5):
        L = lib()
        L.flo_o_stream_new.restype = C.c_void_p
        L.flo_o_stream_new.argtypes = [C.c_uint32, C.c_uint8, C.c_uint8, C.c_uint8]
        L.flo_o_stream_free.argtypes = [C.c_void_p]
        L.flo_o_stream_push.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.flo_o_stream_pending_samples.argtypes = [C.c_void_p]
        L.flo_o_stream_pending_samples.restype = C.c_size_t
        L.flo_o_stream_pending_frames.argtypes = [C.c_void_p]
        L.flo_o_stream_pending_frames.restype = C.c_size_t
        u32p = C.POINTER(C.c_uint32)
        L.flo_o_stream_next_frame.argtypes = [C.c_void_p, u32p, u32p, u32p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.flo_o_stream_flush.argtypes = L.flo_o_stream_next_frame.argtypes
        L.flo_o_stream_finalize.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        self._L = L
        self._h = C.c_void_p(L.flo_o_stream_new(sample_rate, channels, bit_depth, level))

    def push_samples(self, samples):
        p = _f32(samples)
        if self._L.flo_o_stream_push(self._h, p.ctypes.data, p.size) != 0:
            raise RuntimeError("oracle stream push failed")

    def pending_samples(self):
        return self._L.flo_o_stream_pending_samples(self._h)

    def pending_frames(self):
        return self._L.flo_o_stream_pending_frames(self._h)

    def _pull(self, fn):
        idx, ts, ns = C.c_uint32(), C.c_uint32(), C.c_uint32()
        data, n = C.c_void_p(), C.c_size_t()
        r = fn(self._h, C.byref(idx), C.byref(ts), C.byref(ns), C.byref(data), C.byref(n))
        if r < 0:
            raise RuntimeError("oracle stream error")
        if r == 0:
            return None
        b = C.string_at(data.value, n.value)
        self._L.flo_o_free(data)
        return dict(index=idx.value, timestamp_ms=ts.value, samples=ns.value, data=b)

    def next_frame(self):
        return self._pull(self._L.flo_o_stream_next_frame)

    def flush(self):
        return self._pull(self._L.flo_o_stream_flush)

    def finalize(self, meta=b""):
        out, n = C.c_void_p(), C.c_size_t()
        if self._L.flo_o_stream_finalize(self._h, meta, len(meta), C.byref(out), C.byref(n)) != 0:
            raise RuntimeError("oracle stream finalize failed")
        b = C.string_at(out.value, n.value)
        self._L.flo_o_free(out)
        return b

    def __del__(self):
        try:
            self._L.flo_o_stream_free(self._h)
        except Exception:
            pass


def synth_clip(n_sample_frames, channels, seed=0xF10A0D10, clip_id=0):
    """Host instance of include/flo_synth.h — bit-identical to the device generator."""
    out = np.zeros(n_sample_frames * channels, np.float32)
    lib().flo_o_synth_fill(out.ctypes.data, n_sample_frames, channels, seed, clip_id)
    return out
