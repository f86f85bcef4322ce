"""Host-side mirror of libflo's encoder interface, bound to the HIP library through its C ABI.

Same names, argument meaning and error behaviour as the reference (all paths under /root/reference):
  Encoder(sample_rate, channels, bit_depth).with_compression(level).encode(samples, metadata)
        -> libflo/src/lossless/encoder.rs:17-45
  LossyEncoder / TransformEncoder(sample_rate, channels, quality).encode_to_flo(samples, metadata)
        -> libflo/src/lossy/encoder.rs:36-53,167-239 (re-export lib.rs:21-24)
  QualityPreset                      -> libflo/src/lossy/mod.rs:19-128
  encode / encode_lossy / encode_with_bitrate (free functions: analysis metadata first, as the reference)
        -> libflo/src/lib.rs:97-206, 219-283
Errors surface as FloError(message), the analogue of FloResult<T> = Result<T, String> (core/types.rs:281).
"""
import ctypes as C
import enum
import weakref

import numpy as np

from . import _native
from ._native import FloError, MODE_LOSSLESS, MODE_LOSSY


class QualityPreset(enum.IntEnum):
    """lossy/mod.rs:19-128"""
    Low = 0
    Medium = 1
    High = 2
    VeryHigh = 3
    Transparent = 4

    def as_f32(self) -> float:
        return [0.0, 0.35, 0.55, 0.75, 1.0][int(self)]

    @staticmethod
    def from_f32(q: float) -> "QualityPreset":
        if q < 0.2:
            return QualityPreset.Low
        if q < 0.45:
            return QualityPreset.Medium
        if q < 0.65:
            return QualityPreset.High
        if q < 0.85:
            return QualityPreset.VeryHigh
        return QualityPreset.Transparent

    @staticmethod
    def from_bitrate(bitrate_kbps: int, sample_rate: int, channels: int) -> "QualityPreset":
        raw_kbps = (sample_rate * channels * 16) // 1000
        ratio = np.float32(raw_kbps) / np.float32(bitrate_kbps)
        if ratio > 20.0:
            return QualityPreset.Low
        if ratio > 10.0:
            return QualityPreset.Medium
        if ratio > 6.0:
            return QualityPreset.High
        if ratio > 4.0:
            return QualityPreset.VeryHigh
        return QualityPreset.Transparent

    def expected_ratio(self) -> float:
        return [30.0, 10.0, 6.0, 4.0, 3.0][int(self)]

    def equivalent_bitrate(self) -> int:
        return [48, 128, 192, 256, 320][int(self)]


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32).reshape(-1)


class _CBuffer:
    """keeps a buffer the library handed out (flo_free) alive for as long as an array looks at it"""

    def __init__(self, lib, ptr):
        self._lib, self._ptr = lib, C.c_void_p(ptr.value)

    def __del__(self):
        if self._ptr.value:
            self._lib.flo_free(self._ptr)
            self._ptr = C.c_void_p()


class Context:
    """One per host thread / GPU (flo_ctx)."""

    def __init__(self, device: int = 0):
        self._L = _native.lib()
        h = C.c_void_p()
        rc = self._L.flo_ctx_create(device, C.byref(h))
        if rc != 0:
            raise FloError(self._L.flo_last_create_error().decode())
        self._h = h
        self._batches = weakref.WeakSet()   # batches hold device memory of this context: they go first

    def close(self):
        if getattr(self, "_h", None):
            for g in list(getattr(self, "_dists", ())):   # communicators made on this context go first (flo_dist_destroy uses it)
                g.close()
            for b in list(self._batches):
                b.close()
            self._L.flo_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise FloError(self._L.flo_last_error(self._h).decode())

    def _take(self, ptr, n):
        data = C.string_at(ptr.value, n.value) if ptr.value else b""
        self._L.flo_free(ptr)
        return data

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, hbm = C.c_int(), C.c_uint64()
        self._chk(self._L.flo_ctx_device_info(self._h, name, 256, C.byref(cus), C.byref(hbm)))
        return name.value.decode(), cus.value, hbm.value

    def force_path(self, which: int):
        self._chk(self._L.flo_ctx_force_path(self._h, which))

    def reserve_cus(self, n: int):
        """compute units the persistent encode kernels leave free (for RCCL's kernels when ranks exchange files)"""
        self._chk(self._L.flo_ctx_reserve_cus(self._h, n))

    def reserved_cus(self) -> int:
        """compute units the persistent encode kernels currently leave free"""
        return int(self._L.flo_ctx_reserved_cus(self._h))

    def upload_path(self):
        """(path large host uploads take on this host, GB/s the probe measured for pageable-direct, for the pinned ring)"""
        name, a, b = C.create_string_buffer(64), C.c_double(), C.c_double()
        self._chk(self._L.flo_ctx_upload_path(self._h, name, 64, C.byref(a), C.byref(b)))
        return name.value.decode(), a.value, b.value

    # -- one clip ---------------------------------------------------------------------------------------

    def encode_lossy(self, samples, sample_rate, channels, quality, metadata=b"") -> bytes:
        p = _f32(samples)
        out, n = C.c_void_p(), C.c_size_t()
        self._chk(self._L.flo_encode_lossy(self._h, p.ctypes.data, p.size, sample_rate, channels, quality,
                                           metadata, len(metadata), C.byref(out), C.byref(n)))
        return self._take(out, n)

    def encode_lossless(self, samples, sample_rate, channels, bit_depth=16, level=5, metadata=b"") -> bytes:
        p = _f32(samples)
        out, n = C.c_void_p(), C.c_size_t()
        self._chk(self._L.flo_encode_lossless(self._h, p.ctypes.data, p.size, sample_rate, channels, bit_depth, level,
                                              metadata, len(metadata), C.byref(out), C.byref(n)))
        return self._take(out, n)

    def encode_batch(self, mode, clips, sample_rate, channels, quality_or_level):
        arrs = [_f32(c) for c in clips]
        k = len(arrs)
        ptrs = (C.c_void_p * k)(*[a.ctypes.data for a in arrs])
        lens = (C.c_size_t * k)(*[a.size for a in arrs])
        outs, olens = (C.c_void_p * k)(), (C.c_size_t * k)()
        self._chk(self._L.flo_encode_batch(self._h, mode, k, ptrs, lens, sample_rate, channels, quality_or_level, outs, olens))
        res = []
        for i in range(k):
            res.append(C.string_at(outs[i], olens[i]) if outs[i] else b"")
            self._L.flo_free(outs[i])
        return res

    def decode(self, flo: bytes, with_info=False):
        """libflo::decode (lib.rs:296-315): interleaved f32 PCM of a .flo file, decoded on the device."""
        return self._decode(self._L.flo_decode, np.float32, flo, with_info)

    def decode_lossless_i32(self, flo: bytes, with_info=False):
        """the integers a lossless file decodes to, before the 1/32767 scaling (bit-exact parity checks)"""
        return self._decode(self._L.flo_decode_lossless_i32, np.int32, flo, with_info)

    def _decode(self, fn, dtype, flo, with_info):
        flo = bytes(flo)
        out, n = C.c_void_p(), C.c_size_t()
        sr, ch = C.c_uint32(), C.c_uint8()
        self._chk(fn(self._h, flo, len(flo), C.byref(out), C.byref(n), C.byref(sr), C.byref(ch)))
        if n.value:
            # the array IS the library's buffer (freed with it): two more copies of a 3-minute file's 63 MB, each into fresh
            # pages, cost four times what the decode itself does
            buf = (C.c_char * (n.value * 4)).from_address(out.value)
            buf._flo_owner = _CBuffer(self._L, out)
            a = np.frombuffer(buf, dtype=dtype)
        else:
            self._L.flo_free(out)
            a = np.zeros(0, dtype)
        return (a, sr.value, ch.value) if with_info else a

    # -- analysis metadata of libflo::encode* (lib.rs:219-283) ---------------------------------------------------
    def analyze(self, samples, sample_rate, channels, peaks_per_second=50):
        """waveform peaks, spectral fingerprint and EBU R128 integrated loudness, computed on the device"""
        p = _f32(samples)
        peaks = np.zeros(int(np.ceil(p.size // max(channels, 1) * peaks_per_second / max(sample_rate, 1))) + 16, np.float32)
        a = _native.Analysis()
        self._chk(self._L.flo_analyze(self._h, p.ctypes.data, p.size, sample_rate, channels, peaks_per_second,
                                      peaks.ctypes.data, peaks.size, C.byref(a)))
        return dict(peaks=peaks[: a.n_peaks].copy(), hash=bytes(a.hash), duration_ms=a.duration_ms, sample_rate=a.sample_rate,
                    channels=a.channels, frequency_peaks=list(a.frequency_peaks), energy_profile=list(a.energy_profile),
                    avg_loudness=a.avg_loudness, integrated_lufs=a.integrated_lufs, length_ms=a.length_ms,
                    loudness_range_lu=a.loudness_range_lu, true_peak_dbtp=a.true_peak_dbtp, sample_peak_dbfs=a.sample_peak_dbfs,
                    sum_squares=np.float32(a.sum_squares))

    def analysis_metadata(self, samples, sample_rate, channels, peaks_per_second=50) -> bytes:
        """add_analysis_data_if_missing(&[], ...): the MessagePack META libflo::encode* build for an empty input META"""
        p = _f32(samples)
        out, n = C.c_void_p(), C.c_size_t()
        self._chk(self._L.flo_analysis_metadata(self._h, p.ctypes.data, p.size, sample_rate, channels, peaks_per_second,
                                                C.byref(out), C.byref(n)))
        return self._take(out, n)

    # -- stage-level entry points (parity tests) -----------------------------------------------------------
    def mdct_forward(self, frames):
        f = _f32(frames)
        n = f.size // 2048
        out = np.zeros(n * 1024, np.float32)
        self._chk(self._L.flo_mdct_forward(self._h, f.ctypes.data, n, out.ctypes.data))
        return out.reshape(n, 1024)

    def lossy_analyze(self, samples, sample_rate, channels, quality):
        p = _f32(samples)
        hops = ((p.size // channels) + 1024 + 1023) // 1024
        coeffs = np.zeros((hops, channels, 1024), np.float32)
        q = np.zeros((hops, channels, 1024), np.int16)
        sfw = np.zeros((hops, channels, 25), np.uint16)
        nh = C.c_size_t()
        self._chk(self._L.flo_lossy_analyze(self._h, p.ctypes.data, p.size, sample_rate, channels, quality,
                                            coeffs.ctypes.data, q.ctypes.data, sfw.ctypes.data, C.byref(nh)))
        assert nh.value == hops
        return dict(coeffs=coeffs, q=q, sf_words=sfw)

    def lossy_quantize(self, coeffs, sample_rate, quality, exact=False):
        """Device psychoacoustics + quantiser on caller-supplied spectra. exact=False is the quantiser every encode
        runs; exact=True adds the reference's dB-domain re-check next to the threshold (test yardstick)."""
        c = np.ascontiguousarray(coeffs, np.float32)
        hops, channels = c.shape[0], c.shape[1]
        q = np.zeros((hops, channels, 1024), np.int16)
        sfw = np.zeros((hops, channels, 25), np.uint16)
        self._chk(self._L.flo_lossy_quantize(self._h, c.ctypes.data, hops, sample_rate, channels, quality, 1 if exact else 0,
                                             q.ctypes.data, sfw.ctypes.data))
        return dict(q=q, sf_words=sfw)

    def lossy_quantize_smr(self, coeffs, smr, sample_rate, quality):
        """TransformEncoder::quantize_coefficients on the device (encoder.rs:109-154): vectors of 1024 coefficients and the
        caller's signal-to-mask ratios -> (i16 [n][1024], f32 scale factors [n][25]); smr=None: scale factors only."""
        c = np.ascontiguousarray(coeffs, np.float32).reshape(-1, 1024)
        n = c.shape[0]
        sf = np.zeros((n, 25), np.float32)
        if smr is None:
            self._chk(self._L.flo_lossy_quantize_smr(self._h, c.ctypes.data, None, n, sample_rate, quality, None, sf.ctypes.data))
            return None, sf
        m = np.ascontiguousarray(smr, np.float32).reshape(-1, 1024)
        assert m.shape == c.shape
        q = np.zeros((n, 1024), np.int16)
        self._chk(self._L.flo_lossy_quantize_smr(self._h, c.ctypes.data, m.ctypes.data, n, sample_rate, quality, q.ctypes.data, sf.ctypes.data))
        return q, sf

    def sparse_pack(self, q, form=0):
        """serialize_sparse on the device. form 0: as the encoder packs (item form, behind it the block form, behind that the
        general form for dense vectors); form 1: the general form for every vector; form 2: block form, then general."""
        q = np.ascontiguousarray(q, np.int16).reshape(-1, 1024)
        n = q.shape[0]
        out = np.zeros(n * 2080, np.uint8)
        off = np.zeros(n + 1, np.uint32)
        self._chk(self._L.flo_sparse_pack(self._h, q.ctypes.data, n, form, out.ctypes.data, out.size, off.ctypes.data))
        return [out[off[i]:off[i + 1]].tobytes() for i in range(n)]

    # -- profiling hooks --------------------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._chk(self._L.flo_ctx_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        self._chk(self._L.flo_ctx_profile_reset(self._h))

    def profile_query(self, kernel: str):
        ms, n = C.c_double(), C.c_uint64()
        self._chk(self._L.flo_ctx_profile_query(self._h, kernel.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def stream(self):
        return self._L.flo_ctx_stream(self._h)


class Batch:
    """Device-resident batch of clips (flo_batch): PCM stays in HBM, bitstreams are left in HBM."""

    def __init__(self, ctx: Context, mode, n_interleaved, sample_rate, channels, quality_or_level):
        self.ctx, self._L = ctx, ctx._L
        self.n_clips = len(n_interleaved)
        self.n_interleaved = list(int(x) for x in n_interleaved)
        lens = (C.c_size_t * self.n_clips)(*self.n_interleaved)
        h = C.c_void_p()
        ctx._chk(self._L.flo_batch_create(ctx._h, mode, self.n_clips, lens, sample_rate, channels, quality_or_level, C.byref(h)))
        self._h = h
        ctx._batches.add(self)

    def close(self):
        if getattr(self, "_h", None):
            self._L.flo_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def clip_device_ptr(self, clip):
        return self._L.flo_batch_clip_device_ptr(self._h, clip)

    def upload(self, clip, samples):
        p = _f32(samples)
        assert p.size == self.n_interleaved[clip]
        self.ctx._chk(self._L.flo_batch_upload(self._h, clip, p.ctypes.data))
        self.ctx._chk(self._L.flo_batch_sync(self._h))   # p may be released by the caller

    def fill_synthetic(self, seed=0xF10A0D10, clip_id0=0):
        self.ctx._chk(self._L.flo_batch_fill_synthetic(self._h, seed, clip_id0))

    def download_pcm(self, clip):
        """the clip's interleaved f32 PCM as it sits in the batch (after upload or fill_synthetic), as a numpy array"""
        import numpy as np
        self.sync()
        n = int(self.n_interleaved[clip])
        out = np.empty(n, np.float32)
        if n:
            hip = C.CDLL("libamdhip64.so")
            hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            rc = hip.hipMemcpy(out.ctypes.data, self.clip_device_ptr(clip), n * 4, 2)
            if rc != 0:
                raise FloError(f"hipMemcpy (device to host) failed with {rc}")
        return out

    def encode(self, which=0):
        self.ctx._chk(self._L.flo_batch_encode(self._h, which))

    def sync(self):
        self.ctx._chk(self._L.flo_batch_sync(self._h))

    def data_bytes(self):
        t = C.c_uint64()
        self.ctx._chk(self._L.flo_batch_data_bytes(self._h, C.byref(t)))
        return t.value

    def analysis_metadata(self, clip, peaks_per_second=50) -> bytes:
        """the analysis META of a clip already uploaded into this batch (no second trip over PCIe)"""
        out, n = C.c_void_p(), C.c_size_t()
        self.ctx._chk(self._L.flo_batch_analysis_metadata(self._h, clip, peaks_per_second, C.byref(out), C.byref(n)))
        return self.ctx._take(out, n)

    def set_bit_depth(self, bit_depth: int):
        self.ctx._chk(self._L.flo_batch_set_bit_depth(self._h, bit_depth))

    def fetch(self, clip, metadata=b"") -> bytes:
        out, n = C.c_void_p(), C.c_size_t()
        self.ctx._chk(self._L.flo_batch_fetch(self._h, clip, metadata, len(metadata), C.byref(out), C.byref(n)))
        return self.ctx._take(out, n)

    def pack_streams(self, dst_ptr: int, dst_cap: int):
        """Pack all DATA chunks into caller-owned device memory; returns the n_clips + 1 offsets."""
        offs = (C.c_uint64 * (self.n_clips + 1))()
        self.ctx._chk(self._L.flo_batch_pack_streams(self._h, dst_ptr, dst_cap, offs))
        return list(offs)

    def pack_files(self, dst_ptr: int, dst_cap: int):
        """Pack all finished .flo files (empty META) into caller-owned device memory; returns the n_clips + 1 offsets."""
        offs = (C.c_uint64 * (self.n_clips + 1))()
        self.ctx._chk(self._L.flo_batch_pack_files(self._h, dst_ptr, dst_cap, offs))
        return list(offs)

    def decode_to(self, dst_ptr: int, dst_cap_floats: int):
        """Decode every clip of an encoded lossy batch into device memory; returns the per-clip float offsets."""
        offs = (C.c_uint64 * max(self.n_clips, 1))()
        self.ctx._chk(self._L.flo_batch_decode(self._h, dst_ptr, dst_cap_floats, offs))
        return list(offs[: self.n_clips])

    def device_streams(self):
        base = C.c_void_p()
        offs, sizes = C.POINTER(C.c_uint64)(), C.POINTER(C.c_uint64)()
        self.ctx._chk(self._L.flo_batch_device_streams(self._h, C.byref(base), C.byref(offs), C.byref(sizes)))
        return base.value, [offs[i] for i in range(self.n_clips)], [sizes[i] for i in range(self.n_clips)]


_default_ctx = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


class Encoder:
    """lossless::Encoder — lossless/encoder.rs:9-45"""

    def __init__(self, sample_rate: int, channels: int, bit_depth: int, ctx: Context = None):
        self.sample_rate, self.channels, self.bit_depth = sample_rate, channels, bit_depth
        self.compression_level = 5
        self._ctx = ctx

    def with_compression(self, level: int) -> "Encoder":
        self.compression_level = min(int(level), 9)
        return self

    def encode(self, samples, metadata: bytes = b"") -> bytes:
        ctx = self._ctx or default_context()
        return ctx.encode_lossless(samples, self.sample_rate, self.channels, self.bit_depth, self.compression_level, metadata)


class Decoder:
    """lossless::Decoder (lossless/decoder.rs:6-18); like libflo::decode it also accepts transform files."""

    def __init__(self, ctx: Context = None):
        self._ctx = ctx or default_context()

    def decode(self, data: bytes):
        return self._ctx.decode(data)


class TransformFrame:
    """lossy::TransformFrame (lossy/mod.rs): what encode_frame returns - per channel the 1024 quantised coefficients and the
    25 band scale factors; `scale_words` are the u16 words the container stores for them (encoder.rs:262-266)."""

    def __init__(self, coefficients, scale_factors, scale_words):
        self.coefficients, self.scale_factors, self.scale_words = coefficients, scale_factors, scale_words
        self.block_size, self.num_samples = 2048, 1024


class TransformEncoder:
    """lossy::TransformEncoder — lossy/encoder.rs:6-53,63-164,167-239. One fresh encoder per clip is the contract."""

    _HISTORY = 65   # frames whose masking levels can still reach the newest one: the temporal step is max(a_t, 0.7 s_{t-1}),
                    # and the frame-parallel kernels resolve it exactly from a 64-frame warm-up (tests compare them to the chain)

    def __init__(self, sample_rate: int, channels: int, quality: float, ctx: Context = None):
        self.sample_rate, self.channels = sample_rate, channels
        self.quality = float(min(max(quality, 0.0), 1.0))
        self._ctx = ctx
        self._spectra = []   # the last _HISTORY frames' coefficients [ch][1024]: the psychoacoustic model's temporal state

    def set_quality(self, quality: float):
        self.quality = float(min(max(quality, 0.0), 1.0))

    def reset(self):
        """encoder.rs:157-164: forget the temporal masking state (and the transform's, which keeps none here)"""
        self._spectra = []

    def encode_frame(self, samples) -> TransformFrame:
        """encoder.rs:63-106: one block of 2048 sample-frames (interleaved; shorter blocks are zero-padded) -> its quantised
        spectrum. Stateful like the reference: the masking thresholds of a frame depend on the frames encoded before it
        (psychoacoustic.rs:196-203), so the device pass runs over the kept spectra and the newest frame's result is returned."""
        ctx = self._ctx or default_context()
        x = _f32(samples)
        ch = self.channels
        per = x.size // ch + (1 if x.size % ch else 0)
        block = np.zeros((ch, 2048), np.float32)
        for c in range(ch):
            d = x[c::ch][:2048]
            block[c, :d.size] = d
        assert per <= 2048, "encode_frame takes one block (2048 sample-frames)"
        spec = ctx.mdct_forward(block.reshape(-1))            # [ch][1024], device
        self._spectra.append(spec)
        if len(self._spectra) > self._HISTORY:
            self._spectra.pop(0)
        g = ctx.lossy_quantize(np.stack(self._spectra), self.sample_rate, self.quality)
        _, sf = ctx.lossy_quantize_smr(spec, None, self.sample_rate, self.quality)
        return TransformFrame([g["q"][-1, c].copy() for c in range(ch)], [sf[c].copy() for c in range(ch)],
                              [g["sf_words"][-1, c].copy() for c in range(ch)])

    def quantize_coefficients(self, coeffs, smr):
        """encoder.rs:109-154: (quantised i16 [1024], scale factors f32 [25]) of one channel's coefficients under the
        caller's signal-to-mask ratios, on the device"""
        ctx = self._ctx or default_context()
        q, sf = ctx.lossy_quantize_smr(coeffs, smr, self.sample_rate, self.quality)
        return q[0], sf[0]

    def encode_to_flo(self, samples, metadata: bytes = b"") -> bytes:
        ctx = self._ctx or default_context()
        return ctx.encode_lossy(samples, self.sample_rate, self.channels, self.quality, metadata)


LossyEncoder = TransformEncoder


class EncodedFrame:
    """streaming/encoder.rs:18-29"""

    def __init__(self, index, timestamp_ms, data, samples):
        self.index, self.timestamp_ms, self.data, self.samples = index, timestamp_ms, data, samples

    def __repr__(self):
        return f"EncodedFrame(index={self.index}, timestamp_ms={self.timestamp_ms}, samples={self.samples}, {len(self.data)} bytes)"


class StreamingEncoder:
    """streaming::StreamingEncoder - libflo/src/streaming/encoder.rs:6-257, over flo_stream_* of the C ABI."""

    def __init__(self, sample_rate: int, channels: int, bit_depth: int, ctx: Context = None):
        self.sample_rate, self.channels, self.bit_depth = sample_rate, channels, bit_depth
        self.compression_level = 5
        self._ctx = ctx or default_context()
        self._L = self._ctx._L
        self._h = None
        self._open()

    def _open(self):
        if self._h:
            self._L.flo_stream_destroy(self._h)
        h = C.c_void_p()
        self._ctx._chk(self._L.flo_stream_create(self._ctx._h, self.sample_rate, self.channels, self.bit_depth, self.compression_level, C.byref(h)))
        self._h = h

    def with_compression(self, level: int) -> "StreamingEncoder":
        self.compression_level = min(int(level), 9)
        self._open()      # like the reference, meant to be called right after construction
        return self

    def pending_samples(self) -> int:
        return self._L.flo_stream_pending_samples(self._h)

    def pending_frames(self) -> int:
        return self._L.flo_stream_pending_frames(self._h)

    def push_samples(self, samples):
        p = _f32(samples)
        self._ctx._chk(self._L.flo_stream_push(self._h, p.ctypes.data, p.size))

    def _pull(self, fn):
        idx, ts, ns = C.c_uint32(), C.c_uint32(), C.c_uint32()
        data, n = C.c_void_p(), C.c_size_t()
        r = fn(self._h, C.byref(idx), C.byref(ts), C.byref(ns), C.byref(data), C.byref(n))
        if r < 0:
            raise FloError(self._L.flo_last_error(self._ctx._h).decode())
        if r == 0:
            return None
        return EncodedFrame(idx.value, ts.value, self._ctx._take(data, n), ns.value)

    def next_frame(self):
        return self._pull(self._L.flo_stream_next_frame)

    def flush(self):
        return self._pull(self._L.flo_stream_flush)

    def finalize(self, metadata: bytes = b"") -> bytes:
        out, n = C.c_void_p(), C.c_size_t()
        self._ctx._chk(self._L.flo_stream_finalize(self._h, metadata, len(metadata), C.byref(out), C.byref(n)))
        return self._ctx._take(out, n)

    def close(self):
        if getattr(self, "_h", None):
            self._L.flo_stream_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def probe_container(data: bytes):
    """Header and frame census of a .flo file as the container reader sees it (reader.rs:16-256); no device needed.
    Raises FloError with the reader's message for files the reference reader rejects."""
    L = _native.lib()
    info = _native.ContainerInfo()
    err = C.create_string_buffer(256)
    data = bytes(data)
    if L.flo_probe_container(data, len(data), C.byref(info), err, len(err)) != 0:
        raise FloError(err.value.decode() or "not a .flo file")
    return info


def decode(data: bytes):
    """libflo::decode (lib.rs:296-315)"""
    return default_context().decode(data)


def _encode_analysed(mode, samples, sample_rate, channels, quality_or_level, bit_depth, metadata) -> bytes:
    """what the three free functions share (lib.rs:97-206): `add_analysis_data_if_missing(&metadata.unwrap_or_default(), samples,
    sr, ch, 50)`, then the encoder - on ONE copy of the samples: uploaded once, analysed on the device, encoded from there"""
    from . import meta as _meta
    ctx = default_context()
    p = _f32(samples)
    b = Batch(ctx, mode, [p.size], sample_rate, channels, quality_or_level)
    try:
        b.upload(0, p)
        m = _meta.merge_analysis(metadata or b"", b.analysis_metadata(0, 50))
        if mode == MODE_LOSSLESS:
            b.set_bit_depth(bit_depth)
        b.encode(0)
        b.sync()
        return b.fetch(0, m)
    finally:
        b.close()


def encode(samples, sample_rate, channels, bit_depth, metadata=None) -> bytes:
    """libflo::encode (lib.rs:97-117): analysis metadata first (waveform peaks, spectral fingerprint, EBU R128 loudness,
    length), then the lossless encoder at level 5"""
    return _encode_analysed(MODE_LOSSLESS, samples, sample_rate, channels, 5, bit_depth, metadata)


def encode_lossy(samples, sample_rate, channels, _bit_depth, quality: int, metadata=None) -> bytes:
    """libflo::encode_lossy (lib.rs:135-166): quality level 0-4 -> 0.0/0.35/0.55/0.75/1.0"""
    q = {0: 0.0, 1: 0.35, 2: 0.55, 3: 0.75}.get(int(quality), 1.0)
    return _encode_analysed(MODE_LOSSY, samples, sample_rate, channels, q, 16, metadata)


def encode_with_bitrate(samples, sample_rate, channels, _bit_depth, target_bitrate_kbps, metadata=None) -> bytes:
    """libflo::encode_with_bitrate (lib.rs:181-206)"""
    q = QualityPreset.from_bitrate(target_bitrate_kbps, sample_rate, channels).as_f32()
    return _encode_analysed(MODE_LOSSY, samples, sample_rate, channels, q, 16, metadata)
