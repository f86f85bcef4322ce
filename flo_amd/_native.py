"""ctypes loader of the product library flo_amd/libflo_hip.so (C ABI in include/flo_hip.h).

There is no CPU path behind this module: if the shared library is missing, or no gfx950 device can be opened,
every entry point raises. torch is imported first on purpose — its bundled HIP runtime carries the same SONAME
as /opt/rocm's, so loading in this order gives the process ONE HIP runtime and lets torch tensors / RCCL and
this library's kernels share device pointers and streams.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("FLO_HIP_LIB") or os.path.join(_HERE, "libflo_hip.so")   # FLO_HIP_LIB: diagnostic builds
_LIB = None

OK = 0
MODE_LOSSLESS, MODE_LOSSY = 0, 1

EXPORTS = [
    "flo_ctx_create", "flo_ctx_destroy", "flo_last_error", "flo_last_create_error", "flo_free", "flo_ctx_device_info",
    "flo_encode_lossy", "flo_encode_lossless", "flo_encode_batch", "flo_decode", "flo_decode_lossless_i32", "flo_probe_container",
    "flo_batch_create", "flo_batch_destroy", "flo_batch_clip_device_ptr", "flo_batch_upload",
    "flo_batch_fill_synthetic", "flo_batch_encode", "flo_batch_sync", "flo_batch_data_bytes", "flo_batch_fetch",
    "flo_batch_device_streams", "flo_batch_pack_streams", "flo_batch_decode",
    "flo_batch_device_files", "flo_batch_pack_files",
    "flo_ctx_profile_enable", "flo_ctx_profile_query", "flo_ctx_profile_reset", "flo_ctx_force_path", "flo_ctx_stream",
    "flo_mdct_forward", "flo_lossy_analyze", "flo_lossy_quantize", "flo_lossy_quantize_smr", "flo_sparse_pack",
    "flo_dist_unique_id", "flo_dist_create", "flo_dist_destroy", "flo_dist_gather_submit", "flo_dist_gather_flush",
    "flo_dist_gather_result", "flo_dist_stream", "flo_ctx_reserve_cus", "flo_ctx_reserved_cus", "flo_ctx_upload_path",
    "flo_dist_table_submit", "flo_dist_table_flush", "flo_dist_table_result",
    "flo_stream_create", "flo_stream_destroy", "flo_stream_push", "flo_stream_pending_samples", "flo_stream_pending_frames",
    "flo_stream_next_frame", "flo_stream_flush", "flo_stream_finalize",
    "flo_analyze", "flo_analysis_metadata", "flo_batch_analysis_metadata", "flo_batch_set_bit_depth",
]


class Analysis(C.Structure):
    _fields_ = [("n_peaks", C.c_uint32), ("duration_ms", C.c_uint32), ("sample_rate", C.c_uint32), ("channels", C.c_uint8),
                ("avg_loudness", C.c_uint8), ("pad0", C.c_uint8), ("pad1", C.c_uint8), ("hash", C.c_uint8 * 32),
                ("frequency_peaks", C.c_uint8 * 8), ("energy_profile", C.c_uint8 * 16), ("integrated_lufs", C.c_double),
                ("length_ms", C.c_uint64), ("loudness_range_lu", C.c_double), ("true_peak_dbtp", C.c_double),
                ("sample_peak_dbfs", C.c_double), ("sum_squares", C.c_float), ("pad2", C.c_uint32)]


class ContainerInfo(C.Structure):
    _fields_ = [("version_major", C.c_uint8), ("version_minor", C.c_uint8), ("channels", C.c_uint8), ("bit_depth", C.c_uint8),
                ("compression_level", C.c_uint8), ("is_transform", C.c_uint8), ("pad0", C.c_uint8), ("pad1", C.c_uint8),
                ("flags", C.c_uint16), ("pad2", C.c_uint16), ("sample_rate", C.c_uint32), ("data_crc32", C.c_uint32),
                ("n_frames", C.c_uint32), ("total_samples", C.c_uint64), ("data_start", C.c_uint64), ("data_size", C.c_uint64),
                ("frame_samples_sum", C.c_uint64)]


class FloError(RuntimeError):
    pass


def lib_path():
    return _SO


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_SO):
        raise FloError(f"{_SO} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). flo_amd has no CPU fallback.")
    try:
        import torch  # noqa: F401  (one HIP runtime per process, see module docstring)
    except Exception:
        pass
    L = C.CDLL(_SO)
    vp, sz, u8p = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint8)
    L.flo_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.flo_ctx_destroy.argtypes = [vp]
    L.flo_ctx_destroy.restype = None
    L.flo_last_error.argtypes = [vp]
    L.flo_last_error.restype = C.c_char_p
    L.flo_last_create_error.restype = C.c_char_p
    L.flo_free.argtypes = [vp]
    L.flo_free.restype = None
    L.flo_ctx_device_info.argtypes = [vp, C.c_char_p, sz, C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    L.flo_encode_lossy.argtypes = [vp, vp, sz, C.c_uint32, C.c_uint8, C.c_float, C.c_char_p, sz, C.POINTER(vp), C.POINTER(sz)]
    L.flo_encode_lossless.argtypes = [vp, vp, sz, C.c_uint32, C.c_uint8, C.c_uint8, C.c_uint8, C.c_char_p, sz,
                                      C.POINTER(vp), C.POINTER(sz)]
    L.flo_encode_batch.argtypes = [vp, C.c_int, sz, C.POINTER(vp), C.POINTER(sz), C.c_uint32, C.c_uint8, C.c_float,
                                   C.POINTER(vp), C.POINTER(sz)]
    L.flo_batch_create.argtypes = [vp, C.c_int, sz, C.POINTER(sz), C.c_uint32, C.c_uint8, C.c_float, C.POINTER(vp)]
    L.flo_batch_destroy.argtypes = [vp]
    L.flo_batch_destroy.restype = None
    L.flo_batch_clip_device_ptr.argtypes = [vp, sz]
    L.flo_batch_clip_device_ptr.restype = vp
    L.flo_batch_upload.argtypes = [vp, sz, vp]
    L.flo_batch_fill_synthetic.argtypes = [vp, C.c_uint32, C.c_uint64]
    L.flo_batch_encode.argtypes = [vp, C.c_int]
    L.flo_batch_sync.argtypes = [vp]
    L.flo_batch_data_bytes.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.flo_batch_fetch.argtypes = [vp, sz, C.c_char_p, sz, C.POINTER(vp), C.POINTER(sz)]
    L.flo_batch_device_streams.argtypes = [vp, C.POINTER(vp), C.POINTER(C.POINTER(C.c_uint64)),
                                           C.POINTER(C.POINTER(C.c_uint64))]
    L.flo_batch_pack_streams.argtypes = [vp, vp, sz, C.POINTER(C.c_uint64)]
    L.flo_batch_pack_files.argtypes = [vp, vp, sz, C.POINTER(C.c_uint64)]
    L.flo_batch_device_files.argtypes = L.flo_batch_device_streams.argtypes
    L.flo_batch_decode.argtypes = [vp, vp, sz, C.POINTER(C.c_uint64)]
    L.flo_ctx_profile_enable.argtypes = [vp, C.c_int]
    L.flo_ctx_profile_query.argtypes = [vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.flo_ctx_profile_reset.argtypes = [vp]
    L.flo_ctx_force_path.argtypes = [vp, C.c_int]
    L.flo_ctx_stream.argtypes = [vp]
    L.flo_ctx_stream.restype = vp
    L.flo_mdct_forward.argtypes = [vp, vp, sz, vp]
    L.flo_lossy_analyze.argtypes = [vp, vp, sz, C.c_uint32, C.c_uint8, C.c_float, vp, vp, vp, C.POINTER(sz)]
    L.flo_lossy_quantize.argtypes = [vp, vp, sz, C.c_uint32, C.c_uint8, C.c_float, C.c_int, vp, vp]
    L.flo_lossy_quantize_smr.argtypes = [vp, vp, vp, sz, C.c_uint32, C.c_float, vp, vp]
    L.flo_sparse_pack.argtypes = [vp, vp, sz, C.c_int, vp, sz, vp]
    L.flo_dist_unique_id.argtypes = [vp]
    L.flo_dist_create.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.flo_dist_destroy.argtypes = [vp]
    L.flo_dist_destroy.restype = None
    L.flo_dist_gather_submit.argtypes = [vp, vp]
    L.flo_dist_gather_flush.argtypes = [vp]
    L.flo_dist_gather_result.argtypes = [vp, C.POINTER(vp), C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.POINTER(C.c_uint64))]
    L.flo_dist_stream.argtypes = [vp]
    L.flo_dist_stream.restype = vp
    L.flo_ctx_reserve_cus.argtypes = [vp, C.c_int]
    L.flo_ctx_reserve_cus.restype = C.c_int
    L.flo_batch_analysis_metadata.argtypes = [vp, sz, C.c_uint32, C.POINTER(vp), C.POINTER(sz)]
    L.flo_batch_set_bit_depth.argtypes = [vp, C.c_uint8]
    L.flo_ctx_upload_path.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.flo_ctx_reserved_cus.argtypes = [vp]
    L.flo_ctx_reserved_cus.restype = C.c_int
    L.flo_dist_table_submit.argtypes = [vp, vp, C.c_size_t]
    L.flo_dist_table_flush.argtypes = [vp]
    L.flo_dist_table_result.argtypes = [vp, C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    u32p = C.POINTER(C.c_uint32)
    L.flo_stream_create.argtypes = [vp, C.c_uint32, C.c_uint8, C.c_uint8, C.c_uint8, C.POINTER(vp)]
    L.flo_stream_destroy.argtypes = [vp]
    L.flo_stream_destroy.restype = None
    L.flo_stream_push.argtypes = [vp, vp, sz]
    L.flo_stream_pending_samples.argtypes = [vp]
    L.flo_stream_pending_samples.restype = sz
    L.flo_stream_pending_frames.argtypes = [vp]
    L.flo_stream_pending_frames.restype = sz
    L.flo_stream_next_frame.argtypes = [vp, u32p, u32p, u32p, C.POINTER(vp), C.POINTER(sz)]
    L.flo_stream_flush.argtypes = L.flo_stream_next_frame.argtypes
    L.flo_stream_finalize.argtypes = [vp, C.c_char_p, sz, C.POINTER(vp), C.POINTER(sz)]
    L.flo_analyze.argtypes = [vp, vp, sz, C.c_uint32, C.c_uint8, C.c_uint32, vp, sz, C.POINTER(Analysis)]
    L.flo_analysis_metadata.argtypes = [vp, vp, sz, C.c_uint32, C.c_uint8, C.c_uint32, C.POINTER(vp), C.POINTER(sz)]
    L.flo_decode.argtypes = [vp, C.c_char_p, sz, C.POINTER(vp), C.POINTER(sz), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)]
    L.flo_decode_lossless_i32.argtypes = L.flo_decode.argtypes
    L.flo_probe_container.argtypes = [C.c_char_p, sz, C.POINTER(ContainerInfo), C.c_char_p, sz]
    _LIB = L
    return L
