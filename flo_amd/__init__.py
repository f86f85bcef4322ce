"""flo_amd — MI355X-native batch encoder for the .flo audio format (libflo's per-frame encode hot path).

The compute path is flo_amd/libflo_hip.so (hand-written gfx950 HIP kernels behind the C ABI in include/flo_hip.h);
this package is only the host-side mirror of the reference's encoder interface.
"""
from ._native import FloError, MODE_LOSSLESS, MODE_LOSSY  # noqa: F401
from .api import (Batch, Context, Decoder, EncodedFrame, Encoder, LossyEncoder, QualityPreset, StreamingEncoder,  # noqa: F401
                  TransformEncoder, default_context, decode, encode, encode_lossy, encode_with_bitrate, probe_container)
