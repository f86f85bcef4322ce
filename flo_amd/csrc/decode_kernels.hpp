// decode_kernels.hpp — argument blocks and launchers of the device decode path (see decode_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lossy_device.hpp"

namespace flo {

// One transform frame = one wavefront. Frames of a clip are addressed by (clip, local frame index).
struct LossyDecArgs {
    LossyDevTables T;                 // pack (rotation + FFT twiddles) and the coefficient -> band map of the file's sample rate
    const float *window;              // [2048] Vorbis window (mdct.rs:106-113)
    const uint8_t *bytes;             // device copy of the file(s)
    const unsigned long long *blob_off;   // [total_frames] offset of the frame's blob (channel 0 "residuals")
    const unsigned int *blob_len;         // [total_frames]
    const unsigned long long *clip_frame0;  // [n_clips] first frame of the clip
    const unsigned int *clip_frames;        // [n_clips] decodable frames of the clip
    const unsigned long long *clip_out;     // [n_clips] float offset of the clip's PCM in `out`
    int n_clips;
    int channels;                     // header channel count (output interleave)
    float *out;                       // zero-filled: (frames - 1) * 1024 * channels floats per clip
    int *error;                       // set to 1 when a frame cannot be deserialised
};

// One ALPC / raw / silent channel wrapper of one frame = one thread.
struct LlChannelDev {
    unsigned long long off;           // payload offset
    unsigned long long out_off;       // int32 offset of this channel-frame in the planar scratch
    unsigned int len;                 // payload bytes
    unsigned int samples;             // frame_samples
    unsigned char n_coeffs, shift_bits, rice_k, pad;
    int coeffs[12];
};
struct LlDecArgs {
    const uint8_t *bytes;
    const LlChannelDev *ch;
    unsigned int n_ch;
    int *scratch;                     // decoded integers, one run per channel wrapper
};
// Per frame: mid/side, interleave, int -> float.
struct LlFrameDev {
    unsigned long long out_off;       // sample-frame offset of the frame in the output
    unsigned long long scratch_off[2];  // first two channel wrappers (mid/side needs exactly two)
    unsigned int first_channel, n_channels;
    unsigned int samples;
    unsigned int mid_side;
};
struct LlFinishArgs {
    const LlFrameDev *fr;
    const LlChannelDev *ch;
    unsigned int n_frames;
    int channels;
    const int *scratch;
    float *out;                       // zero-filled interleaved f32 (nullable)
    int *out_i32;                     // zero-filled interleaved i32 (nullable; parity tests)
};

int launch_lossy_decode(const LossyDecArgs &A, unsigned max_frames, hipStream_t s);
int launch_ll_decode(const LlDecArgs &A, hipStream_t s);
int launch_ll_finish(const LlFinishArgs &A, unsigned max_samples, hipStream_t s);

}  // namespace flo
