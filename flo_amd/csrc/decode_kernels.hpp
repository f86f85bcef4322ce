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
    float *out;                       // (frames - 1) * 1024 * channels floats per clip, every one written
    int *error;                       // set to 1 when a frame cannot be deserialised
    int run;                          // output blocks per wavefront (set by the launcher)
    unsigned int n_runs;              // runs per clip (set by the launcher)
    unsigned long long *dbg;          // FLO_DEC_STAMPS builds only: phase tick sums (set by the launcher)
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
    const int *only;                  // nullable [n_ch]: decode only the wrappers whose entry is nonzero
};

// Parallel form of the ALPC decode (lldec_kernels.hip). A Rice stream is cut into tiles of kRiceTileBits bits;
// `tile0` is the running tile count over the wrappers (0 tiles for raw / silent wrappers and for those the host
// already handed to the serial kernel through `serial`).
constexpr int kRiceTileBits = 1024;
constexpr int kRiceStates = 16;       // entry states of a tile: skip 0..k bits (k <= 14), or "inside a unary run" (k + 1)
constexpr int kRiceMaxK = kRiceStates - 2;
struct LlParArgs {
    const uint8_t *bytes;
    const LlChannelDev *ch;
    unsigned int n_ch;
    int *scratch;                     // residuals, then samples in place (every wrapper's kernels write all of its samples)
    const unsigned int *tile0;        // [n_ch + 1]
    unsigned int *tabs;               // [tiles][kRiceStates]: exit state | codes started << 5, per entry state
    uint2 *tile_entry;                // [tiles]: (index of the first code that starts in the tile, entry state)
    int *serial;                      // [n_ch] nonzero: the serial kernel decodes this wrapper (set by the host for
                                      // k > 14 or large coefficients, by the device for a 256-ones escape or a sample
                                      // outside i32)
    const unsigned int *others;       // [n_others] the wrappers that are no LPC recurrence (fixed predictors, raw, silent, too short): one
    unsigned int n_others;            // workgroup each in ll_predict behind the LPC groups (a workgroup per wrapper cost 0.2 ms in dispatch alone)
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
int launch_ll_decode_parallel(const LlParArgs &A, unsigned max_tiles, hipStream_t s);
int launch_ll_finish(const LlFinishArgs &A, unsigned max_samples, hipStream_t s);

}  // namespace flo
