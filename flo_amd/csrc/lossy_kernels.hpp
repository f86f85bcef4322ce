// lossy_kernels.hpp — kernel argument block and launchers of the lossy path (see lossy_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lossy_device.hpp"

namespace flo {

struct LossyArgs {
    LossyDevTables T;
    // input
    const float *pcm;                        // all clips, interleaved f32
    const unsigned long long *clip_off;      // [n_clips] float offset of the clip (multiple of 4)
    const unsigned long long *clip_nsf;      // [n_clips] sample-frames per clip
    const unsigned int *clip_hops;           // [n_clips] frames per clip: ceil((nsf + 1024) / 1024)
    const unsigned long long *clip_frame0;   // [n_clips] index of the clip's first frame among all frames
    int nch;
    int n_clips;
    unsigned long long total_frames;
    unsigned int max_hops;                   // longest clip, in frames
    // output
    uint8_t *out;                            // DATA chunks
    const unsigned long long *out_off;       // [n_clips] byte offset of the clip's DATA chunk (16-byte aligned)
    unsigned int *frame_size;                // [total_frames]
    unsigned long long *clip_bytes;          // [n_clips] DATA chunk size
    // frame-parallel form
    float *a_t;                              // [total_frames][nch][32] masking level before temporal masking
    float *bmax_t;                           // [total_frames][nch][32] band maxima (frame-parallel stereo form: pass 1 leaves them for pass 2)
    float4 *coef_t;                          // nullable [total_frames][8][64]: the stereo coefficients as pass 1's lanes hold them (few frames only:
                                             // 8 KB per frame; pass 2 reads them back instead of transforming again)
    float *s_prev_out;                       // [total_frames][nch][32] scan output
    const float *s_prev;                     // same buffer, read by pass 2
    uint8_t *slots;                          // [total_frames][slot_bytes]
    unsigned int slot_bytes;                 // kFrameCap for 1-2 channels, lossy_slot_bytes(nch) beyond
    unsigned long long *frame_off;           // [total_frames] offset of the frame inside its clip's DATA chunk
    // analysis / stage tests (may be null)
    float *dbg_coeffs;                       // [total_frames][nch][1024]
    short *dbg_q;                            // [total_frames][nch][1024]
    unsigned short *dbg_sfw;                 // [total_frames][nch][25]
    const float *in_coeffs;                  // when set: skip the transform, quantise these spectra
    unsigned long long *dbg_stamps;          // diagnostic builds (FLO_STAMPS): per-wave phase cycle sums [wave][16]
    int exact;                               // re-decide near-threshold coefficients with the reference's dB expression
    unsigned int *next_clip;                 // lock-step stereo form: batch-wide counter of claimed clips (zero at launch)
    int n_cus;                               // compute units of the device (persistent workgroups)
};

int launch_lossy_chain(const LossyArgs &A, hipStream_t s);
int launch_lossy_chain2q(const LossyArgs &A, hipStream_t s);  // stereo only: one lock-step transform wave + one quantiser-and-packer wave per clip
int launch_lossy_frames_pass(const LossyArgs &A, int pass, hipStream_t s);
// bytes reserved per frame in the frame-parallel form: header + scale words + every channel's largest sparse blob
inline unsigned int lossy_slot_bytes(int nch) {
    unsigned int need = 12u + 50u * (unsigned)nch + (unsigned)nch * (4u + 2064u) + 16u;
    need = (need + 15u) & ~15u;
    return need < (unsigned)kFrameCap ? (unsigned)kFrameCap : need;
}
constexpr int kMaxLossyChannels = 8;      // more channels than two take the generic frame-parallel kernel
int launch_lossy_scan(const LossyArgs &A, hipStream_t s);
int launch_lossy_compact(const LossyArgs &A, hipStream_t s);
int launch_mdct_only(const LossyDevTables &T, const float *frames, unsigned long long n, float *out, hipStream_t s);
int launch_quantise_smr(const LossyDevTables &T, const float *coeffs, const float *smr, unsigned long long n, short *q, float *sf,
                        hipStream_t s);
int launch_sparse_only(const short *q, unsigned long long n, uint8_t *slots, uint32_t *sizes, int form, hipStream_t s);
int launch_pack_streams(const uint8_t *src, const unsigned long long *src_off, const unsigned long long *dst_off,
                        const unsigned long long *sizes, int n_clips, uint8_t *dst, hipStream_t s);
int launch_synth_fill(float *pcm, const unsigned long long *clip_off, const unsigned long long *clip_nsf, int n_clips,
                      int nch, uint32_t seed, unsigned long long clip_id0, hipStream_t s);

}  // namespace flo
