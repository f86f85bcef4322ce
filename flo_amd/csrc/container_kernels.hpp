// container_kernels.hpp — on-device .flo framing: header, TOC and CRC32 of every clip of a batch, written
// in front of the DATA chunk the encode kernels left in HBM, so that [file_off, file_off + head + data) is a finished
// .flo file (META, if any, is appended by the caller, who then patches meta_size).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace flo {

struct FinishArgs {
    uint8_t *out;                           // the batch's output buffer
    const unsigned long long *data_off;     // [n_clips] DATA chunk start (16-byte aligned); the file starts 74 + 20 frames before
    const unsigned long long *clip_bytes;   // [n_clips] DATA chunk length (written by the encode kernels)
    const unsigned long long *clip_frame0;  // [n_clips] first frame of the clip among all frames
    const unsigned int *clip_frames;        // [n_clips] frames per clip
    const unsigned int *frame_size;         // [total_frames] bytes per frame
    const unsigned int *frame_samples;      // [total_frames] or null: every frame holds const_samples
    unsigned int const_samples;
    unsigned int sample_rate;
    unsigned short flags;                   // header flags (writer.rs:64-68)
    unsigned char channels, bit_depth, level;
    int n_clips;
    unsigned int *crc_out;                  // [n_clips] CRC32 of each DATA chunk (also in the header)
    unsigned int parts;                     // slices per clip for the CRC (1..128): few long clips still fill the chip
    unsigned int *part_reg;                 // [n_clips * parts] scratch: CRC register of every slice
    unsigned int max_frames;                // frames of the longest clip (0 = unknown: one workgroup writes a clip's whole TOC)
    unsigned int toc_chunk;                 // set by launch_finish_files: frames per TOC workgroup (0 = all)
    unsigned int mode;                      // set by launch_finish_files: 0 TOC + CRC + header, 1 the TOC (+ total_samples) only, 2 CRC + header only
    // powers of x modulo the CRC polynomial (reflected), filled in by launch_finish_files
    unsigned int x8pow2[40];                // x^(8 * 2^j)
    unsigned int blk_pow[256];              // x^(8 * 64 * i)
    unsigned int byte_pow[64];              // x^(8 * i)
    unsigned int stripe_pow[256];           // x^(8 * 16384 * i)
    const unsigned int *tables;             // device: tab[4][256] byte tables, then skip[4][256] (times x^(8 * 16320))
};

int launch_finish_files(FinishArgs A, hipStream_t s);
// Location table of a batch's finished files (flo_dist_table_*): row = [0] n | [1 .. max] sizes | [1 + max .. 2 max] offsets
// | [1 + 2 max .. 3 max] CRC32 of DATA. Sizes and offsets are in the row already; this fills the CRC column from the
// headers of the files at base + offset (byte 26).
int launch_table_crcs(const uint8_t *base, unsigned long long *row, unsigned long long n, unsigned long long max_clips, hipStream_t s);
// slices per clip so that about two thousand workgroups run; at most 128, or 512 for the few-long-clips case whose
// finish_files_kernel runs 1024 threads (one per slice register)
inline unsigned finish_parts_for(size_t n_clips) {
    if (n_clips >= 1024) return 1;
    size_t p = (2048 + n_clips - 1) / (n_clips ? n_clips : 1);
    const size_t cap = n_clips < 64 ? 512 : 128;
    return (unsigned)(p > cap ? cap : p);
}

}  // namespace flo
