// container.hpp — host-side .flo framing of the product library (header, TOC, CRC32, chunk concatenation).
// Replaces libflo/src/writer.rs:39-224 and core/crc32.rs:2-30 for files whose DATA chunk was produced on the
// device. Independent of the oracle (oracle/ is test infrastructure and is never linked here).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace flo {

// IEEE CRC32 (poly 0xEDB88320, init/xorout 0xFFFFFFFF), slicing-by-8
uint32_t crc32_ieee(const uint8_t *data, size_t n);

struct FileParams {
    uint32_t sample_rate;
    uint8_t channels;
    uint8_t bit_depth;
    uint8_t compression_level;
    bool lossy;
    uint8_t lossy_quality;  // 0..4
};

// Assemble header + TOC + DATA + META. frame_sizes[i] / frame_samples[i] describe frame i of the DATA chunk
// (frames are back to back). Returns a malloc'ed buffer (caller frees with free()).
uint8_t *assemble_file(const FileParams &p, const uint8_t *data, size_t data_len, const uint32_t *frame_sizes,
                       const uint32_t *frame_samples, size_t n_frames, const uint8_t *meta, size_t meta_len,
                       size_t *out_len);

}  // namespace flo
