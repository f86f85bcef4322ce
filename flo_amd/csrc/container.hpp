// container.hpp — host side of the .flo container in the product library: the reader (Reader::read, reader.rs:16-256)
// that turns a file into flat descriptors for the decode kernels. Writing (header, TOC, CRC32: writer.rs:39-224,
// core/crc32.rs) happens on the device, see container_kernels.hip. Independent of the oracle (oracle/ is test
// infrastructure and is never linked here).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace flo {

// ---- reading (decode side) --------------------------------------------------------------------------------------
// What Reader::read (reader.rs:16-256) extracts from a file, as flat descriptors the decode kernels consume. Payload
// bytes are not copied: descriptors carry offsets into the file.
struct ChannelDesc {
    uint8_t n_coeffs;     // LPC coefficients present (ALPC frames)
    uint8_t shift_bits;   // LPC shift, or 128 + order for the fixed predictors
    uint8_t rice_k;       // Rice parameter (0 unless the channel says Rice)
    uint8_t pad;
    uint32_t len;         // payload ("residuals") bytes
    uint64_t off;         // payload offset in the file
    int32_t coeffs[12];
};
struct FrameDesc {
    uint8_t type;         // types.rs FrameType: 0 silence, 1..12 ALPC, 253 transform, 254 raw
    uint8_t flags;        // bit 0: mid/side
    uint16_t n_channels;  // channel wrappers read (1 for transform frames)
    uint32_t samples;     // frame_samples
    uint32_t first_channel;  // index into ParsedFile::channels
};
struct ParsedFile {
    uint8_t version_major = 0, version_minor = 0;
    uint16_t flags = 0;
    uint32_t sample_rate = 0;
    uint8_t channels = 0, bit_depth = 0, compression_level = 0;
    uint64_t total_samples = 0;
    uint32_t data_crc32 = 0;
    uint64_t data_start = 0, data_size = 0;
    bool is_transform = false;   // any frame of type 253 (lib.rs:302-306)
    std::vector<FrameDesc> frames;
    std::vector<ChannelDesc> channels_desc;
};
// 0 on success; otherwise `err` holds the reference reader's message ("Invalid flo file: bad magic", ...)
int parse_file(const uint8_t *data, size_t len, ParsedFile &out, const char **err);

// CRC32 (IEEE, core/crc32.rs:2-30) of host bytes: only the streaming encoder's finalize uses it, for a DATA chunk it
// assembles on the host from frames the caller may already have pulled (batches compute theirs on the device)
uint32_t host_crc32(const uint8_t *data, size_t len);

}  // namespace flo
