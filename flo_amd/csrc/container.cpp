// container.cpp — see container.hpp.
#include "container.hpp"

#include <cstdlib>
#include <cstring>

namespace flo {

// ---------------------------------------------------------------------------------------------------- reading
namespace {
struct Cur {   // reader.rs uses a Cursor whose reads fail with "Unexpected end of file"
    const uint8_t *d;
    size_t len, pos;
    bool err;
    bool need(size_t n) {
        if (pos + n > len) { err = true; return false; }
        return true;
    }
    uint8_t u8() {
        if (pos >= len) { err = true; return 0; }
        return d[pos++];
    }
    uint32_t u32() {
        if (!need(4)) return 0;
        const uint8_t *p = d + pos;
        pos += 4;
        return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    }
    uint16_t u16() {
        if (!need(2)) return 0;
        uint16_t v = (uint16_t)(d[pos] | (d[pos + 1] << 8));
        pos += 2;
        return v;
    }
    uint64_t u64() {
        uint64_t lo = u32();
        uint64_t hi = u32();
        return lo | (hi << 32);
    }
    void skip(size_t n) { pos = (pos + n < len) ? pos + n : len; }
};
const char *kEof = "Unexpected end of file";
}  // namespace

// reader.rs:168-256 read_channel_data
static int parse_channel(Cur &c, uint8_t frame_type, size_t frame_samples, size_t channel_end, ChannelDesc &ch, const char **err) {
    memset(&ch, 0, sizeof ch);
    if (frame_samples > 2000000) {
        *err = "Invalid frame: too many samples";
        return -1;
    }
    auto rest = [&]() -> int {   // everything up to the end of the channel wrapper is the payload
        size_t rem = channel_end > c.pos ? channel_end - c.pos : 0;
        if (rem) {
            if (!c.need(rem)) return -1;
            ch.off = c.pos;
            ch.len = (uint32_t)rem;
            c.pos += rem;
        }
        return 0;
    };
    if (frame_type == 0) return 0;   // Silence
    if (frame_type == 254) {         // Raw: at most frame_samples i16
        size_t need = frame_samples * 2;
        size_t avail = channel_end > c.pos ? channel_end - c.pos : 0;
        size_t n = need < avail ? need : avail;
        if (!c.need(n)) return -1;
        ch.off = c.pos;
        ch.len = (uint32_t)n;
        c.pos += n;
        return 0;
    }
    if (frame_type == 253) return rest();   // Transform blob
    if (frame_type >= 1 && frame_type <= 12) {
        size_t order = c.u8();
        if (c.err) return -1;
        if (order > 12) {
            *err = "Invalid LPC order";
            return -1;
        }
        for (size_t i = 0; i < order; i++) {
            if (c.pos + 4 > channel_end) break;
            ch.coeffs[ch.n_coeffs++] = (int32_t)c.u32();
        }
        ch.shift_bits = c.u8();
        uint8_t enc = c.u8();
        ch.rice_k = enc == 0 ? c.u8() : 0;
        if (c.err) return -1;
        return rest();
    }
    return 0;   // reserved types read as silence
}

int parse_file(const uint8_t *data, size_t len, ParsedFile &f, const char **err) {
    static const char *none = "";
    *err = none;
    f = ParsedFile();
    if (len < 4) { *err = kEof; return -1; }
    if (memcmp(data, "FLO!", 4) != 0) { *err = "Invalid flo file: bad magic"; return -1; }
    Cur c{data, len, 4, false};
    f.version_major = c.u8();
    f.version_minor = c.u8();
    f.flags = c.u16();
    f.sample_rate = c.u32();
    f.channels = c.u8();
    f.bit_depth = c.u8();
    f.total_samples = c.u64();
    f.compression_level = c.u8();
    c.skip(3);
    f.data_crc32 = c.u32();
    (void)c.u64();   // header_size
    uint64_t toc_size = c.u64();
    f.data_size = c.u64();
    uint64_t extra_size = c.u64();
    uint64_t meta_size = c.u64();
    if (c.err) { *err = kEof; return -1; }
    struct Toc { uint64_t off; uint32_t size; };
    std::vector<Toc> toc;
    if (toc_size >= 4) {   // reader.rs:76-99
        size_t n = c.u32();
        if (c.err) { *err = kEof; return -1; }
        if (n > 100000) { *err = "Invalid TOC: too many entries"; return -1; }
        toc.resize(n);
        for (size_t i = 0; i < n; i++) {
            (void)c.u32();
            toc[i].off = c.u64();
            toc[i].size = c.u32();
            (void)c.u32();
            if (c.err) { *err = kEof; return -1; }
        }
    }
    f.data_start = c.pos;
    const size_t data_end = c.pos + (size_t)f.data_size;
    for (size_t i = 0; i < toc.size(); i++) {   // reader.rs:101-166
        size_t fs = (size_t)f.data_start + (size_t)toc[i].off;
        if (fs >= data_end) break;
        c.pos = fs;
        const size_t frame_end = fs + toc[i].size;
        FrameDesc fr{};
        fr.type = c.u8();
        fr.samples = c.u32();
        fr.flags = c.u8();
        if (c.err) { *err = kEof; return -1; }
        fr.first_channel = (uint32_t)f.channels_desc.size();
        const size_t nch = fr.type == 253 ? 1 : f.channels;
        for (size_t k = 0; k < nch; k++) {
            size_t ch_size = c.u32();
            if (c.err) { *err = kEof; return -1; }
            size_t ch_end = c.pos + ch_size;
            ChannelDesc cd;
            if (parse_channel(c, fr.type, fr.samples, ch_end, cd, err) != 0 || c.err) {
                if (*err == none) *err = kEof;
                return -1;
            }
            f.channels_desc.push_back(cd);
            fr.n_channels++;
            c.pos = ch_end;
        }
        c.pos = frame_end;
        if (fr.type == 253) f.is_transform = true;
        f.frames.push_back(fr);
    }
    c.pos = data_end;
    c.skip((size_t)extra_size);
    if (c.pos + (size_t)meta_size > len) { *err = kEof; return -1; }
    return 0;
}

uint32_t host_crc32(const uint8_t *data, size_t len) {
    struct Table {
        uint32_t v[256];
    };
    static const Table tbl = [] {   // thread-safe one-time initialisation
        Table t{};
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
            t.v[i] = c;
        }
        return t;
    }();
    const uint32_t (&table)[256] = tbl.v;
    uint32_t crc = 0xFFFFFFFFu;
    for (size_t i = 0; i < len; i++) crc = table[(crc ^ data[i]) & 0xFFu] ^ (crc >> 8);
    return crc ^ 0xFFFFFFFFu;
}

}  // namespace flo
