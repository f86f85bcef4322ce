// container.cpp — see container.hpp.
#include "container.hpp"

#include <cstdlib>
#include <cstring>

namespace flo {

static uint32_t g_tab[8][256];
static bool g_tab_ready = false;

static void crc_init() {
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = i;
        for (int j = 0; j < 8; j++) c = (c & 1) ? (c >> 1) ^ 0xEDB88320u : c >> 1;
        g_tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t c = g_tab[0][i];
        for (int t = 1; t < 8; t++) {
            c = g_tab[0][c & 0xFF] ^ (c >> 8);
            g_tab[t][i] = c;
        }
    }
    g_tab_ready = true;
}

uint32_t crc32_ieee(const uint8_t *p, size_t n) {
    if (!g_tab_ready) crc_init();
    uint32_t crc = 0xFFFFFFFFu;
    while (n && (reinterpret_cast<uintptr_t>(p) & 7)) {
        crc = g_tab[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
        n--;
    }
    while (n >= 8) {
        uint32_t a, b;
        memcpy(&a, p, 4);
        memcpy(&b, p + 4, 4);
        a ^= crc;
        crc = g_tab[7][a & 0xFF] ^ g_tab[6][(a >> 8) & 0xFF] ^ g_tab[5][(a >> 16) & 0xFF] ^ g_tab[4][a >> 24] ^
              g_tab[3][b & 0xFF] ^ g_tab[2][(b >> 8) & 0xFF] ^ g_tab[1][(b >> 16) & 0xFF] ^ g_tab[0][b >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) crc = g_tab[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
    return ~crc;
}

static inline void put16(uint8_t *&p, uint16_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p += 2; }
static inline void put32(uint8_t *&p, uint32_t v) { for (int i = 0; i < 4; i++) p[i] = (uint8_t)(v >> (8 * i)); p += 4; }
static inline void put64(uint8_t *&p, uint64_t v) { for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i)); p += 8; }

uint8_t *assemble_file(const FileParams &fp, const uint8_t *data, size_t data_len, const uint32_t *frame_sizes,
                       const uint32_t *frame_samples, size_t n_frames, const uint8_t *meta, size_t meta_len,
                       size_t *out_len) {
    const uint64_t toc_size = 4 + (uint64_t)n_frames * 20;  // writer.rs:51
    const size_t total = 70 + (size_t)toc_size + data_len + meta_len;
    uint8_t *buf = (uint8_t *)malloc(total ? total : 1);
    if (!buf) return nullptr;
    uint8_t *p = buf;
    uint64_t total_samples = 0;
    for (size_t i = 0; i < n_frames; i++) total_samples += frame_samples[i];
    uint16_t flags = 0;
    if (fp.lossy) flags = (uint16_t)(0x01 | ((uint16_t)fp.lossy_quality << 8));  // writer.rs:64-68
    // header (writer.rs:132-191)
    memcpy(p, "FLO!", 4);
    p += 4;
    *p++ = 1;  // version 1.2 (core/types.rs:12-13)
    *p++ = 2;
    put16(p, flags);
    put32(p, fp.sample_rate);
    *p++ = fp.channels;
    *p++ = fp.bit_depth;
    put64(p, total_samples);
    *p++ = fp.compression_level;
    *p++ = 0;
    *p++ = 0;
    *p++ = 0;
    put32(p, crc32_ieee(data, data_len));
    put64(p, 66);
    put64(p, toc_size);
    put64(p, (uint64_t)data_len);
    put64(p, 0);
    put64(p, (uint64_t)meta_len);
    // TOC (writer.rs:193-224)
    put32(p, (uint32_t)n_frames);
    uint64_t byte_offset = 0, cumulative = 0;
    for (size_t i = 0; i < n_frames; i++) {
        put32(p, (uint32_t)i);
        put64(p, byte_offset);
        put32(p, frame_sizes[i]);
        put32(p, (uint32_t)(cumulative * 1000 / (uint64_t)fp.sample_rate));
        byte_offset += frame_sizes[i];
        cumulative += frame_samples[i];
    }
    if (data_len) memcpy(p, data, data_len);
    p += data_len;
    if (meta_len) memcpy(p, meta, meta_len);
    *out_len = total;
    return buf;
}

}  // namespace flo
