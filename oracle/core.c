/* core.c — oracle restatement of libflo/src/core/{crc32,rice,audio_constants,types}.rs
 * TEST INFRASTRUCTURE (see flo_oracle.h). */
#include "internal.h"

/* ------------------------------------------------------------------ buffers */
void buf_init(flo_buf *b) { b->data = NULL; b->len = b->cap = 0; }
void flo_buf_free(flo_buf *b) { free(b->data); buf_init(b); }
void flo_o_free(void *p) { free(p); }
void buf_reserve(flo_buf *b, size_t extra) {
    if (b->len + extra <= b->cap) return;
    size_t nc = b->cap ? b->cap * 2 : 64;
    while (nc < b->len + extra) nc *= 2;
    b->data = (uint8_t *)realloc(b->data, nc);
    if (!b->data) abort();
    b->cap = nc;
}
void buf_push(flo_buf *b, uint8_t v) { buf_reserve(b, 1); b->data[b->len++] = v; }
void buf_extend(flo_buf *b, const void *p, size_t n) {
    if (!n) return;
    buf_reserve(b, n);
    memcpy(b->data + b->len, p, n);
    b->len += n;
}
void buf_u16le(flo_buf *b, uint16_t v) { uint8_t t[2] = {(uint8_t)v, (uint8_t)(v >> 8)}; buf_extend(b, t, 2); }
void buf_u32le(flo_buf *b, uint32_t v) {
    uint8_t t[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)};
    buf_extend(b, t, 4);
}
void buf_u64le(flo_buf *b, uint64_t v) { buf_u32le(b, (uint32_t)v); buf_u32le(b, (uint32_t)(v >> 32)); }

static __thread char g_err[256];
void set_error(const char *msg) { strncpy(g_err, msg, sizeof g_err - 1); g_err[sizeof g_err - 1] = 0; }
const char *flo_o_last_error(void) { return g_err; }

/* ------------------------------------------------------------------ crc32.rs:2-30 */
static uint32_t crc_table[256];
static int crc_table_ready;
static void crc_init(void) {
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t crc = i;
        for (int j = 0; j < 8; j++) crc = (crc & 1) ? (crc >> 1) ^ 0xEDB88320u : crc >> 1;
        crc_table[i] = crc;
    }
    crc_table_ready = 1;
}
uint32_t flo_o_crc32(const uint8_t *data, size_t n) {
    if (!crc_table_ready) crc_init();
    uint32_t crc = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) crc = (crc >> 8) ^ crc_table[(crc ^ data[i]) & 0xFF];
    return ~crc;
}

/* ------------------------------------------------------------------ audio_constants.rs:18-26 */
int32_t flo_o_f32_to_i32(float s) {
    float v = s * 32767.0f;
    /* f32::clamp: NaN passes through, then `as i32` saturates / maps NaN to 0 */
    if (v < -32768.0f) v = -32768.0f;
    if (v > 32767.0f) v = 32767.0f;
    if (v != v) return 0;
    return (int32_t)v; /* truncation toward zero */
}
float flo_o_i32_to_f32(int32_t s) {
    const float scale = 1.0f / 32767.0f; /* I16_TO_F32_SCALE, const-evaluated in f32 */
    return (float)s * scale;
}

/* ------------------------------------------------------------------ rice.rs */
static uint32_t unsigned_abs(int32_t r) { return r < 0 ? (uint32_t)(-(int64_t)r) : (uint32_t)r; }
static unsigned bitlen64(uint64_t v) { return v ? 64u - (unsigned)__builtin_clzll(v) : 0u; }
static unsigned bitlen32(uint32_t v) { return v ? 32u - (unsigned)__builtin_clz(v) : 0u; }

/* rice.rs:29-69 */
uint8_t flo_o_estimate_rice_parameter_i32(const int32_t *res, size_t n) {
    if (n == 0) return 4;
    uint64_t max_abs = 0;
    for (size_t i = 0; i < n; i++) {
        uint64_t a = unsigned_abs(res[i]);
        if (a > max_abs) max_abs = a;
    }
    if (max_abs == 0) return 0;
    uint64_t max_unsigned = 2 * max_abs;
    unsigned min_k = 0;
    if (max_unsigned > 255) {
        unsigned bits_needed = bitlen64(max_unsigned);
        min_k = bits_needed >= 8 ? bits_needed - 8 : 0; /* saturating_sub(8) as u8 */
    }
    uint64_t sum = 0;
    for (size_t i = 0; i < n; i++) sum += unsigned_abs(res[i]);
    uint32_t mean = (uint32_t)(sum / (uint64_t)n);
    unsigned mean_k = mean > 0 ? bitlen32(mean) : 0;
    unsigned k = min_k > mean_k ? min_k : mean_k;
    if (k > 15) k = 15;
    return (uint8_t)k;
}

/* rice.rs:162-202 BitWriter */
typedef struct {
    flo_buf bytes;
    uint8_t current_byte, bit_pos;
} bitwriter;
static void bw_write_bit(bitwriter *w, uint32_t bit) {
    if (bit) w->current_byte |= (uint8_t)(1u << (7 - w->bit_pos));
    w->bit_pos++;
    if (w->bit_pos == 8) {
        buf_push(&w->bytes, w->current_byte);
        w->current_byte = 0;
        w->bit_pos = 0;
    }
}

/* rice.rs:94-114 encode_sample */
static void rice_encode_sample(bitwriter *w, int32_t sample, uint8_t k) {
    uint32_t u = ((uint32_t)sample << 1) ^ (uint32_t)(sample >> 31);
    uint32_t quotient = u >> k;
    uint32_t remainder = u & ((1u << k) - 1u);
    uint32_t q_capped = quotient < 255 ? quotient : 255;
    for (uint32_t i = 0; i < q_capped; i++) bw_write_bit(w, 1);
    bw_write_bit(w, 0);
    for (int i = (int)k - 1; i >= 0; i--) bw_write_bit(w, (remainder >> i) & 1);
}

/* rice.rs:84-92 encode_i32 */
int flo_o_rice_encode_i32(const int32_t *res, size_t n, uint8_t k, uint8_t **out, size_t *out_len) {
    bitwriter w;
    buf_init(&w.bytes);
    w.current_byte = 0;
    w.bit_pos = 0;
    for (size_t i = 0; i < n; i++) rice_encode_sample(&w, res[i], k);
    if (w.bit_pos > 0) buf_push(&w.bytes, w.current_byte); /* into_bytes :197-202 */
    *out = w.bytes.data;
    *out_len = w.bytes.len;
    return 0;
}

/* rice.rs:217-259 BitReader */
typedef struct {
    const uint8_t *bytes;
    size_t len, byte_pos;
    uint8_t bit_pos;
} bitreader;
static uint32_t br_read_bit(bitreader *r) {
    if (r->byte_pos >= r->len) return 0;
    uint32_t bit = (r->bytes[r->byte_pos] >> (7 - r->bit_pos)) & 1u;
    r->bit_pos++;
    if (r->bit_pos == 8) {
        r->bit_pos = 0;
        r->byte_pos++;
    }
    return bit;
}
static int br_exhausted(const bitreader *r) { return r->byte_pos >= r->len; }

/* rice.rs:123-159 decode_i32 */
void flo_o_rice_decode_i32(const uint8_t *enc, size_t enc_len, uint8_t k, size_t target_len, int32_t *out) {
    bitreader r = {enc, enc_len, 0, 0};
    for (size_t n = 0; n < target_len; n++) {
        if (br_exhausted(&r)) {
            out[n] = 0;
            continue;
        }
        uint32_t quotient = 0;
        while (!br_exhausted(&r) && br_read_bit(&r) == 1) {
            quotient++;
            if (quotient > 255) break;
        }
        uint32_t remainder = 0;
        for (uint8_t i = 0; i < k; i++) remainder = (remainder << 1) | br_read_bit(&r);
        uint32_t u = (quotient << k) | remainder;
        out[n] = (int32_t)(u >> 1) ^ -(int32_t)(u & 1);
    }
}

/* ------------------------------------------------------------------ types.rs */
int ft_is_alpc(uint8_t t) { return t >= 1 && t <= 12; }
uint8_t ft_from_order(size_t order) { return (order >= 1 && order <= 12) ? (uint8_t)order : 8; }

void frame_free(o_frame *f) {
    for (size_t c = 0; c < f->n_channels; c++) flo_buf_free(&f->channels[c].residuals);
    free(f->channels);
    f->channels = NULL;
    f->n_channels = 0;
}
void file_free(o_file *f) {
    for (size_t i = 0; i < f->n_frames; i++) frame_free(&f->frames[i]);
    free(f->frames);
    free(f->toc);
    flo_buf_free(&f->metadata);
    memset(f, 0, sizeof *f);
}

/* types.rs:243-267 */
size_t frame_byte_size(const o_frame *f) {
    size_t size = 6;
    for (size_t c = 0; c < f->n_channels; c++) {
        const o_channel *ch = &f->channels[c];
        size += 4;
        if (f->frame_type == FT_TRANSFORM) {
            size += ch->residuals.len;
        } else if (ft_is_alpc(f->frame_type)) {
            size += 1 + ch->n_coeffs * 4 + 1 + 1;
            if (ch->residual_encoding == RE_RICE) size += 1;
            size += ch->residuals.len;
        } else if (f->frame_type == FT_RAW) {
            size += ch->residuals.len;
        }
    }
    return size;
}
