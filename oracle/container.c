/* container.c — oracle restatement of libflo/src/writer.rs and reader.rs (TEST INFRASTRUCTURE). */
#include "internal.h"

/* writer.rs:256-301 */
static void write_channel_data(flo_buf *b, const o_channel *ch, uint8_t frame_type) {
    if (frame_type == FT_SILENCE) {
        /* nothing */
    } else if (frame_type == FT_RAW || frame_type == FT_TRANSFORM) {
        buf_extend(b, ch->residuals.data, ch->residuals.len);
    } else if (ft_is_alpc(frame_type)) {
        buf_push(b, (uint8_t)ch->n_coeffs);
        for (size_t i = 0; i < ch->n_coeffs; i++) buf_u32le(b, (uint32_t)ch->coeffs[i]);
        buf_push(b, ch->shift_bits);
        buf_push(b, ch->residual_encoding);
        if (ch->residual_encoding == RE_RICE) buf_push(b, ch->rice_parameter);
        buf_extend(b, ch->residuals.data, ch->residuals.len);
    }
    /* reserved: nothing */
}

/* writer.rs:236-254 */
static void write_frame(flo_buf *b, const o_frame *f) {
    buf_push(b, f->frame_type);
    buf_u32le(b, f->frame_samples);
    buf_push(b, f->flags);
    for (size_t c = 0; c < f->n_channels; c++) {
        flo_buf chb;
        buf_init(&chb);
        write_channel_data(&chb, &f->channels[c], f->frame_type);
        buf_u32le(b, (uint32_t)chb.len);
        buf_extend(b, chb.data, chb.len);
        flo_buf_free(&chb);
    }
}

/* writer.rs:39-100 (+ :132-191 header, :193-224 toc, :226-234 data) */
void writer_write_ex(uint32_t sample_rate, uint8_t channels, uint8_t bit_depth, uint8_t level, int lossy,
                     uint8_t lossy_quality, const o_frame *frames, size_t n_frames, const uint8_t *meta,
                     size_t meta_len, flo_buf *out) {
    uint64_t toc_size = 4 + (uint64_t)n_frames * 20;
    flo_buf data;
    buf_init(&data);
    for (size_t i = 0; i < n_frames; i++) write_frame(&data, &frames[i]);
    uint64_t data_size = data.len;
    uint32_t crc = flo_o_crc32(data.data, data.len);

    flo_buf toc;
    buf_init(&toc);
    buf_u32le(&toc, (uint32_t)n_frames);
    uint64_t byte_offset = 0, cumulative = 0;
    for (size_t i = 0; i < n_frames; i++) {
        uint32_t fsz = (uint32_t)frame_byte_size(&frames[i]);
        buf_u32le(&toc, (uint32_t)i);
        buf_u64le(&toc, byte_offset);
        buf_u32le(&toc, fsz);
        buf_u32le(&toc, (uint32_t)(cumulative * 1000 / (uint64_t)sample_rate));
        byte_offset += fsz;
        cumulative += frames[i].frame_samples;
    }

    uint16_t flags = 0;
    if (lossy) {
        flags |= 0x01;
        flags |= (uint16_t)((uint16_t)lossy_quality << 8);
    }
    uint64_t total_samples = 0;
    for (size_t i = 0; i < n_frames; i++) total_samples += frames[i].frame_samples;

    static const uint8_t magic[4] = {0x46, 0x4c, 0x4f, 0x21};
    buf_extend(out, magic, 4);
    buf_push(out, FLO_VERSION_MAJOR);
    buf_push(out, FLO_VERSION_MINOR);
    buf_u16le(out, flags);
    buf_u32le(out, sample_rate);
    buf_push(out, channels);
    buf_push(out, bit_depth);
    buf_u64le(out, total_samples);
    buf_push(out, level);
    buf_push(out, 0);
    buf_push(out, 0);
    buf_push(out, 0);
    buf_u32le(out, crc);
    buf_u64le(out, FLO_HEADER_SIZE);
    buf_u64le(out, toc_size);
    buf_u64le(out, data_size);
    buf_u64le(out, 0); /* extra */
    buf_u64le(out, meta_len);

    buf_extend(out, toc.data, toc.len);
    buf_extend(out, data.data, data.len);
    buf_extend(out, meta, meta_len);
    flo_buf_free(&toc);
    flo_buf_free(&data);
}

/* ------------------------------------------------------------------ reader.rs:267-320 Cursor */
typedef struct {
    const uint8_t *data;
    size_t len, pos;
    int err;
} cursor;
static int cur_need(cursor *c, size_t n) {
    if (c->pos + n > c->len) {
        c->err = 1;
        return 0;
    }
    return 1;
}
static uint8_t cur_u8(cursor *c) {
    if (c->pos >= c->len) {
        c->err = 1;
        return 0;
    }
    return c->data[c->pos++];
}
static uint16_t cur_u16(cursor *c) {
    if (!cur_need(c, 2)) return 0;
    uint16_t v = (uint16_t)(c->data[c->pos] | (c->data[c->pos + 1] << 8));
    c->pos += 2;
    return v;
}
static uint32_t cur_u32(cursor *c) {
    if (!cur_need(c, 4)) return 0;
    const uint8_t *p = c->data + c->pos;
    c->pos += 4;
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint64_t cur_u64(cursor *c) {
    uint64_t lo = cur_u32(c);
    uint64_t hi = cur_u32(c);
    return lo | (hi << 32);
}
static void cur_skip(cursor *c, size_t n) { c->pos = (c->pos + n < c->len) ? c->pos + n : c->len; }

/* reader.rs:168-256 */
static int read_channel_data(cursor *c, uint8_t frame_type, size_t frame_samples, size_t channel_end, o_channel *ch) {
    memset(ch, 0, sizeof *ch);
    buf_init(&ch->residuals);
    if (frame_samples > 2000000) {
        set_error("Invalid frame: too many samples");
        return -1;
    }
    if (frame_type == FT_SILENCE) {
        ch->residual_encoding = RE_RICE;
        return 0;
    }
    if (frame_type == FT_RAW) {
        size_t need = frame_samples * 2;
        size_t avail = channel_end > c->pos ? channel_end - c->pos : 0;
        size_t n = need < avail ? need : avail;
        if (!cur_need(c, n)) return -1;
        buf_extend(&ch->residuals, c->data + c->pos, n);
        c->pos += n;
        ch->residual_encoding = RE_RAW;
        return 0;
    }
    if (frame_type == FT_TRANSFORM) {
        size_t rem = channel_end > c->pos ? channel_end - c->pos : 0;
        if (rem) {
            if (!cur_need(c, rem)) return -1;
            buf_extend(&ch->residuals, c->data + c->pos, rem);
            c->pos += rem;
        }
        ch->residual_encoding = RE_RAW;
        return 0;
    }
    if (ft_is_alpc(frame_type)) {
        size_t order = cur_u8(c);
        if (c->err) return -1;
        if (order > 12) {
            set_error("Invalid LPC order");
            return -1;
        }
        for (size_t i = 0; i < order; i++) {
            if (c->pos + 4 > channel_end) break;
            ch->coeffs[ch->n_coeffs++] = (int32_t)cur_u32(c);
        }
        ch->shift_bits = cur_u8(c);
        uint8_t enc = cur_u8(c);
        ch->residual_encoding = enc == 0 ? RE_RICE : (enc == 1 ? RE_GOLOMB : RE_RAW);
        ch->rice_parameter = ch->residual_encoding == RE_RICE ? cur_u8(c) : 0;
        if (c->err) return -1;
        size_t rem = channel_end > c->pos ? channel_end - c->pos : 0;
        if (rem) {
            if (!cur_need(c, rem)) return -1;
            buf_extend(&ch->residuals, c->data + c->pos, rem);
            c->pos += rem;
        }
        return 0;
    }
    ch->residual_encoding = RE_RICE; /* reserved → new_silence */
    return 0;
}

/* reader.rs:130-166 */
static int read_frame(cursor *c, uint8_t channels, size_t frame_size, o_frame *f) {
    size_t frame_start = c->pos, frame_end = frame_start + frame_size;
    memset(f, 0, sizeof *f);
    f->frame_type = cur_u8(c);
    f->frame_samples = cur_u32(c);
    f->flags = cur_u8(c);
    if (c->err) return -1;
    size_t nch = f->frame_type == FT_TRANSFORM ? 1 : channels;
    f->channels = (o_channel *)calloc(nch ? nch : 1, sizeof(o_channel));
    for (size_t i = 0; i < nch; i++) {
        size_t ch_size = cur_u32(c);
        if (c->err) return -1;
        size_t ch_end = c->pos + ch_size;
        if (read_channel_data(c, f->frame_type, f->frame_samples, ch_end, &f->channels[i]) != 0) {
            f->n_channels = i + 1;
            return -1;
        }
        f->n_channels = i + 1;
        c->pos = ch_end;
    }
    c->pos = frame_end;
    return 0;
}

/* reader.rs:16-128 */
int reader_read(const uint8_t *data, size_t len, o_file *out) {
    memset(out, 0, sizeof *out);
    buf_init(&out->metadata);
    cursor c = {data, len, 0, 0};
    if (len < 4 || memcmp(data, "FLO!", 4) != 0) {
        set_error(len < 4 ? "Unexpected end of file" : "Invalid flo file: bad magic");
        return -1;
    }
    c.pos = 4;
    flo_o_info *h = &out->hdr;
    h->version_major = cur_u8(&c);
    h->version_minor = cur_u8(&c);
    h->flags = cur_u16(&c);
    h->sample_rate = cur_u32(&c);
    h->channels = cur_u8(&c);
    h->bit_depth = cur_u8(&c);
    h->total_samples = cur_u64(&c);
    h->compression_level = cur_u8(&c);
    cur_skip(&c, 3);
    h->data_crc32 = cur_u32(&c);
    h->header_size = cur_u64(&c);
    h->toc_size = cur_u64(&c);
    h->data_size = cur_u64(&c);
    h->extra_size = cur_u64(&c);
    h->meta_size = cur_u64(&c);
    if (c.err) {
        set_error("Unexpected end of file");
        return -1;
    }
    /* toc (reader.rs:76-99) */
    if (h->toc_size >= 4) {
        size_t n = cur_u32(&c);
        if (c.err) {
            set_error("Unexpected end of file");
            return -1;
        }
        if (n > 100000) {
            set_error("Invalid TOC: too many entries");
            return -1;
        }
        out->toc = (o_toc_entry *)calloc(n ? n : 1, sizeof(o_toc_entry));
        for (size_t i = 0; i < n; i++) {
            out->toc[i].frame_index = cur_u32(&c);
            out->toc[i].byte_offset = cur_u64(&c);
            out->toc[i].frame_size = cur_u32(&c);
            out->toc[i].timestamp_ms = cur_u32(&c);
            if (c.err) {
                set_error("Unexpected end of file");
                file_free(out);
                return -1;
            }
        }
        out->n_toc = n;
    }
    /* data chunk (reader.rs:101-128) */
    size_t data_start = c.pos, data_end = c.pos + (size_t)h->data_size;
    out->frames = (o_frame *)calloc(out->n_toc ? out->n_toc : 1, sizeof(o_frame));
    for (size_t i = 0; i < out->n_toc; i++) {
        size_t fs = data_start + (size_t)out->toc[i].byte_offset;
        if (fs >= data_end) break;
        c.pos = fs;
        if (read_frame(&c, h->channels, out->toc[i].frame_size, &out->frames[out->n_frames]) != 0 || c.err) {
            out->n_frames++;
            if (c.err) set_error("Unexpected end of file");
            file_free(out);
            return -1;
        }
        out->n_frames++;
    }
    if (data_start <= len) {
        size_t de = data_end <= len ? data_end : len;
        h->crc_computed = flo_o_crc32(data + data_start, de - data_start);
    }
    h->num_frames = (uint32_t)out->n_frames;
    c.pos = data_end;
    cur_skip(&c, (size_t)h->extra_size);
    if (c.pos + (size_t)h->meta_size > len) {
        set_error("Unexpected end of file");
        file_free(out);
        return -1;
    }
    buf_extend(&out->metadata, data + c.pos, (size_t)h->meta_size);
    return 0;
}

int flo_o_info_read(const uint8_t *flo, size_t len, flo_o_info *info) {
    o_file f;
    if (reader_read(flo, len, &f) != 0) return -1;
    *info = f.hdr;
    file_free(&f);
    return 0;
}
