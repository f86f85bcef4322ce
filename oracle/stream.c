/* stream.c — oracle restatement of libflo/src/streaming/encoder.rs (TEST INFRASTRUCTURE).
 *
 * StreamingEncoder buffers pushed samples, encodes every complete one-second frame through the lossless Encoder into a
 * temporary one-frame .flo file, re-reads that file and re-serialises its frame (encoder.rs:191-257), and can build a
 * complete file from the frames that have not been taken out yet (finalize, encoder.rs:113-185). Reproduced as it is,
 * including the channel layout of serialize_channel (encoder.rs:243-257), which differs from the container writer's
 * (no coefficient count, shift or encoding byte: [rice_parameter][coeffs ...][residuals]).
 */
#include "internal.h"
#include <math.h>

typedef struct {
    uint32_t index, timestamp_ms, samples;
    flo_buf data;
} s_frame;

struct flo_o_stream {
    uint32_t sample_rate;
    uint8_t channels, bit_depth, level;
    float *buf;
    size_t buf_len, buf_cap;
    size_t samples_per_frame;
    s_frame *pending;
    size_t n_pending, cap_pending;
    uint64_t total_samples;
    uint32_t frame_index;
};

flo_o_stream *flo_o_stream_new(uint32_t sample_rate, uint8_t channels, uint8_t bit_depth, uint8_t level) { /* :33-56 */
    flo_o_stream *s = (flo_o_stream *)calloc(1, sizeof *s);
    s->sample_rate = sample_rate;
    s->channels = channels;
    s->bit_depth = bit_depth;
    s->level = level > 9 ? 9 : level;
    s->samples_per_frame = sample_rate;
    return s;
}

void flo_o_stream_free(flo_o_stream *s) {
    if (!s) return;
    for (size_t i = 0; i < s->n_pending; i++) flo_buf_free(&s->pending[i].data);
    free(s->pending);
    free(s->buf);
    free(s);
}

/* encoder.rs:215-257: encode through a temporary file, re-read, re-serialise the first frame */
static int encode_frame_data(const flo_o_stream *s, const float *samples, size_t n, flo_buf *out) {
    uint8_t *tmp = NULL;
    size_t tmp_len = 0;
    if (flo_o_encode_lossless(samples, n, s->sample_rate, s->channels, s->bit_depth, s->level, NULL, 0, &tmp, &tmp_len) != 0) return -1;
    o_file f;
    if (reader_read(tmp, tmp_len, &f) != 0) {
        free(tmp);
        return -1;
    }
    free(tmp);
    if (f.n_frames == 0) {
        file_free(&f);
        set_error("No frames encoded");
        return -1;
    }
    const o_frame *fr = &f.frames[0];
    buf_init(out);
    buf_push(out, fr->frame_type);
    buf_u32le(out, fr->frame_samples);
    buf_push(out, fr->flags);
    for (size_t c = 0; c < fr->n_channels; c++) {
        const o_channel *ch = &fr->channels[c];
        flo_buf cd;
        buf_init(&cd);
        if (fr->frame_type == FT_SILENCE) {
            /* empty */
        } else if (fr->frame_type == FT_RAW || fr->frame_type == FT_TRANSFORM) {
            buf_extend(&cd, ch->residuals.data, ch->residuals.len);
        } else { /* every other type value, ALPC and reserved alike (match arm `_`) */
            buf_push(&cd, ch->rice_parameter);
            for (size_t k = 0; k < ch->n_coeffs; k++) buf_u32le(&cd, (uint32_t)ch->coeffs[k]);
            buf_extend(&cd, ch->residuals.data, ch->residuals.len);
        }
        buf_u32le(out, (uint32_t)cd.len);
        buf_extend(out, cd.data, cd.len);
        flo_buf_free(&cd);
    }
    file_free(&f);
    return 0;
}

static void push_pending(flo_o_stream *s, s_frame fr) {
    if (s->n_pending == s->cap_pending) {
        s->cap_pending = s->cap_pending ? 2 * s->cap_pending : 8;
        s->pending = (s_frame *)realloc(s->pending, s->cap_pending * sizeof(s_frame));
    }
    s->pending[s->n_pending++] = fr;
}

/* encoder.rs:71-75 + :191-213 */
int flo_o_stream_push(flo_o_stream *s, const float *samples, size_t n) {
    if (s->buf_len + n > s->buf_cap) {
        s->buf_cap = (s->buf_len + n) * 2 + 16;
        s->buf = (float *)realloc(s->buf, s->buf_cap * sizeof(float));
    }
    memcpy(s->buf + s->buf_len, samples, n * sizeof(float));
    s->buf_len += n;
    const size_t frame_samples = s->samples_per_frame * s->channels;
    size_t off = 0;
    while (frame_samples && s->buf_len - off >= frame_samples) {
        s_frame fr;
        fr.index = s->frame_index;
        fr.timestamp_ms = (uint32_t)((double)s->total_samples / (double)s->sample_rate * 1000.0);
        fr.samples = (uint32_t)s->samples_per_frame;
        if (encode_frame_data(s, s->buf + off, frame_samples, &fr.data) != 0) return -1;
        push_pending(s, fr);
        s->total_samples += s->samples_per_frame;
        s->frame_index++;
        off += frame_samples;
    }
    memmove(s->buf, s->buf + off, (s->buf_len - off) * sizeof(float));
    s->buf_len -= off;
    return 0;
}

size_t flo_o_stream_pending_samples(const flo_o_stream *s) { return s->channels ? s->buf_len / s->channels : 0; } /* :59-61 */
size_t flo_o_stream_pending_frames(const flo_o_stream *s) { return s->n_pending; }                               /* :64-66 */

/* encoder.rs:78-85: returns 1 and fills the outputs (data is malloc'ed, free with flo_o_free), or 0 */
int flo_o_stream_next_frame(flo_o_stream *s, uint32_t *index, uint32_t *timestamp_ms, uint32_t *samples, uint8_t **data, size_t *len) {
    if (s->n_pending == 0) return 0;
    s_frame fr = s->pending[0];
    memmove(s->pending, s->pending + 1, (s->n_pending - 1) * sizeof(s_frame));
    s->n_pending--;
    *index = fr.index;
    *timestamp_ms = fr.timestamp_ms;
    *samples = fr.samples;
    *data = fr.data.data;
    *len = fr.data.len;
    return 1;
}

/* encoder.rs:88-110: 1 = a frame was produced (returned, not queued), 0 = nothing buffered, -1 error */
static int flush_into(flo_o_stream *s, s_frame *out) {
    if (s->buf_len == 0) return 0;
    const size_t spc = s->buf_len / s->channels;
    out->index = s->frame_index;
    out->timestamp_ms = (uint32_t)((double)s->total_samples / (double)s->sample_rate * 1000.0);
    out->samples = (uint32_t)spc;
    if (encode_frame_data(s, s->buf, s->buf_len, &out->data) != 0) return -1;
    s->total_samples += spc;
    s->frame_index++;
    s->buf_len = 0;
    return 1;
}
int flo_o_stream_flush(flo_o_stream *s, uint32_t *index, uint32_t *timestamp_ms, uint32_t *samples, uint8_t **data, size_t *len) {
    s_frame fr;
    int rc = flush_into(s, &fr);
    if (rc != 1) return rc;
    *index = fr.index;
    *timestamp_ms = fr.timestamp_ms;
    *samples = fr.samples;
    *data = fr.data.data;
    *len = fr.data.len;
    return 1;
}

/* encoder.rs:113-185 */
int flo_o_stream_finalize(flo_o_stream *s, const uint8_t *meta, size_t meta_len, uint8_t **out, size_t *out_len) {
    s_frame fr;
    int rc = flush_into(s, &fr);
    if (rc < 0) return -1;
    if (rc == 1) push_pending(s, fr);
    flo_buf toc, data, o;
    buf_init(&toc);
    buf_init(&data);
    buf_init(&o);
    buf_u32le(&toc, (uint32_t)s->n_pending);
    uint64_t off = 0, total = 0;
    for (size_t i = 0; i < s->n_pending; i++) {
        const s_frame *p = &s->pending[i];
        buf_u32le(&toc, p->index);
        buf_u64le(&toc, off);
        buf_u32le(&toc, (uint32_t)p->data.len);
        buf_u32le(&toc, p->timestamp_ms);
        off += p->data.len;
        buf_extend(&data, p->data.data, p->data.len);
        total += p->samples;
    }
    const uint32_t crc = flo_o_crc32(data.data, data.len);
    buf_extend(&o, "FLO!", 4);
    buf_push(&o, 1);
    buf_push(&o, 2);
    buf_u16le(&o, 0);
    buf_u32le(&o, s->sample_rate);
    buf_push(&o, s->channels);
    buf_push(&o, s->bit_depth);
    buf_u64le(&o, total);
    buf_push(&o, s->level);
    buf_push(&o, 0);
    buf_push(&o, 0);
    buf_push(&o, 0);
    buf_u32le(&o, crc);
    buf_u64le(&o, 66);
    buf_u64le(&o, toc.len);
    buf_u64le(&o, data.len);
    buf_u64le(&o, 0);
    buf_u64le(&o, meta_len);
    buf_extend(&o, toc.data, toc.len);
    buf_extend(&o, data.data, data.len);
    buf_extend(&o, meta, meta_len);
    flo_buf_free(&toc);
    flo_buf_free(&data);
    for (size_t i = 0; i < s->n_pending; i++) flo_buf_free(&s->pending[i].data);
    s->n_pending = 0;
    *out = o.data;
    *out_len = o.len;
    return 0;
}
