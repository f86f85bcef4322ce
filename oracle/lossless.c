/* lossless.c — oracle restatement of libflo/src/lossless/{lpc,encoder,decoder}.rs (TEST INFRASTRUCTURE). */
#include "internal.h"
#include <math.h>

/* ------------------------------------------------------------------ lpc.rs:213-221 */
void flo_o_autocorr_int(const int32_t *s, size_t n, size_t order, int64_t *out) {
    for (size_t lag = 0; lag <= order; lag++) {
        int64_t acc = 0;
        for (size_t i = lag; i < n; i++) acc += (int64_t)s[i] * (int64_t)s[i - lag];
        out[lag] = acc;
    }
}

static uint8_t f64_as_u8(double v) { /* Rust `as u8`: saturating, NaN -> 0 */
    if (v != v) return 0;
    if (v <= 0.0) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)v;
}
static int32_t f64_as_i32(double v) {
    if (v != v) return 0;
    if (v <= -2147483648.0) return INT32_MIN;
    if (v >= 2147483647.0) return INT32_MAX;
    return (int32_t)v;
}

/* lpc.rs:225-276 */
int flo_o_levinson_durbin_int(const int64_t *autocorr, size_t n_autocorr, size_t order, int32_t *coeffs_out,
                              uint8_t *shift_out) {
    if (n_autocorr == 0 || autocorr[0] == 0) return 0;
    double coeffs[32] = {0};
    double new_coeffs[32];
    if (order > 32) return 0;
    double error = (double)autocorr[0];
    for (size_t i = 0; i < order; i++) {
        double lambda = (i + 1 < n_autocorr) ? (double)autocorr[i + 1] : 0.0;
        for (size_t j = 0; j < i; j++) {
            double r = (i - j < n_autocorr) ? (double)autocorr[i - j] : 0.0;
            lambda -= coeffs[j] * r;
        }
        if (fabs(error) < 1e-10) return 0;
        double gamma = lambda / error;
        if (fabs(gamma) >= 1.0) return 0;
        new_coeffs[i] = gamma;
        for (size_t j = 0; j < i; j++) new_coeffs[j] = coeffs[j] - gamma * coeffs[i - 1 - j];
        memcpy(coeffs, new_coeffs, (i + 1) * sizeof(double));
        error *= 1.0 - gamma * gamma;
    }
    double max_coeff = 0.0;
    for (size_t i = 0; i < order; i++) max_coeff = fmax(max_coeff, fabs(coeffs[i]));
    if (max_coeff == 0.0 || !isfinite(max_coeff)) return 0;
    uint8_t shift = f64_as_u8(floor(log2((double)(1 << 30) / max_coeff)));
    if (shift > 15) shift = 15;
    double scale = (double)((int64_t)1 << shift);
    for (size_t i = 0; i < order; i++) coeffs_out[i] = f64_as_i32(round(coeffs[i] * scale));
    *shift_out = shift;
    return 1;
}

/* lpc.rs:279-298 */
void flo_o_calc_residuals_int(const int32_t *s, size_t n, const int32_t *coeffs, size_t n_coeffs, uint8_t shift,
                              size_t order, int32_t *out) {
    size_t warm = order < n ? order : n;
    for (size_t i = 0; i < warm; i++) out[i] = s[i];
    for (size_t i = order; i < n; i++) {
        int64_t prediction = 0;
        for (size_t j = 0; j < n_coeffs; j++) prediction += (int64_t)coeffs[j] * (int64_t)s[i - j - 1];
        prediction >>= shift; /* arithmetic */
        out[i] = (int32_t)((uint32_t)s[i] - (uint32_t)(int32_t)prediction);
    }
}

static int32_t wrap32(int64_t v) { return (int32_t)(uint32_t)(uint64_t)v; }

/* lpc.rs:301-359 */
void flo_o_fixed_predictor_residuals(const int32_t *s, size_t n, size_t order, int32_t *out) {
    if (n == 0) return;
    if (order == 0 || order > 4) {
        memcpy(out, s, n * sizeof(int32_t));
        return;
    }
    out[0] = s[0];
    if (n > 1 && order >= 1) out[1] = wrap32((int64_t)s[1] - s[0]);
    if (order == 1) {
        for (size_t i = 1; i < n; i++) out[i] = wrap32((int64_t)s[i] - s[i - 1]);
        return;
    }
    if (n > 2 && order >= 3) out[2] = wrap32((int64_t)s[2] - 2 * (int64_t)s[1] + s[0]);
    if (order == 2) {
        for (size_t i = 2; i < n; i++) out[i] = wrap32((int64_t)s[i] - 2 * (int64_t)s[i - 1] + s[i - 2]);
        return;
    }
    if (n > 3 && order >= 4) out[3] = wrap32((int64_t)s[3] - 3 * (int64_t)s[2] + 3 * (int64_t)s[1] - s[0]);
    if (order == 3) {
        for (size_t i = 3; i < n; i++)
            out[i] = wrap32((int64_t)s[i] - 3 * (int64_t)s[i - 1] + 3 * (int64_t)s[i - 2] - s[i - 3]);
        return;
    }
    for (size_t i = 4; i < n; i++)
        out[i] = wrap32((int64_t)s[i] - 4 * (int64_t)s[i - 1] + 6 * (int64_t)s[i - 2] - 4 * (int64_t)s[i - 3] +
                        s[i - 4]);
}

/* ------------------------------------------------------------------ lossless/encoder.rs */

/* encoder.rs:289-302 */
static size_t lpc_order_from_level(uint8_t level) {
    switch (level) {
    case 0: return 0;
    case 1: return 2;
    case 2: return 4;
    case 3: return 4;
    case 4: return 6;
    case 5: return 8;
    case 6: return 8;
    case 7: return 10;
    case 8: return 12;
    default: return 12;
    }
}

static void channel_set(o_channel *dst, o_channel *src) { /* move */
    flo_buf_free(&dst->residuals);
    *dst = *src;
    buf_init(&src->residuals);
}

/* encoder.rs:173-217 encode_channel_int; returns order_used */
static size_t encode_channel_int(const int32_t *s, size_t n, size_t max_order, uint8_t level, o_channel *best) {
    memset(best, 0, sizeof *best);
    buf_init(&best->residuals);
    if (n == 0) {
        best->residual_encoding = RE_RICE; /* new_silence */
        return 0;
    }
    size_t best_size = (size_t)-1, best_order = 0;
    int32_t *res = (int32_t *)malloc(n * sizeof(int32_t));

    /* Strategy 1: raw PCM (encoder.rs:220-226) */
    {
        o_channel c;
        memset(&c, 0, sizeof c);
        buf_init(&c.residuals);
        buf_reserve(&c.residuals, n * 2);
        for (size_t i = 0; i < n; i++) buf_u16le(&c.residuals, (uint16_t)(int16_t)s[i]); /* `as i16` wraps */
        c.residual_encoding = RE_RAW;
        if (c.residuals.len < best_size) {
            best_size = c.residuals.len;
            channel_set(best, &c);
            best_order = 0;
        }
        flo_buf_free(&c.residuals);
    }
    /* Strategy 2: fixed predictors (encoder.rs:193-201, 229-251) */
    size_t fmax_order = max_order < 4 ? max_order : 4;
    for (size_t order = 0; order <= fmax_order; order++) {
        flo_o_fixed_predictor_residuals(s, n, order, res);
        uint8_t k = flo_o_estimate_rice_parameter_i32(res, n);
        o_channel c;
        memset(&c, 0, sizeof c);
        buf_init(&c.residuals);
        flo_o_rice_encode_i32(res, n, k, &c.residuals.data, &c.residuals.len);
        c.residuals.cap = c.residuals.len;
        c.shift_bits = (uint8_t)(128 + order);
        c.residual_encoding = RE_RICE;
        c.rice_parameter = k;
        if (c.residuals.len < best_size) {
            best_size = c.residuals.len;
            channel_set(best, &c);
            best_order = order;
        }
        flo_buf_free(&c.residuals);
    }
    /* Strategy 3: LPC (encoder.rs:204-214, 254-287) */
    if (level >= 3 && max_order > 4) {
        for (size_t order = 5; order <= max_order; order++) {
            if (n <= order) continue;
            int64_t ac[16];
            flo_o_autocorr_int(s, n, order, ac);
            o_channel c;
            memset(&c, 0, sizeof c);
            buf_init(&c.residuals);
            uint8_t shift;
            if (!flo_o_levinson_durbin_int(ac, order + 1, order, c.coeffs, &shift)) continue;
            c.n_coeffs = order;
            flo_o_calc_residuals_int(s, n, c.coeffs, order, shift, order, res);
            int64_t max_res = 0;
            for (size_t i = 0; i < n; i++) {
                int64_t a = res[i] < 0 ? -(int64_t)res[i] : res[i];
                if (a > max_res) max_res = a;
            }
            if (max_res > 1000000) continue;
            uint8_t k = flo_o_estimate_rice_parameter_i32(res, n);
            flo_o_rice_encode_i32(res, n, k, &c.residuals.data, &c.residuals.len);
            c.residuals.cap = c.residuals.len;
            c.shift_bits = shift;
            c.residual_encoding = RE_RICE;
            c.rice_parameter = k;
            if (c.residuals.len < best_size) {
                best_size = c.residuals.len;
                channel_set(best, &c);
                best_order = order;
            }
            flo_buf_free(&c.residuals);
        }
    }
    free(res);
    return best_order;
}

/* encoder.rs:66-128 encode_frame */
static void encode_frame(const float *samples, size_t len, uint8_t channels, uint8_t level, o_frame *f) {
    size_t ch = channels;
    size_t num_samples = len / ch;
    memset(f, 0, sizeof *f);

    int silent = 1;
    for (size_t i = 0; i < len; i++)
        if (!(fabsf(samples[i]) < 1e-7f)) {
            silent = 0;
            break;
        }
    if (silent) {
        f->frame_type = FT_SILENCE;
        f->frame_samples = (uint32_t)num_samples;
        f->channels = (o_channel *)calloc(ch ? ch : 1, sizeof(o_channel));
        f->n_channels = ch;
        for (size_t c = 0; c < ch; c++) f->channels[c].residual_encoding = RE_RICE;
        return;
    }

    /* f32 -> i32, de-interleave (encoder.rs:79-91) */
    int32_t **cd = (int32_t **)calloc(ch, sizeof(int32_t *));
    size_t *cn = (size_t *)calloc(ch, sizeof(size_t));
    for (size_t c = 0; c < ch; c++) {
        cn[c] = len > c ? (len - c + ch - 1) / ch : 0;
        cd[c] = (int32_t *)malloc((cn[c] ? cn[c] : 1) * sizeof(int32_t));
        for (size_t i = 0; i < cn[c]; i++) cd[c][i] = flo_o_f32_to_i32(samples[i * ch + c]);
    }

    /* mid/side (encoder.rs:94-99, 131-170) */
    int use_ms = 0;
    if (channels == 2) {
        size_t m = cn[0] < cn[1] ? cn[0] : cn[1];
        int64_t var_l = 0, var_r = 0, var_side = 0;
        for (size_t i = 0; i < m; i++) {
            int64_t l = cd[0][i], r = cd[1][i];
            var_l += l * l;
            var_r += r * r;
            int64_t side = (int32_t)(l - r);
            var_side += side * side;
        }
        use_ms = var_side < (var_l + var_r) / 2;
        if (use_ms) {
            for (size_t i = 0; i < m; i++) {
                int32_t l = cd[0][i], r = cd[1][i];
                cd[0][i] = l + r;
                cd[1][i] = l - r;
            }
            cn[0] = cn[1] = m;
        }
    }

    size_t lpc_order = lpc_order_from_level(level);
    f->channels = (o_channel *)calloc(ch, sizeof(o_channel));
    f->n_channels = ch;
    int all_raw = 1;
    for (size_t c = 0; c < ch; c++) {
        size_t used = encode_channel_int(cd[c], cn[c], lpc_order, level, &f->channels[c]);
        if (used > 0) all_raw = 0;
    }
    f->frame_type = all_raw ? FT_RAW : ft_from_order(lpc_order);
    f->frame_samples = (uint32_t)num_samples;
    f->flags = use_ms ? 0x01 : 0;
    for (size_t c = 0; c < ch; c++) free(cd[c]);
    free(cd);
    free(cn);
}

/* encoder.rs:47-64 encode_frames */
void lossless_encode_frames(const float *samples, size_t n, uint32_t sample_rate, uint8_t channels, uint8_t level,
                            o_frame **frames, size_t *n_frames) {
    size_t spf = sample_rate, ch = channels;
    size_t total = n / ch;
    size_t nf = spf ? (total + spf - 1) / spf : 0;
    *frames = (o_frame *)calloc(nf ? nf : 1, sizeof(o_frame));
    *n_frames = nf;
    for (size_t fi = 0; fi < nf; fi++) {
        size_t start = fi * spf * ch;
        size_t end = (fi + 1) * spf * ch;
        if (end > n) end = n;
        encode_frame(samples + start, end - start, channels, level, &(*frames)[fi]);
    }
}

int flo_o_encode_lossless(const float *pcm, size_t n, uint32_t sample_rate, uint8_t channels, uint8_t bit_depth,
                          uint8_t level, const uint8_t *meta, size_t meta_len, uint8_t **out, size_t *out_len) {
    if (channels == 0 || sample_rate == 0) {
        set_error("invalid arguments");
        return -1;
    }
    if (level > 9) level = 9; /* with_compression: level.min(9) */
    o_frame *frames;
    size_t nf;
    lossless_encode_frames(pcm, n, sample_rate, channels, level, &frames, &nf);
    flo_buf b;
    buf_init(&b);
    writer_write_ex(sample_rate, channels, bit_depth, level, 0, 0, frames, nf, meta, meta_len, &b);
    for (size_t i = 0; i < nf; i++) frame_free(&frames[i]);
    free(frames);
    *out = b.data;
    *out_len = b.len;
    return 0;
}

/* ------------------------------------------------------------------ lossless/decoder.rs */

/* decoder.rs:187-273 */
static void reconstruct_fixed(size_t order, const int32_t *r, size_t n, int32_t *s) {
    /* residuals.len() == target_len == n here (rice decode yields exactly frame_samples values) */
    if (n == 0) return;
    if (order == 0 || order > 4) {
        memcpy(s, r, n * sizeof(int32_t));
        return;
    }
    s[0] = r[0];
    if (order == 1) {
        for (size_t i = 1; i < n; i++) s[i] = wrap32((int64_t)r[i] + s[i - 1]);
        return;
    }
    if (n > 1) s[1] = wrap32((int64_t)r[1] + s[0]);
    if (order == 2) {
        for (size_t i = 2; i < n; i++) {
            int32_t pred = wrap32(2 * (int64_t)s[i - 1] - (int64_t)s[i - 2]);
            s[i] = wrap32((int64_t)r[i] + pred);
        }
        return;
    }
    if (n > 2) {
        int32_t pred = wrap32(2 * (int64_t)s[1] - (int64_t)s[0]);
        s[2] = wrap32((int64_t)r[2] + pred);
    }
    if (order == 3) {
        for (size_t i = 3; i < n; i++) {
            int32_t pred = wrap32(3 * (int64_t)s[i - 1] - 3 * (int64_t)s[i - 2] + (int64_t)s[i - 3]);
            s[i] = wrap32((int64_t)r[i] + pred);
        }
        return;
    }
    if (n > 3) {
        int32_t pred = wrap32(3 * (int64_t)s[2] - 3 * (int64_t)s[1] + (int64_t)s[0]);
        s[3] = wrap32((int64_t)r[3] + pred);
    }
    for (size_t i = 4; i < n; i++) {
        int32_t pred =
            wrap32(4 * (int64_t)s[i - 1] - 6 * (int64_t)s[i - 2] + 4 * (int64_t)s[i - 3] - (int64_t)s[i - 4]);
        s[i] = wrap32((int64_t)r[i] + pred);
    }
}

/* decoder.rs:92-148 */
static void decode_channel_int(const o_channel *ch, size_t frame_samples, int32_t *out) {
    int has_coeffs = ch->n_coeffs > 0;
    int has_res = ch->residuals.len > 0;
    uint8_t shift_bits = ch->shift_bits;
    if (!has_coeffs && has_res && shift_bits >= 128) {
        size_t fixed_order = shift_bits - 128;
        int32_t *r = (int32_t *)malloc((frame_samples ? frame_samples : 1) * sizeof(int32_t));
        flo_o_rice_decode_i32(ch->residuals.data, ch->residuals.len, ch->rice_parameter, frame_samples, r);
        reconstruct_fixed(fixed_order, r, frame_samples, out);
        free(r);
        return;
    }
    if (has_coeffs) {
        /* decoder.rs:152-184 reconstruct_lpc_int */
        int32_t *r = (int32_t *)malloc((frame_samples ? frame_samples : 1) * sizeof(int32_t));
        flo_o_rice_decode_i32(ch->residuals.data, ch->residuals.len, ch->rice_parameter, frame_samples, r);
        size_t order = ch->n_coeffs;
        size_t warm = order < frame_samples ? order : frame_samples;
        for (size_t i = 0; i < warm; i++) out[i] = r[i];
        for (size_t i = order; i < frame_samples; i++) {
            int64_t prediction = 0;
            for (size_t j = 0; j < order; j++) prediction += (int64_t)ch->coeffs[j] * (int64_t)out[i - j - 1];
            out[i] = wrap32((int64_t)(int32_t)(prediction >> shift_bits) + r[i]);
        }
        free(r);
        return;
    }
    if (has_res) {
        size_t n = 0;
        for (size_t i = 0; i + 1 < ch->residuals.len && n < frame_samples; i += 2)
            out[n++] = (int16_t)(ch->residuals.data[i] | (ch->residuals.data[i + 1] << 8));
        /* note: the reference pushes every complete pair; frames never carry more than frame_samples pairs
         * because the reader reads at most frame_samples*2 bytes for Raw frames (reader.rs:183-186). */
        while (n < frame_samples) out[n++] = 0;
        return;
    }
    for (size_t i = 0; i < frame_samples; i++) out[i] = 0;
}

/* decoder.rs:21-72 decode_file, integer domain */
int lossless_decode_file_i32(const o_file *file, int32_t **out, size_t *n_interleaved) {
    size_t channels = file->hdr.channels;
    if (channels == 0) {
        *out = NULL;
        *n_interleaved = 0;
        return 0;
    }
    size_t *len = (size_t *)calloc(channels, sizeof(size_t));
    size_t total = 0;
    for (size_t i = 0; i < file->n_frames; i++) total += file->frames[i].frame_samples;
    int32_t **all = (int32_t **)calloc(channels, sizeof(int32_t *));
    for (size_t c = 0; c < channels; c++) all[c] = (int32_t *)malloc((total ? total : 1) * sizeof(int32_t));

    for (size_t fi = 0; fi < file->n_frames; fi++) {
        const o_frame *fr = &file->frames[fi];
        size_t fs = fr->frame_samples;
        int use_ms = channels == 2 && (fr->flags & 1);
        size_t nfc = fr->n_channels;
        int32_t **tmp = (int32_t **)calloc(nfc ? nfc : 1, sizeof(int32_t *));
        for (size_t c = 0; c < nfc; c++) {
            tmp[c] = (int32_t *)malloc((fs ? fs : 1) * sizeof(int32_t));
            decode_channel_int(&fr->channels[c], fs, tmp[c]);
        }
        if (use_ms && nfc == 2) {
            for (size_t i = 0; i < fs; i++) {
                int32_t m = tmp[0][i], s = tmp[1][i];
                all[0][len[0] + i] = (m + s) / 2; /* truncating division, decoder.rs:81,86 */
                all[1][len[1] + i] = (m - s) / 2;
            }
            len[0] += fs;
            len[1] += fs;
        } else {
            for (size_t c = 0; c < nfc; c++) {
                if (c < channels) {
                    memcpy(all[c] + len[c], tmp[c], fs * sizeof(int32_t));
                    len[c] += fs;
                }
            }
        }
        for (size_t c = 0; c < nfc; c++) free(tmp[c]);
        free(tmp);
    }
    size_t max_len = 0;
    for (size_t c = 0; c < channels; c++)
        if (len[c] > max_len) max_len = len[c];
    int32_t *il = (int32_t *)malloc(((max_len * channels) > 0 ? max_len * channels : 1) * sizeof(int32_t));
    for (size_t i = 0; i < max_len; i++)
        for (size_t c = 0; c < channels; c++) il[i * channels + c] = i < len[c] ? all[c][i] : 0;
    for (size_t c = 0; c < channels; c++) free(all[c]);
    free(all);
    free(len);
    *out = il;
    *n_interleaved = max_len * channels;
    return 0;
}
