/* lossy.c — oracle restatement of libflo/src/lossy/{mdct,psychoacoustic,encoder,decoder}.rs
 * TEST INFRASTRUCTURE (see flo_oracle.h).
 *
 * Third-party arithmetic: the reference's 512-point complex FFT is rustfft 6.4.1
 * (libflo/Cargo.lock:215-216; call site mdct.rs:200,252). It is not vendored; this file uses its own
 * iterative radix-2 f32 FFT with the same sign convention (forward, e^{-2*pi*i*jk/N}, unnormalised).
 * Everything around the FFT follows the reference's f32 evaluation order exactly.
 */
#include "internal.h"
#include <math.h>

#define PI_F32 3.14159265358979323846f /* std::f32::consts::PI */

/* ------------------------------------------------------------------ FFT (stands in for rustfft) */
typedef struct { float re, im; } cpx;

static void fft_forward(cpx *z, size_t n) {
    /* bit reversal */
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            cpx t = z[i];
            z[i] = z[j];
            z[j] = t;
        }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        size_t half = len >> 1;
        for (size_t k = 0; k < half; k++) {
            double ang = -2.0 * M_PI * (double)k / (double)len;
            float wr = (float)cos(ang), wi = (float)sin(ang);
            for (size_t s = 0; s < n; s += len) {
                cpx a = z[s + k], b = z[s + k + half];
                float tr = b.re * wr - b.im * wi;
                float ti = b.re * wi + b.im * wr;
                z[s + k].re = a.re + tr;
                z[s + k].im = a.im + ti;
                z[s + k + half].re = a.re - tr;
                z[s + k + half].im = a.im - ti;
            }
        }
    }
}

/* ------------------------------------------------------------------ mdct.rs */
void flo_o_sine_window(size_t n, float *out) { /* :99-103 */
    for (size_t i = 0; i < n; i++) out[i] = sinf(PI_F32 * ((float)i + 0.5f) / (float)n);
}
void flo_o_vorbis_window(size_t n, float *out) { /* :106-113 */
    for (size_t i = 0; i < n; i++) {
        float x = sinf(PI_F32 * ((float)i + 0.5f) / (float)n);
        out[i] = sinf(PI_F32 / 2.0f * x * x);
    }
}

typedef struct {
    size_t n, n2, n4;
    float *window;
    cpx *twiddle;
} mdct_transform;

/* mdct.rs:64-96 */
static void mdct_init(mdct_transform *t, size_t n, int window_type) {
    t->n = n;
    t->n2 = n / 2;
    t->n4 = n / 4;
    t->window = (float *)malloc(n * sizeof(float));
    if (window_type == 0) flo_o_sine_window(n, t->window);
    else flo_o_vorbis_window(n, t->window); /* KBD (type 1) is never used by the encoder; not restated */
    t->twiddle = (cpx *)malloc(t->n4 * sizeof(cpx));
    for (size_t k = 0; k < t->n4; k++) {
        float theta = PI_F32 / (float)t->n2 * ((float)k + 0.125f);
        t->twiddle[k].re = cosf(theta);
        t->twiddle[k].im = sinf(theta);
    }
}
static void mdct_free(mdct_transform *t) {
    free(t->window);
    free(t->twiddle);
}

/* mdct.rs:166-226 */
static void mdct_fwd(const mdct_transform *t, const float *samples, float *output) {
    size_t n = t->n, n2 = t->n2, n4 = t->n4, n8 = n4 / 2, n3 = 3 * n4;
    float *x = (float *)malloc(n * sizeof(float));
    cpx *z = (cpx *)malloc(n4 * sizeof(cpx));
    for (size_t i = 0; i < n; i++) x[i] = samples[i] * t->window[i];
    for (size_t i = 0; i < n8; i++) {
        float re = -x[2 * i + n3] - x[n3 - 1 - 2 * i];
        float im = -x[n4 + 2 * i] + x[n4 - 1 - 2 * i];
        cpx w = t->twiddle[i];
        z[i].re = -re * w.re - im * w.im;
        z[i].im = re * w.im - im * w.re;
        float re2 = x[2 * i] - x[n2 - 1 - 2 * i];
        float im2 = -x[n2 + 2 * i] - x[n - 1 - 2 * i];
        cpx w2 = t->twiddle[n8 + i];
        z[n8 + i].re = -re2 * w2.re - im2 * w2.im;
        z[n8 + i].im = re2 * w2.im - im2 * w2.re;
    }
    fft_forward(z, n4);
    for (size_t i = 0; i < n8; i++) {
        size_t idx1 = n8 - i - 1, idx2 = n8 + i;
        cpx w1 = t->twiddle[idx1], z1 = z[idx1];
        float i1 = -z1.re * w1.im + z1.im * w1.re;
        float r0 = -z1.re * w1.re - z1.im * w1.im;
        cpx w2 = t->twiddle[idx2], z2 = z[idx2];
        float i0 = -z2.re * w2.im + z2.im * w2.re;
        float r1 = -z2.re * w2.re - z2.im * w2.im;
        output[2 * idx1] = r0;
        output[2 * idx1 + 1] = i0;
        output[2 * idx2] = r1;
        output[2 * idx2 + 1] = i1;
    }
    free(x);
    free(z);
}

/* mdct.rs:231-290 */
static void mdct_inv(const mdct_transform *t, const float *spec, float *output) {
    size_t n = t->n, n2 = t->n2, n4 = t->n4, n8 = n4 / 2;
    cpx *z = (cpx *)malloc(n4 * sizeof(cpx));
    for (size_t i = 0; i < n4; i++) {
        float even = spec[i * 2];
        float odd = -spec[n2 - 1 - i * 2];
        cpx w = t->twiddle[i];
        z[i].re = odd * w.im - even * w.re;
        z[i].im = odd * w.re + even * w.im;
    }
    fft_forward(z, n4);
    float scale = 2.0f / (float)n2;
    const float *win = t->window;
    for (size_t i = 0; i < n; i++) output[i] = 0.0f;
    for (size_t i = 0; i < n8; i++) {
        cpx w = t->twiddle[i];
        float val_re = w.re * z[i].re + w.im * z[i].im;
        float val_im = w.im * z[i].re - w.re * z[i].im;
        size_t fi = 2 * i, ri = n4 - 1 - 2 * i;
        output[ri] = -val_im * scale * win[ri];
        output[n4 + fi] = val_im * scale * win[n4 + fi];
        output[n2 + ri] = val_re * scale * win[n2 + ri];
        output[n2 + n4 + fi] = val_re * scale * win[n2 + n4 + fi];
    }
    for (size_t i = 0; i < n8; i++) {
        size_t idx = n8 + i;
        cpx w = t->twiddle[idx];
        float val_re = w.re * z[idx].re + w.im * z[idx].im;
        float val_im = w.im * z[idx].re - w.re * z[idx].im;
        size_t fi = 2 * i, ri = n4 - 1 - 2 * i;
        output[fi] = -val_re * scale * win[fi];
        output[n4 + ri] = val_re * scale * win[n4 + ri];
        output[n2 + fi] = val_im * scale * win[n2 + fi];
        output[n2 + n4 + ri] = val_im * scale * win[n2 + n4 + ri];
    }
    free(z);
}

void flo_o_mdct_forward(const float *samples, size_t n, int window_type, float *out) {
    mdct_transform t;
    mdct_init(&t, n, window_type);
    mdct_fwd(&t, samples, out);
    mdct_free(&t);
}
void flo_o_mdct_inverse(const float *spec, size_t n, int window_type, float *out) {
    mdct_transform t;
    mdct_init(&t, n, window_type);
    mdct_inv(&t, spec, out);
    mdct_free(&t);
}
/* X[k] = sum x[n] w[n] cos(pi/N2 (n + 0.5 + N2/2)(k + 0.5)), N2 = n/2  (doc comment mdct.rs:336) */
void flo_o_mdct_forward_direct_f64(const float *samples, size_t n, int window_type, double *out) {
    float *w = (float *)malloc(n * sizeof(float));
    if (window_type == 0) flo_o_sine_window(n, w);
    else flo_o_vorbis_window(n, w);
    size_t n2 = n / 2;
    for (size_t k = 0; k < n2; k++) {
        double acc = 0.0;
        for (size_t i = 0; i < n; i++) {
            double xw = (double)(samples[i] * w[i]); /* the windowing product is f32 in the reference */
            acc += xw * cos(M_PI / (double)n2 * ((double)i + 0.5 + (double)n2 / 2.0) * ((double)k + 0.5));
        }
        out[k] = acc;
    }
    free(w);
}

/* The same evaluation for the long block with the cosine taken from an exact-period table: the argument is
 * pi/4096 * (2i + 1 + 1024)(2k + 1), so the integer product mod 8192 indexes cos(pi j / 4096). Output rounded to f32:
 * "the transform the reference would compute with an exact FFT" — the yardstick the kept-integer tests measure both
 * the oracle's f32 FFT and the device's against (test infrastructure only; not a reference function). */
static void mdct_fwd_f64_long(const float *window, const float *samples, float *out) {
    static double ctab[8192];
    static int have = 0;
    if (!have) {
        for (int j = 0; j < 8192; j++) ctab[j] = cos(M_PI * (double)j / 4096.0);
        have = 1;
    }
    double xw[2048];
    for (size_t i = 0; i < 2048; i++) xw[i] = (double)(samples[i] * window[i]); /* windowing product is f32 (mdct.rs:174-178) */
    for (size_t k = 0; k < 1024; k++) {
        double acc = 0.0;
        const unsigned kk = 2u * (unsigned)k + 1u;
        for (size_t i = 0; i < 2048; i++) acc += xw[i] * ctab[((2u * (unsigned)i + 1025u) * kk) & 8191u];
        out[k] = (float)acc;
    }
}

/* ------------------------------------------------------------------ psychoacoustic.rs */
static const float BARK_BAND_EDGES[26] = {0.0f,    100.0f,  200.0f,  300.0f,  400.0f,  510.0f,   630.0f,
                                          770.0f,  920.0f,  1080.0f, 1270.0f, 1480.0f, 1720.0f,  2000.0f,
                                          2320.0f, 2700.0f, 3150.0f, 3700.0f, 4400.0f, 5300.0f,  6400.0f,
                                          7700.0f, 9500.0f, 12000.0f, 15500.0f, 20500.0f}; /* :5-9 */

float flo_o_ath(float freq) { /* :90-104 */
    if (!(freq >= 20.0f && freq <= 20000.0f)) return 96.0f;
    float f_khz = freq / 1000.0f;
    float term1 = 3.64f * powf(f_khz, -0.8f);
    float d = f_khz - 3.3f;
    float term2 = 6.5f * expf(-0.6f * (d * d));
    float f2 = f_khz * f_khz;
    float term3 = 0.001f * (f2 * f2);
    float v = term1 - term2 + term3;
    if (v < -10.0f) v = -10.0f;
    if (v > 96.0f) v = 96.0f;
    return v;
}
float flo_o_freq_to_bark(float freq) { /* :107-111 */
    float bark = ((26.81f * freq) / (1960.0f + freq)) - 0.53f;
    if (bark < 0.0f) bark = 0.0f;
    if (bark > 24.0f) bark = 24.0f;
    return bark;
}
size_t flo_o_freq_to_bark_band(float freq) { /* :114-121 */
    for (size_t i = 1; i < 26; i++)
        if (freq < BARK_BAND_EDGES[i]) return i - 1;
    return NUM_BARK_BANDS - 1;
}

typedef struct {
    uint32_t sample_rate;
    size_t fft_size, num_coeffs;
    float freq_resolution;
    float *ath;
    size_t *bark_band;
    float spreading[NUM_BARK_BANDS][NUM_BARK_BANDS];
    float prev_energy[NUM_BARK_BANDS];
} psy_model;

/* :35-71, :125-147 */
static void psy_init(psy_model *p, uint32_t sample_rate, size_t fft_size) {
    p->sample_rate = sample_rate;
    p->fft_size = fft_size;
    p->num_coeffs = fft_size / 2;
    p->freq_resolution = (float)sample_rate / (float)fft_size;
    p->ath = (float *)malloc(p->num_coeffs * sizeof(float));
    p->bark_band = (size_t *)malloc(p->num_coeffs * sizeof(size_t));
    for (size_t k = 0; k < p->num_coeffs; k++) {
        float freq = ((float)k + 0.5f) * p->freq_resolution;
        p->ath[k] = flo_o_ath(freq);
        p->bark_band[k] = flo_o_freq_to_bark_band(freq);
    }
    for (size_t i = 0; i < NUM_BARK_BANDS; i++)
        for (size_t j = 0; j < NUM_BARK_BANDS; j++) {
            float delta_bark = (float)j - (float)i;
            float spread = delta_bark >= 0.0f ? -25.0f * delta_bark : -10.0f * delta_bark;
            float v = powf(10.0f, spread / 10.0f);
            p->spreading[i][j] = v < 1.0f ? v : 1.0f; /* .min(1.0) */
        }
    for (size_t i = 0; i < NUM_BARK_BANDS; i++) p->prev_energy[i] = 0.0f;
}
static void psy_free(psy_model *p) {
    free(p->ath);
    free(p->bark_band);
}

void flo_o_psy_tables(uint32_t sample_rate, float *ath, uint8_t *band, float *spreading) {
    psy_model p;
    psy_init(&p, sample_rate, 2048);
    for (size_t k = 0; k < 1024; k++) {
        if (ath) ath[k] = p.ath[k];
        if (band) band[k] = (uint8_t)p.bark_band[k];
    }
    if (spreading) memcpy(spreading, p.spreading, sizeof p.spreading);
    psy_free(&p);
}

/* :151-214 */
static void psy_masking_threshold(psy_model *p, const float *coeffs, float *thresholds) {
    float band_energy[NUM_BARK_BANDS] = {0};
    size_t band_count[NUM_BARK_BANDS] = {0};
    for (size_t k = 0; k < p->num_coeffs; k++) {
        size_t band = p->bark_band[k];
        float energy = coeffs[k] * coeffs[k];
        band_energy[band] += energy;
        band_count[band] += 1;
    }
    float band_db[NUM_BARK_BANDS];
    for (size_t b = 0; b < NUM_BARK_BANDS; b++) {
        if (band_count[b] > 0 && band_energy[b] > 1e-10f)
            band_db[b] = 10.0f * log10f(band_energy[b] / (float)band_count[b]);
        else
            band_db[b] = -100.0f;
    }
    float spread_threshold[NUM_BARK_BANDS];
    for (size_t i = 0; i < NUM_BARK_BANDS; i++) spread_threshold[i] = -100.0f;
    for (size_t i = 0; i < NUM_BARK_BANDS; i++)
        for (size_t j = 0; j < NUM_BARK_BANDS; j++) {
            float masking = band_db[j] + 10.0f * log10f(p->spreading[j][i]);
            spread_threshold[i] = fmaxf(spread_threshold[i], masking);
        }
    const float masking_offset = -6.0f;
    for (size_t i = 0; i < NUM_BARK_BANDS; i++) spread_threshold[i] += masking_offset;
    const float temporal_decay = 0.7f;
    for (size_t i = 0; i < NUM_BARK_BANDS; i++) {
        float temporal_mask = p->prev_energy[i] * temporal_decay;
        spread_threshold[i] = fmaxf(spread_threshold[i], temporal_mask);
        p->prev_energy[i] = spread_threshold[i];
    }
    for (size_t k = 0; k < p->num_coeffs; k++) {
        size_t band = p->bark_band[k];
        thresholds[k] = fmaxf(spread_threshold[band], p->ath[k]) - 10.0f;
    }
}

/* :218-235 */
static void psy_calculate_smr(psy_model *p, const float *coeffs, float *smr) {
    float *thr = (float *)malloc(p->num_coeffs * sizeof(float));
    psy_masking_threshold(p, coeffs, thr);
    for (size_t k = 0; k < p->num_coeffs; k++) {
        float a = fabsf(coeffs[k]);
        float signal_db = a > 1e-10f ? 20.0f * log10f(a) : -100.0f;
        smr[k] = signal_db - thr[k];
    }
    free(thr);
}

/* ------------------------------------------------------------------ lossy/encoder.rs */
float flo_o_smr_threshold(float quality) { /* :130-136 */
    if (quality >= 0.99f) return -100.0f;
    float t = fmaxf(1.0f - quality, 0.001f);
    return -60.0f * (1.0f - powf(t, 0.5f));
}

static int16_t f32_as_i16(float v) { /* Rust `as i16` */
    if (v != v) return 0;
    if (v <= -32768.0f) return INT16_MIN;
    if (v >= 32767.0f) return INT16_MAX;
    return (int16_t)v;
}

/* :109-154 */
static void quantize_coefficients(uint32_t sample_rate, float quality, const float *coeffs, const float *smr,
                                  int16_t *quantized, float *scale_factors) {
    float band_max[NUM_BARK_BANDS] = {0};
    float freq_resolution = (float)sample_rate / 2048.0f;
    for (size_t k = 0; k < 1024; k++) {
        float freq = ((float)k + 0.5f) * freq_resolution;
        size_t band = flo_o_freq_to_bark_band(freq);
        band_max[band] = fmaxf(band_max[band], fabsf(coeffs[k]));
    }
    for (size_t b = 0; b < NUM_BARK_BANDS; b++)
        scale_factors[b] = band_max[b] > 1e-10f ? 30000.0f / band_max[b] : 1.0f;
    float smr_threshold = flo_o_smr_threshold(quality);
    for (size_t k = 0; k < 1024; k++) {
        float freq = ((float)k + 0.5f) * freq_resolution;
        size_t band = flo_o_freq_to_bark_band(freq);
        quantized[k] = 0;
        if (smr[k] > smr_threshold) {
            float scaled = coeffs[k] * scale_factors[band];
            float r = roundf(scaled);
            if (r < -32768.0f) r = -32768.0f;
            if (r > 32767.0f) r = 32767.0f;
            quantized[k] = f32_as_i16(r);
        }
    }
}

uint16_t flo_o_scale_factor_word(float s) { /* :262-266 */
    if (s > 1e-10f) {
        float v = (log2f(s) * 256.0f) + 32768.0f;
        if (v < 0.0f) v = 0.0f;
        if (v > 65535.0f) v = 65535.0f;
        if (v != v) return 0;
        return (uint16_t)v;
    }
    return 0;
}

/* :317-329 */
static size_t encode_varint(uint8_t *out, uint32_t value) {
    size_t n = 0;
    for (;;) {
        uint8_t byte = (uint8_t)(value & 0x7F);
        value >>= 7;
        if (value != 0) byte |= 0x80;
        out[n++] = byte;
        if (value == 0) break;
    }
    return n;
}

/* :284-314 ; returns bytes needed (writes only while within cap) */
size_t flo_o_serialize_sparse(const int16_t *coeffs, size_t n, uint8_t *out, size_t cap) {
    size_t len = 0, i = 0;
    while (i < n) {
        size_t zero_start = i;
        while (i < n && coeffs[i] == 0) i++;
        size_t zero_count = i - zero_start;
        size_t nz_start = i;
        while (i < n && coeffs[i] != 0 && (i - nz_start) < 255) i++;
        size_t nz_count = i - nz_start;
        uint8_t vi[5];
        size_t vn = encode_varint(vi, (uint32_t)zero_count);
        for (size_t j = 0; j < vn; j++, len++)
            if (len < cap) out[len] = vi[j];
        if (len < cap) out[len] = (uint8_t)nz_count;
        len++;
        for (size_t j = nz_start; j < nz_start + nz_count; j++) {
            uint16_t v = (uint16_t)coeffs[j];
            if (len < cap) out[len] = (uint8_t)v;
            len++;
            if (len < cap) out[len] = (uint8_t)(v >> 8);
            len++;
        }
    }
    return len;
}

/* :243-280 */
static void serialize_frame(const int16_t *q, const float *sf, size_t channels, flo_buf *data) {
    buf_push(data, 0); /* BlockSize::Long */
    buf_push(data, (uint8_t)channels);
    for (size_t c = 0; c < channels; c++)
        for (size_t b = 0; b < NUM_BARK_BANDS; b++) buf_u16le(data, flo_o_scale_factor_word(sf[c * NUM_BARK_BANDS + b]));
    uint8_t tmp[1024 * 2 + 1024 * 3 + 16];
    for (size_t c = 0; c < channels; c++) {
        size_t len = flo_o_serialize_sparse(q + c * 1024, 1024, tmp, sizeof tmp);
        buf_u32le(data, (uint32_t)len);
        buf_extend(data, tmp, len);
    }
}

size_t flo_o_lossy_num_hops(size_t n_interleaved, uint8_t channels) { /* :174-179 */
    size_t per_ch = n_interleaved / channels;
    size_t total = per_ch + 1024;
    return (total + 1023) / 1024;
}

/* encoder.rs:167-239 (driver) + :63-106 (encode_frame). Captures intermediates when asked. */
static int lossy_drive_ex(const float *samples, size_t n, uint32_t sample_rate, uint8_t channels, float quality,
                          o_frame **frames_out, size_t *n_frames_out, float *cap_coeffs, float *cap_smr,
                          int16_t *cap_q, float *cap_sf, uint16_t *cap_sfw, int f64_mdct) {
    const size_t block_samples = 2048, hop_size = 1024;
    size_t ch = channels;
    if (quality < 0.0f) quality = 0.0f; /* TransformEncoder::new: quality.clamp(0,1) */
    if (quality > 1.0f) quality = 1.0f;
    size_t per_ch = n / ch;
    size_t pre_roll = hop_size;
    size_t total_samples = per_ch + pre_roll;
    size_t num_hops = (total_samples + hop_size - 1) / hop_size;
    size_t needed = (num_hops + 1) * hop_size;
    float *padded = (float *)calloc(needed * ch, sizeof(float));
    size_t lim = per_ch < needed - pre_roll ? per_ch : needed - pre_roll;
    for (size_t c = 0; c < ch; c++)
        for (size_t i = 0; i < lim; i++) {
            size_t src = i * ch + c, dst = (i + pre_roll) * ch + c;
            if (src < n && dst < needed * ch) padded[dst] = samples[src];
        }

    mdct_transform mdct;
    mdct_init(&mdct, 2048, 2 /* Vorbis */);
    psy_model *psy = (psy_model *)calloc(ch, sizeof(psy_model));
    for (size_t c = 0; c < ch; c++) psy_init(&psy[c], sample_rate, 2048);

    o_frame *frames = frames_out ? (o_frame *)calloc(num_hops ? num_hops : 1, sizeof(o_frame)) : NULL;
    size_t nf = 0;
    float *frame_data = (float *)malloc(block_samples * sizeof(float));
    float *coeffs = (float *)malloc(1024 * sizeof(float));
    float *smr = (float *)malloc(1024 * sizeof(float));
    int16_t *q = (int16_t *)malloc(ch * 1024 * sizeof(int16_t));
    float *sf = (float *)malloc(ch * NUM_BARK_BANDS * sizeof(float));

    for (size_t hop = 0; hop < num_hops; hop++) {
        size_t start = hop * hop_size * ch;
        size_t end = start + block_samples * ch;
        if (end > needed * ch) break;
        const float *fs = padded + start;
        for (size_t c = 0; c < ch; c++) {
            for (size_t i = 0; i < block_samples; i++) frame_data[i] = fs[i * ch + c];
            if (f64_mdct) mdct_fwd_f64_long(mdct.window, frame_data, coeffs);
            else mdct_fwd(&mdct, frame_data, coeffs);
            psy_calculate_smr(&psy[c], coeffs, smr);
            quantize_coefficients(sample_rate, quality, coeffs, smr, q + c * 1024, sf + c * NUM_BARK_BANDS);
            size_t o = (hop * ch + c);
            if (cap_coeffs) memcpy(cap_coeffs + o * 1024, coeffs, 1024 * sizeof(float));
            if (cap_smr) memcpy(cap_smr + o * 1024, smr, 1024 * sizeof(float));
            if (cap_q) memcpy(cap_q + o * 1024, q + c * 1024, 1024 * sizeof(int16_t));
            if (cap_sf) memcpy(cap_sf + o * NUM_BARK_BANDS, sf + c * NUM_BARK_BANDS, NUM_BARK_BANDS * sizeof(float));
            if (cap_sfw)
                for (size_t b = 0; b < NUM_BARK_BANDS; b++)
                    cap_sfw[o * NUM_BARK_BANDS + b] = flo_o_scale_factor_word(sf[c * NUM_BARK_BANDS + b]);
        }
        if (frames) {
            o_frame *f = &frames[nf];
            f->frame_type = FT_TRANSFORM;
            f->frame_samples = (uint32_t)hop_size;
            f->flags = 0;
            f->channels = (o_channel *)calloc(1, sizeof(o_channel));
            f->n_channels = 1;
            f->channels[0].residual_encoding = RE_RAW;
            buf_init(&f->channels[0].residuals);
            serialize_frame(q, sf, ch, &f->channels[0].residuals);
        }
        nf++;
    }
    free(frame_data);
    free(coeffs);
    free(smr);
    free(q);
    free(sf);
    for (size_t c = 0; c < ch; c++) psy_free(&psy[c]);
    free(psy);
    mdct_free(&mdct);
    free(padded);
    if (frames_out) *frames_out = frames;
    if (n_frames_out) *n_frames_out = nf;
    return 0;
}

static int lossy_drive(const float *samples, size_t n, uint32_t sample_rate, uint8_t channels, float quality,
                       o_frame **frames_out, size_t *n_frames_out, float *cap_coeffs, float *cap_smr,
                       int16_t *cap_q, float *cap_sf, uint16_t *cap_sfw) {
    return lossy_drive_ex(samples, n, sample_rate, channels, quality, frames_out, n_frames_out, cap_coeffs, cap_smr, cap_q,
                          cap_sf, cap_sfw, 0);
}

int lossy_encode_frames(const float *samples, size_t n, uint32_t sample_rate, uint8_t channels, float quality,
                        o_frame **frames, size_t *n_frames) {
    return lossy_drive(samples, n, sample_rate, channels, quality, frames, n_frames, NULL, NULL, NULL, NULL, NULL);
}

size_t flo_o_lossy_analyze(const float *pcm, size_t n, uint32_t sample_rate, uint8_t channels, float quality,
                           float *coeffs, float *smr, int16_t *q, float *sf, uint16_t *sf_words) {
    size_t nf = 0;
    lossy_drive(pcm, n, sample_rate, channels, quality, NULL, &nf, coeffs, smr, q, sf, sf_words);
    return nf;
}

/* The clip driver with the transform evaluated in double precision (see mdct_fwd_f64_long): same psychoacoustic
 * model, quantiser and scale words as the reference path behind it. */
size_t flo_o_lossy_analyze_f64mdct(const float *pcm, size_t n, uint32_t sample_rate, uint8_t channels, float quality,
                                   float *coeffs, float *smr, int16_t *q, float *sf, uint16_t *sf_words) {
    size_t nf = 0;
    lossy_drive_ex(pcm, n, sample_rate, channels, quality, NULL, &nf, coeffs, smr, q, sf, sf_words, 1);
    return nf;
}

int flo_o_encode_lossy(const float *pcm, size_t n, uint32_t sample_rate, uint8_t channels, float quality,
                       const uint8_t *meta, size_t meta_len, uint8_t **out, size_t *out_len) {
    if (channels == 0 || sample_rate == 0) {
        set_error("invalid arguments");
        return -1;
    }
    if (quality < 0.0f) quality = 0.0f;
    if (quality > 1.0f) quality = 1.0f;
    o_frame *frames;
    size_t nf;
    lossy_encode_frames(pcm, n, sample_rate, channels, quality, &frames, &nf);
    float ql = roundf(quality * 4.0f);
    uint8_t qlevel = ql >= 255.0f ? 255 : (ql <= 0.0f ? 0 : (uint8_t)ql);
    if (qlevel > 4) qlevel = 4;
    flo_buf b;
    buf_init(&b);
    writer_write_ex(sample_rate, channels, 16, 5, 1, qlevel, frames, nf, meta, meta_len, &b);
    for (size_t i = 0; i < nf; i++) frame_free(&frames[i]);
    free(frames);
    *out = b.data;
    *out_len = b.len;
    return 0;
}

/* ------------------------------------------------------------------ lossy/decoder.rs */

/* decoder.rs:170-188 */
static uint32_t decode_varint(const uint8_t *data, size_t len, size_t *bytes_read) {
    uint32_t value = 0;
    unsigned shift = 0;
    size_t n = 0;
    for (size_t i = 0; i < len; i++) {
        uint8_t byte = data[i];
        value |= (uint32_t)(byte & 0x7F) << shift;
        n++;
        if ((byte & 0x80) == 0) break;
        shift += 7;
        if (shift >= 32) break;
    }
    *bytes_read = n;
    return value;
}

/* decoder.rs:134-167 */
void flo_o_deserialize_sparse(const uint8_t *data, size_t len, size_t num_coeffs, int16_t *output) {
    memset(output, 0, num_coeffs * sizeof(int16_t));
    size_t pos = 0, out_idx = 0;
    while (pos < len && out_idx < num_coeffs) {
        size_t br;
        uint32_t zero_count = decode_varint(data + pos, len - pos, &br);
        pos += br;
        out_idx += zero_count;
        if (pos >= len) break;
        size_t nz = data[pos++];
        for (size_t i = 0; i < nz; i++) {
            if (pos + 2 > len || out_idx >= num_coeffs) break;
            output[out_idx] = (int16_t)(data[pos] | (data[pos + 1] << 8));
            pos += 2;
            out_idx++;
        }
    }
}

/* decoder.rs:61-131 + :29-52 + mdct.rs:437-468; lib.rs:325-352 */
int lossy_decode_file(const o_file *file, float **out, size_t *n_interleaved) {
    size_t channels = file->hdr.channels;
    uint32_t sample_rate = file->hdr.sample_rate;
    mdct_transform mdct;
    mdct_init(&mdct, 2048, 2);
    float *overlap = (float *)calloc((channels ? channels : 1) * 1024, sizeof(float));
    size_t cap = file->n_frames * 1024 * (channels ? channels : 1);
    float *all = (float *)malloc((cap ? cap : 1) * sizeof(float));
    size_t all_len = 0;
    size_t frame_count = 0;
    float freq_resolution = (float)sample_rate / 2048.0f;
    int rc = 0;

    int16_t *quant = (int16_t *)malloc(1024 * sizeof(int16_t));
    float *coeffs = (float *)malloc(1024 * sizeof(float));
    float *recon = (float *)malloc(2048 * sizeof(float));

    for (size_t fi = 0; fi < file->n_frames && rc == 0; fi++) {
        const o_frame *fr = &file->frames[fi];
        if (fr->n_channels == 0) continue;
        const uint8_t *data = fr->channels[0].residuals.data;
        size_t len = fr->channels[0].residuals.len;
        /* deserialize_frame */
        if (len < 2 || data[0] != 0 /* only Long blocks are ever produced */) {
            rc = -1;
            break;
        }
        size_t pos = 1;
        size_t nch = data[pos++];
        if (nch > channels) { /* the reference would index past overlap_buffer and panic */
            rc = -1;
            break;
        }
        float *sfs = (float *)calloc((nch ? nch : 1) * NUM_BARK_BANDS, sizeof(float));
        for (size_t c = 0; c < nch && rc == 0; c++)
            for (size_t b = 0; b < NUM_BARK_BANDS; b++) {
                if (pos + 2 > len) {
                    rc = -1;
                    break;
                }
                uint16_t log_sf = (uint16_t)(data[pos] | (data[pos + 1] << 8));
                pos += 2;
                if (log_sf > 0) sfs[c * NUM_BARK_BANDS + b] = powf(2.0f, ((float)log_sf - 32768.0f) / 256.0f);
            }
        float *frame_out = (float *)calloc(1024 * (channels ? channels : 1), sizeof(float));
        for (size_t c = 0; c < nch && rc == 0; c++) {
            if (pos + 4 > len) {
                rc = -1;
                break;
            }
            size_t blen = (size_t)data[pos] | ((size_t)data[pos + 1] << 8) | ((size_t)data[pos + 2] << 16) |
                          ((size_t)data[pos + 3] << 24);
            pos += 4;
            if (pos + blen > len) {
                rc = -1;
                break;
            }
            flo_o_deserialize_sparse(data + pos, blen, 1024, quant);
            pos += blen;
            /* decode_frame: dequantise (decoder.rs:35-48) */
            for (size_t k = 0; k < 1024; k++) {
                float freq = ((float)k + 0.5f) * freq_resolution;
                size_t band = flo_o_freq_to_bark_band(freq);
                float s = sfs[c * NUM_BARK_BANDS + band];
                coeffs[k] = s > 0.0f ? (float)quant[k] / s : 0.0f;
            }
            /* synthesize (mdct.rs:437-468) */
            mdct_inv(&mdct, coeffs, recon);
            for (size_t i = 0; i < 1024; i++) {
                frame_out[i * channels + c] = recon[i] + overlap[c * 1024 + i];
            }
            memcpy(overlap + c * 1024, recon + 1024, 1024 * sizeof(float));
        }
        if (rc == 0) {
            if (frame_count > 0) {
                /* synthesize interleaves `channel_outputs` = one per coefficient vector (nch) */
                for (size_t i = 0; i < 1024; i++)
                    for (size_t c = 0; c < channels; c++) all[all_len++] = frame_out[i * channels + c];
            }
            frame_count++;
        }
        free(frame_out);
        free(sfs);
    }
    free(quant);
    free(coeffs);
    free(recon);
    free(overlap);
    mdct_free(&mdct);
    if (rc != 0) {
        free(all);
        set_error("Failed to deserialize transform frame");
        return -1;
    }
    *out = all;
    *n_interleaved = all_len;
    return 0;
}
