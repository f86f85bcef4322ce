/* analysis.c — oracle restatement of the analysis metadata libflo's free encode functions add (TEST INFRASTRUCTURE).
 *
 *   add_analysis_data_if_missing ........ libflo/src/lib.rs:219-283
 *   extract_waveform_peaks .............. libflo/src/core/analysis.rs:38-115
 *   extract_spectral_fingerprint ........ libflo/src/core/analysis.rs:223-357
 *   compute_ebu_r128_loudness ........... libflo/src/core/ebu_r128.rs:182-355 (the integrated loudness; the loudness
 *                                         range and true peak it also computes do not reach the META chunk)
 *   FloMetadata / WaveformData / LoudnessPoint as MessagePack (rmp_serde::to_vec_named) ... core/metadata.rs:164-174,
 *                                         230-235, 328-665
 *
 * Third-party pieces, none of them vendored under /root/reference:
 *   blake3 (libflo/Cargo.toml) ........... restated from the published BLAKE3 specification; pinned by the spec's
 *                                          known answers for "" and "abc" (tests/test_oracle_analysis.py)
 *   rustfft 6.4.1, 256-point forward FFT . replaced by this file's radix-2 f32 FFT (as for the MDCT, exact bits unpinned)
 *   rmp-serde ............................ the MessagePack layout follows the format specification
 * PARITY UNPINNED for this file as a whole: no reference-made file carries analysis metadata (every .flo file under Examples/ has
 * the CLI's five-field META), and the reference's tests hold no known answers for it.
 */
#include "internal.h"
#include <math.h>

/* ------------------------------------------------------------------ BLAKE3 (hash mode, 32-byte output) */
static const uint32_t B3_IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const uint8_t B3_PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
enum { B3_CHUNK_START = 1, B3_CHUNK_END = 2, B3_PARENT = 4, B3_ROOT = 8 };
static uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
#define B3_G(a, b, c, d, mx, my)            \
    do {                                    \
        v[a] = v[a] + v[b] + (mx);          \
        v[d] = rotr32(v[d] ^ v[a], 16);     \
        v[c] = v[c] + v[d];                 \
        v[b] = rotr32(v[b] ^ v[c], 12);     \
        v[a] = v[a] + v[b] + (my);          \
        v[d] = rotr32(v[d] ^ v[a], 8);      \
        v[c] = v[c] + v[d];                 \
        v[b] = rotr32(v[b] ^ v[c], 7);      \
    } while (0)
static void b3_compress(const uint32_t cv[8], const uint32_t block[16], uint64_t counter, uint32_t block_len, uint32_t flags, uint32_t out[16]) {
    uint32_t v[16], m[16], t[16];
    for (int i = 0; i < 8; i++) v[i] = cv[i];
    for (int i = 0; i < 4; i++) v[8 + i] = B3_IV[i];
    v[12] = (uint32_t)counter;
    v[13] = (uint32_t)(counter >> 32);
    v[14] = block_len;
    v[15] = flags;
    for (int i = 0; i < 16; i++) m[i] = block[i];
    for (int r = 0; r < 7; r++) {
        B3_G(0, 4, 8, 12, m[0], m[1]);
        B3_G(1, 5, 9, 13, m[2], m[3]);
        B3_G(2, 6, 10, 14, m[4], m[5]);
        B3_G(3, 7, 11, 15, m[6], m[7]);
        B3_G(0, 5, 10, 15, m[8], m[9]);
        B3_G(1, 6, 11, 12, m[10], m[11]);
        B3_G(2, 7, 8, 13, m[12], m[13]);
        B3_G(3, 4, 9, 14, m[14], m[15]);
        for (int i = 0; i < 16; i++) t[i] = m[B3_PERM[i]];
        for (int i = 0; i < 16; i++) m[i] = t[i];
    }
    for (int i = 0; i < 8; i++) {
        out[i] = v[i] ^ v[i + 8];
        out[i + 8] = v[i + 8] ^ cv[i];
    }
}
/* chaining value of chunk `index` (up to 1024 bytes); is_root: the whole input is this one chunk */
static void b3_chunk_cv(const uint8_t *data, size_t len, uint64_t index, int is_root, uint32_t cv_out[8]) {
    uint32_t cv[8], out[16];
    memcpy(cv, B3_IV, sizeof cv);
    size_t nblocks = len ? (len + 63) / 64 : 1;
    for (size_t b = 0; b < nblocks; b++) {
        uint8_t buf[64] = {0};
        size_t take = len - b * 64 < 64 ? len - b * 64 : 64;
        if (len == 0) take = 0;
        memcpy(buf, data + b * 64, take);
        uint32_t words[16];
        for (int i = 0; i < 16; i++) words[i] = (uint32_t)buf[4 * i] | ((uint32_t)buf[4 * i + 1] << 8) | ((uint32_t)buf[4 * i + 2] << 16) | ((uint32_t)buf[4 * i + 3] << 24);
        uint32_t flags = (b == 0 ? B3_CHUNK_START : 0) | (b == nblocks - 1 ? B3_CHUNK_END : 0) | ((is_root && b == nblocks - 1) ? B3_ROOT : 0);
        b3_compress(cv, words, index, (uint32_t)take, flags, out);
        memcpy(cv, out, sizeof cv);
    }
    memcpy(cv_out, cv, 32);
}
void flo_o_blake3(const uint8_t *data, size_t len, uint8_t out32[32]) {
    size_t nchunks = len ? (len + 1023) / 1024 : 1;
    uint32_t(*cvs)[8] = (uint32_t(*)[8])malloc(nchunks * 32);
    for (size_t c = 0; c < nchunks; c++) {
        size_t off = c * 1024, take = len - off < 1024 ? len - off : 1024;
        if (len == 0) take = 0;
        b3_chunk_cv(data + off, take, c, nchunks == 1, cvs[c]);
    }
    /* the tree: pairs are merged level by level, an odd last node is carried up unchanged; the last merge is the root */
    size_t n = nchunks;
    while (n > 1) {
        size_t m = 0;
        for (size_t i = 0; i + 1 < n; i += 2) {
            uint32_t block[16], o[16];
            memcpy(block, cvs[i], 32);
            memcpy(block + 8, cvs[i + 1], 32);
            b3_compress(B3_IV, block, 0, 64, B3_PARENT | (n == 2 ? B3_ROOT : 0), o);
            memcpy(cvs[m++], o, 32);
        }
        if (n & 1) {
            memmove(cvs[m], cvs[n - 1], 32);
            m++;
        }
        n = m;
    }
    for (int i = 0; i < 8; i++)
        for (int k = 0; k < 4; k++) out32[4 * i + k] = (uint8_t)(cvs[0][i] >> (8 * k));
    free(cvs);
}

/* ------------------------------------------------------------------ Rust cast helpers */
static uint8_t f32_as_u8(float v) { /* saturating, NaN -> 0 */
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}
static float f32_max_rust(float a, float b) { /* f32::max: ignores a NaN operand */
    if (a != a) return b;
    if (b != b) return a;
    return a > b ? a : b;
}

/* ------------------------------------------------------------------ analysis.rs:38-115 */
size_t flo_o_waveform_peaks(const float *samples, size_t len, uint8_t channels, uint32_t sample_rate, uint32_t peaks_per_second,
                            float *peaks, size_t cap) {
    if (len == 0) return 0;
    const double spp = (double)sample_rate / (double)peaks_per_second;
    const double tp = ceil((double)len / (spp * (double)channels));
    size_t total_peaks = tp >= 0 ? (size_t)tp : 0, np = 0;
    for (size_t idx = 0; idx < total_peaks; idx++) {
        size_t start = (size_t)((double)idx * spp), end = (size_t)(((double)idx + 1.0) * spp);
        start *= channels;
        end *= channels;
        if (end > len) end = len;
        if (start >= len) break;
        float peak;
        if (channels == 1) {
            peak = 0.0f;
            for (size_t i = start; i < end; i++) peak = f32_max_rust(peak, fabsf(samples[i]));
        } else if (channels == 2) {
            float l = 0.0f, r = 0.0f;
            for (size_t i = start; i + 1 < end; i += 2) {
                l = f32_max_rust(l, fabsf(samples[i]));
                r = f32_max_rust(r, fabsf(samples[i + 1]));
            }
            peak = (l + r) / 2.0f;
        } else {
            peak = 0.0f;
            for (size_t i = start; i < end; i += channels) {
                size_t n = end - i < channels ? end - i : channels;
                float s = 0.0f;
                for (size_t k = 0; k < n; k++) s += samples[i + k];
                peak = f32_max_rust(peak, s / (float)n);
            }
        }
        if (np < cap) peaks[np] = peak;
        np++;
    }
    float mx = 0.0f;
    for (size_t i = 0; i < np && i < cap; i++) mx = f32_max_rust(mx, peaks[i]);
    if (mx > 0.0f)
        for (size_t i = 0; i < np && i < cap; i++) peaks[i] /= mx;
    return np;
}

/* ------------------------------------------------------------------ 256-point FFT of the fingerprint */
typedef struct { float re, im; } acpx;
void flo_o_fft256_twiddles(float *tw /* [8][128][2], stage s (len = 2 << s), index k < len/2 */) {
    for (int s = 0; s < 8; s++) {
        int len = 2 << s;
        for (int k = 0; k < len / 2; k++) {
            double ang = -2.0 * M_PI * (double)k / (double)len;
            tw[(s * 128 + k) * 2] = (float)cos(ang);
            tw[(s * 128 + k) * 2 + 1] = (float)sin(ang);
        }
    }
}
static void fft256(acpx *z) {
    static float tw[8 * 128 * 2];
    static int have = 0;
    if (!have) {
        flo_o_fft256_twiddles(tw);
        have = 1;
    }
    const size_t n = 256;
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            acpx t = z[i];
            z[i] = z[j];
            z[j] = t;
        }
    }
    for (int s = 0; s < 8; s++) {
        size_t len = (size_t)2 << s, half = len >> 1;
        for (size_t k = 0; k < half; k++) {
            float wr = tw[(s * 128 + k) * 2], wi = tw[(s * 128 + k) * 2 + 1];
            for (size_t st = 0; st < n; st += len) {
                acpx a = z[st + k], b = z[st + k + half];
                float tr = b.re * wr - b.im * wi;
                float ti = b.re * wi + b.im * wr;
                z[st + k].re = a.re + tr;
                z[st + k].im = a.im + ti;
                z[st + k + half].re = a.re - tr;
                z[st + k + half].im = a.im - ti;
            }
        }
    }
}

/* ------------------------------------------------------------------ analysis.rs:223-357 */
void flo_o_spectral_fingerprint(const float *samples, size_t len, uint8_t channels, uint32_t sample_rate, flo_o_fingerprint *fp) {
    memset(fp, 0, sizeof *fp);
    fp->sample_rate = sample_rate;
    fp->channels = channels;
    if (len == 0) return;
    const size_t spc = len / channels;
    double dms = (double)spc / (double)sample_rate * 1000.0;
    uint32_t d = dms >= 4294967295.0 ? 4294967295u : (dms <= 0 ? 0u : (uint32_t)dms);
    fp->duration_ms = d < 1 ? 1 : d;
    {
        size_t total = 9 + len * 4;
        uint8_t *buf = (uint8_t *)malloc(total);
        buf[0] = channels;
        for (int k = 0; k < 4; k++) buf[1 + k] = (uint8_t)(sample_rate >> (8 * k));
        uint32_t l32 = (uint32_t)len;
        for (int k = 0; k < 4; k++) buf[5 + k] = (uint8_t)(l32 >> (8 * k));
        memcpy(buf + 9, samples, len * 4);
        flo_o_blake3(buf, total, fp->hash);
        free(buf);
    }
    const size_t fft_size = 256;
    const size_t points[3] = {spc / 4, spc / 2, spc * 3 / 4};
    float bands[16] = {0};
    uint8_t peak_bands[8] = {0};
    acpx z[256];
    for (int p = 0; p < 3; p++) {
        const size_t si = points[p];
        if (!(si + fft_size < spc)) continue;
        for (size_t i = 0; i < fft_size; i++) {
            float s = 0.0f;
            for (size_t c = 0; c < channels; c++) {
                size_t idx = (si + i) * channels + c;
                if (idx < len) s += samples[idx];
            }
            s /= (float)channels;
            z[i].re = s;
            z[i].im = 0.0f;
        }
        fft256(z);
        for (size_t band = 0; band < 16; band++) {
            size_t sb = band * fft_size / 32, eb = (band + 1) * fft_size / 32;
            if (eb > fft_size / 2) eb = fft_size / 2;
            float energy = 0.0f;
            for (size_t b = sb; b < eb; b++) energy += z[b].re * z[b].re + z[b].im * z[b].im;
            bands[band] += sqrtf(energy);
        }
        for (size_t band = 0; band < 8; band++) {
            size_t sb = band * fft_size / 16, eb = (band + 1) * fft_size / 16;
            if (eb > fft_size / 2) eb = fft_size / 2;
            size_t best = 0;
            float bestv = 0.0f;
            int have = 0;
            for (size_t b = sb; b < eb; b++) { /* Iterator::max_by: the LAST of several equal maxima; incomparable = equal */
                float v = sqrtf(z[b].re * z[b].re + z[b].im * z[b].im);
                if (!have || !(v < bestv)) {
                    best = b;
                    bestv = v;
                    have = 1;
                }
            }
            uint8_t pv = f32_as_u8((float)best / (float)fft_size * 255.0f);
            if (pv > peak_bands[band]) peak_bands[band] = pv;
        }
    }
    float mx = 0.0f;
    for (int i = 0; i < 16; i++) mx = f32_max_rust(mx, bands[i]);
    for (int i = 0; i < 16; i++) fp->energy_profile[i] = mx > 0.0f ? f32_as_u8(bands[i] / mx * 255.0f) : 0;
    memcpy(fp->frequency_peaks, peak_bands, 8);
    float acc = 0.0f;
    for (size_t i = 0; i < len; i++) acc += samples[i] * samples[i];
    float rms = acc / (float)len;
    float v = -20.0f * log10f(rms + 1e-10f);
    if (v != v) v = v; /* clamp keeps NaN; `as u8` of NaN is 0 */
    else if (v < -60.0f) v = -60.0f;
    else if (v > 0.0f) v = 0.0f;
    fp->avg_loudness = f32_as_u8(v + 60.0f);
}

/* ------------------------------------------------------------------ ebu_r128.rs:43-111 (K-weighting), :182-318 */
void flo_o_kweighting_coeffs(double sample_rate, double shelf[5], double hp[5]) { /* b0 b1 b2 a1 a2 */
    const double f0 = 1681.974450955533, g_db = 3.999843853973347, q = 0.7071752369554196;
    const double k = tan(M_PI * f0 / sample_rate);
    const double vh = pow(10.0, g_db / 20.0);
    const double vb = pow(vh, 0.4996667741545416);
    const double a0 = 1.0 + k / q + k * k;
    shelf[0] = (vh + vb * k / q + k * k) / a0;
    shelf[1] = 2.0 * (k * k - vh) / a0;
    shelf[2] = (vh - vb * k / q + k * k) / a0;
    shelf[3] = 2.0 * (k * k - 1.0) / a0;
    shelf[4] = (1.0 - k / q + k * k) / a0;
    const double f0h = 38.13547087602444, qh = 0.5003270373238773;
    const double kh = tan(M_PI * f0h / sample_rate);
    const double a0h = 1.0 + kh / qh + kh * kh;
    hp[0] = 1.0;
    hp[1] = -2.0;
    hp[2] = 1.0;
    hp[3] = 2.0 * (kh * kh - 1.0) / a0h;
    hp[4] = (1.0 - kh / qh + kh * kh) / a0h;
}
/* gating of the block energies (ebu_r128.rs:268-318); energies[k] already summed over channels */
double flo_o_gated_lufs(const double *energies, size_t n) {
    if (n == 0) return -23.0;
    const double abs_gate = pow(10.0, (-70.0 + 0.691) / 10.0);
    double sum = 0.0;
    size_t cnt = 0;
    for (size_t i = 0; i < n; i++)
        if (energies[i] >= abs_gate) {
            sum += energies[i];
            cnt++;
        }
    if (cnt == 0) return -23.0;
    const double ungated = -0.691 + 10.0 * log10(sum / (double)cnt);
    const double rel_gate = pow(10.0, (ungated - 10.0 + 0.691) / 10.0);
    double s2 = 0.0;
    size_t c2 = 0;
    for (size_t i = 0; i < n; i++)
        if (energies[i] >= abs_gate && energies[i] >= rel_gate) {
            s2 += energies[i];
            c2++;
        }
    if (c2 == 0) return ungated;
    return -0.691 + 10.0 * log10(s2 / (double)c2);
}
double flo_o_integrated_lufs(const float *samples, size_t len, uint8_t channels, uint32_t sample_rate) {
    if (len == 0 || channels == 0) return -23.0;
    const double sr = (double)sample_rate;
    const size_t hop = (size_t)round(sr * 0.1), block = hop * 4;
    const size_t frames = len / channels;
    double shelf[5], hp[5];
    flo_o_kweighting_coeffs(sr, shelf, hp);
    double *kw = (double *)malloc((frames ? frames : 1) * channels * sizeof(double)); /* [ch][frames] */
    for (size_t c = 0; c < channels; c++) {
        double s1 = 0, s2 = 0, h1 = 0, h2 = 0;
        for (size_t i = 0; i < frames; i++) {
            double x = (double)samples[i * channels + c];
            double y = shelf[0] * x + s1;
            s1 = shelf[1] * x - shelf[3] * y + s2;
            s2 = shelf[2] * x - shelf[4] * y;
            double y2 = hp[0] * y + h1;
            h1 = hp[1] * y - hp[3] * y2 + h2;
            h2 = hp[2] * y - hp[4] * y2;
            kw[c * frames + i] = y2;
        }
    }
    double *en = NULL;
    size_t nb = 0, capb = 0;
    size_t start = 0;
    while (start < frames) {
        size_t end = start + block < frames ? start + block : frames;
        if (end <= start) break;
        double energy = 0.0;
        const size_t l = end - start;
        for (size_t c = 0; c < channels; c++) {
            double ss = 0.0;
            for (size_t i = start; i < end; i++) ss += kw[c * frames + i] * kw[c * frames + i];
            energy += ss / (double)l;
        }
        if (nb == capb) {
            capb = capb ? 2 * capb : 64;
            en = (double *)realloc(en, capb * sizeof(double));
        }
        en[nb++] = energy;
        if (end == frames) break;
        start += hop;
        if (hop == 0) break;
    }
    double r = flo_o_gated_lufs(en, nb);
    free(en);
    free(kw);
    return r;
}

/* compute_true_peak (ebu_r128.rs:112-179): a 49-tap Hann-windowed sinc (cutoff 0.45 fs, designed at 4 fs, normalised to
 * unit sum) evaluated at the four sub-sample positions of every sample. The tap index is (pos - 24 + k) cast to usize -
 * a floor - so the four positions of a sample read the SAME samples with the SAME taps: the result is the largest
 * |output| of that FIR at the plain rate (the loop over `sub` is kept as the reference has it). */
double flo_o_true_peak_dbtp(const float *samples, size_t len, uint8_t channels, uint32_t sample_rate) {
    if (len == 0 || channels == 0) return -150.0;
    const unsigned factor = 4;
    const double oversample_rate = (double)sample_rate * (double)factor;
    const double cutoff = (double)sample_rate * 0.45;
    enum { TAPS = 49 };
    double coeffs[TAPS];
    const double center = (double)(TAPS - 1) / 2.0;
    for (int i = 0; i < TAPS; i++) {
        const double n = (double)i - center;
        const double sinc = fabs(n) < 1e-12 ? 2.0 * cutoff / oversample_rate : sin(2.0 * cutoff * n / oversample_rate) / (M_PI * n);
        const double window = 0.5 * (1.0 - cos(2.0 * M_PI * (double)i / (double)(TAPS - 1)));
        coeffs[i] = sinc * window;
    }
    double sum = 0.0;
    for (int i = 0; i < TAPS; i++) sum += coeffs[i];
    for (int i = 0; i < TAPS; i++) coeffs[i] /= sum;
    double max_peak = 0.0;
    /* `samples.iter().skip(ch).step_by(channels)`: every channels-th value from ch on (a trailing partial frame counts) */
    for (size_t ch = 0; ch < channels; ch++) {
        const size_t n_ch = len > ch ? (len - ch + channels - 1) / channels : 0;
        if (n_ch == 0) continue;
        for (size_t i = 0; i < n_ch; i++)
            for (unsigned sub = 0; sub < factor; sub++) {
                const double pos = (double)i + (double)sub / (double)factor;
                double acc = 0.0;
                for (int k = 0; k < TAPS; k++) {
                    const double src = pos - center + (double)k;
                    if (src >= 0.0 && src < (double)n_ch) acc += (double)samples[(size_t)src * channels + ch] * coeffs[k];
                }
                if (fabs(acc) > max_peak) max_peak = fabs(acc);
            }
    }
    return max_peak > 1e-9 ? 20.0 * log10(max_peak) : -150.0;
}

/* compute_ebu_r128_loudness (ebu_r128.rs:182-355): out = {integrated_lufs, loudness_range_lu, true_peak_dbtp,
 * sample_peak_dbfs} */
static int cmp_double(const void *a, const void *b) {
    const double x = *(const double *)a, y = *(const double *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
void flo_o_loudness_metrics(const float *samples, size_t len, uint8_t channels, uint32_t sample_rate, double out[4]) {
    out[0] = -23.0;
    out[1] = 0.0;
    out[2] = -150.0;
    out[3] = -150.0;
    if (len == 0 || channels == 0) return;
    const double sr = (double)sample_rate;
    const size_t hop = (size_t)round(sr * 0.1), block = hop * 4;
    const size_t frames = len / channels;
    /* sample peak: per channel, over whole frames */
    double sample_peak = -150.0;
    for (size_t c = 0; c < channels; c++) {
        double peak = 0.0;
        for (size_t i = 0; i < frames; i++) {
            const double a = fabs((double)samples[i * channels + c]);
            if (a > peak) peak = a;   /* f64::max ignores NaN the same way */
        }
        if (peak > 1e-6) {
            const double db = 20.0 * log10(peak);
            if (db > sample_peak) sample_peak = db;
        }
    }
    out[3] = sample_peak;
    double shelf[5], hp[5];
    flo_o_kweighting_coeffs(sr, shelf, hp);
    double *kw = (double *)malloc((frames ? frames : 1) * channels * sizeof(double));
    for (size_t c = 0; c < channels; c++) {
        double s1 = 0, s2 = 0, h1 = 0, h2 = 0;
        for (size_t i = 0; i < frames; i++) {
            double x = (double)samples[i * channels + c];
            double y = shelf[0] * x + s1;
            s1 = shelf[1] * x - shelf[3] * y + s2;
            s2 = shelf[2] * x - shelf[4] * y;
            double y2 = hp[0] * y + h1;
            h1 = hp[1] * y - hp[3] * y2 + h2;
            h2 = hp[2] * y - hp[4] * y2;
            kw[c * frames + i] = y2;
        }
    }
    double *en = NULL, *bl = NULL;
    size_t nb = 0, capb = 0, start = 0;
    while (start < frames) {
        size_t end = start + block < frames ? start + block : frames;
        if (end <= start) break;
        double energy = 0.0;
        const size_t l = end - start;
        for (size_t c = 0; c < channels; c++) {
            double ss = 0.0;
            for (size_t i = start; i < end; i++) ss += kw[c * frames + i] * kw[c * frames + i];
            energy += ss / (double)l;
        }
        if (nb == capb) {
            capb = capb ? 2 * capb : 64;
            en = (double *)realloc(en, capb * sizeof(double));
            bl = (double *)realloc(bl, capb * sizeof(double));
        }
        en[nb] = energy;
        bl[nb] = energy > 0.0 ? -0.691 + 10.0 * log10(energy) : -150.0;
        nb++;
        if (end == frames) break;
        start += hop;
        if (hop == 0) break;
    }
    free(kw);
    out[2] = flo_o_true_peak_dbtp(samples, len, channels, sample_rate);
    if (nb) {
        const double abs_gate = pow(10.0, (-70.0 + 0.691) / 10.0);
        double sum = 0.0;
        size_t cnt = 0;
        for (size_t i = 0; i < nb; i++)
            if (en[i] >= abs_gate) {
                sum += en[i];
                cnt++;
            }
        if (cnt) {
            const double ungated = -0.691 + 10.0 * log10(sum / (double)cnt);
            const double rel_gate = pow(10.0, (ungated - 10.0 + 0.691) / 10.0);
            double *vals = (double *)malloc(nb * sizeof(double));
            double s2 = 0.0;
            size_t c2 = 0;
            for (size_t i = 0; i < nb; i++)
                if (en[i] >= abs_gate && en[i] >= rel_gate) {
                    s2 += en[i];
                    vals[c2++] = bl[i];
                }
            out[0] = c2 == 0 ? ungated : -0.691 + 10.0 * log10(s2 / (double)c2);
            if (c2 >= 2) {   /* LRA: 10th - 95th percentile of the gated block loudness, linear interpolation */
                qsort(vals, c2, sizeof(double), cmp_double);
                const double n = (double)c2;
                const double pos[2] = {0.10 * (n - 1.0), 0.95 * (n - 1.0)};
                double pv[2];
                for (int q = 0; q < 2; q++) {
                    const size_t i = (size_t)floor(pos[q]);
                    const double frac = pos[q] - (double)i;
                    pv[q] = i + 1 < c2 ? vals[i] * (1.0 - frac) + vals[i + 1] * frac : vals[i];
                }
                out[1] = pv[1] - pv[0];
            }
            free(vals);
        }
    }
    free(en);
    free(bl);
}

/* ------------------------------------------------------------------ MessagePack (named maps) */
static void mp_uint(flo_buf *b, uint64_t v) {
    if (v < 128) buf_push(b, (uint8_t)v);
    else if (v < 256) { buf_push(b, 0xcc); buf_push(b, (uint8_t)v); }
    else if (v < 65536) { buf_push(b, 0xcd); buf_push(b, (uint8_t)(v >> 8)); buf_push(b, (uint8_t)v); }
    else if (v < 4294967296ull) { buf_push(b, 0xce); for (int k = 3; k >= 0; k--) buf_push(b, (uint8_t)(v >> (8 * k))); }
    else { buf_push(b, 0xcf); for (int k = 7; k >= 0; k--) buf_push(b, (uint8_t)(v >> (8 * k))); }
}
static void mp_str(flo_buf *b, const char *s) {
    size_t n = strlen(s);
    if (n < 32) buf_push(b, (uint8_t)(0xa0 | n));
    else { buf_push(b, 0xd9); buf_push(b, (uint8_t)n); }
    buf_extend(b, s, n);
}
static void mp_f32(flo_buf *b, float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    buf_push(b, 0xca);
    for (int k = 3; k >= 0; k--) buf_push(b, (uint8_t)(u >> (8 * k)));
}
static void mp_array(flo_buf *b, size_t n) {
    if (n < 16) buf_push(b, (uint8_t)(0x90 | n));
    else if (n < 65536) { buf_push(b, 0xdc); buf_push(b, (uint8_t)(n >> 8)); buf_push(b, (uint8_t)n); }
    else { buf_push(b, 0xdd); for (int k = 3; k >= 0; k--) buf_push(b, (uint8_t)(n >> (8 * k))); }
}
static void mp_bin(flo_buf *b, const uint8_t *p, size_t n) {
    if (n < 256) { buf_push(b, 0xc4); buf_push(b, (uint8_t)n); }
    else if (n < 65536) { buf_push(b, 0xc5); buf_push(b, (uint8_t)(n >> 8)); buf_push(b, (uint8_t)n); }
    else { buf_push(b, 0xc6); for (int k = 3; k >= 0; k--) buf_push(b, (uint8_t)(n >> (8 * k))); }
    buf_extend(b, p, n);
}

/* lib.rs:219-283 with empty input metadata: FloMetadata::default() + the analysis fields, in declaration order
 * (metadata.rs:442 length_ms, :589 waveform_data, :594 spectrum_fingerprint, :607 loudness_profile) */
int flo_o_analysis_metadata(const float *samples, size_t len, uint32_t sample_rate, uint8_t channels, uint32_t peaks_per_second,
                            uint8_t **out, size_t *out_len) {
    flo_buf b, fpb;
    buf_init(&b);
    buf_init(&fpb);
    buf_push(&b, 0x84);
    /* length_ms */
    mp_str(&b, "length_ms");
    {
        double ms = (double)((uint64_t)len / channels) / (double)sample_rate * 1000.0;
        mp_uint(&b, ms <= 0 ? 0 : (uint64_t)ms);
    }
    /* waveform_data */
    mp_str(&b, "waveform_data");
    {
        size_t cap = len / (channels ? channels : 1) + 16;
        float *peaks = (float *)malloc(cap * sizeof(float));
        size_t np = flo_o_waveform_peaks(samples, len, channels, sample_rate, peaks_per_second, peaks, cap);
        buf_push(&b, 0x83);
        mp_str(&b, "peaks_per_second");
        mp_uint(&b, peaks_per_second);
        mp_str(&b, "peaks");
        mp_array(&b, np);
        for (size_t i = 0; i < np; i++) mp_f32(&b, peaks[i]);
        mp_str(&b, "channels");
        mp_uint(&b, channels);
        free(peaks);
    }
    /* spectrum_fingerprint: bytes of the named-map serialisation of SpectralFingerprint (analysis.rs:10-26) */
    mp_str(&b, "spectrum_fingerprint");
    {
        flo_o_fingerprint fp;
        flo_o_spectral_fingerprint(samples, len, channels, sample_rate, &fp);
        buf_push(&fpb, 0x87);
        mp_str(&fpb, "hash");
        mp_array(&fpb, 32);
        for (int i = 0; i < 32; i++) mp_uint(&fpb, fp.hash[i]);
        mp_str(&fpb, "duration_ms");
        mp_uint(&fpb, fp.duration_ms);
        mp_str(&fpb, "sample_rate");
        mp_uint(&fpb, fp.sample_rate);
        mp_str(&fpb, "channels");
        mp_uint(&fpb, fp.channels);
        mp_str(&fpb, "frequency_peaks");
        mp_array(&fpb, 8);
        for (int i = 0; i < 8; i++) mp_uint(&fpb, fp.frequency_peaks[i]);
        mp_str(&fpb, "energy_profile");
        mp_array(&fpb, 16);
        for (int i = 0; i < 16; i++) mp_uint(&fpb, fp.energy_profile[i]);
        mp_str(&fpb, "avg_loudness");
        mp_uint(&fpb, fp.avg_loudness);
        mp_bin(&b, fpb.data, fpb.len);
    }
    /* loudness_profile: one LoudnessPoint { timestamp_ms: 0, lufs: integrated as f32 } */
    mp_str(&b, "loudness_profile");
    buf_push(&b, 0x91);
    buf_push(&b, 0x82);
    mp_str(&b, "timestamp_ms");
    mp_uint(&b, 0);
    mp_str(&b, "lufs");
    mp_f32(&b, (float)flo_o_integrated_lufs(samples, len, channels, sample_rate));
    flo_buf_free(&fpb);
    *out = b.data;
    *out_len = b.len;
    return 0;
}
