/*
 * flo_oracle.h — CPU restatement of libflo's encode/decode hot path (TEST INFRASTRUCTURE).
 *
 * This is the parity oracle for the MI355X-native encoder in flo_amd/. It restates, in plain C and in
 * the reference's own evaluation order and numeric types, the Rust sources under
 * /root/reference/libflo/src (cited per function as file:line). It is NOT part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Pinning: the oracle is checked against the reference's own fixtures (Examples/ .flo files, committed
 * as data under tests/golden/) and the known-answer values in the reference's test-suite
 * (tests/test_oracle_*.py).  The one piece that cannot be pinned bit-for-bit is the 512-point FFT inside
 * the MDCT: the reference calls rustfft 6.4.1 (libflo/Cargo.lock:215-216), which is not vendored and
 * whose kernel choice is CPU-dependent; the oracle uses its own radix-2 f32 FFT, so lossy parity is
 * tolerance-based ("exact FFT bits: parity unpinned"), while the lossless path and all framing are
 * bit-exact.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off; Rust never fuses mul+add).
 */
#ifndef FLO_ORACLE_H
#define FLO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- byte buffer (stands in for Rust's Vec<u8>) ---- */
typedef struct {
    uint8_t *data;
    size_t len, cap;
} flo_buf;
void flo_buf_free(flo_buf *b);
void flo_o_free(void *p);

/* ---- core/crc32.rs:23-30 ---- */
uint32_t flo_o_crc32(const uint8_t *data, size_t n);

/* ---- core/audio_constants.rs:18-26 ---- */
int32_t flo_o_f32_to_i32(float s);
float flo_o_i32_to_f32(int32_t s);

/* ---- core/rice.rs ---- */
uint8_t flo_o_estimate_rice_parameter_i32(const int32_t *res, size_t n);         /* :29-69  */
int flo_o_rice_encode_i32(const int32_t *res, size_t n, uint8_t k, uint8_t **out, size_t *out_len); /* :84-114 */
void flo_o_rice_decode_i32(const uint8_t *enc, size_t enc_len, uint8_t k, size_t target_len, int32_t *out); /* :123-159 */

/* ---- lossless/lpc.rs (integer half) ---- */
void flo_o_autocorr_int(const int32_t *s, size_t n, size_t order, int64_t *out);                 /* :213-221 */
int flo_o_levinson_durbin_int(const int64_t *autocorr, size_t n_autocorr, size_t order,
                              int32_t *coeffs_out, uint8_t *shift_out);                           /* :225-276; 1=Some 0=None */
void flo_o_calc_residuals_int(const int32_t *s, size_t n, const int32_t *coeffs, size_t n_coeffs,
                              uint8_t shift, size_t order, int32_t *out);                         /* :279-298 */
void flo_o_fixed_predictor_residuals(const int32_t *s, size_t n, size_t order, int32_t *out);     /* :301-359 */

/* ---- lossy/mdct.rs ---- */
void flo_o_vorbis_window(size_t n, float *out);                /* :106-113 */
void flo_o_sine_window(size_t n, float *out);                  /* :99-103  */
/* window_type: 0 = sine, 2 = vorbis (mdct.rs:10-17); n = 2048 or 256 */
void flo_o_mdct_forward(const float *samples, size_t n, int window_type, float *out /* n/2 */);   /* :166-226 */
void flo_o_mdct_inverse(const float *spec, size_t n, int window_type, float *out /* n */);        /* :231-290 */
/* O(N^2) double-precision evaluation of the formula in mdct.rs:336 using the f32 window — accuracy yardstick */
void flo_o_mdct_forward_direct_f64(const float *samples, size_t n, int window_type, double *out);

/* ---- lossy/psychoacoustic.rs ---- */
float flo_o_ath(float freq);                                   /* :90-104  */
size_t flo_o_freq_to_bark_band(float freq);                    /* :114-121 */
float flo_o_freq_to_bark(float freq);                          /* :107-111 */
void flo_o_psy_tables(uint32_t sample_rate, float *ath /*1024*/, uint8_t *band /*1024*/,
                      float *spreading /*25*25 row-major [i][j]*/);                              /* :35-71,125-147 */

/* ---- lossy/encoder.rs ---- */
size_t flo_o_serialize_sparse(const int16_t *coeffs, size_t n, uint8_t *out, size_t cap);         /* :284-314 */
void flo_o_deserialize_sparse(const uint8_t *data, size_t len, size_t num_coeffs, int16_t *out);  /* decoder.rs:134-167 */
float flo_o_smr_threshold(float quality);                      /* encoder.rs:130-136 */
uint16_t flo_o_scale_factor_word(float sf);                    /* encoder.rs:262-266 */

/*
 * Run the reference clip driver (lossy/encoder.rs:167-239) and expose every per-frame intermediate:
 * arrays are [num_hops][channels][1024] (coeffs/smr/thr/q) and [num_hops][channels][25] (scale factors).
 * Any output pointer may be NULL. Returns num_hops.
 */
size_t flo_o_lossy_num_hops(size_t n_interleaved, uint8_t channels);
size_t flo_o_lossy_analyze(const float *pcm, size_t n_interleaved, uint32_t sample_rate, uint8_t channels,
                           float quality, float *coeffs, float *smr, int16_t *q, float *sf, uint16_t *sf_words);

/* the same driver with the MDCT evaluated in double precision from its definition and rounded to f32 (an accuracy
 * yardstick for the f32 FFTs of the oracle and of the device; not a reference function) */
size_t flo_o_lossy_analyze_f64mdct(const float *pcm, size_t n_interleaved, uint32_t sample_rate, uint8_t channels,
                                   float quality, float *coeffs, float *smr, int16_t *q, float *sf, uint16_t *sf_words);

/* ---- top-level API (lossless/encoder.rs:32-45, lossy/encoder.rs:167-239, lib.rs:296-352) ---- */
int flo_o_encode_lossless(const float *pcm, size_t n_interleaved, uint32_t sample_rate, uint8_t channels,
                          uint8_t bit_depth, uint8_t level, const uint8_t *meta, size_t meta_len,
                          uint8_t **out, size_t *out_len);
int flo_o_encode_lossy(const float *pcm, size_t n_interleaved, uint32_t sample_rate, uint8_t channels,
                       float quality, const uint8_t *meta, size_t meta_len, uint8_t **out, size_t *out_len);
/* returns 0 ok, else error; message via flo_o_last_error() */
int flo_o_decode(const uint8_t *flo, size_t len, float **pcm, size_t *n_interleaved,
                 uint32_t *sample_rate, uint8_t *channels);
/* lossless only, integer domain (before i32_to_f32), interleaved; used to regenerate fixture inputs */
int flo_o_decode_lossless_i32(const uint8_t *flo, size_t len, int32_t **pcm, size_t *n_interleaved,
                              uint32_t *sample_rate, uint8_t *channels);
const char *flo_o_last_error(void);

/* ---- analysis metadata of libflo::encode* (lib.rs:219-283; core/analysis.rs, core/ebu_r128.rs) ---- */
typedef struct {
    uint8_t hash[32];
    uint32_t duration_ms, sample_rate;
    uint8_t channels;
    uint8_t frequency_peaks[8], energy_profile[16];
    uint8_t avg_loudness;
} flo_o_fingerprint; /* analysis.rs:10-26 */
void flo_o_blake3(const uint8_t *data, size_t len, uint8_t out32[32]);              /* blake3 crate, hash mode */
size_t flo_o_waveform_peaks(const float *samples, size_t len, uint8_t channels, uint32_t sample_rate,
                            uint32_t peaks_per_second, float *peaks, size_t cap);     /* analysis.rs:38-115 */
void flo_o_fft256_twiddles(float *tw /* [8][128][2] */);
void flo_o_spectral_fingerprint(const float *samples, size_t len, uint8_t channels, uint32_t sample_rate,
                                flo_o_fingerprint *fp);                               /* analysis.rs:223-357 */
void flo_o_kweighting_coeffs(double sample_rate, double shelf[5], double hp[5]);      /* ebu_r128.rs:51-103 */
double flo_o_gated_lufs(const double *energies, size_t n);                            /* ebu_r128.rs:268-318 */
double flo_o_integrated_lufs(const float *samples, size_t len, uint8_t channels, uint32_t sample_rate); /* :182-318 */
double flo_o_true_peak_dbtp(const float *samples, size_t len, uint8_t channels, uint32_t sample_rate); /* ebu_r128.rs:112-179 */
/* out = {integrated_lufs, loudness_range_lu, true_peak_dbtp, sample_peak_dbfs}                   ebu_r128.rs:182-355 */
void flo_o_loudness_metrics(const float *samples, size_t len, uint8_t channels, uint32_t sample_rate, double out[4]);
/* add_analysis_data_if_missing(&[], samples, sr, ch, peaks_per_second): the META bytes (lib.rs:219-283) */
int flo_o_analysis_metadata(const float *samples, size_t len, uint32_t sample_rate, uint8_t channels,
                            uint32_t peaks_per_second, uint8_t **out, size_t *out_len);

/* ---- streaming/encoder.rs: StreamingEncoder (lossless frames of one second, pushed and pulled) ---- */
typedef struct flo_o_stream flo_o_stream;
flo_o_stream *flo_o_stream_new(uint32_t sample_rate, uint8_t channels, uint8_t bit_depth, uint8_t level);     /* :33-56 */
void flo_o_stream_free(flo_o_stream *s);
int flo_o_stream_push(flo_o_stream *s, const float *samples, size_t n);                                       /* :71-75 */
size_t flo_o_stream_pending_samples(const flo_o_stream *s);                                                   /* :59-61 */
size_t flo_o_stream_pending_frames(const flo_o_stream *s);                                                    /* :64-66 */
/* 1 = a frame came out (data malloc'ed: flo_o_free), 0 = none, -1 = error */
int flo_o_stream_next_frame(flo_o_stream *s, uint32_t *index, uint32_t *timestamp_ms, uint32_t *samples,
                            uint8_t **data, size_t *len);                                                     /* :78-85 */
int flo_o_stream_flush(flo_o_stream *s, uint32_t *index, uint32_t *timestamp_ms, uint32_t *samples, uint8_t **data,
                       size_t *len);                                                                          /* :88-110 */
int flo_o_stream_finalize(flo_o_stream *s, const uint8_t *meta, size_t meta_len, uint8_t **out, size_t *out_len); /* :113-185 */

/* ---- parsed view of a file (reader.rs:16-256), flattened for ctypes ---- */
typedef struct {
    uint8_t version_major, version_minor;
    uint16_t flags;
    uint32_t sample_rate;
    uint8_t channels, bit_depth;
    uint64_t total_samples;
    uint8_t compression_level;
    uint32_t data_crc32;
    uint64_t header_size, toc_size, data_size, extra_size, meta_size;
    uint32_t num_frames;      /* frames actually parsed */
    uint32_t crc_computed;    /* crc32 over the DATA chunk as found */
} flo_o_info;
int flo_o_info_read(const uint8_t *flo, size_t len, flo_o_info *info);

#ifdef __cplusplus
}
#endif
#endif
