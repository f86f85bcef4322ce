/*
 * flo_hip.h — C ABI of the MI355X-native flo encoder (libflo_hip.so).
 *
 * Drop-in boundary for libflo's per-clip encode path. Every entry point is `extern "C"`, takes plain
 * pointers and sizes, and replaces one reference interface (cited as file:line under /root/reference):
 *
 *   flo_encode_lossy      <- lossy::TransformEncoder::new(sr,ch,q).encode_to_flo(samples, meta)
 *                            libflo/src/lossy/encoder.rs:36-53,167-239 (re-exported as LossyEncoder, lib.rs:21-24)
 *   flo_encode_lossless   <- lossless::Encoder::new(sr,ch,bits).with_compression(level).encode(samples, meta)
 *                            libflo/src/lossless/encoder.rs:17-45
 *   flo_encode_batch      <- the same two calls, once per clip (callers loop in reflo/src/lib.rs:286-306)
 *   flo_decode            <- libflo::decode(data) / lossless::Decoder::new().decode(data)
 *                            libflo/src/lib.rs:296-352, lossless/decoder.rs:14-72, lossy/decoder.rs:29-188
 *   flo_free              <- drop of the returned Vec<u8> / Vec<f32>
 *   error codes + flo_last_error <- FloResult<T> = Result<T, String>   (core/types.rs:281)
 *
 * Conventions mirror the reference: inputs are interleaved f32 PCM in [-1,1], length
 * n_interleaved = sample_frames * channels (a trailing partial sample-frame is ignored, as the reference's
 * integer division does); metadata is an opaque byte string appended verbatim as the META chunk; the output
 * is one malloc'ed buffer holding a complete .flo file, released with flo_free. One flo_ctx per host
 * thread / GPU; contexts are independent. A "fresh encoder per clip" is the contract for lossy encodes
 * (the reference never resets the psychoacoustic state between calls; all its callers build a new encoder).
 *
 * Sample rates: any rate the reference accepts (tested 8 kHz .. 384 kHz for lossy encode and decode, 8 .. 192 kHz
 * lossless); channels: 1 .. 8 lossy, 1 .. 255 lossless.
 *
 * There is NO CPU fallback: every encode and decode entry point runs the HIP kernels on the context's device and
 * fails with a non-zero code if no gfx950 device is usable.
 */
#ifndef FLO_HIP_H
#define FLO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLO_OK 0
#define FLO_ERR_ARG 1      /* invalid argument */
#define FLO_ERR_DEVICE 2   /* HIP runtime / device error (text in flo_last_error) */
#define FLO_ERR_NOMEM 3
#define FLO_ERR_STATE 4    /* call sequence error on a batch object */
#define FLO_ERR_FORMAT 5   /* not a decodable .flo file; flo_last_error holds the reference reader's message */

#define FLO_MODE_LOSSLESS 0
#define FLO_MODE_LOSSY 1

typedef struct flo_ctx flo_ctx; /* owns device id, stream, constant tables, scratch */

int flo_ctx_create(int device, flo_ctx **out);
void flo_ctx_destroy(flo_ctx *ctx);
const char *flo_last_error(const flo_ctx *ctx); /* message of the last failing call on this ctx */
/* message of the last failing flo_ctx_create (no ctx exists yet to hold it) */
const char *flo_last_create_error(void);
void flo_free(void *p);
/* device facts for reports: name, CU count, total HBM bytes */
int flo_ctx_device_info(const flo_ctx *ctx, char *name, size_t name_cap, int *compute_units, uint64_t *hbm_bytes);

/* ---- one clip in, one .flo file out (host buffers; includes H2D/D2H) ------------------------------- */
int flo_encode_lossy(flo_ctx *ctx, const float *pcm, size_t n_interleaved, uint32_t sample_rate, uint8_t channels,
                     float quality, const uint8_t *meta, size_t meta_len, uint8_t **out, size_t *out_len);
int flo_encode_lossless(flo_ctx *ctx, const float *pcm, size_t n_interleaved, uint32_t sample_rate,
                        uint8_t channels, uint8_t bit_depth, uint8_t level, const uint8_t *meta, size_t meta_len,
                        uint8_t **out, size_t *out_len);
/* many clips in one launch sequence; outs[i]/out_lens[i] receive one malloc'ed .flo per clip */
int flo_encode_batch(flo_ctx *ctx, int mode, size_t n_clips, const float *const *pcm, const size_t *n_interleaved,
                     uint32_t sample_rate, uint8_t channels, float quality_or_level, uint8_t **outs,
                     size_t *out_lens);

/* ---- one .flo file in, interleaved f32 PCM out (host buffers) ---------------------------------------------
 * Replaces libflo::decode (lib.rs:296-315): files with a transform frame go through the device inverse MDCT
 * (first frame dropped as the reference does: (frames - 1) * 1024 sample-frames come back), all others through the
 * device Rice / predictor kernels (bit-exact integers, then * 1/32767). The container is parsed on the host like
 * Reader::read (reader.rs:16-256): its error strings come back through flo_last_error with FLO_ERR_FORMAT. Like the
 * reference, the CRC is not verified by decode. *pcm is malloc'ed (flo_free); sample_rate / channels may be NULL. */
int flo_decode(flo_ctx *ctx, const uint8_t *flo, size_t len, float **pcm, size_t *n_interleaved,
               uint32_t *sample_rate, uint8_t *channels);
/* the integers before the float conversion (lossless files only): what parity tests compare bit for bit */
int flo_decode_lossless_i32(flo_ctx *ctx, const uint8_t *flo, size_t len, int32_t **pcm, size_t *n_interleaved,
                            uint32_t *sample_rate, uint8_t *channels);

/* What the container reader (Reader::read, reader.rs:16-256) extracts from a file, without touching the device: the
 * host half of flo_decode on its own. Returns FLO_OK or FLO_ERR_FORMAT with the reader's message in err (may be NULL,
 * err_cap bytes). No context needed. */
typedef struct flo_container_info {
    uint8_t version_major, version_minor, channels, bit_depth;
    uint8_t compression_level, is_transform, pad0, pad1;
    uint16_t flags, pad2;
    uint32_t sample_rate;
    uint32_t data_crc32;
    uint32_t n_frames;          /* frames the reader accepted */
    uint64_t total_samples;     /* header field */
    uint64_t data_start, data_size;
    uint64_t frame_samples_sum; /* sum of frame_samples over the frames read */
} flo_container_info;
int flo_probe_container(const uint8_t *flo, size_t len, flo_container_info *out, char *err, size_t err_cap);

/* ---- device-resident batch (the throughput path: PCM already in HBM, bitstreams left in HBM) -------- */
typedef struct flo_batch flo_batch;

/* Plan a batch: clip i has n_interleaved[i] samples. Allocates the HBM input buffer (clips back to back,
 * each start 16-byte aligned), the output bitstream buffer and scratch. mode = FLO_MODE_LOSSY/LOSSLESS. */
int flo_batch_create(flo_ctx *ctx, int mode, size_t n_clips, const size_t *n_interleaved, uint32_t sample_rate,
                     uint8_t channels, float quality_or_level, flo_batch **out);
void flo_batch_destroy(flo_batch *b);
/* device pointer to clip i's interleaved f32 PCM (n_interleaved[i] floats) — fill it however you like */
float *flo_batch_clip_device_ptr(flo_batch *b, size_t clip);
/* H2D copy of one clip */
int flo_batch_upload(flo_batch *b, size_t clip, const float *pcm);
/* fill every clip with the integer-exact synthetic signal of flo_synth.h (device kernel), seeded by seed;
 * clip ids start at clip_id0 so that ranks of a sharded job generate disjoint parts of one corpus */
int flo_batch_fill_synthetic(flo_batch *b, uint32_t seed, uint64_t clip_id0);
/* launch the encode kernels on the ctx stream (asynchronous). which = 0: auto, 1: clip-chain kernel with one wave
 * per channel, 2: frame-parallel kernels, 5: clip-chain kernel with one transform wave that carries both channels in
 * lock-step (packed f32 arithmetic) + one quantiser-and-packer wave per stereo clip, persistent workgroups that deal
 * the clips dynamically: the form auto picks for stereo batches; 3 and 4 (stereo chain forms of earlier rounds,
 * retired) mean 5 (lossy only; all forms produce identical bytes; 5 falls back to 1 for mono) */
int flo_batch_encode(flo_batch *b, int which);
int flo_batch_sync(flo_batch *b);
/* after sync: total compressed DATA bytes of the batch, and of one clip */
int flo_batch_data_bytes(flo_batch *b, uint64_t *total);
/* D2H + container assembly (header, TOC, CRC32, META) of one clip -> malloc'ed .flo */
int flo_batch_fetch(flo_batch *b, size_t clip, const uint8_t *meta, size_t meta_len, uint8_t **out,
                    size_t *out_len);
/* device address and size of the packed per-clip DATA chunks, for a zero-copy hand-off (e.g. an RCCL gather):
 * clip i's DATA chunk is at base + offsets[i], sizes[i] bytes (both arrays host-side, n_clips entries) */
int flo_batch_device_streams(flo_batch *b, const uint8_t **base, const uint64_t **offsets, const uint64_t **sizes);

/* pack every clip's DATA chunk back to back (16-byte aligned offsets) into a caller-owned device buffer on the ctx
 * stream: the single contiguous payload a rank contributes to the RCCL gather. offsets has n_clips + 1 entries. */
int flo_batch_pack_streams(flo_batch *b, void *dst_device, size_t dst_cap, uint64_t *offsets);
/* The same for FINISHED FILES: after flo_batch_sync every clip's header (version 1.2, CRC32 of DATA, sizes), TOC and
 * DATA sit contiguously in HBM — writer.rs:132-224 and core/crc32.rs done on the device — i.e. a complete .flo file
 * with an empty META chunk. device_files exposes them in place; pack_files copies them back to back (16-byte aligned
 * offsets) into caller-owned device memory: the payload of the multi-GPU gather. (bit_depth in the header is 16;
 * flo_batch_fetch / flo_encode_lossless patch the caller's value and the META size when they copy a file out.) */
int flo_batch_device_files(flo_batch *b, const uint8_t **base, const uint64_t **offsets, const uint64_t **sizes);
int flo_batch_pack_files(flo_batch *b, void *dst_device, size_t dst_cap, uint64_t *offsets);
/* after sync: decode every clip from its device bitstream into dst (device memory, dst_cap floats). Clip i's PCM —
 * exactly what flo_decode returns for its file: (frames_i - 1) * 1024 * channels floats for a lossy clip, the
 * clip's interleaved samples for a lossless one — starts at offsets[i] floats (host array, n_clips entries). The
 * payload never leaves HBM and nothing is parsed: frame and wrapper descriptions come from the batch's own encode
 * records. Full-size round-trip checks and the decode throughput figures. */
int flo_batch_decode(flo_batch *b, float *dst_device, size_t dst_cap_floats, uint64_t *offsets);

/* ---- multi-GPU: one process per GPU, one exchange step per batch -------------------------
 * Clips shard across ranks with no communication during the encode. The exchange step is the variable-size gather of
 * every rank's finished .flo files to the root, written directly against RCCL over xGMI: ncclAllGather of the packed
 * sizes, then grouped ncclSend / ncclRecv on a communication stream of the library's own, double-buffered so that the
 * transfer of step k overlaps the encode of step k + 1. (The reference has no counterpart: its callers encode clips one
 * after the other in one thread, reflo/src/lib.rs:286-306.)
 *   rank 0:   flo_dist_unique_id(id); share id with the other ranks by any side channel (file, socket, MPI, ...)
 *   all:      flo_dist_create(ctx, id, rank, world, root, &d);
 *   per step: flo_batch_encode(b, 0); flo_batch_sync(b); flo_dist_gather_submit(d, b);
 *   at the end: flo_dist_gather_flush(d);   root: flo_dist_gather_result(d, &base, &offs, &sizes)
 * gather_submit never waits on the host for the device: it packs the batch's files (device), all-gathers the packed
 * sizes into pinned host memory (asynchronous) and posts the point-to-point transfers of the PREVIOUS submit, whose
 * sizes arrived while this step was being encoded. */
typedef struct flo_dist flo_dist;
#define FLO_DIST_ID_BYTES 128
int flo_dist_unique_id(uint8_t *id /* FLO_DIST_ID_BYTES */);
int flo_dist_create(flo_ctx *ctx, const uint8_t *id, int rank, int world, int root, flo_dist **out);
void flo_dist_destroy(flo_dist *d);
int flo_dist_gather_submit(flo_dist *d, flo_batch *b);
int flo_dist_gather_flush(flo_dist *d);
/* root, after a flush: rank r's packed files (each a complete .flo file without META, starts 16-byte aligned, lengths
 * in their headers) lie at base + rank_offsets[r], rank_sizes[r] bytes, in device memory owned by d */
int flo_dist_gather_result(flo_dist *d, const uint8_t **base, const uint64_t **rank_offsets, const uint64_t **rank_sizes);
void *flo_dist_stream(flo_dist *d);   /* hipStream_t of the communication stream */
/* Second exchange mode, offered beside the gather (the gather is what the path's single exchange step is; this shows what
 * the encode scales to when the root's link ingress is not in the way): the files stay on the ranks that made them and
 * ONE ncclAllGather of 24 bytes per clip tells every rank where each file of every rank lies (byte offset in its
 * owner's device buffer), how long it is and the CRC32 of its DATA chunk. max_clips = the largest clip count of any
 * rank (the same value on all ranks). Asynchronous on the communication stream, double-buffered like the gather.
 *   per step: flo_batch_encode(b, 0); flo_batch_sync(b); flo_dist_table_submit(d, b, max_clips);
 *   at the end: flo_dist_table_flush(d);  every rank: flo_dist_table_result(d, &rows, &row_words, &max_clips)
 * rows = host memory owned by d, `world` rows of row_words u64: [0] the rank's clip count | [1 .. max] sizes |
 * [1 + max .. 2 max] offsets | [1 + 2 max .. 3 max] CRC32 values (of the last submitted step). */
int flo_dist_table_submit(flo_dist *d, flo_batch *b, size_t max_clips);
int flo_dist_table_flush(flo_dist *d);
int flo_dist_table_result(flo_dist *d, const uint64_t **rows, size_t *row_words, size_t *max_clips);
/* The persistent encode kernels start one workgroup per compute unit; RCCL's send / receive are kernels too, so with
 * more than one rank the library leaves `n` compute units free for them (default 8 once a communicator with world > 1
 * exists, 0 otherwise; the environment variable FLO_RESERVE_CUS sets it at context creation). Costs n / 256 of the
 * single-GPU rate; without it the transfer of step k cannot start before the encode of step k + 1 has ended. */
int flo_ctx_reserve_cus(flo_ctx *ctx, int n);
/* Host-buffer entry points (flo_encode_*): which path uploads beyond 8 MB take on this host - "pageable-direct" (the
 * runtime's copy straight from the caller's memory) or "pinned-ring" (copy threads + pinned staging) - and the rates the
 * one-time probe measured for both (GB/s; 0 and "not measured yet" before the first large upload). A single clip always
 * goes direct. FLO_UPLOAD_PATH=direct|ring overrides the probe. */
int flo_ctx_upload_path(flo_ctx *ctx, char *name, size_t name_cap, double *direct_gbs, double *ring_gbs);
/* compute units the persistent encode kernels currently leave free (0 when nothing is reserved). A reservation that
 * flo_dist_create made by default ends with flo_dist_destroy; one set by the caller or FLO_RESERVE_CUS stays. */
int flo_ctx_reserved_cus(flo_ctx *ctx);

/* ---- analysis metadata: what libflo::encode / encode_lossy / encode_with_bitrate add to META (lib.rs:219-283) -------
 * flo_analyze computes, on the device, what add_analysis_data_if_missing computes from the samples: waveform peaks
 * (core/analysis.rs:38-115), the spectral fingerprint (analysis.rs:223-357: BLAKE3 of the content, band energies and
 * peak bins of three 256-point FFT sections, average loudness) and the EBU R128 integrated loudness
 * (core/ebu_r128.rs:182-318). flo_analysis_metadata frames them as the MessagePack FloMetadata the reference serialises
 * for an empty input META (fields length_ms, waveform_data, spectrum_fingerprint, loudness_profile; malloc'ed, flo_free):
 * pass the result as `meta` to flo_encode_lossless / flo_encode_lossy to get what libflo::encode* return. */
typedef struct flo_analysis {
    uint32_t n_peaks;            /* waveform peaks written to `peaks` (normalised to the largest) */
    uint32_t duration_ms, sample_rate;
    uint8_t channels, avg_loudness, pad0, pad1;
    uint8_t hash[32];
    uint8_t frequency_peaks[8];
    uint8_t energy_profile[16];
    double integrated_lufs;
    uint64_t length_ms;
    /* the rest of compute_ebu_r128_loudness (ebu_r128.rs:182-355; not part of the META chunk, printed by the CLI's
     * `analysis` command like reflo's): range of the gated block loudness, the 49-tap "true peak", the sample peak */
    double loudness_range_lu, true_peak_dbtp, sample_peak_dbfs;
    /* the f32 accumulator avg_loudness is derived from (analysis.rs:338: the sequential sum of s * s over all samples),
     * bit for bit the reference's value at any length; exposed for the tests */
    float sum_squares;
    uint32_t pad2;
} flo_analysis;
int flo_analyze(flo_ctx *ctx, const float *pcm, size_t n_interleaved, uint32_t sample_rate, uint8_t channels,
                uint32_t peaks_per_second, float *peaks, size_t peaks_cap, flo_analysis *out);
int flo_analysis_metadata(flo_ctx *ctx, const float *pcm, size_t n_interleaved, uint32_t sample_rate, uint8_t channels,
                          uint32_t peaks_per_second, uint8_t **out, size_t *out_len);
/* the same for a clip that already sits in a batch (after flo_batch_upload): libflo::encode* analyse and encode the SAME
 * samples (lib.rs:105-116), so the free functions upload once, analyse on the device copy, then encode it:
 *   flo_batch_create(1 clip); flo_batch_upload; flo_batch_analysis_metadata -> META (merge with the caller's);
 *   flo_batch_set_bit_depth (lossless); flo_batch_encode; flo_batch_sync; flo_batch_fetch(meta) */
int flo_batch_analysis_metadata(flo_batch *b, size_t clip, uint32_t peaks_per_second, uint8_t **out, size_t *out_len);
/* the bit depth a lossless batch's files declare (16 unless set; flo_encode_lossless's argument) */
int flo_batch_set_bit_depth(flo_batch *b, uint8_t bit_depth);

/* ---- streaming encoder: StreamingEncoder of libflo/src/streaming/encoder.rs:6-257 -----------------------------
 * Samples are pushed (interleaved f32); every complete one-second frame is encoded losslessly - all frames a push
 * completes in ONE device batch - and queued; frames are pulled one by one, or assembled into a complete .flo file.
 * Frame bytes are the reference's encode_frame_data / serialize_channel bytes (encoder.rs:215-257).
 *   flo_stream_create  <- StreamingEncoder::new(sr, ch, bits).with_compression(level)      encoder.rs:33-56
 *   flo_stream_push    <- push_samples                                                     :71-75
 *   flo_stream_next_frame <- next_frame (1: a frame came out, data malloc'ed / flo_free; 0: none; -1: error)   :78-85
 *   flo_stream_flush   <- flush: the buffered remainder as one partial frame, returned, not queued   :88-110
 *   flo_stream_finalize <- finalize: header + TOC + DATA (+ META) of the frames not pulled yet       :113-185 */
typedef struct flo_stream flo_stream;
int flo_stream_create(flo_ctx *ctx, uint32_t sample_rate, uint8_t channels, uint8_t bit_depth, uint8_t level,
                      flo_stream **out);
void flo_stream_destroy(flo_stream *s);
int flo_stream_push(flo_stream *s, const float *samples, size_t n_interleaved);
size_t flo_stream_pending_samples(const flo_stream *s);   /* sample-frames waiting for a full second */
size_t flo_stream_pending_frames(const flo_stream *s);
int flo_stream_next_frame(flo_stream *s, uint32_t *index, uint32_t *timestamp_ms, uint32_t *samples, uint8_t **data,
                          size_t *len);
int flo_stream_flush(flo_stream *s, uint32_t *index, uint32_t *timestamp_ms, uint32_t *samples, uint8_t **data,
                     size_t *len);
int flo_stream_finalize(flo_stream *s, const uint8_t *meta, size_t meta_len, uint8_t **out, size_t *out_len);

/* ---- measurement hooks ----------------------------------------------------------------------------- */
/* When enabled, every launch of a named kernel on the ctx stream is bracketed by hipEvents on that stream. */
int flo_ctx_profile_enable(flo_ctx *ctx, int on);
/* sum and count of bracketed launches of `kernel` since the last reset (call after a sync) */
int flo_ctx_profile_query(flo_ctx *ctx, const char *kernel, double *total_ms, uint64_t *launches);
int flo_ctx_profile_reset(flo_ctx *ctx);
/* test hook: force the lossy kernel form (0 auto, 1 .. 5 as in flo_batch_encode) */
int flo_ctx_force_path(flo_ctx *ctx, int which);
/* stream handle (hipStream_t) of the context, for callers that enqueue their own work around the encode */
void *flo_ctx_stream(flo_ctx *ctx);

/* ---- stage-level entry points (parity tests call the kernels through these) -------------------------- */
/* forward MDCT of n_frames windows of 2048 samples (mono, back to back) -> n_frames*1024 coefficients.
 * Replaces Mdct::forward(samples, BlockSize::Long) with the Vorbis window — lossy/mdct.rs:337-347,166-226 */
int flo_mdct_forward(flo_ctx *ctx, const float *frames, size_t n_frames, float *coeffs);
/* per-frame intermediates of one clip's lossy encode, device path: arrays [hops][ch][1024] / [hops][ch][25];
 * any pointer may be NULL. Replaces TransformEncoder::encode_frame — lossy/encoder.rs:63-106 */
int flo_lossy_analyze(flo_ctx *ctx, const float *pcm, size_t n_interleaved, uint32_t sample_rate,
                      uint8_t channels, float quality, float *coeffs, int16_t *quantized, uint16_t *sf_words,
                      size_t *num_hops);
/* quantise + serialise given MDCT coefficients (device quantiser fed caller-supplied spectra):
 * coeffs [hops][ch][1024] -> quantized [hops][ch][1024], sf_words [hops][ch][25]. Replaces
 * PsychoacousticModel::calculate_smr + TransformEncoder::quantize_coefficients
 * (lossy/psychoacoustic.rs:151-235, lossy/encoder.rs:109-154). exact = 0 runs the quantiser exactly as every encode
 * entry point does (keep test in the amplitude domain); exact = 1 additionally re-decides coefficients within 1e-5
 * of the threshold with the reference's own dB-domain f32 expression (a test yardstick, never used by an encode). */
int flo_lossy_quantize(flo_ctx *ctx, const float *coeffs, size_t num_hops, uint32_t sample_rate, uint8_t channels,
                       float quality, int exact, int16_t *quantized, uint16_t *sf_words);
/* TransformEncoder::quantize_coefficients as the reference exposes it (lossy/encoder.rs:109-154): n_vec vectors of 1024
 * coefficients with the caller's own signal-to-mask ratios -> i16 (kept iff smr > the quality's threshold, c * scale factor
 * rounded half away from zero) and the 25 band scale factors (30000 / band maximum, 1.0 for a silent band) per vector.
 * smr = NULL: scale factors only (quantized is not written). */
int flo_lossy_quantize_smr(flo_ctx *ctx, const float *coeffs, const float *smr, size_t n_vec, uint32_t sample_rate, float quality,
                           int16_t *quantized, float *scale_factors);
/* serialize_sparse on device: n_vec vectors of 1024 i16 -> bytes; out_off[n_vec+1] prefix offsets.
 * Replaces lossy/encoder.rs:284-314. form = 0: the packer as the encoder runs it (item form up to 128 non-zeros, behind it
 * the block form, behind that the general form for the dense vectors it declines); form = 1: the general form for every
 * vector; form = 2: block form, then general (tests compare the three). */
int flo_sparse_pack(flo_ctx *ctx, const int16_t *q, size_t n_vec, int form, uint8_t *out, size_t out_cap,
                    uint32_t *out_off);

#ifdef __cplusplus
}
#endif
#endif
