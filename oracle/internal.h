/* internal.h — shared types of the oracle (TEST INFRASTRUCTURE, see flo_oracle.h). */
#ifndef FLO_ORACLE_INTERNAL_H
#define FLO_ORACLE_INTERNAL_H

#include "flo_oracle.h"
#include <stdlib.h>
#include <string.h>

/* core/types.rs:6-13 */
#define FLO_HEADER_SIZE 66u
#define FLO_VERSION_MAJOR 1
#define FLO_VERSION_MINOR 2

/* core/types.rs:28-45 */
enum { FT_SILENCE = 0, FT_TRANSFORM = 253, FT_RAW = 254, FT_RESERVED = 255 };
/* core/types.rs:114-118 */
enum { RE_RICE = 0, RE_GOLOMB = 1, RE_RAW = 2 };

#define NUM_BARK_BANDS 25

/* core/types.rs:183-189 */
typedef struct {
    int32_t coeffs[12];
    size_t n_coeffs;
    uint8_t shift_bits;
    uint8_t residual_encoding;
    uint8_t rice_parameter;
    flo_buf residuals;
} o_channel;

/* core/types.rs:225-230 */
typedef struct {
    uint8_t frame_type;
    uint32_t frame_samples;
    uint8_t flags;
    o_channel *channels;
    size_t n_channels;
} o_frame;

typedef struct {
    uint32_t frame_index;
    uint64_t byte_offset;
    uint32_t frame_size;
    uint32_t timestamp_ms;
} o_toc_entry;

typedef struct {
    flo_o_info hdr;
    o_toc_entry *toc;
    size_t n_toc;
    o_frame *frames;
    size_t n_frames;
    flo_buf metadata;
} o_file;

/* buf helpers */
void buf_init(flo_buf *b);
void buf_reserve(flo_buf *b, size_t extra);
void buf_push(flo_buf *b, uint8_t v);
void buf_extend(flo_buf *b, const void *p, size_t n);
void buf_u16le(flo_buf *b, uint16_t v);
void buf_u32le(flo_buf *b, uint32_t v);
void buf_u64le(flo_buf *b, uint64_t v);

/* frames */
int ft_is_alpc(uint8_t t);                 /* types.rs:59-61 */
uint8_t ft_from_order(size_t order);       /* types.rs:69-85 */
void frame_free(o_frame *f);
void file_free(o_file *f);
size_t frame_byte_size(const o_frame *f);  /* types.rs:243-267 */

/* container */
void writer_write_ex(uint32_t sample_rate, uint8_t channels, uint8_t bit_depth, uint8_t level, int lossy,
                     uint8_t lossy_quality, const o_frame *frames, size_t n_frames, const uint8_t *meta,
                     size_t meta_len, flo_buf *out);              /* writer.rs:39-100 */
int reader_read(const uint8_t *data, size_t len, o_file *out);    /* reader.rs:16-52; 0 ok */
void set_error(const char *msg);

/* lossless */
void lossless_encode_frames(const float *samples, size_t n, uint32_t sample_rate, uint8_t channels,
                            uint8_t level, o_frame **frames, size_t *n_frames);
int lossless_decode_file_i32(const o_file *f, int32_t **out, size_t *n_interleaved);

/* lossy */
int lossy_encode_frames(const float *samples, size_t n, uint32_t sample_rate, uint8_t channels, float quality,
                        o_frame **frames, size_t *n_frames);
int lossy_decode_file(const o_file *f, float **out, size_t *n_interleaved);

#endif
