/* api.c — oracle decode dispatch, restating libflo/src/lib.rs:296-352 (TEST INFRASTRUCTURE). */
#include "internal.h"

/* lib.rs:296-315 decode + lossless/decoder.rs:49-71 (i32 -> f32) */
int flo_o_decode(const uint8_t *flo, size_t len, float **pcm, size_t *n_interleaved, uint32_t *sample_rate,
                 uint8_t *channels) {
    o_file f;
    if (reader_read(flo, len, &f) != 0) return -1;
    if (sample_rate) *sample_rate = f.hdr.sample_rate;
    if (channels) *channels = f.hdr.channels;
    int is_transform = 0;
    for (size_t i = 0; i < f.n_frames; i++)
        if (f.frames[i].frame_type == FT_TRANSFORM) is_transform = 1;
    int rc;
    if (is_transform) {
        rc = lossy_decode_file(&f, pcm, n_interleaved);
    } else {
        int32_t *ipcm = NULL;
        size_t n = 0;
        rc = lossless_decode_file_i32(&f, &ipcm, &n);
        if (rc == 0) {
            float *o = (float *)malloc((n ? n : 1) * sizeof(float));
            for (size_t i = 0; i < n; i++) o[i] = flo_o_i32_to_f32(ipcm[i]);
            free(ipcm);
            *pcm = o;
            *n_interleaved = n;
        }
    }
    file_free(&f);
    return rc;
}

int flo_o_decode_lossless_i32(const uint8_t *flo, size_t len, int32_t **pcm, size_t *n_interleaved,
                              uint32_t *sample_rate, uint8_t *channels) {
    o_file f;
    if (reader_read(flo, len, &f) != 0) return -1;
    if (sample_rate) *sample_rate = f.hdr.sample_rate;
    if (channels) *channels = f.hdr.channels;
    int rc = lossless_decode_file_i32(&f, pcm, n_interleaved);
    file_free(&f);
    return rc;
}
