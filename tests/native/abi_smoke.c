/*
 * abi_smoke.c — a caller of libflo_hip.so written in plain C: no Python, ctypes or numpy between the program and
 * the C ABI of include/flo_hip.h. It does what a libflo maintainer's Rust shim would do through `extern "C"`:
 *
 *   Encoder::new(44100, 2, 16).encode(&silence, &[])                 -> flo_encode_lossless   (lossless/encoder.rs:17-45)
 *   LossyEncoder::new(44100, 2, 0.6).encode_to_flo(&silence, &[])     -> flo_encode_lossy      (lossy/encoder.rs:167-239)
 *   libflo::decode(&file)                                              -> flo_decode            (lib.rs:296-352)
 *
 * on the input of BASELINE configs[0] (Examples/audio.wav: one second of stereo silence) and compares the bytes with
 * the files the reference itself wrote from that input (tests/golden/examples/audio_lossless.flo, audio_lossy.flo).
 *
 *   usage: abi_smoke <audio_lossless.flo> <audio_lossy.flo>
 *   exit:  0 ok | 1 mismatch | 2 usage / io | 3 no usable device (the library has no CPU path)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "flo_hip.h"

static unsigned char *slurp(const char *path, size_t *n) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char *p = (unsigned char *)malloc(len > 0 ? (size_t)len : 1);
    if (p && fread(p, 1, (size_t)len, f) != (size_t)len) {
        free(p);
        p = NULL;
    }
    fclose(f);
    *n = (size_t)len;
    return p;
}

static unsigned long long rd_u64(const unsigned char *p) {
    unsigned long long v = 0;
    for (int i = 7; i >= 0; i--) v = (v << 8) | p[i];
    return v;
}

#define FAIL(...)                     \
    do {                              \
        fprintf(stderr, __VA_ARGS__); \
        fprintf(stderr, "\n");        \
        return 1;                     \
    } while (0)

int main(int argc, char **argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s audio_lossless.flo audio_lossy.flo\n", argv[0]);
        return 2;
    }
    size_t n_ll = 0, n_ly = 0;
    unsigned char *ref_ll = slurp(argv[1], &n_ll), *ref_ly = slurp(argv[2], &n_ly);
    if (!ref_ll || !ref_ly || n_ll < 70 || n_ly < 70) {
        fprintf(stderr, "cannot read the fixture files\n");
        return 2;
    }
    flo_ctx *ctx = NULL;
    if (flo_ctx_create(0, &ctx) != FLO_OK) {
        fprintf(stderr, "flo_ctx_create: %s\n", flo_last_create_error());
        return 3;
    }
    const size_t n = 44100 * 2;
    float *pcm = (float *)calloc(n, sizeof(float));

    /* ---- lossless: header + TOC + DATA identical to the reference's file (its META chunk is the CLI's) ---- */
    uint8_t *out = NULL;
    size_t out_len = 0;
    if (flo_encode_lossless(ctx, pcm, n, 44100, 2, 16, 5, NULL, 0, &out, &out_len) != FLO_OK)
        FAIL("flo_encode_lossless: %s", flo_last_error(ctx));
    const unsigned long long ref_meta = rd_u64(ref_ll + 62);
    if (out_len != n_ll - ref_meta) FAIL("lossless length %zu, reference %zu + %llu META", out_len, n_ll - (size_t)ref_meta, ref_meta);
    if (memcmp(out, ref_ll, 62) != 0) FAIL("lossless header differs from the reference file");
    if (rd_u64(out + 62) != 0) FAIL("meta_size of a file without META must be 0");
    if (memcmp(out + 70, ref_ll + 70, out_len - 70) != 0) FAIL("lossless TOC/DATA differ from the reference file");
    float *dec = NULL;
    size_t dec_n = 0;
    uint32_t sr = 0;
    uint8_t ch = 0;
    if (flo_decode(ctx, out, out_len, &dec, &dec_n, &sr, &ch) != FLO_OK) FAIL("flo_decode(lossless): %s", flo_last_error(ctx));
    if (dec_n != n || sr != 44100 || ch != 2) FAIL("lossless decode geometry %zu %u %u", dec_n, sr, (unsigned)ch);
    for (size_t i = 0; i < dec_n; i++)
        if (dec[i] != 0.0f) FAIL("lossless decode of silence is not silence at %zu", i);
    flo_free(dec);
    flo_free(out);

    /* ---- lossy (the CLI's "high" = 0.6): DATA chunk identical; the fixture's TOC timestamps are from an older writer ---- */
    if (flo_encode_lossy(ctx, pcm, n, 44100, 2, 0.6f, NULL, 0, &out, &out_len) != FLO_OK)
        FAIL("flo_encode_lossy: %s", flo_last_error(ctx));
    const unsigned long long toc = rd_u64(out + 38), data = rd_u64(out + 46);
    const unsigned long long rtoc = rd_u64(ref_ly + 38), rdata = rd_u64(ref_ly + 46);
    if (toc != rtoc || data != rdata) FAIL("lossy chunk sizes %llu/%llu, reference %llu/%llu", toc, data, rtoc, rdata);
    if (memcmp(out, ref_ly, 38) != 0) FAIL("lossy header (magic .. CRC32 of DATA) differs from the reference file");
    if (memcmp(out + 70 + toc, ref_ly + 70 + rtoc, data) != 0) FAIL("lossy DATA differs from the reference file");
    if (flo_decode(ctx, out, out_len, &dec, &dec_n, &sr, &ch) != FLO_OK) FAIL("flo_decode(lossy): %s", flo_last_error(ctx));
    if (dec_n != (size_t)44 * 1024 * 2 || sr != 44100 || ch != 2) FAIL("lossy decode geometry %zu", dec_n);
    for (size_t i = 0; i < dec_n; i++)
        if (dec[i] != 0.0f) FAIL("lossy decode of silence is not silence at %zu", i);
    flo_free(dec);
    /* the reference-made file decodes too */
    if (flo_decode(ctx, ref_ly, n_ly, &dec, &dec_n, &sr, &ch) != FLO_OK) FAIL("flo_decode(reference lossy): %s", flo_last_error(ctx));
    if (dec_n != (size_t)44 * 1024 * 2) FAIL("reference lossy decode geometry %zu", dec_n);
    flo_free(dec);
    flo_free(out);

    /* ---- error behaviour: Err(String) becomes a code + message ---- */
    if (flo_decode(ctx, (const uint8_t *)"RIFFxxxx", 8, &dec, &dec_n, NULL, NULL) != FLO_ERR_FORMAT) FAIL("bad magic must be FLO_ERR_FORMAT");
    if (!strlen(flo_last_error(ctx))) FAIL("a failing call leaves a message");

    free(pcm);
    free(ref_ll);
    free(ref_ly);
    flo_ctx_destroy(ctx);
    printf("abi_smoke ok\n");
    return 0;
}
