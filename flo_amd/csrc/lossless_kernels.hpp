// lossless_kernels.hpp — device side of the lossless (ALPC + Rice) path; see lossless_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace flo {

struct LosslessPlan;

// n_il[i] = interleaved sample count of clip i, clip_off[i] = float offset of the clip inside d_pcm
LosslessPlan *lossless_plan_create(const std::vector<uint64_t> &n_il, const std::vector<uint64_t> &clip_off,
                                   uint32_t sample_rate, uint8_t channels, uint8_t level, const float *d_pcm,
                                   std::string &err);
void lossless_plan_destroy(LosslessPlan *p);
// enqueue all kernels of one encode on `s`; profile != 0 brackets them with events
int lossless_encode_launch(LosslessPlan *p, hipStream_t s, int profile, std::string &err);
// after the stream is idle: bring sizes/metadata to the host
int lossless_collect(LosslessPlan *p, std::string &err);
uint64_t lossless_total_bytes(const LosslessPlan *p);
int lossless_device_streams(LosslessPlan *p, const uint8_t **base, const uint64_t **offsets, const uint64_t **sizes);
int lossless_device_files(LosslessPlan *p, const uint8_t **base, const uint64_t **offsets, const uint64_t **sizes);
// What a reader of the finished files would find (reader.rs:150-256 applied to what ll_layout / ll_pack wrote), from
// the encoder's own frame and channel records: the device decode of a batch needs no parse of its files.
struct LosslessWrapperInfo {
    uint64_t off;            // residual / raw payload, byte offset in the batch's output buffer
    uint32_t len;
    uint8_t n_coeffs, shift_bits, rice_k;
    int32_t coeffs[12];
};
struct LosslessFrameInfo {
    uint32_t clip, samples, flags, first_wrapper, n_wrappers;
};
int lossless_describe(LosslessPlan *p, std::vector<LosslessFrameInfo> &frames, std::vector<LosslessWrapperInfo> &wrappers,
                      const uint8_t **base, std::string &err);
int lossless_fetch(LosslessPlan *p, size_t clip, uint8_t bit_depth, const uint8_t *meta, size_t meta_len, uint8_t **out,
                   size_t *out_len, std::string &err);

}  // namespace flo
