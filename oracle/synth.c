/* synth.c — host instance of the integer-exact synthetic PCM generator (include/flo_synth.h), so that tests and
 * the CPU baseline feed the oracle exactly the samples the device kernel generates. TEST INFRASTRUCTURE. */
#include "../include/flo_synth.h"
#include <stddef.h>

void flo_o_synth_fill(float *pcm, size_t n_sample_frames, unsigned channels, uint32_t seed, uint64_t clip_id) {
    for (size_t i = 0; i < n_sample_frames; i++)
        for (unsigned c = 0; c < channels; c++) pcm[i * channels + c] = flo_synth_sample(seed, clip_id, c, i);
}
