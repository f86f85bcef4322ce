/*
 * flo_synth.h — integer-exact synthetic PCM, one definition for host and device.
 *
 * Sample n of channel h of clip c is a pure function of (seed, c, h, n) built from u32/u64 integer
 * arithmetic only, so the HIP generator kernel and the CPU baseline see bit-identical f32 input:
 *   v = tri(phi1)>>2 + tri(phi2)>>3 + tri(phi3)>>4 + noise,   sample = clamp16(v) / 32768.0f  (exact)
 * with three triangle partials (50 Hz .. 12 kHz at 44.1 kHz, chosen per (clip, channel) by splitmix64),
 * slowly gated in 1/4-second blocks so that frames differ in level, plus a low-level counter-based hash
 * noise (about -66 dBFS) standing in for dither.
 */
#ifndef FLO_SYNTH_H
#define FLO_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define FLO_HD __host__ __device__ static inline
#else
#define FLO_HD static inline
#endif

#define FLO_SYNTH_DEFAULT_SEED 0xF10A0D10u

FLO_HD uint64_t flo_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

/* triangle wave in [-16384, 16383] from a u32 phase */
FLO_HD int32_t flo_tri(uint32_t phase) {
    uint32_t t = phase >> 16; /* 0..65535 */
    int32_t up = (int32_t)(t < 32768u ? t : 65535u - t);
    return up - 16384;
}

/* phase increment for a frequency drawn log-ish in [50 Hz, 12 kHz] at 44.1 kHz: inc = f * 2^32 / 44100 */
FLO_HD uint32_t flo_synth_inc(uint64_t r) {
    /* 8 octaves above 50 Hz: 50 * 2^(x/2^16 * 7.9), done in integers: base inc for 50 Hz is 4869588 */
    uint32_t oct = (uint32_t)(r & 7u);                 /* 0..7 */
    uint32_t frac = (uint32_t)((r >> 3) & 0xFFFFu);    /* 0..65535 */
    uint64_t base = 4869588ull << oct;                  /* 50 Hz * 2^oct */
    return (uint32_t)(base + ((base * frac) >> 16));    /* up to (almost) the next octave; max ~12.8 kHz */
}

FLO_HD float flo_synth_sample(uint32_t seed, uint64_t clip, uint32_t ch, uint64_t n) {
    uint64_t k0 = flo_splitmix64(((uint64_t)seed << 32) ^ (clip * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)ch * 0xD1B54A32D192ED03ull));
    uint64_t k1 = flo_splitmix64(k0);
    uint64_t k2 = flo_splitmix64(k1);
    uint32_t inc1 = flo_synth_inc(k0), inc2 = flo_synth_inc(k0 >> 20), inc3 = flo_synth_inc(k1);
    uint32_t n32 = (uint32_t)n;
    uint32_t p1 = (uint32_t)(k1 >> 32) + inc1 * n32;
    uint32_t p2 = (uint32_t)(k2) + inc2 * n32;
    uint32_t p3 = (uint32_t)(k2 >> 32) + inc3 * n32;
    /* level gates per 11025-sample block: each partial is attenuated by 0..3 bits */
    uint64_t g = flo_splitmix64(k0 ^ (n / 11025ull));
    int32_t v = (flo_tri(p1) >> (2 + (int)(g & 3u))) + (flo_tri(p2) >> (3 + (int)((g >> 2) & 3u))) +
                (flo_tri(p3) >> (4 + (int)((g >> 4) & 3u)));
    /* counter-based noise, 5 bits: about +-16 LSB */
    uint64_t h = flo_splitmix64(k2 ^ (n * 0x9E3779B97F4A7C15ull));
    v += (int32_t)(h & 31u) - 16;
    if (v > 32767) v = 32767;
    if (v < -32768) v = -32768;
    return (float)v / 32768.0f;
}

#endif
