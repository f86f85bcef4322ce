// tables.hpp — host-side constant tables of the lossy path, uploaded once per (sample_rate, quality).
//
// window / twiddle / ath / bark_band / spreading follow the reference's f32 formulas so the device sees
// bit-identical constants to what libflo builds at encoder construction:
//   window   lossy/mdct.rs:106-113      twiddle  lossy/mdct.rs:81-86
//   ath      lossy/psychoacoustic.rs:90-104   band map :114-121   spreading :125-147
// The FFT-stage twiddles (t1, t2) and the band-segment bookkeeping are this implementation's own.
#pragma once
#include <cstdint>
#include <vector>

namespace flo {

constexpr int kNumBands = 25;
constexpr int kN = 2048;       // long block
constexpr int kHop = 1024;     // coefficients per frame-channel
constexpr int kMaxSlots = 96;  // per-lane band segments: 64 lanes + at most 24 interior boundaries (+pad)

struct LossyTablesHost {
    uint32_t sample_rate = 0;
    float quality = 0.f;        // clamped to [0,1]
    float smr_threshold = 0.f;  // lossy/encoder.rs:130-136
    int q_transparent = 0;      // quality >= 0.99
    uint8_t q_level = 0;        // min(round(q*4),4) for the header

    std::vector<float> window;      // [2048]
    std::vector<float> twiddle;     // [512][2] (cos, sin)
    std::vector<float> t1;          // [7][64][2]  e^{-2 pi i lane*k/512}, k = 1..7
    std::vector<float> t2;          // [7][8][2]   e^{-2 pi i n*k/64},     k = 1..7
    std::vector<float> ath_db;      // [1024]
    std::vector<float> ath_lin;     // [1024] 10^((smr_thr + fl(ath-10))/20): amplitude below which ATH alone drops
    std::vector<float> pack;        // [kPackRows][64][4] per-lane constant pack (pack_rows.h)
    std::vector<float> pack_ext;    // [2][64][4] slot-list groups 6 and 7 (bands of more than 48 lane segments: 128 kHz and up)
    std::vector<uint8_t> band;      // [1024]
    std::vector<float> band_count;  // [25]
    std::vector<float> s10d;        // [25] 10*log10f(spreading) as a function of (i - j) >= 0
    // band-segment bookkeeping for the "16 contiguous coefficients per lane" layout
    std::vector<uint32_t> lane_bnd;    // [64] bit e set: a band segment ends after element e (bit 15 always)
    std::vector<uint32_t> lane_slot0;  // [64] index of the lane's first segment slot
    std::vector<uint32_t> band_slot0;  // [26] slots of band b are [band_slot0[b], band_slot0[b+1])
    int max_band_slots = 0;            // max over bands of slot count
    uint32_t dirty = 0;                // OR of lane_bnd over the lanes: element positions at which some lane closes a segment
    int n_slots = 0;
};

void build_lossy_tables(uint32_t sample_rate, float quality, LossyTablesHost &t);

// reference scalar helpers (f32, same expression order as the Rust sources)
float ref_ath(float freq);
int ref_freq_to_bark_band(float freq);
float ref_smr_threshold(float quality);

}  // namespace flo
