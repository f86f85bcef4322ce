// tables.cpp — see tables.hpp. Pure host code; built with -ffp-contract=off so every f32 expression rounds
// step by step exactly as the Rust reference does.
#include "tables.hpp"

#include "pack_rows.h"

#include <cmath>
#include <cstring>

namespace flo {

static const int kSlotCapHost = 96;  // == kSlotCap in lossy_device.hpp

static const float kPi = 3.14159265358979323846f;  // std::f32::consts::PI

static const float kBarkEdges[26] = {0.0f,    100.0f,  200.0f,  300.0f,  400.0f,  510.0f,   630.0f,
                                     770.0f,  920.0f,  1080.0f, 1270.0f, 1480.0f, 1720.0f,  2000.0f,
                                     2320.0f, 2700.0f, 3150.0f, 3700.0f, 4400.0f, 5300.0f,  6400.0f,
                                     7700.0f, 9500.0f, 12000.0f, 15500.0f, 20500.0f};  // psychoacoustic.rs:5-9

float ref_ath(float freq) {  // psychoacoustic.rs:90-104
    if (!(freq >= 20.0f && freq <= 20000.0f)) return 96.0f;
    float f_khz = freq / 1000.0f;
    float term1 = 3.64f * powf(f_khz, -0.8f);
    float d = f_khz - 3.3f;
    float term2 = 6.5f * expf(-0.6f * (d * d));
    float f2 = f_khz * f_khz;
    float term3 = 0.001f * (f2 * f2);
    float v = term1 - term2 + term3;
    if (v < -10.0f) v = -10.0f;
    if (v > 96.0f) v = 96.0f;
    return v;
}

int ref_freq_to_bark_band(float freq) {  // psychoacoustic.rs:114-121
    for (int i = 1; i < 26; i++)
        if (freq < kBarkEdges[i]) return i - 1;
    return kNumBands - 1;
}

float ref_smr_threshold(float quality) {  // encoder.rs:130-136
    if (quality >= 0.99f) return -100.0f;
    float t = fmaxf(1.0f - quality, 0.001f);
    return -60.0f * (1.0f - powf(t, 0.5f));
}

void build_lossy_tables(uint32_t sample_rate, float quality, LossyTablesHost &t) {
    if (quality < 0.0f) quality = 0.0f;  // TransformEncoder::new clamps (encoder.rs:50)
    if (quality > 1.0f) quality = 1.0f;
    t.sample_rate = sample_rate;
    t.quality = quality;
    t.smr_threshold = ref_smr_threshold(quality);
    t.q_transparent = quality >= 0.99f;
    float ql = roundf(quality * 4.0f);  // encoder.rs:235
    t.q_level = (uint8_t)(ql > 4.0f ? 4 : (ql < 0.0f ? 0 : (int)ql));

    // mdct.rs:106-113 (Vorbis window), :81-86 (twiddle)
    t.window.resize(kN);
    for (int i = 0; i < kN; i++) {
        float x = sinf(kPi * ((float)i + 0.5f) / (float)kN);
        t.window[i] = sinf(kPi / 2.0f * x * x);
    }
    t.twiddle.resize(512 * 2);
    for (int k = 0; k < 512; k++) {
        float theta = kPi / 1024.0f * ((float)k + 0.125f);
        t.twiddle[2 * k] = cosf(theta);
        t.twiddle[2 * k + 1] = sinf(theta);
    }
    // FFT-512 inter-stage twiddles (own): stage 1 -> W512^(lane*k), stage 2 -> W64^(n*k)
    t.t1.resize(7 * 64 * 2);
    for (int k = 1; k < 8; k++)
        for (int l = 0; l < 64; l++) {
            double a = -2.0 * M_PI * (double)(l * k) / 512.0;
            t.t1[((k - 1) * 64 + l) * 2] = (float)cos(a);
            t.t1[((k - 1) * 64 + l) * 2 + 1] = (float)sin(a);
        }
    t.t2.resize(7 * 8 * 2);
    for (int k = 1; k < 8; k++)
        for (int n = 0; n < 8; n++) {
            double a = -2.0 * M_PI * (double)(n * k) / 64.0;
            t.t2[((k - 1) * 8 + n) * 2] = (float)cos(a);
            t.t2[((k - 1) * 8 + n) * 2 + 1] = (float)sin(a);
        }

    // psychoacoustic.rs:35-71
    float freq_resolution = (float)sample_rate / (float)kN;
    t.ath_db.resize(kHop);
    t.ath_lin.resize(kHop);
    t.band.resize(kHop);
    t.band_count.assign(kNumBands, 0.0f);
    for (int k = 0; k < kHop; k++) {
        float freq = ((float)k + 0.5f) * freq_resolution;
        t.ath_db[k] = ref_ath(freq);
        int b = ref_freq_to_bark_band(freq);
        t.band[k] = (uint8_t)b;
        t.band_count[b] += 1.0f;
        // keep iff 20 log10|c| - (max(s, ath) - 10) > smr_thr  <=>  |c| > 10^((smr_thr + fl(max(s,ath) - 10)) / 20);
        // the ATH half of that max is a constant of (sample_rate, quality):
        float thr = t.ath_db[k] - 10.0f;
        t.ath_lin[k] = (float)pow(10.0, ((double)t.smr_threshold + (double)thr) / 20.0);
    }
    // psychoacoustic.rs:125-147,184: 10*log10(spreading[j][i]) depends only on delta = i - j (0 for i < j)
    t.s10d.resize(kNumBands);
    for (int d = 0; d < kNumBands; d++) {
        float delta_bark = (float)d;
        float spread = -25.0f * delta_bark;
        float v = powf(10.0f, spread / 10.0f);
        if (!(v < 1.0f)) v = 1.0f;
        t.s10d[d] = 10.0f * log10f(v);
    }

    // per-lane constant pack [row][lane][4] (layout documented in lossy_device.hpp)
    t.pack.assign((size_t)kPackRows * 64 * 4, 0.0f);
    t.pack_ext.assign(2 * 64 * 4, 0.0f);
    auto P = [&](int row, int lane, int i) -> float & { return t.pack[((size_t)row * 64 + lane) * 4 + i]; };
    // row kRowS10 is not per lane: its first 24 floats are s10d[1..24], read uniformly by every lane
    for (int d = 1; d < kNumBands; d++) P(kRowS10, (d - 1) / 4, (d - 1) % 4) = t.s10d[d];
    for (int lane = 0; lane < 64; lane++) {
        for (int r = 0; r < 8; r++) {
            int eo, oo;
            if (r < 4) {
                int i = lane + 64 * r;
                eo = 512 + 2 * i;
                oo = 511 - 2 * i;
            } else {
                int i = lane + 64 * (r - 4);
                eo = 2 * i;
                oo = 1023 - 2 * i;
            }
            P(kRowWin + r, lane, 0) = t.window[eo];
            P(kRowWin + r, lane, 1) = t.window[oo];
            P(kRowWin + r, lane, 2) = t.window[1024 + eo];
            P(kRowWin + r, lane, 3) = t.window[1024 + oo];
            int m = lane + 64 * r;
            P(kRowTw + (r >> 1), lane, 2 * (r & 1)) = t.twiddle[2 * m];
            P(kRowTw + (r >> 1), lane, 2 * (r & 1) + 1) = t.twiddle[2 * m + 1];
        }
        for (int k = 1; k < 8; k++) {
            int idx = 2 * (k - 1);  // float index inside the 16-float group
            P(kRowF1 + idx / 4, lane, idx % 4) = t.t1[((k - 1) * 64 + lane) * 2];
            P(kRowF1 + idx / 4, lane, idx % 4 + 1) = t.t1[((k - 1) * 64 + lane) * 2 + 1];
            P(kRowF2 + idx / 4, lane, idx % 4) = t.t2[((k - 1) * 8 + (lane & 7)) * 2];
            P(kRowF2 + idx / 4, lane, idx % 4 + 1) = t.t2[((k - 1) * 8 + (lane & 7)) * 2 + 1];
        }
        for (int e = 0; e < 16; e++) P(kRowAth + e / 4, lane, e % 4) = t.ath_lin[16 * lane + e];
        // the same thresholds in the packer wave's natural layout: entry 2 k + j = position 128 k + 2 lane + j
        for (int e = 0; e < 16; e++) P(kRowAthN + e / 4, lane, e % 4) = t.ath_lin[128 * (e >> 1) + 2 * lane + (e & 1)];
    }

    // bands present in each block of 128 positions (the packer wave skips blocks none of whose bands can keep anything)
    for (int k = 0; k < 8; k++) {
        uint32_t m = 0;
        for (int i = 0; i < 128; i++) m |= 1u << t.band[128 * k + i];
        memcpy(&t.pack[((size_t)kRowBlk * 64) * 4 + k], &m, 4);
    }
    // segments of the contiguous layout (lane j owns k in [16j, 16j+16))
    t.lane_bnd.assign(64, 0);
    t.lane_slot0.assign(64, 0);
    std::vector<int> slot_band;
    for (int j = 0; j < 64; j++) {
        t.lane_slot0[j] = (uint32_t)slot_band.size();
        for (int e = 0; e < 16; e++) {
            int k = 16 * j + e;
            bool end = (e == 15) || (t.band[k + 1] != t.band[k]);
            if (end) {
                t.lane_bnd[j] |= 1u << e;
                slot_band.push_back(t.band[k]);
            }
        }
    }
    t.n_slots = (int)slot_band.size();
    t.dirty = 0;
    for (int j = 0; j < 64; j++) t.dirty |= t.lane_bnd[j];
    t.band_slot0.assign(kNumBands + 1, 0);
    // slots are generated in k order, so each band's slots are consecutive
    int s = 0;
    for (int b = 0; b < kNumBands; b++) {
        t.band_slot0[b] = (uint32_t)s;
        while (s < t.n_slots && slot_band[s] == b) s++;
    }
    t.band_slot0[kNumBands] = (uint32_t)s;
    t.max_band_slots = 0;
    for (int b = 0; b < kNumBands; b++) {
        int n = (int)(t.band_slot0[b + 1] - t.band_slot0[b]);
        if (n > t.max_band_slots) t.max_band_slots = n;
    }
    // per-lane band bookkeeping, stored as raw 32-bit patterns
    auto PU = [&](int row, int lane, int i, uint32_t v) { memcpy(&t.pack[((size_t)row * 64 + lane) * 4 + i], &v, 4); };
    for (int lane = 0; lane < 64; lane++) {
        const int bl = lane & 31;
        const int b = bl < 25 ? bl : 24;
        const float cnt = t.band_count[b];
        const float rcount = cnt > 0.f ? 1.0f / cnt : 0.f;
        uint32_t rc;
        memcpy(&rc, &rcount, 4);
        // band b is reduced by lane b (even slots) and lane 32 + b (odd slots)
        const uint32_t bs0 = t.band_slot0[b] + (uint32_t)(lane >> 5);
        const uint32_t bs1 = bl < 25 ? t.band_slot0[b + 1] : 0u;
        // kRowBo: byte offsets into bandv (float2 per band); kRowKeep: restart multipliers; kRowDst: slot destinations
        {
            uint32_t slot = t.lane_slot0[lane];
            for (int e = 0; e < 16; e++) {
                const bool end = (t.lane_bnd[lane] >> e) & 1u;
                PU(kRowBo + e / 4, lane, e % 4, (uint32_t)t.band[16 * lane + e] * 8u);
                P(kRowKeep + e / 4, lane, e % 4) = end ? 0.0f : 1.0f;
                PU(kRowDst + e / 4, lane, e % 4, (end ? slot : (uint32_t)(kSlotCapHost + lane)) * 8u);
                if (end) slot++;
            }
        }
        // kRowLst (first 12) / kRowLstCold (next 12): the slots this band lane adds (every other slot of its band), padded
        // with the zero slot
        for (int u = 0; u < 24; u++) {
            const uint32_t sidx = bs0 + 2u * (uint32_t)u;
            PU((u < 12 ? kRowLst : kRowLstCold - 3) + u / 4, lane, u % 4, (sidx < bs1 ? sidx : (uint32_t)(kSlotCapHost + 64)) * 8u);
        }
        // entries 24..31 (a band can span all 64 lanes: 32 slots per reducer lane) live outside the LDS pack: only the
        // widest band of sample rates from 128 kHz up reaches them, and the kernels read them from global memory
        for (int u = 24; u < 32; u++) {
            const uint32_t sidx = bs0 + 2u * (uint32_t)u;
            const uint32_t v = (sidx < bs1 ? sidx : (uint32_t)(kSlotCapHost + 64)) * 8u;
            memcpy(&t.pack_ext[((size_t)((u - 24) / 4) * 64 + lane) * 4 + u % 4], &v, 4);
        }
        for (int e = 0; e < 16; e++) PU(kRowBoN + e / 4, lane, e % 4, (uint32_t)t.band[128 * (e >> 1) + 2 * lane + (e & 1)] * 16u);
        PU(kRowLane, lane, 0, t.lane_bnd[lane]);
        PU(kRowLane, lane, 1, t.lane_slot0[lane]);
        PU(kRowLane, lane, 2, rc);
        PU(kRowLane, lane, 3, bs0 | (bs1 << 16));
    }
}

}  // namespace flo
