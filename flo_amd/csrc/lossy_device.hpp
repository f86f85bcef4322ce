// lossy_device.hpp — device functions of the lossy encode path, written for gfx950 (wave64, one wave per
// stereo/mono frame stream, no workgroup barriers: every workgroup is exactly one wavefront).
//
// Reference behaviour replaced (files under /root/reference/libflo/src):
//   fold + pre-rotation + FFT-512 + post-rotation .... lossy/mdct.rs:166-226 (MdctTransform::forward)
//   band energies, spreading, temporal masking ....... lossy/psychoacoustic.rs:151-214
//   SMR keep/drop test ................................ lossy/psychoacoustic.rs:218-235, lossy/encoder.rs:129-151
//   scale factors + quantiser ......................... lossy/encoder.rs:109-154
//   scale-factor words, sparse RLE + varint, blob ..... lossy/encoder.rs:243-329
//   frame header ...................................... writer.rs:236-254 (type 253, one channel wrapper)
//
// Data layout in a wave ("lane" = 0..63, CH = 1 or 2 channels processed in lock-step):
//   FFT input/output : lane l, register r  <->  z[l + 64 r]           (stride-64, three radix-8 passes)
//   coefficients     : lane j, register e  <->  c[16 j + e]           (contiguous, after an LDS transpose)
// The 1024-sample overlap between consecutive frames never goes through memory twice: the raw second-half
// samples a lane folds in frame h are exactly the first-half samples the same lane folds in frame h+1, so
// they are carried in registers (see DESIGN.md "fold symmetry").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace flo {

// Per-lane constant pack, laid out [row][lane] as float4 so that one wave-wide load is 1 KiB contiguous:
//   rows 0..7   : window values of fold row r: (w[eo], w[oo], w[1024+eo], w[1024+oo])      (mdct.rs:106-113)
//   rows 8..11  : pre/post-rotation twiddles tw[lane + 64 r], two rows per float4           (mdct.rs:81-86)
//   rows 12..15 : FFT pass-1 twiddles W512^(lane k), k = 1..7 (re,im pairs, last pair unused)
//   rows 16..19 : FFT pass-2 twiddles W64^((lane&7) k), k = 1..7
//   rows 20..23 : ATH amplitude thresholds of coefficients 16 lane .. 16 lane + 15
constexpr int kPackRows = 24;

struct LossyDevTables {
    const float4 *pack;      // [kPackRows][64]
    const float *ath_db;     // [1024]
    const uint8_t *band;     // [1024]
    const float *band_count; // [25]
    const float *s10d;       // [25]
    const uint32_t *lane_bnd;    // [64]
    const uint32_t *lane_slot0;  // [64]
    const uint32_t *band_slot0;  // [26]
    int max_band_slots;
    float smr_thr;
    int q_transparent;
};

constexpr int kXchStride = 9;            // complex elements per exchange row (8 + 1 pad)
constexpr int kXchFloats = 64 * kXchStride * 2;  // 1152 floats per channel
constexpr int kCoefFloats = 1280;        // 1024 coefficients, 4 floats of padding per 16
constexpr int kSlotCap = 96;
constexpr int kFrameCap = 4352;          // >= 16 + 6+4+2+100+2*(4+2064), multiple of 16

// per-wave LDS
template <int CH>
struct LossyLds {
    union {
        float xch[CH][kXchFloats];     // FFT exchanges (float2 pairs)
        float coef[CH][kCoefFloats];   // transposition to the contiguous layout
        uint8_t stage[kFrameCap + 64]; // assembled frame bytes (+ carried tail)
    } u;
    float2 slots[CH][kSlotCap];        // (sum c^2, max |c|) per lane segment
    float4 bandv[CH][32];              // per band: (amplitude threshold, scale factor, s dB, unused)
};

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// ------------------------------------------------------------------------------------------------ DFT-8
// forward (e^{-2 pi i nk/8}), natural order in and out
__device__ __forceinline__ void dft8(float *xr, float *xi) {
    const float s = 0.70710678118654752440f;
    float a0r = xr[0] + xr[4], a0i = xi[0] + xi[4];
    float a1r = xr[0] - xr[4], a1i = xi[0] - xi[4];
    float a2r = xr[2] + xr[6], a2i = xi[2] + xi[6];
    float a3r = xr[2] - xr[6], a3i = xi[2] - xi[6];
    float a4r = xr[1] + xr[5], a4i = xi[1] + xi[5];
    float a5r = xr[1] - xr[5], a5i = xi[1] - xi[5];
    float a6r = xr[3] + xr[7], a6i = xi[3] + xi[7];
    float a7r = xr[3] - xr[7], a7i = xi[3] - xi[7];
    float b0r = a0r + a2r, b0i = a0i + a2i;
    float b2r = a0r - a2r, b2i = a0i - a2i;
    float b1r = a1r + a3i, b1i = a1i - a3r;  // a1 - i a3
    float b3r = a1r - a3i, b3i = a1i + a3r;  // a1 + i a3
    float c0r = a4r + a6r, c0i = a4i + a6i;
    float c2r = a4r - a6r, c2i = a4i - a6i;
    float c1r = a5r + a7i, c1i = a5i - a7r;
    float c3r = a5r - a7i, c3i = a5i + a7r;
    // W8^1 c1 = ((c1r + c1i) s, (c1i - c1r) s);  W8^3 c3 = ((c3i - c3r) s, -(c3r + c3i) s);  W8^2 c2 = (c2i, -c2r)
    const float e1r = c1r + c1i, e1i = c1i - c1r;
    const float e3r = c3i - c3r, e3i = c3r + c3i;
    xr[0] = b0r + c0r; xi[0] = b0i + c0i;
    xr[4] = b0r - c0r; xi[4] = b0i - c0i;
    xr[1] = fmaf(e1r, s, b1r); xi[1] = fmaf(e1i, s, b1i);
    xr[5] = fmaf(-e1r, s, b1r); xi[5] = fmaf(-e1i, s, b1i);
    xr[2] = b2r + c2i; xi[2] = b2i - c2r;
    xr[6] = b2r - c2i; xi[6] = b2i + c2r;
    xr[3] = fmaf(e3r, s, b3r); xi[3] = fmaf(-e3i, s, b3i);
    xr[7] = fmaf(-e3r, s, b3r); xi[7] = fmaf(e3i, s, b3i);
}

__device__ __forceinline__ void cmul(float &xr, float &xi, float wr, float wi) {
    float r = fmaf(xr, wr, -(xi * wi));
    float i = fmaf(xr, wi, xi * wr);
    xr = r;
    xi = i;
}

// ------------------------------------------------------------------------------------------------ FFT-512
// One 512-point complex FFT per channel per wave. In: lane l register r = z[l + 64 r]; out: same layout of Z.
//   z index n = 64 na + 8 nb + nc, Z index k = ka + 8 kb + 64 kc.
//   pass 1: DFT over na (registers) -> ka, twiddle W512^(lane*ka); LDS exchange nb <-> ka
//   pass 2: DFT over nb -> kb, twiddle W64^(nc*kb);             LDS exchange nc <-> kb
//   pass 3: DFT over nc -> kc.
template <int CH>
__device__ __forceinline__ void fft512(float (&zr)[CH][8], float (&zi)[CH][8], float (*xch)[kXchFloats],
                                       const LossyDevTables &T) {
    const int lane = lane_id();
    // pass 1
#pragma unroll
    for (int c = 0; c < CH; c++) dft8(zr[c], zi[c]);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        const float4 w = T.pack[(12 + kk) * 64 + lane];
#pragma unroll
        for (int c = 0; c < CH; c++) {
            cmul(zr[c][2 * kk + 1], zi[c][2 * kk + 1], w.x, w.y);
            if (kk < 3) cmul(zr[c][2 * kk + 2], zi[c][2 * kk + 2], w.z, w.w);
        }
    }
    {
        const int nb = lane >> 3, nc = lane & 7;
#pragma unroll
        for (int c = 0; c < CH; c++) {
            float2 *x = reinterpret_cast<float2 *>(xch[c]);
#pragma unroll
            for (int ka = 0; ka < 8; ka++) x[(8 * ka + nc) * kXchStride + nb] = make_float2(zr[c][ka], zi[c][ka]);
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const float2 *x = reinterpret_cast<const float2 *>(xch[c]);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                float2 v = x[lane * kXchStride + r];
                zr[c][r] = v.x;
                zi[c][r] = v.y;
            }
        }
        __syncthreads();
    }
    // pass 2 (lane = 8 ka + nc, register = nb)
#pragma unroll
    for (int c = 0; c < CH; c++) dft8(zr[c], zi[c]);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        const float4 w = T.pack[(16 + kk) * 64 + lane];
#pragma unroll
        for (int c = 0; c < CH; c++) {
            cmul(zr[c][2 * kk + 1], zi[c][2 * kk + 1], w.x, w.y);
            if (kk < 3) cmul(zr[c][2 * kk + 2], zi[c][2 * kk + 2], w.z, w.w);
        }
    }
    {
        const int ka = lane >> 3, nc = lane & 7;
#pragma unroll
        for (int c = 0; c < CH; c++) {
            float2 *x = reinterpret_cast<float2 *>(xch[c]);
#pragma unroll
            for (int kb = 0; kb < 8; kb++) x[(ka + 8 * kb) * kXchStride + nc] = make_float2(zr[c][kb], zi[c][kb]);
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const float2 *x = reinterpret_cast<const float2 *>(xch[c]);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                float2 v = x[lane * kXchStride + r];
                zr[c][r] = v.x;
                zi[c][r] = v.y;
            }
        }
        __syncthreads();
    }
    // pass 3 (lane = ka + 8 kb, register = nc)
#pragma unroll
    for (int c = 0; c < CH; c++) dft8(zr[c], zi[c]);
}

// ------------------------------------------------------------------------------------------------ input
// Half-frame offsets a lane folds: row r < 4: i = lane + 64 r  -> even 512 + 2i, odd 511 - 2i
//                                  row r >= 4: i = lane + 64(r-4) -> even 2i, odd 1023 - 2i
__device__ __forceinline__ void half_offsets(int lane, int r, int &eo, int &oo) {
    if (r < 4) {
        int i = lane + 64 * r;
        eo = 512 + 2 * i;
        oo = 511 - 2 * i;
    } else {
        int i = lane + 64 * (r - 4);
        eo = 2 * i;
        oo = 1023 - 2 * i;
    }
}

// Load the raw samples of one half-frame (1024 sample-frames starting at real sample s0; s0 may be negative
// = pre-roll) that this lane folds: even-offset and odd-offset sample of each of its 8 rows, CH channels
// starting at channel c0 of an nch-channel interleaved clip of n_frames_total sample-frames.
template <int CH>
__device__ __forceinline__ void load_half(const float *__restrict__ pcm, long long n_sf, int nch, int c0,
                                          long long s0, float (&he)[CH][8], float (&ho)[CH][8]) {
    const int lane = lane_id();
#pragma unroll
    for (int r = 0; r < 8; r++) {
        int eo, oo;
        half_offsets(lane, r, eo, oo);
        long long se = s0 + eo, so = s0 + oo;
        bool ve = se >= 0 && se < n_sf, vo = so >= 0 && so < n_sf;
        if (CH == 2 && nch == 2) {
            float2 a = ve ? *reinterpret_cast<const float2 *>(pcm + 2 * se) : make_float2(0.f, 0.f);
            float2 b = vo ? *reinterpret_cast<const float2 *>(pcm + 2 * so) : make_float2(0.f, 0.f);
            he[0][r] = a.x;
            he[CH - 1][r] = a.y;
            ho[0][r] = b.x;
            ho[CH - 1][r] = b.y;
        } else {
#pragma unroll
            for (int c = 0; c < CH; c++) {
                he[c][r] = ve ? pcm[se * nch + c0 + c] : 0.f;
                ho[c][r] = vo ? pcm[so * nch + c0 + c] : 0.f;
            }
        }
    }
}

// Fold first half (ae, ao) and second half (be, bo) into the 8 complex FFT inputs of this lane
// (window, butterflies and pre-rotation of mdct.rs:174-197, same operation order per element).
template <int CH>
__device__ __forceinline__ void fold(const float (&ae)[CH][8], const float (&ao)[CH][8], const float (&be)[CH][8],
                                     const float (&bo)[CH][8], float (&zr)[CH][8], float (&zi)[CH][8],
                                     const LossyDevTables &T) {
    const int lane = lane_id();
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const float4 ww = T.pack[r * 64 + lane];
        const float wae = ww.x, wao = ww.y, wbe = ww.z, wbo = ww.w;
        const float4 t4 = T.pack[(8 + (r >> 1)) * 64 + lane];
        const float2 w = (r & 1) ? make_float2(t4.z, t4.w) : make_float2(t4.x, t4.y);
#pragma unroll
        for (int c = 0; c < CH; c++) {
            float re, im;
            if (r < 4) {
                // re = -x[2i+n3] - x[n3-1-2i];  im = -x[n4+2i] + x[n4-1-2i]
                re = -(be[c][r] * wbe) - (bo[c][r] * wbo);
                im = -(ae[c][r] * wae) + (ao[c][r] * wao);
            } else {
                // re2 = x[2i] - x[n2-1-2i];  im2 = -x[n2+2i] - x[n-1-2i]
                re = (ae[c][r] * wae) - (ao[c][r] * wao);
                im = -(be[c][r] * wbe) - (bo[c][r] * wbo);
            }
            zr[c][r] = -re * w.x - im * w.y;
            zi[c][r] = re * w.y - im * w.x;
        }
    }
}

// Post-rotation (mdct.rs:203-223) + transpose to the contiguous layout through LDS:
//   out[2m] = -Z.re w.re - Z.im w.im,  out[1023 - 2m] = -Z.re w.im + Z.im w.re,   m = lane + 64 r
template <int CH>
__device__ __forceinline__ void post_rotate_transpose(const float (&zr)[CH][8], const float (&zi)[CH][8],
                                                      float (*coef)[kCoefFloats], float (&c)[CH][16],
                                                      const LossyDevTables &T) {
    const int lane = lane_id();
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int m = lane + 64 * r;
        const float4 t4 = T.pack[(8 + (r >> 1)) * 64 + lane];
        const float2 w = (r & 1) ? make_float2(t4.z, t4.w) : make_float2(t4.x, t4.y);
        const int k0 = 2 * m, k1 = 1023 - 2 * m;
        const int p0 = k0 + 4 * (k0 >> 4), p1 = k1 + 4 * (k1 >> 4);
#pragma unroll
        for (int ch = 0; ch < CH; ch++) {
            float R = -zr[ch][r] * w.x - zi[ch][r] * w.y;
            float I = -zr[ch][r] * w.y + zi[ch][r] * w.x;
            coef[ch][p0] = R;
            coef[ch][p1] = I;
        }
    }
    __syncthreads();
#pragma unroll
    for (int ch = 0; ch < CH; ch++) {
        const float4 *p = reinterpret_cast<const float4 *>(&coef[ch][20 * lane]);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float4 v = p[q];
            c[ch][4 * q + 0] = v.x;
            c[ch][4 * q + 1] = v.y;
            c[ch][4 * q + 2] = v.z;
            c[ch][4 * q + 3] = v.w;
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ bands
// Per-lane constants of the contiguous layout
struct LaneConst {
    uint32_t bnd;      // bit e: segment ends after element e
    uint32_t slot0;    // first slot of this lane
    uint32_t boff[8];  // 16 x u16: byte offset (band * 16) into bandv for element e
    float bcount;      // lanes 0..24: bins in band `lane`
    uint32_t bs0, bs1; // lanes 0..24: slot range of band `lane`
};

__device__ __forceinline__ void load_lane_const(LaneConst &L, const LossyDevTables &T) {
    const int lane = lane_id();
    L.bnd = T.lane_bnd[lane];
    L.slot0 = T.lane_slot0[lane];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t b0 = T.band[16 * lane + 2 * i], b1 = T.band[16 * lane + 2 * i + 1];
        L.boff[i] = (b0 * 16u) | ((b1 * 16u) << 16);
    }
    const int b = lane < 25 ? lane : 24;
    L.bcount = T.band_count[b];
    L.bs0 = T.band_slot0[b];
    L.bs1 = T.band_slot0[b + 1];
}

// Band energy (sum of c^2) and band maximum |c| (psychoacoustic.rs:155-163, encoder.rs:111-118).
// Each lane accumulates its 16 coefficients in ascending order and closes a partial at every band boundary
// into its own LDS slot; lane b < 25 then adds the slots of band b in ascending order: a fixed summation
// tree, independent of run and of grid shape. Returns (energy, max) of band `lane` in lanes 0..24.
template <int CH>
__device__ __forceinline__ void band_stats(const float (&c)[CH][16], float2 (*slots)[kSlotCap], const LaneConst &L,
                                           int max_band_slots, float (&energy)[CH], float (&bmax)[CH]) {
    float acc[CH], mx[CH];
    uint32_t slot = L.slot0;
#pragma unroll
    for (int ch = 0; ch < CH; ch++) {
        acc[ch] = 0.f;
        mx[ch] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 16; e++) {
#pragma unroll
        for (int ch = 0; ch < CH; ch++) {
            acc[ch] = fmaf(c[ch][e], c[ch][e], acc[ch]);
            mx[ch] = fmaxf(mx[ch], fabsf(c[ch][e]));
        }
        if (L.bnd & (1u << e)) {
#pragma unroll
            for (int ch = 0; ch < CH; ch++) {
                slots[ch][slot] = make_float2(acc[ch], mx[ch]);
                acc[ch] = 0.f;
                mx[ch] = 0.f;
            }
            slot++;
        }
    }
    __syncthreads();
#pragma unroll
    for (int ch = 0; ch < CH; ch++) {
        energy[ch] = 0.f;
        bmax[ch] = 0.f;
    }
    for (int i = 0; i < max_band_slots; i++) {
        uint32_t s = L.bs0 + i;
        if (s < L.bs1) {
#pragma unroll
            for (int ch = 0; ch < CH; ch++) {
                float2 v = slots[ch][s];
                energy[ch] += v.x;
                bmax[ch] = fmaxf(bmax[ch], v.y);
            }
        }
    }
    __syncthreads();
}

// Spreading + masking offset (psychoacoustic.rs:166-194): lanes 0..24 in, a[band] out (before temporal masking).
__device__ __forceinline__ float spread_threshold(float energy, float bcount, const LossyDevTables &T) {
    const int lane = lane_id();
    const bool is_band = lane < 25;
    float band_db = -100.0f;
    if (is_band && bcount > 0.f && energy > 1e-10f) band_db = 10.0f * log10f(__fdiv_rn(energy, bcount));
    if (!is_band) band_db = -__builtin_inff();
    // suffix maximum: bands j >= i mask band i at full strength (spreading[j][i] = 1 for j >= i)
    float sm = band_db;
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
        float t = __shfl_down(sm, d);
        if (lane + d < 25) sm = fmaxf(sm, t);
    }
    // bands j < i: band_db[j] + s10d[i-j]; only deltas with band_db_max + s10d[d] > -100 can matter
    float gmax = __shfl(sm, 0);
    int dmax = 24;
    if (gmax < 500.f) {
        int d = (int)((gmax + 100.0f) * (1.0f / 24.9f)) + 1;
        dmax = d < 1 ? 1 : (d > 24 ? 24 : d);
    }
    dmax = __builtin_amdgcn_readfirstlane(dmax);
    float m = fmaxf(-100.0f, sm);
    for (int d = 1; d <= dmax; d++) {
        float v = __shfl_up(band_db, d);
        float s = T.s10d[d];
        if (lane >= d) m = fmaxf(m, v + s);
    }
    return m + (-6.0f);
}

// scale-factor word (encoder.rs:262-266)
__device__ __forceinline__ uint32_t sf_word(float sf) {
    if (sf > 1e-10f) {
        float v = log2f(sf) * 256.0f + 32768.0f;
        v = fminf(fmaxf(v, 0.0f), 65535.0f);
        return (uint32_t)v;
    }
    return 0u;
}

// round half away from zero, exactly (f32::round)
__device__ __forceinline__ float round_away(float x) {
    float t = truncf(x);
    float d = x - t;  // exact
    return t + truncf(d + d);
}

// ------------------------------------------------------------------------------------------------ quantise
// Keep/drop + quantise 16 contiguous coefficients per channel (psychoacoustic.rs:205-234, encoder.rs:138-151).
// bandv[band] = (amplitude threshold from the masking level, scale factor, masking level s in dB, -).
// The keep test |c| > max(T_band, T_ath) is the reference's dB-domain test 20 log10|c| - (max(s, ath) - 10) > thr
// moved to the amplitude domain; coefficients within 1e-5 (relative) of the threshold are re-decided with the
// reference's exact f32 expression so that rounding of the reformulation never decides.
template <int CH>
__device__ __forceinline__ void quantise(const float (&c)[CH][16], const float4 (*bandv)[32], const LaneConst &L,
                                         const LossyDevTables &T, int (&q)[CH][16]) {
    const int lane = lane_id();
    float al[16];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float4 v = T.pack[(20 + i) * 64 + lane];
        al[4 * i] = v.x;
        al[4 * i + 1] = v.y;
        al[4 * i + 2] = v.z;
        al[4 * i + 3] = v.w;
    }
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const uint32_t off = (e & 1) ? (L.boff[e >> 1] >> 16) : (L.boff[e >> 1] & 0xFFFFu);
#pragma unroll
        for (int ch = 0; ch < CH; ch++) {
            const float4 bv = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(bandv[ch]) + off);
            const float x = c[ch][e];
            const float ax = fabsf(x);
            const float thr = fmaxf(bv.x, al[e]);
            bool keep = ax > thr;
            // |c| <= 1e-10 takes the reference's "-100 dB" branch; it can only be kept at quality >= 0.99
            const bool near = fabsf(ax - thr) <= 1e-5f * thr || (T.q_transparent && !(ax > 1e-10f));
            if (near) {
                // exact reference expression
                float signal_db = ax > 1e-10f ? 20.0f * log10f(ax) : -100.0f;
                float t = fmaxf(bv.z, T.ath_db[16 * lane + e]) - 10.0f;
                keep = (signal_db - t) > T.smr_thr;
            }
            int v = 0;
            if (keep) {
                float r = round_away(x * bv.y);
                r = fminf(fmaxf(r, -32768.0f), 32767.0f);
                v = (int)r;
            }
            q[ch][e] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ sparse RLE
// inclusive prefix sum over the wave
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v) {
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}
__device__ __forceinline__ int wave_excl_max_up(int v, int ident) {  // max over lanes < lane
    const int lane = lane_id();
    int x = __shfl_up(v, 1);
    if (lane == 0) x = ident;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(x, d);
        if (lane >= d) x = max(x, t);
    }
    return x;
}
__device__ __forceinline__ int wave_excl_min_down(int v, int ident) {  // min over lanes > lane
    const int lane = lane_id();
    int x = __shfl_down(v, 1);
    if (lane == 63) x = ident;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_down(x, d);
        if (lane + d < 64) x = min(x, t);
    }
    return x;
}

// Size and per-lane offsets of serialize_sparse (encoder.rs:284-314) for the 1024 values of one channel held
// 16 per lane. Records: [varint zero_run][u8 n <= 255][n x i16]; a trailing zero run closes with [varint][0].
struct SparsePlan {
    uint32_t m;        // non-zero mask of this lane's 16 values
    uint32_t zs;       // zero-run starts in this lane
    int nn;            // position of the first non-zero after this lane (1024 if none)
    int nz_end;        // position of the first zero after this lane (1024 if none)
    int t_in;          // length of the non-zero run ending just before this lane
    int cc_pos;        // local position of a 255-cap continuation record (or -1)
    uint32_t off0;     // byte offset (within the channel's sparse blob) of this lane's first byte
    uint32_t total;    // total sparse bytes (uniform)
};

__device__ __forceinline__ void sparse_plan(const int (&q)[16], SparsePlan &P) {
    const int lane = lane_id();
    uint32_t m = 0;
#pragma unroll
    for (int e = 0; e < 16; e++) m |= (q[e] != 0 ? 1u : 0u) << e;
    P.m = m;
    const uint32_t prev_m = __shfl_up(m, 1);
    const uint32_t prev_nz = lane == 0 ? 0u : (prev_m >> 15) & 1u;
    uint32_t zs = ~m & ((m << 1) | prev_nz) & 0xFFFFu;
    if (lane == 0 && !(m & 1u)) zs |= 1u;  // the walk starts with a (possibly empty... no: non-empty) zero run
    P.zs = zs;
    const int base = 16 * lane;
    // look-ahead / look-behind over the other lanes
    const int first_nz = m ? base + __builtin_ctz(m) : 1024;
    const uint32_t inv = ~m & 0xFFFFu;
    const int first_z = inv ? base + __builtin_ctz(inv) : 1024;
    const int last_z = inv ? base + 31 - __builtin_clz(inv) : -1;
    P.nn = wave_excl_min_down(first_nz, 1024);
    P.nz_end = wave_excl_min_down(first_z, 1024);
    const int lz_before = wave_excl_max_up(last_z, -1);
    P.t_in = base - 1 - lz_before;
    // 255-cap continuation: a position i in the leading non-zeros with (t_in + i) % 255 == 0 and t_in + i > 0
    const int ln = inv ? __builtin_ctz(inv) : 16;  // leading non-zeros
    P.cc_pos = -1;
    if (P.t_in > 0) {
        int x = (255 - (P.t_in % 255)) % 255;
        if (x < ln) P.cc_pos = x;
    }
    // header bytes of the records that start in this lane
    uint32_t hdr = 2u * (uint32_t)__builtin_popcount(zs);
    if (zs) {
        // only the last zero run of a lane can leave it; its length decides the varint size
        int s = 31 - __builtin_clz(zs);
        uint32_t above = m >> s;  // bit 0 is the zero at s
        int end = above ? base + s + __builtin_ctz(above) : P.nn;
        if (end - (base + s) >= 128) hdr += 1u;
    }
    if (P.cc_pos >= 0) hdr += 2u;
    if (lane == 0 && (m & 1u)) hdr += 2u;  // record that starts the walk on a non-zero
    const uint32_t bytes = hdr + 2u * (uint32_t)__builtin_popcount(m);
    const uint32_t incl = wave_incl_sum(bytes);
    P.off0 = incl - bytes;
    P.total = __shfl(incl, 63);
}

// Emit this lane's part of the sparse blob to `dst` (LDS bytes; the blob starts at dst[0]). Values are fetched by
// run-time position from qv (this lane's 16 values parked in LDS) so no register array is indexed dynamically.
// A zero-run record reserves its count byte; the first non-zero of the following run fills it in.
__device__ __forceinline__ void sparse_emit(const int16_t *qv, const SparsePlan &P, uint8_t *dst) {
    const int lane = lane_id();
    const int base = 16 * lane;
    uint32_t off = P.off0;
    uint32_t ev = P.zs | P.m;
    const uint32_t prev_nz_bit = (P.t_in > 0) ? 1u : 0u;
    while (ev) {
        const int i = __builtin_ctz(ev);
        ev &= ev - 1;
        if ((P.zs >> i) & 1u) {
            const uint32_t above = P.m >> i;
            const int end = above ? base + i + __builtin_ctz(above) : P.nn;
            const uint32_t zc = (uint32_t)(end - (base + i));
            if (zc >= 128u) {
                dst[off] = (uint8_t)((zc & 0x7Fu) | 0x80u);
                dst[off + 1] = (uint8_t)(zc >> 7);
                off += 2;
            } else {
                dst[off] = (uint8_t)zc;
                off += 1;
            }
            if (end >= 1024) dst[off] = 0;  // trailing zeros: [varint][0]
            off += 1;
        } else {
            const bool prev_is_nz = i > 0 ? ((P.m >> (i - 1)) & 1u) : prev_nz_bit;
            const uint32_t rest = P.m >> i;
            const int run_local = __builtin_ctz(~rest);
            const int run_end = (i + run_local >= 16) ? P.nz_end : base + i + run_local;
            const uint32_t remaining = (uint32_t)(run_end - (base + i));
            const uint8_t cnt = (uint8_t)(remaining < 255u ? remaining : 255u);
            if (i == P.cc_pos || (lane == 0 && i == 0)) {
                dst[off] = 0;
                dst[off + 1] = cnt;
                off += 2;
            } else if (!prev_is_nz) {
                dst[off - 1] = cnt;
            }
            const uint32_t v = (uint16_t)qv[i];
            dst[off] = (uint8_t)v;
            dst[off + 1] = (uint8_t)(v >> 8);
            off += 2;
        }
    }
}

}  // namespace flo
