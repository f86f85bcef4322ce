// lossless_kernels.hip — gfx950 kernels and host orchestration of the lossless (ALPC + Rice) encode path.
//
// Replaces, bit for bit (files under /root/reference/libflo/src):
//   frame split, silence test, f32->i32, de-interleave, mid/side ...... lossless/encoder.rs:47-170
//   candidate search raw / fixed 0-4 / LPC 5..max, strict '<' order .... lossless/encoder.rs:173-287
//   integer autocorrelation, f64 Levinson-Durbin, fixed-point LPC ....... lossless/lpc.rs:213-298
//   fixed predictors with warm-up ......................................... lossless/lpc.rs:301-359
//   Rice parameter estimate, zigzag, unary/binary MSB-first packing ..... core/rice.rs:29-114,162-202
//   frame/channel framing ................................................. writer.rs:236-301, core/types.rs:243-267
//
// Kernels (one launch each per encode):
//   ll_prepare   one workgroup per 1-second frame: silence flag, f32->i32 planes, M/S decision and transform
//   ll_analyze   one workgroup per (frame, channel): all candidates' residual statistics in three sweeps over the
//                integer plane (L2-resident), Levinson for every LPC order by one lane in FP64, winner selection
//   ll_layout    one thread per clip: frame types, payload sizes, byte offsets (exclusive scan inside the clip)
//   ll_pack      one workgroup per (frame, channel): headers + residual bitstream of the winner only
// The reference fully Rice-encodes ~10 candidates per channel bit by bit just to learn their lengths; here the
// lengths come from exact integer sums and only the winner is ever packed.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "container.hpp"
#include "lossless_kernels.hpp"
#include "devpool.hpp"

#include "container_kernels.hpp"

namespace flo {

constexpr int kLLThreads = 256;
constexpr int kMaxOrder = 12;
constexpr int kNumCand = 1 + 5 + 8;  // raw, fixed 0..4, lpc 5..12
constexpr unsigned int kStageWords = 4096;   // 16 KiB: a tile of 4096 samples at up to 32 bits per sample

struct LLFrame {              // one 1-second frame of one clip (host-planned)
    unsigned long long pcm_off;   // float offset of the frame slice from the PCM base
    unsigned int slice_len;       // floats in the slice (all channels)
    unsigned int frame_samples;   // slice_len / channels
    unsigned long long plane_off; // i16 offset of this frame's planes in the scratch (a multiple of 16)
    unsigned int plane_stride;    // i16 elements per channel plane (a multiple of 16)
    unsigned int clip;
    unsigned int first_chan;      // index of this frame's first LLChan record
};

struct LLFrameOut {           // device-decided
    int silent;
    int use_ms;
    unsigned int frame_type;
    unsigned int flags;
    unsigned long long byte_off;  // offset of the frame inside the clip's DATA chunk
    unsigned int size;
};

struct LLChan {               // per (frame, channel)
    int kind;                 // 0 raw PCM, 1 fixed, 2 LPC, 3 empty
    int order;
    int k;
    int shift;
    int coefs[kMaxOrder];
    unsigned int res_bytes;   // residual payload bytes of the winner
    unsigned int n;           // samples in this channel plane
    unsigned long long payload_off;  // byte offset (inside the clip's DATA chunk) of the u32 size prefix
    unsigned int payload_size;       // bytes after the size prefix
    unsigned int alpc_layout;        // 1: ALPC channel layout, 0: bare residual bytes (Raw frame) / nothing (Silence)
};

struct LLArgs {
    const float *pcm;
    short *planes;                // f32_to_i32 of every sample is within i16 (audio_constants.rs:18-20 clamps): 2 B per sample
    const LLFrame *frames;
    LLFrameOut *fout;
    LLChan *chans;
    unsigned int n_frames;
    int nch;
    int level;
    int max_order;
    // layout
    const unsigned int *clip_first_frame;  // [n_clips + 1]
    const unsigned long long *clip_out_off;
    unsigned long long *clip_bytes;
    unsigned int n_clips;
    unsigned char *out;
    unsigned int *frame_size;              // [n_frames] bytes of each frame (for the TOC)
};

// ------------------------------------------------------------------------------------------------ helpers
__device__ __forceinline__ int f32_to_i32(float s) {  // core/audio_constants.rs:18-20
    float v = s * 32767.0f;
    v = v < -32768.0f ? -32768.0f : v;
    v = v > 32767.0f ? 32767.0f : v;
    return (v != v) ? 0 : (int)v;  // NaN -> 0, truncation toward zero
}

__device__ __forceinline__ unsigned int uabs(int r) { return r < 0 ? 0u - (unsigned int)r : (unsigned int)r; }
__device__ __forceinline__ unsigned int zigzag(int s) { return ((unsigned int)s << 1) ^ (unsigned int)(s >> 31); }

template <typename T>
__device__ __forceinline__ T block_sum(T v, T *scratch) {
    // deterministic tree: wave shuffle then LDS across the 4 waves
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    T r = scratch[0];
    for (int i = 1; i < kLLThreads / 64; i++) r += scratch[i];
    return r;
}
__device__ __forceinline__ unsigned int block_max(unsigned int v, unsigned int *scratch) {
    for (int d = 32; d > 0; d >>= 1) {
        unsigned int t = __shfl_down(v, d);
        v = v > t ? v : t;
    }
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    unsigned int r = scratch[0];
    for (int i = 1; i < kLLThreads / 64; i++) r = r > scratch[i] ? r : scratch[i];
    return r;
}

// rice.rs:29-69 from the exact statistics of a residual sequence
__device__ __forceinline__ int rice_k(unsigned long long sum_abs, unsigned int max_abs, unsigned int n) {
    if (n == 0) return 4;
    if (max_abs == 0) return 0;
    unsigned long long max_unsigned = 2ull * max_abs;
    int min_k = 0;
    if (max_unsigned > 255) {
        int bits = 64 - __clzll((long long)max_unsigned);
        min_k = bits >= 8 ? bits - 8 : 0;
    }
    unsigned int mean = (unsigned int)(sum_abs / (unsigned long long)n);
    int mean_k = mean > 0 ? 32 - __clz((int)mean) : 0;
    int k = min_k > mean_k ? min_k : mean_k;
    return k > 15 ? 15 : k;
}

// residual of fixed predictor `order` at position i (lpc.rs:301-359, warm-up uses lower orders); s = plane
__device__ __forceinline__ int fixed_residual(const int *__restrict__ s, unsigned int i, int order) {
    int o = order < (int)i ? order : (int)i;  // warm-up: position i < order uses order i
    long long v;
    switch (o) {
        case 0: v = s[i]; break;
        case 1: v = (long long)s[i] - s[i - 1]; break;
        case 2: v = (long long)s[i] - 2ll * s[i - 1] + s[i - 2]; break;
        case 3: v = (long long)s[i] - 3ll * s[i - 1] + 3ll * s[i - 2] - s[i - 3]; break;
        default: v = (long long)s[i] - 4ll * s[i - 1] + 6ll * s[i - 2] - 4ll * s[i - 3] + s[i - 4]; break;
    }
    return (int)(unsigned int)(unsigned long long)v;  // wrapping i32
}

// LPC residual (lpc.rs:279-298): warm-up copies samples; i64 dot, arithmetic shift, truncating cast, wrapping sub
__device__ __forceinline__ int lpc_residual(const int *__restrict__ s, unsigned int i, const int *coef, int order, int shift) {
    if (i < (unsigned int)order) return s[i];
    long long pred = 0;
    for (int j = 0; j < order; j++) pred += (long long)coef[j] * (long long)s[i - j - 1];
    pred >>= shift;
    return (int)((unsigned int)s[i] - (unsigned int)(int)pred);
}

// Levinson-Durbin in f64 then fixed point (lpc.rs:225-276). No fused multiply-add anywhere: Rust never contracts.
#pragma clang fp contract(off)
__device__ bool levinson_fixed(const long long *autocorr, int order, int *coefs_out, int *shift_out, double *err_out) {
    if (autocorr[0] == 0) return false;
    double coeffs[kMaxOrder], nc[kMaxOrder];
    for (int i = 0; i < kMaxOrder; i++) coeffs[i] = 0.0;
    double error = (double)autocorr[0];
    for (int i = 0; i < order; i++) {
        double lambda = (double)autocorr[i + 1];
        for (int j = 0; j < i; j++) {
            double prod = coeffs[j] * (double)autocorr[i - j];
            lambda = lambda - prod;
        }
        if (fabs(error) < 1e-10) return false;
        double gamma = lambda / error;
        if (fabs(gamma) >= 1.0) return false;
        nc[i] = gamma;
        for (int j = 0; j < i; j++) {
            double prod = gamma * coeffs[i - 1 - j];
            nc[j] = coeffs[j] - prod;
        }
        for (int j = 0; j <= i; j++) coeffs[j] = nc[j];
        double g2 = gamma * gamma;
        double f = 1.0 - g2;
        error = error * f;
    }
    double max_coeff = 0.0;
    for (int i = 0; i < order; i++) max_coeff = fmax(max_coeff, fabs(coeffs[i]));
    if (max_coeff == 0.0 || !isfinite(max_coeff)) return false;
    // shift = min(floor(log2(2^30 / max_coeff)) as u8, 15). |gamma| < 1 bounds every coefficient by C(12,6) = 924,
    // so the quotient is >= 2^20 and the answer is 15; the general expression is kept for completeness.
    double quot = 1073741824.0 / max_coeff;
    int shift;
    if (quot >= 65536.0) {
        shift = 15;
    } else {
        double fl = floor(log2(quot));
        shift = fl != fl ? 0 : (fl <= 0.0 ? 0 : (fl >= 255.0 ? 255 : (int)fl));
        if (shift > 15) shift = 15;
    }
    double scale = (double)(1ll << shift);
    for (int i = 0; i < order; i++) {
        double v = round(coeffs[i] * scale);
        int c;
        if (v != v) c = 0;
        else if (v <= -2147483648.0) c = (int)0x80000000;
        else if (v >= 2147483647.0) c = 0x7FFFFFFF;
        else c = (int)v;
        coefs_out[i] = c;
    }
    *shift_out = shift;
    *err_out = error;
    return true;
}
#pragma clang fp contract(fast)

// ------------------------------------------------------------------------------------------------ plane access
// The planes hold the converted samples per channel, L and R as they are even when the frame is coded mid/side
// (encoder.rs:156-170: mid = l + r, side = l - r need 17 bits): the consumers form mid or side while loading, so the
// planes are written once, at 2 bytes per sample. ms: 0 = the plane P itself, 1 = P + Q (mid), 2 = P - Q (side).
__device__ __forceinline__ int plane_at(const short *__restrict__ P, const short *__restrict__ Q, int ms, unsigned int i) {
    const int a = P[i];
    if (ms == 0) return a;
    const int b = Q[i];
    return ms == 1 ? a + b : a - b;
}
__device__ __forceinline__ void unpack8(const uint4 v, int (&e)[8]) {
    e[0] = (int)(short)(v.x & 0xFFFFu); e[1] = (int)v.x >> 16;
    e[2] = (int)(short)(v.y & 0xFFFFu); e[3] = (int)v.y >> 16;
    e[4] = (int)(short)(v.z & 0xFFFFu); e[5] = (int)v.z >> 16;
    e[6] = (int)(short)(v.w & 0xFFFFu); e[7] = (int)v.w >> 16;
}
// w[0 .. KH) = the KH samples before i0 (zeros in front of the plane), w[KH .. KH + 16) = the run from i0 (zeros past n).
// i0 is a multiple of 16 and the planes are 32-byte aligned, so away from the plane's ends a run is three or four
// 16-byte loads per plane instead of one load per sample.
template <int KH>
__device__ __forceinline__ void load_run(const short *__restrict__ P, const short *__restrict__ Q, int ms, unsigned int i0,
                                         unsigned int n, int (&w)[16 + KH]) {
    static_assert(KH >= 1 && KH <= 16, "history of at most 16 samples");
    if (i0 >= 16u && i0 + 16u <= n) {
        constexpr int G0 = KH > 8 ? 0 : 1;   // first 8-sample group needed, counted from i0 - 16
        int e[4][8];
#pragma unroll
        for (int g = G0; g < 4; g++) {
            unpack8(*reinterpret_cast<const uint4 *>(P + i0 - 16 + 8 * g), e[g]);
            if (ms) {
                int q[8];
                unpack8(*reinterpret_cast<const uint4 *>(Q + i0 - 16 + 8 * g), q);
#pragma unroll
                for (int j = 0; j < 8; j++) e[g][j] = ms == 1 ? e[g][j] + q[j] : e[g][j] - q[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 16 + KH; j++) {
            const int src = 16 - KH + j;   // position counted from i0 - 16
            w[j] = e[src >> 3][src & 7];
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 16 + KH; j++) {
        const long long idx = (long long)i0 - KH + j;
        w[j] = (idx >= 0 && idx < (long long)n) ? plane_at(P, Q, ms, (unsigned int)idx) : 0;
    }
}

// ------------------------------------------------------------------------------------------------ ll_prepare
__global__ __launch_bounds__(kLLThreads) void ll_prepare_kernel(LLArgs A) {
    __shared__ long long red[kLLThreads / 64];
    __shared__ int s_flag;
    const unsigned int f = blockIdx.x;
    if (f >= A.n_frames) return;
    const LLFrame fr = A.frames[f];
    const float *src = A.pcm + fr.pcm_off;
    const int ch = A.nch;
    short *planes = A.planes + fr.plane_off;
    // One pass over the slice: silence test over every float (encoder.rs:70), f32 -> i32 -> i16 planes, and for stereo
    // the i64 energies of L, R and L - R over the common length (encoder.rs:131-153) from the same registers.
    int loud = 0;
    long long vl = 0, vr = 0, vs = 0;
    if (ch == 2) {
        const unsigned int n1 = fr.slice_len / 2;   // complete sample-frames; an odd slice ends with a lone L sample
        short *L = planes, *R = planes + fr.plane_stride;
        for (unsigned int p = threadIdx.x; p < n1; p += kLLThreads) {
            const float a = src[2 * p], b = src[2 * p + 1];
            if (!(fabsf(a) < 1e-7f) || !(fabsf(b) < 1e-7f)) loud = 1;
            const int l = f32_to_i32(a), r = f32_to_i32(b);
            L[p] = (short)l;
            R[p] = (short)r;
            vl += (long long)(l * l);
            vr += (long long)(r * r);
            const int d = l - r;
            vs += (long long)d * d;
        }
        if ((fr.slice_len & 1u) && threadIdx.x == 0) {
            const float a = src[fr.slice_len - 1];
            if (!(fabsf(a) < 1e-7f)) loud = 1;
            L[n1] = (short)f32_to_i32(a);
        }
    } else {
        for (unsigned int i = threadIdx.x; i < fr.slice_len; i += kLLThreads) {
            const float v = src[i];
            if (!(fabsf(v) < 1e-7f)) loud = 1;
            const unsigned int c = i % (unsigned int)ch, p = i / (unsigned int)ch;
            planes[(size_t)c * fr.plane_stride + p] = (short)f32_to_i32(v);
        }
    }
    if (threadIdx.x == 0) s_flag = 0;
    __syncthreads();
    if (loud) atomicOr(&s_flag, 1);
    __syncthreads();
    const int silent = !s_flag;
    int use_ms = 0;
    if (ch == 2) {   // uniform: every thread takes part in the sums
        vl = block_sum(vl, red);
        vr = block_sum(vr, red);
        vs = block_sum(vs, red);
        use_ms = !silent && vs < (vl + vr) / 2;
    }
    if (threadIdx.x == 0) {
        A.fout[f].silent = silent;
        A.fout[f].use_ms = use_ms;
    }
    // per-channel sample counts
    if (threadIdx.x < (unsigned int)ch) {
        unsigned int c = threadIdx.x;
        unsigned int n = fr.slice_len > c ? (fr.slice_len - c + ch - 1) / ch : 0;
        if (use_ms) {
            unsigned int n0 = (fr.slice_len + 1) / 2, n1 = fr.slice_len / 2;
            n = n0 < n1 ? n0 : n1;
        }
        A.chans[fr.first_chan + c].n = n;
    }
}

// ------------------------------------------------------------------------------------------------ ll_analyze
struct CandStats {
    unsigned long long sum_abs;
    unsigned int max_abs;
};

// MAXO = level_to_order(level) (encoder.rs:289-302): a compile-time bound for every lag / order loop
template <int MAXO>
__global__ __launch_bounds__(kLLThreads) void ll_analyze_kernel(LLArgs A) {
    __shared__ unsigned long long red64[kLLThreads / 64];
    __shared__ long long redi64[kLLThreads / 64];
    __shared__ unsigned int red32[kLLThreads / 64];
    __shared__ long long s_ac[kMaxOrder + 1];
    __shared__ int s_k[kNumCand];
    __shared__ int s_valid[kNumCand];
    __shared__ int s_coef[8][kMaxOrder];
    __shared__ int s_shift[8];
    __shared__ int s_kw0[8];      // first Rice parameter of the three-wide window tried during sweep 2
    __shared__ int s_need3[8];    // the window missed: the candidate still needs sweep 3
    __shared__ unsigned long long s_bits[kNumCand];

    const unsigned int f = blockIdx.x / (unsigned int)A.nch, c = blockIdx.x % (unsigned int)A.nch;
    if (f >= A.n_frames) return;
    const LLFrame fr = A.frames[f];
    LLChan *out = &A.chans[fr.first_chan + c];
    if (A.fout[f].silent) {
        if (threadIdx.x == 0) {
            out->kind = 3;
            out->order = 0;
            out->res_bytes = 0;
        }
        return;
    }
    const unsigned int n = out->n;
    const int ms = A.fout[f].use_ms ? (c == 0 ? 1 : 2) : 0;
    const short *P = A.planes + fr.plane_off + (ms ? 0 : (size_t)c * fr.plane_stride);
    const short *Q = P + fr.plane_stride;
    constexpr int max_order = MAXO;
    constexpr int kLpcOrders = MAXO > 4 ? MAXO - 4 : 0;   // LPC candidates: orders 5..MAXO
    constexpr int fixed_max = max_order < 4 ? max_order : 4;
    const bool try_lpc = A.level >= 3 && max_order > 4;
    // The plane is walked in tiles of 256 threads x 16 consecutive samples. A thread keeps its 16 samples and the 12
    // before them in registers (kWin), so every predictor tap of every candidate is a register operand: the plane is
    // read once per sweep instead of once per tap. Every statistic is an exact integer sum or maximum, so how the
    // samples are dealt to threads cannot change a result.
#ifndef FLO_LL_RUN
#define FLO_LL_RUN 16
#endif
    constexpr int kRun = FLO_LL_RUN, kHist = MAXO > 4 ? MAXO : 4, kWin = kRun + kHist;
    const unsigned int tile = kLLThreads * kRun;
    static_assert(kRun == 16, "load_run deals 16-sample runs");
    auto load_window = [&](unsigned int i0, int (&w)[kWin]) { load_run<kHist>(P, Q, ms, i0, n, w); };
    // All fixed-predictor residuals of a run by repeated differencing: order o is the difference of two order o - 1
    // residuals (in wrapping 32-bit arithmetic this equals the binomial form truncated from i64), 4.4 subtractions
    // per sample for the five orders together. D[o][j] is the order-o residual of sample i0 + j computed with zeros in
    // front of the plane; positions below the order take the lower order (lpc.rs:301-359), which only ever concerns
    // the first four samples of the plane.
    auto fixed_all = [&](const int (&w)[kWin], unsigned int i0, unsigned int (&D)[5][kRun]) {
        unsigned int d1[kRun + 3], d2[kRun + 2], d3[kRun + 1];
#pragma unroll
        for (int j = 0; j < kRun + 3; j++) d1[j] = (unsigned int)w[kHist + j - 3] - (unsigned int)w[kHist + j - 4];
#pragma unroll
        for (int j = 0; j < kRun + 2; j++) d2[j] = d1[j + 1] - d1[j];
#pragma unroll
        for (int j = 0; j < kRun + 1; j++) d3[j] = d2[j + 1] - d2[j];
#pragma unroll
        for (int j = 0; j < kRun; j++) {
            D[0][j] = (unsigned int)w[kHist + j];
            D[1][j] = d1[j + 3];
            D[2][j] = d2[j + 2];
            D[3][j] = d3[j + 1];
            D[4][j] = d3[j + 1] - d3[j];
        }
        if (i0 == 0) {   // warm-up of the plane's first samples: position i uses order min(o, i)
            static_assert(kRun >= 4, "the warm-up fix-up touches the first four samples of a run");
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int o = 1; o < 5; o++)
                    if (o > j) D[o][j] = D[j][j];
        }
    };

    // ---- sweep 1: autocorrelation lags 0..max_order (lpc.rs:213-221) and fixed-predictor statistics
    {
        // The autocorrelation products are exact in double precision (|s| <= 2^16: the planes are i16, mid/side 17 bits),
        // and so is a thread's running sum while it stays below 2^53 (it is flushed into the i64 total long before:
        // every 2^12 tiles). v_fma_f64 issues at four times the rate of v_mad_i64_i32.
        long long ac[MAXO + 1];
        double acd[MAXO + 1];
        unsigned long long fs[5];
        unsigned int fm[5];
        for (int l = 0; l <= MAXO; l++) { ac[l] = 0; acd[l] = 0.0; }
        for (int o = 0; o < 5; o++) { fs[o] = 0; fm[o] = 0; }
        // A run in the interior of the plane (everything but a plane's first and last run) needs no bounds test, no
        // warm-up rule and has zeros nowhere in its window: FULL drops all of that, element by element (it was five
        // scalar / compare instructions per element and candidate). Partial sums of a run stay in 32 bits.
        auto run1 = [&](const unsigned int i0, auto FULL) {
            constexpr bool full = decltype(FULL)::value;
            int w[kWin];
            load_window(i0, w);
            unsigned int D[5][kRun];
            fixed_all(w, i0, D);
            unsigned int rs[5] = {0, 0, 0, 0, 0};
            if (try_lpc) {
                // (samples in front of the plane and behind its end are zeros in the window: no bounds test needed)
                double wd[kWin];
#pragma unroll
                for (int j = 0; j < kWin; j++) wd[j] = (double)w[j];
#pragma unroll
                for (int j = 0; j < kRun; j++)
#pragma unroll
                    for (int l = 0; l <= MAXO; l++) acd[l] = fma(wd[kHist + j], wd[kHist + j - l], acd[l]);
            }
#pragma unroll
            for (int j = 0; j < kRun; j++) {
                const unsigned int i = i0 + j;
                if (!full && i >= n) break;
#pragma unroll
                for (int o = 0; o < 5; o++)
                    if (o <= fixed_max) {
                        const unsigned int a = uabs((int)D[o][j]);
                        rs[o] += a;   // planes hold i16 (mid/side: 17 bits): |r| < 2^22, sixteen of them fit 32 bits
                        fm[o] = fm[o] > a ? fm[o] : a;
                    }
            }
#pragma unroll
            for (int o = 0; o < 5; o++) fs[o] += rs[o];
        };
        unsigned int tiles_done = 0;
        for (unsigned int t0 = 0; t0 < n; t0 += tile) {
            const unsigned int i0 = t0 + threadIdx.x * kRun;
            if (i0 < n) {
                if (i0 >= 16u && i0 + kRun <= n) run1(i0, std::true_type{});
                else run1(i0, std::false_type{});
            }
            if ((++tiles_done & 4095u) == 0u)   // 2^12 tiles x 16 samples x 2^32 < 2^53
                for (int l = 0; l <= MAXO; l++) { ac[l] += (long long)acd[l]; acd[l] = 0.0; }
        }
        for (int l = 0; l <= MAXO; l++) ac[l] += (long long)acd[l];
        if (try_lpc)
            for (int l = 0; l <= max_order; l++) {
                long long t = block_sum(ac[l], redi64);
                if (threadIdx.x == 0) s_ac[l] = t;
            }
        for (int o = 0; o <= fixed_max; o++) {
            unsigned long long t = block_sum(fs[o], red64);
            unsigned int m = block_max(fm[o], red32);
            if (threadIdx.x == 0) {
                s_k[1 + o] = rice_k(t, m, n);
                s_valid[1 + o] = 1;
            }
        }
    }
    __syncthreads();
    // Levinson for every LPC order (one lane; tiny)
    if (threadIdx.x == 0) {
        for (int o = fixed_max + 1; o <= 4; o++) s_valid[1 + o] = 0;
        for (int ord = 5; ord <= kMaxOrder; ord++) {
            int ci = 6 + (ord - 5);
            s_valid[ci] = 0;
            if (try_lpc && ord <= max_order && n > (unsigned int)ord) {
                int sh;
                double err;
                if (levinson_fixed(s_ac, ord, s_coef[ord - 5], &sh, &err)) {
                    s_shift[ord - 5] = sh;
                    s_valid[ci] = 1;
                    // Guess of the Rice parameter from the predicted residual energy (mean |r| of a Laplacian is
                    // sigma / sqrt 2): sweep 2 counts code lengths for the guess and its two neighbours, so that
                    // the third sweep is only needed when the true parameter falls outside (never wrong, only slower).
                    double mean = err > 0.0 ? sqrt(err / (double)n) * 0.70710678 : 0.0;
                    unsigned int mi = mean < 4.0e9 ? (unsigned int)mean : 0xFFFFFFFFu;
                    int kg = mi ? 32 - __clz((int)mi) : 0;
                    kg = kg > 15 ? 15 : kg;
                    s_kw0[ord - 5] = kg >= 14 ? 13 : (kg > 0 ? kg - 1 : 0);
                }
            }
        }
    }
    __syncthreads();

    // LPC residual of order ORD at position i = i0 + j from the window (lpc.rs:279-298)
    auto lpc_res = [&](const int (&w)[kWin], unsigned int i, int j, int ord, const int *coef, int shift) {
        if (i < (unsigned int)ord) return w[kHist + j];
        long long pred = 0;
#pragma unroll
        for (int q = 0; q < MAXO; q++)
            if (q < ord) pred += (long long)coef[q] * (long long)w[kHist + j - 1 - q];
        pred >>= shift;
        return (int)((unsigned int)w[kHist + j] - (unsigned int)(int)pred);
    };

    // the same for a run in the interior of the plane: no warm-up (i >= 16 > ord)
    auto lpc_res_full = [&](const int (&w)[kWin], int j, int ord, const int *coef, int shift) {
        long long pred = 0;
#pragma unroll
        for (int q = 0; q < MAXO; q++)
            if (q < ord) pred += (long long)coef[q] * (long long)w[kHist + j - 1 - q];
        pred >>= shift;
        return (int)((unsigned int)w[kHist + j] - (unsigned int)(int)pred);
    };

    // ---- sweep 2: code lengths of the fixed candidates, statistics of the LPC candidates
    {
        unsigned long long fb[5];
        constexpr int NL = kLpcOrders > 0 ? kLpcOrders : 1;
        unsigned long long ls[NL];
        unsigned int lm[NL];
        unsigned int lw[NL][3];   // code-length sums for the window of Rice parameters (a thread sees < 2^23 / 256 samples)
        for (int o = 0; o < 5; o++) fb[o] = 0;
        for (int o = 0; o < NL; o++) { ls[o] = 0; lm[o] = 0; lw[o][0] = lw[o][1] = lw[o][2] = 0; }
        int kf[5];
        for (int o = 0; o < 5; o++) kf[o] = s_k[1 + o];
        auto run2 = [&](const unsigned int i0, auto FULL) {
            constexpr bool full = decltype(FULL)::value;
            int w[kWin];
            load_window(i0, w);
            {
                unsigned int D[5][kRun];
                fixed_all(w, i0, D);
#pragma unroll
                for (int o = 0; o < 5; o++)
                    if (o <= fixed_max) {
                        unsigned int rs = 0;   // at most 16 x 255
#pragma unroll
                        for (int j = 0; j < kRun; j++)
                            if (full || i0 + j < n) {
                                unsigned int q = zigzag((int)D[o][j]) >> kf[o];
                                rs += q < 255u ? q : 255u;
                            }
                        fb[o] += rs;
                    }
            }
            if (try_lpc) {
#pragma unroll
                for (int oi = 0; oi < kLpcOrders; oi++) {
                    const int ord = 5 + oi;
                    if (!s_valid[6 + oi]) continue;   // uniform across the block
                    int coef[NL + 4];
#pragma unroll
                    for (int q = 0; q < NL + 4; q++) coef[q] = q < ord ? s_coef[oi][q] : 0;
                    const int sh = s_shift[oi], k0 = s_kw0[oi];
                    unsigned int rsum = 0;   // |r| <= 1e6 for a candidate that stays valid; a larger one is discarded
                                             // by its maximum, whatever its sum
#pragma unroll
                    for (int j = 0; j < kRun; j++)
                        if (full || i0 + j < n) {
                            const int rr = full ? lpc_res_full(w, j, ord, coef, sh) : lpc_res(w, i0 + j, j, ord, coef, sh);
                            const unsigned int a = uabs(rr);
                            rsum += a;
                            lm[oi] = lm[oi] > a ? lm[oi] : a;
                            const unsigned int u0 = zigzag(rr) >> k0;
                            lw[oi][0] += u0 < 255u ? u0 : 255u;
                            lw[oi][1] += (u0 >> 1) < 255u ? (u0 >> 1) : 255u;
                            lw[oi][2] += (u0 >> 2) < 255u ? (u0 >> 2) : 255u;
                        }
                    ls[oi] += rsum;
                }
            }
        };
        for (unsigned int t0 = 0; t0 < n; t0 += tile) {
            const unsigned int i0 = t0 + threadIdx.x * kRun;
            if (i0 >= n) continue;
            if (i0 >= 16u && i0 + kRun <= n) run2(i0, std::true_type{});
            else run2(i0, std::false_type{});
        }
        for (int o = 0; o <= fixed_max; o++) {
            unsigned long long t = block_sum(fb[o], red64);
            if (threadIdx.x == 0) s_bits[1 + o] = t + (unsigned long long)n * (1 + kf[o]);
        }
        if (try_lpc)
            for (int ord = 5; ord <= max_order; ord++) {
                const int ci = 6 + ord - 5;
                if (!s_valid[ci]) continue;  // uniform across the block
                unsigned long long t = block_sum(ls[ord - 5], red64);
                unsigned int m = block_max(lm[ord - 5], red32);
                unsigned long long w0 = block_sum((unsigned long long)lw[ord - 5][0], red64);
                unsigned long long w1 = block_sum((unsigned long long)lw[ord - 5][1], red64);
                unsigned long long w2 = block_sum((unsigned long long)lw[ord - 5][2], red64);
                if (threadIdx.x == 0) {
                    s_need3[ord - 5] = 0;
                    if (m > 1000000u) {
                        s_valid[ci] = 0;  // encoder.rs:269-272
                    } else {
                        const int k = rice_k(t, m, n), d = k - s_kw0[ord - 5];
                        s_k[ci] = k;
                        if (d >= 0 && d <= 2) s_bits[ci] = (d == 0 ? w0 : (d == 1 ? w1 : w2)) + (unsigned long long)n * (1 + k);
                        else s_need3[ord - 5] = 1;
                    }
                }
            }
    }
    __syncthreads();
    // ---- sweep 3: code lengths of the surviving LPC candidates whose Rice parameter fell outside the window
    bool any3 = false;
    if (try_lpc)
        for (int ord = 5; ord <= max_order; ord++) any3 |= s_valid[6 + ord - 5] && s_need3[ord - 5];
    if (any3) {
        constexpr int NL = kLpcOrders > 0 ? kLpcOrders : 1;
        unsigned long long lb[NL];
        for (int o = 0; o < NL; o++) lb[o] = 0;
        auto run3 = [&](const unsigned int i0, auto FULL) {
            constexpr bool full = decltype(FULL)::value;
            int w[kWin];
            load_window(i0, w);
#pragma unroll
            for (int oi = 0; oi < kLpcOrders; oi++) {
                const int ord = 5 + oi;
                if (!s_valid[6 + oi] || !s_need3[oi]) continue;
                int coef[NL + 4];
#pragma unroll
                for (int q = 0; q < NL + 4; q++) coef[q] = q < ord ? s_coef[oi][q] : 0;
                const int sh = s_shift[oi], kk = s_k[6 + oi];
                unsigned int rs = 0;
#pragma unroll
                for (int j = 0; j < kRun; j++)
                    if (full || i0 + j < n) {
                        const int rr = full ? lpc_res_full(w, j, ord, coef, sh) : lpc_res(w, i0 + j, j, ord, coef, sh);
                        unsigned int q = zigzag(rr) >> kk;
                        rs += q < 255u ? q : 255u;
                    }
                lb[oi] += rs;
            }
        };
        for (unsigned int t0 = 0; t0 < n; t0 += tile) {
            const unsigned int i0 = t0 + threadIdx.x * kRun;
            if (i0 >= n) continue;
            if (i0 >= 16u && i0 + kRun <= n) run3(i0, std::true_type{});
            else run3(i0, std::false_type{});
        }
        for (int ord = 5; ord <= max_order; ord++) {
            const int ci = 6 + ord - 5;
            if (!s_valid[ci] || !s_need3[ord - 5]) continue;
            unsigned long long t = block_sum(lb[ord - 5], red64);
            if (threadIdx.x == 0) s_bits[ci] = t + (unsigned long long)n * (1 + s_k[ci]);
        }
    }
    __syncthreads();
    // ---- winner: raw, fixed 0..4, LPC 5..max; strictly smaller byte length wins (encoder.rs:183-216)
    if (threadIdx.x == 0) {
        unsigned long long best = 2ull * n;
        int kind = 0, order = 0, k = 0, ci_best = 0;
        for (int o = 0; o <= fixed_max; o++) {
            unsigned long long bytes = (s_bits[1 + o] + 7) >> 3;
            if (bytes < best) { best = bytes; kind = 1; order = o; k = s_k[1 + o]; ci_best = 1 + o; }
        }
        if (try_lpc)
            for (int ord = 5; ord <= max_order; ord++) {
                const int ci = 6 + ord - 5;
                if (!s_valid[ci]) continue;
                unsigned long long bytes = (s_bits[ci] + 7) >> 3;
                if (bytes < best) { best = bytes; kind = 2; order = ord; k = s_k[ci]; ci_best = ci; }
            }
        out->kind = kind;
        out->order = order;
        out->k = k;
        out->shift = kind == 2 ? s_shift[order - 5] : 0;
        for (int j = 0; j < kMaxOrder; j++) out->coefs[j] = (kind == 2 && j < order) ? s_coef[order - 5][j] : 0;
        out->res_bytes = (unsigned int)best;
        (void)ci_best;
    }
}

// ------------------------------------------------------------------------------------------------ ll_layout
// frame types, channel payload sizes and byte offsets inside each clip's DATA chunk (writer.rs:236-301)
__global__ void ll_layout_kernel(LLArgs A) {
    const unsigned int clip = blockIdx.x * blockDim.x + threadIdx.x;
    if (clip >= A.n_clips) return;
    unsigned long long off = 0;
    for (unsigned int f = A.clip_first_frame[clip]; f < A.clip_first_frame[clip + 1]; f++) {
        const LLFrame fr = A.frames[f];
        LLFrameOut &fo = A.fout[f];
        LLChan *ch = &A.chans[fr.first_chan];
        unsigned int type, flags = 0;
        if (fo.silent) {
            type = 0;  // FrameType::Silence
        } else {
            bool all_raw = true;
            for (int c = 0; c < A.nch; c++)
                if (ch[c].order > 0) all_raw = false;  // order_used (raw and fixed order 0 both report 0)
            type = all_raw ? 254u : (A.max_order >= 1 && A.max_order <= 12 ? (unsigned int)A.max_order : 8u);
            if (fo.use_ms) flags |= 1u;
        }
        fo.frame_type = type;
        fo.flags = flags;
        fo.byte_off = off;
        unsigned long long pos = off + 6;
        for (int c = 0; c < A.nch; c++) {
            unsigned int psize = 0, layout = 0;
            if (type == 254u) {
                psize = ch[c].res_bytes;
            } else if (type != 0u) {
                layout = 1;
                const int ncoef = ch[c].kind == 2 ? ch[c].order : 0;
                psize = 1 + 4 * ncoef + 1 + 1 + (ch[c].kind == 0 ? 0 : 1) + ch[c].res_bytes;
            }
            ch[c].payload_off = pos;
            ch[c].payload_size = psize;
            ch[c].alpc_layout = layout;
            pos += 4 + psize;
        }
        fo.size = (unsigned int)(pos - off);
        A.frame_size[f] = fo.size;
        off = pos;
    }
    A.clip_bytes[clip] = off;
}

// ------------------------------------------------------------------------------------------------ ll_pack
// All byte-granular pieces go through 32-bit atomic ORs into the zero-initialised output, so words shared by
// neighbouring payloads (different workgroups, possibly different XCDs) are composed correctly in any order.
__device__ __forceinline__ void or_bytes(unsigned char *out, unsigned long long pos, unsigned long long value, int nbytes) {
    for (int i = 0; i < nbytes; i++) {
        unsigned long long p = pos + i;
        unsigned int b = (unsigned int)((value >> (8 * i)) & 0xFF);
        if (b) atomicOr(reinterpret_cast<unsigned int *>(out + (p & ~3ull)), b << (8 * (p & 3)));
    }
}

// MSB-first bit writer over 32-bit big-endian words at an absolute bit position of `out`
struct BitSink {
    unsigned char *out;
    unsigned long long word;   // index of the 32-bit word being filled
    unsigned int acc;          // bits filled from the MSB side
    int fill;                  // number of valid bits in acc
    bool first;                // the first word may be shared with whoever wrote the preceding bits
    __device__ __forceinline__ void flush_word(bool shared) {
        unsigned int v = __builtin_bswap32(acc);
        unsigned int *p = reinterpret_cast<unsigned int *>(out) + word;
        if (shared) { if (v) atomicOr(p, v); }
        else *p = v;
        word++;
        acc = 0;
        fill = 0;
        first = false;
    }
    __device__ __forceinline__ void put(unsigned int value, int nbits) {  // nbits <= 32, value < 2^nbits
        while (nbits > 0) {
            int room = 32 - fill;
            int take = nbits < room ? nbits : room;
            unsigned int part = (take == 32) ? value : ((value >> (nbits - take)) & ((1u << take) - 1u));
            acc |= (take == 32) ? part : (part << (room - take));
            fill += take;
            nbits -= take;
            if (fill == 32) flush_word(first);
        }
    }
    __device__ __forceinline__ void finish() {
        if (fill > 0) flush_word(true);
    }
};

// Zigzag codes u[j] of the 16 samples of a run and the sum of their Rice code lengths. ORD = 0: fixed predictor of
// order ch.order (lpc.rs:301-359; positions below the order use order i); ORD >= 5: LPC of exactly ORD taps
// (lpc.rs:279-298: the first ORD samples of the plane are copied, then i64 dot product, arithmetic shift, wrapping
// subtraction). w holds the 12 samples before the run and the run itself.
// FULL: a run in the interior of the plane (i0 >= 16, all 16 samples present): no bounds test, no warm-up rule.
template <int ORD, bool FULL = false>
__device__ __forceinline__ unsigned long long run_codes(const LLChan &ch, const int (&w)[16 + kMaxOrder], unsigned int i0,
                                                        unsigned int i1, int k, unsigned int (&u)[16]) {
    unsigned int bits = 0;   // 16 codes of at most 256 + 15 bits
#pragma unroll
    for (unsigned int j = 0; j < 16; j++) {
        const unsigned int i = i0 + j;
        u[j] = 0;
        if (FULL || i < i1) {
            const int x0 = w[kMaxOrder + j];
            int r;
            if (ORD == 0) {
                const int oo = (FULL || ch.order < (int)i) ? ch.order : (int)i;
                const long long x1 = w[kMaxOrder + j - 1], x2 = w[kMaxOrder + j - 2], x3 = w[kMaxOrder + j - 3],
                                x4 = w[kMaxOrder + j - 4];
                long long v;
                switch (oo) {
                    case 0: v = x0; break;
                    case 1: v = (long long)x0 - x1; break;
                    case 2: v = (long long)x0 - 2ll * x1 + x2; break;
                    case 3: v = (long long)x0 - 3ll * x1 + 3ll * x2 - x3; break;
                    default: v = (long long)x0 - 4ll * x1 + 6ll * x2 - 4ll * x3 + x4; break;
                }
                r = (int)(unsigned int)(unsigned long long)v;
            } else if (!FULL && i < (unsigned int)ORD) {
                r = x0;
            } else {
                long long pred = 0;
#pragma unroll
                for (int q = 0; q < ORD; q++) pred += (long long)ch.coefs[q] * (long long)w[kMaxOrder + j - 1 - q];
                pred >>= ch.shift;
                r = (int)((unsigned int)x0 - (unsigned int)(int)pred);
            }
            u[j] = zigzag(r);
            unsigned int q = u[j] >> k;
            bits += (q < 255u ? q : 255u) + 1u + (unsigned int)k;
        }
    }
    return bits;
}

__global__ __launch_bounds__(kLLThreads) void ll_pack_kernel(LLArgs A) {
    __shared__ unsigned long long sc[kLLThreads];
    __shared__ unsigned int stage[kStageWords];   // one tile's bits (big-endian words), composed here, stored whole
    const unsigned int f = blockIdx.x / (unsigned int)A.nch, c = blockIdx.x % (unsigned int)A.nch;
    if (f >= A.n_frames) return;
    const LLFrame fr = A.frames[f];
    const LLFrameOut fo = A.fout[f];
    const LLChan ch = A.chans[fr.first_chan + c];
    unsigned char *out = A.out + A.clip_out_off[fr.clip];
    // frame header by the first channel's workgroup (writer.rs:240-242)
    if (c == 0 && threadIdx.x == 0) {
        or_bytes(out, fo.byte_off, fo.frame_type, 1);
        or_bytes(out, fo.byte_off + 1, fr.frame_samples, 4);
        or_bytes(out, fo.byte_off + 5, fo.flags, 1);
    }
    unsigned long long pos = ch.payload_off;
    if (threadIdx.x == 0) {
        or_bytes(out, pos, ch.payload_size, 4);
        if (ch.alpc_layout) {  // writer.rs:274-296
            unsigned long long p = pos + 4;
            const int ncoef = ch.kind == 2 ? ch.order : 0;
            or_bytes(out, p++, (unsigned int)ncoef, 1);
            // indexed through the global record: a run-time index into the register copy would push all of it to scratch
            for (int j = 0; j < ncoef; j++, p += 4) or_bytes(out, p, (unsigned int)A.chans[fr.first_chan + c].coefs[j], 4);
            const unsigned int shift_bits = ch.kind == 1 ? 128u + (unsigned int)ch.order : (ch.kind == 2 ? (unsigned int)ch.shift : 0u);
            or_bytes(out, p++, shift_bits, 1);
            or_bytes(out, p++, ch.kind == 0 ? 2u : 0u, 1);  // ResidualEncoding::Raw = 2, Rice = 0
            if (ch.kind != 0) or_bytes(out, p++, (unsigned int)ch.k, 1);
        }
    }
    if (fo.frame_type == 0u || ch.kind == 3) return;
    unsigned long long res_pos = pos + 4;
    if (ch.alpc_layout) res_pos += 1 + 4 * (ch.kind == 2 ? ch.order : 0) + 1 + 1 + (ch.kind == 0 ? 0 : 1);
    const unsigned int n = ch.n;
    const int ms = fo.use_ms ? (c == 0 ? 1 : 2) : 0;
    const short *P = A.planes + fr.plane_off + (ms ? 0 : (size_t)c * fr.plane_stride);
    const short *Q = P + fr.plane_stride;
    if (ch.kind == 0) {
        // raw PCM: (s as i16).to_le_bytes() (encoder.rs:220-226), wrapping truncation
        for (unsigned int i = threadIdx.x; i < n; i += kLLThreads)
            or_bytes(out, res_pos + 2ull * i, (unsigned int)(unsigned short)(short)plane_at(P, Q, ms, i), 2);
        return;
    }
    // Rice (rice.rs:94-114), in tiles of 256 threads x 16 consecutive samples: a tile spans 16 KiB of the plane that the
    // workgroup reads completely (every cache line used by neighbouring lanes), each thread counts the bits of its
    // 16 samples, an exclusive scan places them behind the previous tile, then the bits go out MSB-first.
    constexpr unsigned int kPer = 16;
    const int k = ch.k;
    const unsigned long long bit0 = 8ull * (unsigned long long)(reinterpret_cast<uintptr_t>(out + res_pos) & 3ull);
    unsigned char *base4 = reinterpret_cast<unsigned char *>(reinterpret_cast<uintptr_t>(out + res_pos) & ~(uintptr_t)3);
    unsigned long long tile_bit = 0;   // bits written by earlier tiles
    for (unsigned int t0 = 0; t0 < n; t0 += kLLThreads * kPer) {
        const unsigned int i0 = t0 + threadIdx.x * kPer < n ? t0 + threadIdx.x * kPer : n;
        const unsigned int i1 = i0 + kPer < n ? i0 + kPer : n;
        unsigned int u[kPer];
        unsigned long long bits = 0;
        if (i0 < i1) {
            // the thread's 16 samples and the 12 before them, once; every predictor tap is then a register operand
            int w[kPer + kMaxOrder];
            load_run<kMaxOrder>(P, Q, ms, i0, n, w);
            // residuals of the run: one straight-line instantiation per predictor (the choice is uniform over the
            // workgroup), so every tap and every window index is a compile-time constant
            const bool full = i0 >= 16u && i0 + kPer <= n;   // everything but a plane's first and last run
#define FLO_RUN_CODES(ORD) bits = full ? run_codes<ORD, true>(ch, w, i0, i1, k, u) : run_codes<ORD, false>(ch, w, i0, i1, k, u)
            if (ch.kind == 1) {
                FLO_RUN_CODES(0);
            } else {
                switch (ch.order) {
                    case 5: FLO_RUN_CODES(5); break;
                    case 6: FLO_RUN_CODES(6); break;
                    case 7: FLO_RUN_CODES(7); break;
                    case 8: FLO_RUN_CODES(8); break;
                    case 9: FLO_RUN_CODES(9); break;
                    case 10: FLO_RUN_CODES(10); break;
                    case 11: FLO_RUN_CODES(11); break;
                    default: FLO_RUN_CODES(12); break;
                }
            }
#undef FLO_RUN_CODES
        }
        // exclusive scan of the threads' bit counts inside the tile: in the wave by shuffles (a run is at most
        // 16 x 271 bits, a tile below 2^21: 32 bits are plenty), across the four waves through LDS - two barriers per tile
        // where the 256-entry Hillis-Steele scan took sixteen
        unsigned int inc = bits;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned int o = __shfl_up(inc, d, 64);
            if ((int)(threadIdx.x & 63u) >= d) inc += o;
        }
        __syncthreads();   // the previous tile's readers of sc are done
        if ((threadIdx.x & 63u) == 63u) sc[threadIdx.x >> 6] = inc;
        __syncthreads();
        unsigned int wave_base = 0, total32 = 0;
#pragma unroll
        for (int wv = 0; wv < kLLThreads / 64; wv++) {
            const unsigned int t = (unsigned int)sc[wv];
            if (wv < (int)(threadIdx.x >> 6)) wave_base += t;
            total32 += t;
        }
        const unsigned long long tile_total = total32;
        const unsigned long long rel_bit = (unsigned long long)(wave_base + inc) - bits;   // inside the tile
        const unsigned long long tile_abs = bit0 + tile_bit;                   // absolute bit position of the tile
        tile_bit += tile_total;
        // The tile's bits are composed in LDS and leave as whole words: only the first and the last word of a tile can
        // be shared with a neighbour and need a global atomic OR (it was two atomics per THREAD before, and the L2's
        // atomic rate was what bounded the kernel). A tile too long for the staging buffer (very noisy material at a
        // small Rice parameter) keeps the direct path.
        const unsigned int lead = (unsigned int)(tile_abs & 31ull);            // the tile starts inside this word
        const unsigned long long span = lead + tile_total;
        const bool staged = span <= 32ull * kStageWords;
        if (staged) {
            const unsigned int nw = (unsigned int)((span + 31ull) >> 5);
            for (unsigned int i = threadIdx.x; i < nw; i += kLLThreads) stage[i] = 0u;
            __syncthreads();
            if (i0 < i1) {
                // The thread's bits go through a 64-bit shift register (bit 63 first): a code of at most 32 bits - unary
                // part, terminator and remainder as ONE field for every quotient below 32 - k - is appended with one shift and
                // one or, and the upper word leaves whenever it is complete. The first word a thread touches is shared with
                // its predecessor and the last with its successor: those two are or-ed in, the ones between are stored.
                const unsigned long long pos = lead + rel_bit;
                unsigned int word = (unsigned int)(pos >> 5);
                unsigned long long acc = 0;
                int fill = (int)(pos & 31ull);   // bits of the word in front of this thread's first bit
                bool first = true;
                auto put = [&](unsigned int value, int nbits) {   // 1 <= nbits <= 32, value < 2^nbits, fill < 32 on entry
                    acc |= (unsigned long long)value << (64 - fill - nbits);
                    fill += nbits;
                    if (fill >= 32) {
                        const unsigned int hi = (unsigned int)(acc >> 32);
                        if (first) { if (hi) atomicOr(&stage[word], hi); }
                        else stage[word] = hi;
                        first = false;
                        word++;
                        acc <<= 32;
                        fill -= 32;
                    }
                };
#pragma unroll
                for (unsigned int j = 0; j < kPer; j++) {
                    if (i0 + j < i1) {
                        unsigned int q = u[j] >> k;
                        q = q < 255u ? q : 255u;
                        const unsigned int rem = k ? (u[j] & ((1u << k) - 1u)) : 0u;
                        if (q + 1u + (unsigned int)k <= 32u) {
                            put(((((1u << q) - 1u) << 1) << k) | rem, (int)q + 1 + k);
                        } else {
                            unsigned int ones = q;
                            while (ones >= 32) {
                                put(0xFFFFFFFFu, 32);
                                ones -= 32;
                            }
                            put(ones ? (((1u << ones) - 1u) << 1) : 0u, (int)ones + 1);
                            if (k) put(rem, k);
                        }
                    }
                }
                if (fill > 0) {
                    const unsigned int hi = (unsigned int)(acc >> 32);
                    if (hi) atomicOr(&stage[word], hi);
                }
            }
            __syncthreads();
            unsigned int *gw = reinterpret_cast<unsigned int *>(base4) + (tile_abs >> 5);
            for (unsigned int i = threadIdx.x; i < nw; i += kLLThreads) {
                const unsigned int v = __builtin_bswap32(stage[i]);     // words are big-endian bit containers
                if (i == 0 || i + 1 == nw) { if (v) atomicOr(gw + i, v); }
                else gw[i] = v;
            }
            // the next tile zeroes `stage` only after its own scan, i.e. behind two barriers: no extra one needed here
        } else if (i0 < i1) {
            const unsigned long long abs_bit = tile_abs + rel_bit;
            BitSink bs;
            bs.out = base4;
            bs.word = abs_bit >> 5;
            bs.acc = 0;
            bs.fill = (int)(abs_bit & 31);
            bs.first = true;
#pragma unroll
            for (unsigned int j = 0; j < kPer; j++) {
                if (i0 + j < i1) {
                    unsigned int q = u[j] >> k;
                    q = q < 255u ? q : 255u;
                    unsigned int ones = q;
                    while (ones >= 32) {
                        bs.put(0xFFFFFFFFu, 32);
                        ones -= 32;
                    }
                    // remaining ones, the terminating zero, then k remainder bits
                    if (ones + 1 <= 32) bs.put(ones ? (((1u << ones) - 1u) << 1) : 0u, (int)ones + 1);
                    if (k) bs.put(u[j] & ((1u << k) - 1u), k);
                }
            }
            bs.finish();
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
struct LosslessPlan {
    uint32_t sr = 0;
    uint8_t ch = 0, level = 0;
    int max_order = 0;
    size_t n_clips = 0;
    const float *d_pcm = nullptr;
    std::vector<LLFrame> frames;
    std::vector<uint32_t> clip_first_frame;
    std::vector<uint64_t> clip_out_off, clip_out_cap, clip_file_off;
    uint64_t out_bytes = 0, plane_ints = 0;
    size_t n_chans = 0;
    // device
    LLFrame *d_frames = nullptr;
    LLFrameOut *d_fout = nullptr;
    LLChan *d_chans = nullptr;
    short *d_planes = nullptr;
    uint32_t *d_cff = nullptr;
    uint64_t *d_coo = nullptr, *d_clip_bytes = nullptr, *d_cf0 = nullptr;
    uint32_t *d_fsize = nullptr, *d_fsamp = nullptr, *d_cfn = nullptr, *d_crc = nullptr, *d_part = nullptr;
    uint8_t *d_out = nullptr;
    // host results
    std::vector<LLFrameOut> h_fout;
    std::vector<uint64_t> h_clip_bytes, h_file_bytes;
    LLChan *h_chans_pin = nullptr;   // pinned mirror of d_chans for lossless_describe (allocated on first use)
    hipStream_t stream = nullptr;
};

static int level_to_order(int level) {  // encoder.rs:289-302
    static const int t[10] = {0, 2, 4, 4, 6, 8, 8, 10, 12, 12};
    return t[level < 0 ? 0 : (level > 9 ? 9 : level)];
}

void lossless_plan_destroy(LosslessPlan *p) {
    if (!p) return;
    void *ptrs[] = {p->d_frames, p->d_fout, p->d_chans, p->d_planes, p->d_cff, p->d_coo, p->d_clip_bytes, p->d_out,
                    p->d_cf0, p->d_fsize, p->d_fsamp, p->d_cfn, p->d_crc, p->d_part};
    for (void *q : ptrs)
        if (q) pool_free(q);
    if (p->h_chans_pin) (void)hipHostFree(p->h_chans_pin);
    delete p;
}

LosslessPlan *lossless_plan_create(const std::vector<uint64_t> &n_il, const std::vector<uint64_t> &clip_off,
                                      uint32_t sr, uint8_t ch, uint8_t level, const float *d_pcm, std::string &err) {
    LosslessPlan *p = new LosslessPlan();
    p->sr = sr;
    p->ch = ch;
    p->level = level;
    p->max_order = level_to_order(level);
    p->n_clips = n_il.size();
    p->d_pcm = d_pcm;
    p->clip_first_frame.resize(p->n_clips + 1);
    p->clip_out_off.resize(p->n_clips);
    p->clip_out_cap.resize(p->n_clips);
    p->clip_file_off.resize(p->n_clips);
    const uint64_t spf = sr;
    uint64_t out = 0, planes = 0;
    for (size_t i = 0; i < p->n_clips; i++) {
        p->clip_first_frame[i] = (uint32_t)p->frames.size();
        const uint64_t len = n_il[i], total = len / ch;
        const uint64_t nf = (total + spf - 1) / spf;  // encoder.rs:48-49
        uint64_t cap = 0;
        for (uint64_t f = 0; f < nf; f++) {
            uint64_t start = f * spf * ch, end = (f + 1) * spf * ch;
            if (end > len) end = len;
            LLFrame fr{};
            fr.pcm_off = clip_off[i] + start;
            fr.slice_len = (uint32_t)(end - start);
            fr.frame_samples = (uint32_t)((end - start) / ch);
            fr.plane_stride = (uint32_t)(((end - start + ch - 1) / ch + 15) & ~15ull);
            fr.plane_off = planes;
            fr.clip = (uint32_t)i;
            fr.first_chan = (uint32_t)(p->frames.size() * ch);
            planes += (uint64_t)fr.plane_stride * ch;
            cap += 6 + (uint64_t)ch * (4 + 56 + 2ull * fr.plane_stride);
            p->frames.push_back(fr);
        }
        // header + TOC of the finished file sit right in front of the (16-byte aligned) DATA chunk
        const uint64_t head = 74 + 20 * nf;
        out += (head + 15) & ~15ull;
        p->clip_out_off[i] = out;
        p->clip_file_off[i] = out - head;
        p->clip_out_cap[i] = (cap + 8 + 15) & ~15ull;
        out += p->clip_out_cap[i];
    }
    p->clip_first_frame[p->n_clips] = (uint32_t)p->frames.size();
    p->out_bytes = out;
    p->plane_ints = planes;
    p->n_chans = p->frames.size() * ch;
    hipError_t e;
#define LCHK(x)                                                         \
    if ((e = (x)) != hipSuccess) {                                      \
        err = std::string(#x) + ": " + hipGetErrorString(e);            \
        lossless_plan_destroy(p);                                       \
        return nullptr;                                                 \
    }
    const size_t nf = p->frames.size();
    LCHK(pool_alloc(&p->d_frames, (nf + 1) * sizeof(LLFrame)));
    LCHK(pool_alloc(&p->d_fout, (nf + 1) * sizeof(LLFrameOut)));
    LCHK(pool_alloc(&p->d_chans, (p->n_chans + 1) * sizeof(LLChan)));
    LCHK(pool_alloc(&p->d_planes, (planes + 16) * sizeof(short)));
    LCHK(pool_alloc(&p->d_cff, (p->n_clips + 1) * 4));
    LCHK(pool_alloc(&p->d_coo, (p->n_clips + 1) * 8));
    LCHK(pool_alloc(&p->d_clip_bytes, (p->n_clips + 1) * 8));
    LCHK(pool_alloc(&p->d_out, out + 64));
    if (nf) LCHK(hipMemcpy(p->d_frames, p->frames.data(), nf * sizeof(LLFrame), hipMemcpyHostToDevice));
    LCHK(hipMemcpy(p->d_cff, p->clip_first_frame.data(), (p->n_clips + 1) * 4, hipMemcpyHostToDevice));
    if (p->n_clips) LCHK(hipMemcpy(p->d_coo, p->clip_out_off.data(), p->n_clips * 8, hipMemcpyHostToDevice));
    {   // what the on-device file assembly needs: frames per clip, first frame, samples per frame
        std::vector<uint64_t> cf0(p->n_clips + 1);
        std::vector<uint32_t> cfn(p->n_clips + 1), fsamp(nf + 1);
        for (size_t i = 0; i < p->n_clips; i++) {
            cf0[i] = p->clip_first_frame[i];
            cfn[i] = p->clip_first_frame[i + 1] - p->clip_first_frame[i];
        }
        for (size_t f = 0; f < nf; f++) fsamp[f] = p->frames[f].frame_samples;
        LCHK(pool_alloc(&p->d_cf0, cf0.size() * 8));
        LCHK(pool_alloc(&p->d_cfn, cfn.size() * 4));
        LCHK(pool_alloc(&p->d_fsamp, fsamp.size() * 4));
        LCHK(pool_alloc(&p->d_fsize, (nf + 1) * 4));
        LCHK(pool_alloc(&p->d_crc, (p->n_clips + 1) * 4));
        LCHK(pool_alloc(&p->d_part, (p->n_clips * finish_parts_for(p->n_clips) + 1) * 4));
        LCHK(hipMemcpy(p->d_cf0, cf0.data(), cf0.size() * 8, hipMemcpyHostToDevice));
        LCHK(hipMemcpy(p->d_cfn, cfn.data(), cfn.size() * 4, hipMemcpyHostToDevice));
        LCHK(hipMemcpy(p->d_fsamp, fsamp.data(), fsamp.size() * 4, hipMemcpyHostToDevice));
    }
#undef LCHK
    return p;
}

int lossless_encode_launch(LosslessPlan *p, hipStream_t s, int profile, std::string &err) {
    (void)profile;
    p->stream = s;
    const unsigned nf = (unsigned)p->frames.size();
    hipError_t e;
    if ((e = hipMemsetAsync(p->d_out, 0, p->out_bytes + 64, s)) != hipSuccess) {
        err = hipGetErrorString(e);
        return -1;
    }
    if ((e = hipMemsetAsync(p->d_clip_bytes, 0, (p->n_clips + 1) * 8, s)) != hipSuccess) {
        err = hipGetErrorString(e);
        return -1;
    }
    LLArgs A{};
    A.pcm = p->d_pcm;
    A.planes = p->d_planes;
    A.frames = p->d_frames;
    A.fout = p->d_fout;
    A.chans = p->d_chans;
    A.n_frames = nf;
    A.nch = p->ch;
    A.level = p->level;
    A.max_order = p->max_order;
    A.clip_first_frame = p->d_cff;
    A.clip_out_off = (const unsigned long long *)p->d_coo;
    A.clip_bytes = (unsigned long long *)p->d_clip_bytes;
    A.n_clips = (unsigned)p->n_clips;
    A.out = p->d_out;
    A.frame_size = p->d_fsize;
    if (nf) {
        hipLaunchKernelGGL(ll_prepare_kernel, dim3(nf), dim3(kLLThreads), 0, s, A);
        switch (p->max_order) {   // encoder.rs:289-302 yields exactly these orders
            case 0: hipLaunchKernelGGL(ll_analyze_kernel<0>, dim3(nf * p->ch), dim3(kLLThreads), 0, s, A); break;
            case 2: hipLaunchKernelGGL(ll_analyze_kernel<2>, dim3(nf * p->ch), dim3(kLLThreads), 0, s, A); break;
            case 4: hipLaunchKernelGGL(ll_analyze_kernel<4>, dim3(nf * p->ch), dim3(kLLThreads), 0, s, A); break;
            case 6: hipLaunchKernelGGL(ll_analyze_kernel<6>, dim3(nf * p->ch), dim3(kLLThreads), 0, s, A); break;
            case 8: hipLaunchKernelGGL(ll_analyze_kernel<8>, dim3(nf * p->ch), dim3(kLLThreads), 0, s, A); break;
            case 10: hipLaunchKernelGGL(ll_analyze_kernel<10>, dim3(nf * p->ch), dim3(kLLThreads), 0, s, A); break;
            default: hipLaunchKernelGGL(ll_analyze_kernel<12>, dim3(nf * p->ch), dim3(kLLThreads), 0, s, A); break;
        }
        hipLaunchKernelGGL(ll_layout_kernel, dim3((A.n_clips + 63) / 64), dim3(64), 0, s, A);
        hipLaunchKernelGGL(ll_pack_kernel, dim3(nf * p->ch), dim3(kLLThreads), 0, s, A);
    }
    if ((e = hipGetLastError()) != hipSuccess) {
        err = hipGetErrorString(e);
        return -1;
    }
    // header, TOC and CRC32 in front of every DATA chunk (writer.rs:132-224; encoder.rs:36-44 parameters)
    FinishArgs F{};
    F.out = p->d_out;
    F.data_off = (const unsigned long long *)p->d_coo;
    F.clip_bytes = (const unsigned long long *)p->d_clip_bytes;
    F.clip_frame0 = (const unsigned long long *)p->d_cf0;
    F.clip_frames = p->d_cfn;
    F.frame_size = p->d_fsize;
    F.frame_samples = p->d_fsamp;
    F.const_samples = 0;
    F.sample_rate = p->sr;
    F.flags = 0;
    F.channels = p->ch;
    F.bit_depth = 16;   // echoed from the caller when the file is fetched
    F.level = p->level;
    F.n_clips = (int)p->n_clips;
    F.crc_out = p->d_crc;
    F.parts = finish_parts_for(p->n_clips);
    F.part_reg = p->d_part;
    F.max_frames = 0;
    for (size_t i = 0; i < p->n_clips; i++) {
        const unsigned nfc = (unsigned)(p->clip_first_frame[i + 1] - p->clip_first_frame[i]);
        F.max_frames = nfc > F.max_frames ? nfc : F.max_frames;
    }
    if (launch_finish_files(F, s) != 0) {
        err = "finish_files launch failed";
        return -1;
    }
    return 0;
}

int lossless_collect(LosslessPlan *p, std::string &err) {
    hipError_t e;
    p->h_fout.resize(p->frames.size());
    p->h_clip_bytes.assign(p->n_clips, 0);
    if (!p->frames.empty() &&
        (e = hipMemcpy(p->h_fout.data(), p->d_fout, p->frames.size() * sizeof(LLFrameOut), hipMemcpyDeviceToHost)) != hipSuccess) {
        err = hipGetErrorString(e);
        return -1;
    }
    if (p->n_clips && (e = hipMemcpy(p->h_clip_bytes.data(), p->d_clip_bytes, p->n_clips * 8, hipMemcpyDeviceToHost)) != hipSuccess) {
        err = hipGetErrorString(e);
        return -1;
    }
    for (size_t i = 0; i < p->n_clips; i++)
        if (p->h_clip_bytes[i] > p->clip_out_cap[i]) {
            err = "bitstream overran its buffer";
            return -1;
        }
    return 0;
}

uint64_t lossless_total_bytes(const LosslessPlan *p) {
    uint64_t t = 0;
    for (auto v : p->h_clip_bytes) t += v;
    return t;
}

int lossless_device_streams(LosslessPlan *p, const uint8_t **base, const uint64_t **offsets, const uint64_t **sizes) {
    if (base) *base = p->d_out;
    if (offsets) *offsets = p->clip_out_off.data();
    if (sizes) *sizes = p->h_clip_bytes.data();
    return 0;
}

int lossless_device_files(LosslessPlan *p, const uint8_t **base, const uint64_t **offsets, const uint64_t **sizes) {
    p->h_file_bytes.resize(p->n_clips);
    for (size_t i = 0; i < p->n_clips; i++)
        p->h_file_bytes[i] = 74 + 20 * (uint64_t)(p->clip_first_frame[i + 1] - p->clip_first_frame[i]) + p->h_clip_bytes[i];
    if (base) *base = p->d_out;
    if (offsets) *offsets = p->clip_file_off.data();
    if (sizes) *sizes = p->h_file_bytes.data();
    return 0;
}

int lossless_describe(LosslessPlan *p, std::vector<LosslessFrameInfo> &frames, std::vector<LosslessWrapperInfo> &wrappers,
                      const uint8_t **base, std::string &err) {
    frames.clear();
    wrappers.clear();
    *base = p->d_out;
    if (p->h_fout.size() != p->frames.size()) {
        err = "lossless_describe: the batch has not been collected";
        return -1;
    }
    // (into pinned memory: a pageable destination made this copy 0.1 ms of every batch decode)
    if (p->n_chans && !p->h_chans_pin && hipHostMalloc((void **)&p->h_chans_pin, p->n_chans * sizeof(LLChan), hipHostMallocDefault) != hipSuccess) {
        p->h_chans_pin = nullptr;
        err = "lossless_describe: no pinned memory for the channel records";
        return -1;
    }
    const LLChan *chans = p->h_chans_pin;
    if (p->n_chans) {
        hipError_t e = hipMemcpy(p->h_chans_pin, p->d_chans, p->n_chans * sizeof(LLChan), hipMemcpyDeviceToHost);
        if (e != hipSuccess) {
            err = hipGetErrorString(e);
            return -1;
        }
    }
    frames.reserve(p->frames.size());
    wrappers.reserve(p->n_chans);
    for (size_t f = 0; f < p->frames.size(); f++) {
        const LLFrame &fr = p->frames[f];
        const LLFrameOut &fo = p->h_fout[f];
        LosslessFrameInfo fi{fr.clip, fr.frame_samples, fo.flags, (uint32_t)wrappers.size(), (uint32_t)p->ch};
        for (int c = 0; c < p->ch; c++) {
            const LLChan &ch = chans[fr.first_chan + c];
            LosslessWrapperInfo w{};
            const uint64_t payload = p->clip_out_off[fr.clip] + ch.payload_off + 4;   // behind the u32 size
            if (fo.frame_type == 0u) {                 // Silence: nothing to read
                w.off = payload;
            } else if (fo.frame_type == 254u) {        // Raw: at most frame_samples i16 (reader.rs raw branch)
                const uint64_t need = 2ull * fr.frame_samples;
                w.off = payload;
                w.len = (uint32_t)(need < ch.payload_size ? need : ch.payload_size);
            } else {                                   // ALPC wrapper: [order][coefficients][shift][encoding][k?][residuals]
                const int ncoef = ch.kind == 2 ? ch.order : 0;
                const uint32_t head = 1u + 4u * (uint32_t)ncoef + 1u + 1u + (ch.kind == 0 ? 0u : 1u);
                w.n_coeffs = (uint8_t)ncoef;
                for (int q = 0; q < ncoef && q < 12; q++) w.coeffs[q] = ch.coefs[q];
                w.shift_bits = (uint8_t)(ch.kind == 1 ? 128 + ch.order : (ch.kind == 2 ? ch.shift : 0));
                w.rice_k = (uint8_t)(ch.kind == 0 ? 0 : ch.k);   // encoding byte 2 (raw) carries no parameter
                w.off = payload + head;
                w.len = ch.payload_size > head ? ch.payload_size - head : 0u;
            }
            wrappers.push_back(w);
        }
        frames.push_back(fi);
    }
    return 0;
}

int lossless_fetch(LosslessPlan *p, size_t clip, uint8_t bit_depth, const uint8_t *meta, size_t meta_len, uint8_t **out,
                   size_t *out_len, std::string &err) {
    // the file was finished on the device: copy it, append META, patch bit_depth (header byte 13) and meta_size (62..69)
    const size_t nf = p->clip_first_frame[clip + 1] - p->clip_first_frame[clip];
    const size_t n = 74 + 20 * nf + (size_t)p->h_clip_bytes[clip];
    uint8_t *f = (uint8_t *)malloc(n + meta_len);
    if (!f) {
        err = "malloc failed";
        return -1;
    }
    hipError_t e = hipMemcpy(f, p->d_out + p->clip_file_off[clip], n, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        free(f);
        err = hipGetErrorString(e);
        return -1;
    }
    if (meta_len) memcpy(f + n, meta, meta_len);
    f[13] = bit_depth;
    for (int i = 0; i < 8; i++) f[62 + i] = (uint8_t)((uint64_t)meta_len >> (8 * i));
    *out = f;
    *out_len = n + meta_len;
    return 0;
}

}  // namespace flo
