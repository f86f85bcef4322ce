// analysis_kernels.hip — gfx950 kernels behind flo_analysis_metadata (see analysis_kernels.hpp).
//
// Reference behaviour replaced (files under /root/reference/libflo/src):
//   extract_waveform_peaks ............. core/analysis.rs:38-115      an_peaks_kernel (one wave per peak window)
//   extract_spectral_fingerprint ....... core/analysis.rs:223-357     an_blake3_chunks / an_blake3_tree (the hash; BLAKE3
//                                        is a tree of 1 KiB chunks, so chunks hash in parallel), an_fft_kernel (the three
//                                        256-point sections), the f32 sum of squares in an_sumsq_kernel
//   compute_ebu_r128_loudness .......... core/ebu_r128.rs:112-266      an_loud_kernel: K-weighting (two biquads, f64), the
//                                        400 ms block sums, the sample peak and the true-peak FIR
// Everything the reference accumulates sequentially is accumulated in the same order here (the sums feed truncating
// casts to u8 and an f32 cast of the loudness, so the order matters for byte equality) - exactly so for clips up to one
// segment (65 536 frames; 65 536 interleaved samples for the sum of squares), in segments with a filter warm-up beyond (see "order-bound scans" below); only order-free work
// (maxima, the hash tree, butterflies, the FIR outputs) is spread over lanes. Compiled with -ffp-contract=off.
#include "analysis_kernels.hpp"

namespace flo {

#define AN_LAUNCH_CHECK()                      \
    do {                                       \
        hipError_t e_ = hipGetLastError();     \
        if (e_ != hipSuccess) return (int)e_;  \
    } while (0)

__device__ __forceinline__ float max_rust(float a, float b) {   // f32::max: a NaN operand is ignored
    if (a != a) return b;
    if (b != b) return a;
    return a > b ? a : b;
}

// ------------------------------------------------------------------------------------------------ waveform peaks
__global__ __launch_bounds__(64) void an_peaks_kernel(AnalysisArgs A) {
    const unsigned idx = blockIdx.x;
    if (idx >= A.n_peaks) return;
    const unsigned lane = threadIdx.x, ch = A.channels;
    unsigned long long start = (unsigned long long)((double)idx * A.samples_per_peak);
    unsigned long long end = (unsigned long long)(((double)idx + 1.0) * A.samples_per_peak);
    start *= ch;
    end *= ch;
    if (end > A.n) end = A.n;
    float peak = 0.f;
    if (start < A.n) {
        if (ch == 1) {
            float m = 0.f;
            for (unsigned long long i = start + lane; i < end; i += 64) m = max_rust(m, fabsf(A.pcm[i]));
            for (int o = 32; o; o >>= 1) m = max_rust(m, __shfl_xor(m, o));
            peak = m;
        } else if (ch == 2) {
            float l = 0.f, r = 0.f;
            for (unsigned long long i = start + 2ull * lane; i + 1 < end; i += 128) {
                l = max_rust(l, fabsf(A.pcm[i]));
                r = max_rust(r, fabsf(A.pcm[i + 1]));
            }
            for (int o = 32; o; o >>= 1) {
                l = max_rust(l, __shfl_xor(l, o));
                r = max_rust(r, __shfl_xor(r, o));
            }
            peak = (l + r) / 2.0f;
        } else {
            float m = 0.f;
            for (unsigned long long i = start + (unsigned long long)lane * ch; i < end; i += 64ull * ch) {
                const unsigned n = end - i < ch ? (unsigned)(end - i) : ch;
                float s = 0.f;
                for (unsigned k = 0; k < n; k++) s += A.pcm[i + k];
                m = max_rust(m, s / (float)n);
            }
            for (int o = 32; o; o >>= 1) m = max_rust(m, __shfl_xor(m, o));
            peak = m;
        }
    }
    if (lane == 0) A.peaks[idx] = peak;
}

// ------------------------------------------------------------------------------------------------ order-bound scans
// What the reference accumulates sample after sample - the f32 sum of squares (analysis.rs:338), the K-weighting
// recurrence and the 400 ms block sums (ebu_r128.rs:219-262) - is a dependent chain: one lane walks it. Round 2 walked a
// whole clip with ONE lane reading global memory sample by sample (180 ms for a 10 s clip, 3.2 s for three minutes: the
// load latency, not the arithmetic). Now:
//   * the samples come through LDS in tiles the whole workgroup loads (coalesced), the walking lane reads LDS;
//   * a clip is cut into SEGMENTS that run in parallel. The first segment starts from the reference's zero state, so a
//     clip shorter than one segment is bit for bit the reference's sequential result. Every further segment runs the two
//     biquads `warm_frames` ahead of its first frame from a zero state: the filters' slowest mode (the 38 Hz high-pass,
//     pole radius exp(-2 pi 38 / fs)) has decayed by exp(-59) over the quarter second of warm-up, twenty orders of
//     magnitude below a double's resolution, so the states agree with the sequential ones to the last bit or the one
//     before it; a block that straddles two segments is the sum of two partial sums. Loudness enters the META chunk as
//     an f32: long clips equal the sequential result to ~1e-15 relative in the block energies.
//   * the partial f32 sums of squares of the segments are added in order on the host.
// A second wave of the same workgroup evaluates, on the same tiles, what is order-free: the sample peak and the 49-tap
// FIR of compute_true_peak (each output is its own short sequential sum, taps in the reference's order).
constexpr int kAnTile = 2048;   // frames per tile
constexpr int kAnHalo = 24;     // (taps - 1) / 2

__device__ __forceinline__ void atomic_max_f64_bits(unsigned long long *p, double v) {
    if (v > 0.0) atomicMax(p, (unsigned long long)__double_as_longlong(v));
}

__global__ __launch_bounds__(128) void an_loud_kernel(AnalysisArgs A) {
    __shared__ double xt[kAnTile + 2 * kAnHalo];
    const unsigned seg = blockIdx.x, c = blockIdx.y, ch = A.channels;
    const unsigned tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned long long frames = A.n / ch;                                   // whole sample-frames (K-weighting, sample peak)
    const unsigned long long n_ch = A.n > c ? (A.n - c + ch - 1) / ch : 0;        // samples of channel c (the FIR walks these)
    const unsigned long long own0 = (unsigned long long)seg * A.seg_frames;
    unsigned long long own1 = own0 + A.seg_frames;
    const bool last = seg + 1 == A.n_seg;
    if (last) own1 = n_ch;                                                        // (n_ch >= frames: the last segment owns the tail)
    if (own0 >= n_ch) return;
    const unsigned long long run0 = own0 > A.warm_frames ? own0 - A.warm_frames : 0;
    const unsigned hop = A.hop;
    // walking lane's state. a0..a3 are the sums of the (up to four) blocks alive: a_j belongs to block k_start - j. They are
    // named registers that rotate when a block starts - an array indexed by k_start & 3 lives in scratch memory, and every
    // one of its four updates per sample then costs a memory round trip (that, not the arithmetic, was this kernel's time).
    double s1 = 0, s2 = 0, h1 = 0, h2 = 0, a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    unsigned long long k_start = hop ? own0 / hop : 0, next_edge = hop ? (k_start + 1) * (unsigned long long)hop : ~0ull;
    double *part = A.block_part + (unsigned long long)c * A.n_blocks * 2;
    auto put_block = [&](long long k, double v) {   // block k's sum as far as this segment saw it
        if (k >= 0 && (unsigned long long)k < A.n_blocks) part[2 * k + (((unsigned long long)k * hop >= own0) ? 0 : 1)] = v;
    };
    double peak_x = 0.0, peak_fir = 0.0;
    for (unsigned long long t0 = run0; t0 < own1; t0 += kAnTile) {
        __syncthreads();
        for (unsigned i = tid; i < kAnTile + 2 * kAnHalo; i += 128) {
            const long long f = (long long)t0 - kAnHalo + (long long)i;
            xt[i] = (f >= 0 && (unsigned long long)f < n_ch) ? (double)A.pcm[(unsigned long long)f * ch + c] : 0.0;
        }
        __syncthreads();
        const unsigned long long t1 = t0 + kAnTile < own1 ? t0 + kAnTile : own1;
        if (wave == 0) {
            if (lane == 0) {
                const unsigned long long e1 = t1 < frames ? t1 : frames;
                for (unsigned long long i8 = t0; i8 < e1; i8 += 8) {
                    double xv[8];   // eight LDS reads in flight ahead of the dependent chain
#pragma unroll
                    for (int j = 0; j < 8; j++) xv[j] = xt[(unsigned)(i8 - t0) + kAnHalo + j];   // (the tile has 24 spare entries behind it)
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const unsigned long long i = i8 + j;
                        if (i >= e1) break;
                        const double x = xv[j];
                        const double y = A.shelf[0] * x + s1;
                        s1 = A.shelf[1] * x - A.shelf[3] * y + s2;
                        s2 = A.shelf[2] * x - A.shelf[4] * y;
                        const double y2 = A.hp[0] * y + h1;
                        h1 = A.hp[1] * y - A.hp[3] * y2 + h2;
                        h2 = A.hp[2] * y - A.hp[4] * y2;
                        if (i < own0) continue;   // warm-up: the filters run, nothing is summed
                        if (i == next_edge) {
                            // a new block starts here; the block that started four hops ago ended with the previous sample
                            k_start++;
                            put_block((long long)k_start - 4, a3);
                            a3 = a2;
                            a2 = a1;
                            a1 = a0;
                            a0 = 0.0;
                            next_edge += hop;
                        }
                        const double e = y2 * y2;
                        // every block alive at this sample (accumulators of blocks before the clip's first are never read)
                        a0 += e;
                        a1 += e;
                        a2 += e;
                        a3 += e;
                    }
                }
            }
        } else {
            const unsigned long long b0 = t0 > own0 ? t0 : own0;
            for (unsigned long long i = b0 + lane; i < t1; i += 64) {
                const unsigned o = (unsigned)(i - t0);
                if (i < frames) {
                    const double a = fabs(xt[o + kAnHalo]);
                    if (a > peak_x) peak_x = a;   // (a NaN sample never wins, as with f64::max)
                }
                double a2 = 0.0;
#pragma unroll 7
                for (int k = 0; k < 49; k++) {
                    // taps whose sample lies outside the channel are skipped by the reference: the tile holds zeros there,
                    // and adding x * 0 = +-0 leaves the sum unchanged (a NaN or infinite sample cannot sit outside)
                    a2 += xt[o + k] * A.tp_coef[k];
                }
                a2 = fabs(a2);
                if (a2 > peak_fir) peak_fir = a2;
            }
        }
    }
    if (wave == 0) {
        if (lane == 0 && hop && own0 < frames) {
            // blocks still open when the segment ends: partial sums (a later segment adds its share) or, in the last
            // segment, the blocks that reach the end of the clip (n_blocks counts the ones the reference makes)
            put_block((long long)k_start - 3, a3);
            put_block((long long)k_start - 2, a2);
            put_block((long long)k_start - 1, a1);
            put_block((long long)k_start, a0);
        }
    } else {
        for (int o = 32; o; o >>= 1) {
            const double px = __shfl_xor(peak_x, o), pf = __shfl_xor(peak_fir, o);
            peak_x = px > peak_x ? px : peak_x;
            peak_fir = pf > peak_fir ? pf : peak_fir;
        }
        if (lane == 0) {
            atomic_max_f64_bits(A.peak_bits, peak_x);
            atomic_max_f64_bits(A.peak_bits + 1, peak_fir);
        }
    }
}

// ------------------------------------------------------------------------------------------------ K-weighting, long clips
// The recurrence is a chain, but a LINEAR one: with v = (s1, s2, h1, h2) the state of the two biquads, one step is
// v' = M v + b x. Over a segment of L frames, v_end = M^L v_start + z, where z is the end state the segment reaches from a
// ZERO start state. So: pass 1 walks every segment from zero (all segments at once, one LANE per segment: 64 walks per
// wavefront instead of one), a scan over the segments applies M^L (a 4 x 4 product per segment; M^L is made on the host by
// walking the homogeneous system) and leaves every segment's true start state, and pass 2 walks every segment again
// from that state, this time summing y^2 into the 100 ms quanta the 400 ms blocks are made of. No warm-up, two walks of
// L = 2048 frames instead of one of 65 536 + 11 025: the 10 ms this stage took per clip whatever its length become 0.3 ms.
// Against the sequential recurrence the start states differ by rounding (1e-16 relative, decaying with the filters'
// memory) and a block is the sum of its four quanta, themselves sums of the segments' shares, instead of one running sum:
// the block energies agree to ~1e-15, the f32 loudness of the META chunk is the same. Clips up to 65 536 frames keep the
// one-lane walk in the reference's own order (an_loud_kernel): bit for bit.
template <int PASS>
__global__ __launch_bounds__(64) void an_kw_pass_kernel(AnalysisArgs A) {
    const unsigned c = blockIdx.y, ch = A.channels;
    const unsigned long long s = (unsigned long long)blockIdx.x * 64 + threadIdx.x;
    if (s >= A.n_kseg) return;
    const unsigned long long frames = A.n / ch;
    const unsigned long long f0 = s * A.kseg_frames;
    const unsigned cnt = (unsigned)(f0 + A.kseg_frames < frames ? A.kseg_frames : frames - f0);   // frames of this segment
    double *st = A.kstate + ((unsigned long long)c * A.n_kseg + s) * 4;
    double s1 = 0, s2 = 0, h1 = 0, h2 = 0;
    if (PASS == 2) s1 = st[0], s2 = st[1], h1 = st[2], h2 = st[3];
    const double b0 = A.shelf[0], b1 = A.shelf[1], b2 = A.shelf[2], a1 = A.shelf[3], a2 = A.shelf[4];
    const double c0 = A.hp[0], c1 = A.hp[1], c2 = A.hp[2], d1 = A.hp[3], d2 = A.hp[4];
    const unsigned hop = A.hop;
    double acc = 0.0;
    unsigned slot = 0;
    // first quantum boundary behind f0, as an index into this segment (32-bit arithmetic inside the walk)
    unsigned edge = hop ? (unsigned)((f0 / hop + 1) * (unsigned long long)hop - f0) : 0xFFFFFFFFu;
    double *qp = A.kqpart + ((unsigned long long)c * A.n_kseg + s) * A.kq;
    const float *p = A.pcm + f0 * ch + c;
    for (unsigned i8 = 0; i8 < cnt; i8 += 8) {
        float xv[8];   // eight loads in flight ahead of the dependent chain (the lanes of a wave read 64 different lines: L1 hits from the second frame of a line on)
#pragma unroll
        for (int j = 0; j < 8; j++) xv[j] = i8 + j < cnt ? p[(unsigned long long)(i8 + j) * ch] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const unsigned i = i8 + j;
            // (frames behind the clip's end - the last segment only - run the filters on zeros: their states are never
            // used, and their squares are kept out of the sums by a select instead of a branch)
            const double x = (double)xv[j];
            const double y = b0 * x + s1;
            s1 = b1 * x - a1 * y + s2;
            s2 = b2 * x - a2 * y;
            const double y2 = c0 * y + h1;
            h1 = c1 * y - d1 * y2 + h2;
            h2 = c2 * y - d2 * y2;
            if (PASS == 2) {
                if (i == edge) {   // a quantum ends with the previous frame
                    qp[slot++] = acc;
                    acc = 0.0;
                    edge += hop;
                }
                const double e2 = y2 * y2;
                acc += i < cnt ? e2 : 0.0;
            }
        }
    }
    if (PASS == 1) st[0] = s1, st[1] = s2, st[2] = h1, st[3] = h2;
    else if (cnt) qp[slot] = acc;
}
// start states: v_0 = 0, v_{s+1} = M^L v_s + z_s (z_s = what pass 1 left). One wave per channel: 64 segments' z at a time
// (one coalesced read), handed to the chain by shuffles; each lane keeps the start state of its segment and writes it back.
__global__ __launch_bounds__(64) void an_kw_scan_kernel(AnalysisArgs A) {
    const unsigned c = blockIdx.x, lane = threadIdx.x;
    double v0 = 0, v1 = 0, v2 = 0, v3 = 0;
    double *st = A.kstate + (unsigned long long)c * A.n_kseg * 4;
    double P[16];
#pragma unroll
    for (int i = 0; i < 16; i++) P[i] = A.kpow[i];
    for (unsigned long long s0 = 0; s0 < A.n_kseg; s0 += 64) {
        const unsigned long long mine = s0 + lane;
        const bool have = mine < A.n_kseg;
        const double z0 = have ? st[4 * mine] : 0.0, z1 = have ? st[4 * mine + 1] : 0.0, z2 = have ? st[4 * mine + 2] : 0.0, z3 = have ? st[4 * mine + 3] : 0.0;
        double k0 = 0, k1 = 0, k2 = 0, k3 = 0;
        const unsigned cnt = A.n_kseg - s0 < 64 ? (unsigned)(A.n_kseg - s0) : 64u;
        for (unsigned jj = 0; jj < cnt; jj++) {
            const unsigned j = (unsigned)__builtin_amdgcn_readfirstlane((int)jj);
            if (lane == j) k0 = v0, k1 = v1, k2 = v2, k3 = v3;   // segment s0 + j starts here
            // (j is uniform: v_readlane, no trip through the LDS crossbar)
            auto bcast = [&](double v) {
                const long long b = __double_as_longlong(v);
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, (int)j), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), (int)j);
                return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
            };
            const double y0 = bcast(z0), y1 = bcast(z1), y2 = bcast(z2), y3 = bcast(z3);
            // (fused: these are start states, right to 1e-16 either way; half the operations on the chain)
            const double w0 = fma(P[0], v0, fma(P[1], v1, fma(P[2], v2, fma(P[3], v3, y0))));
            const double w1 = fma(P[4], v0, fma(P[5], v1, fma(P[6], v2, fma(P[7], v3, y1))));
            const double w2 = fma(P[8], v0, fma(P[9], v1, fma(P[10], v2, fma(P[11], v3, y2))));
            const double w3 = fma(P[12], v0, fma(P[13], v1, fma(P[14], v2, fma(P[15], v3, y3))));
            v0 = w0, v1 = w1, v2 = w2, v3 = w3;
        }
        if (have) st[4 * mine] = k0, st[4 * mine + 1] = k1, st[4 * mine + 2] = k2, st[4 * mine + 3] = k3;
    }
}
// sample peak and true-peak FIR of the whole clip (order-free: every output is its own sum, taps in the reference's order).
// A thread makes eight consecutive outputs from a window of 56 samples held in registers: seven LDS reads per output instead of 49.
__global__ __launch_bounds__(256) void an_peak_kernel(AnalysisArgs A) {
    __shared__ double xt[kAnTile + 2 * kAnHalo + 8];
    __shared__ double taps[49];
    const unsigned c = blockIdx.y, ch = A.channels, tid = threadIdx.x;
    const unsigned long long frames = A.n / ch;
    const unsigned long long n_ch = A.n > c ? (A.n - c + ch - 1) / ch : 0;
    const unsigned long long t0 = (unsigned long long)blockIdx.x * kAnTile;
    if (t0 >= n_ch) return;
    if (tid < 49) taps[tid] = A.tp_coef[tid];
    for (unsigned i = tid; i < kAnTile + 2 * kAnHalo + 8; i += 256) {
        const long long f = (long long)t0 - kAnHalo + (long long)i;
        xt[i] = (f >= 0 && (unsigned long long)f < n_ch) ? (double)A.pcm[(unsigned long long)f * ch + c] : 0.0;
    }
    __syncthreads();
    const unsigned long long t1 = t0 + kAnTile < n_ch ? t0 + kAnTile : n_ch;
    const unsigned o = 8 * tid;   // first of this thread's eight outputs (kAnTile = 8 x 256)
    double w[56];
#pragma unroll
    for (int k = 0; k < 56; k++) w[k] = xt[o + k];
    double a2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 49; k++) {   // every output adds its taps in ascending order, as the reference does
        const double ck = taps[k];   // (one broadcast read serves eight outputs; from the kernel arguments it was a scalar load per use)
#pragma unroll
        for (int q = 0; q < 8; q++) a2[q] += w[q + k] * ck;   // taps outside the channel meet zeros: x * 0 = +-0 leaves the sum unchanged
    }
    double peak_x = 0.0, peak_fir = 0.0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const unsigned long long i = t0 + o + q;
        if (i < t1) {
            if (i < frames) {
                const double a = fabs(w[q + kAnHalo]);
                if (a > peak_x) peak_x = a;   // (a NaN sample never wins, as with f64::max)
            }
            const double f = fabs(a2[q]);
            if (f > peak_fir) peak_fir = f;
        }
    }
    for (int sh = 32; sh; sh >>= 1) {
        const double px = __shfl_xor(peak_x, sh), pf = __shfl_xor(peak_fir, sh);
        peak_x = px > peak_x ? px : peak_x;
        peak_fir = pf > peak_fir ? pf : peak_fir;
    }
    // one pair of maxima per workgroup, reduced by an_peak_reduce_kernel (thousands of atomics on two addresses took
    // longer than the filter itself)
    __shared__ double wmax[2][4];
    if ((tid & 63) == 0) wmax[0][tid >> 6] = peak_x, wmax[1][tid >> 6] = peak_fir;
    __syncthreads();
    if (tid < 2) {
        double m = wmax[tid][0];
        for (int k = 1; k < 4; k++) m = wmax[tid][k] > m ? wmax[tid][k] : m;
        A.peak_part[2ull * ((unsigned long long)blockIdx.y * gridDim.x + blockIdx.x) + tid] = m;
    }
}
__global__ __launch_bounds__(256) void an_peak_reduce_kernel(AnalysisArgs A, unsigned long long n_part) {
    __shared__ double wmax[2][4];
    double mx = 0.0, mf = 0.0;
    for (unsigned long long i = threadIdx.x; i < n_part; i += 256) {
        const double a = A.peak_part[2 * i], b = A.peak_part[2 * i + 1];
        mx = a > mx ? a : mx;
        mf = b > mf ? b : mf;
    }
    for (int sh = 32; sh; sh >>= 1) {
        const double px = __shfl_xor(mx, sh), pf = __shfl_xor(mf, sh);
        mx = px > mx ? px : mx;
        mf = pf > mf ? pf : mf;
    }
    if ((threadIdx.x & 63) == 0) wmax[0][threadIdx.x >> 6] = mx, wmax[1][threadIdx.x >> 6] = mf;
    __syncthreads();
    if (threadIdx.x < 2) {
        double m = wmax[threadIdx.x][0];
        for (int k = 1; k < 4; k++) m = wmax[threadIdx.x][k] > m ? wmax[threadIdx.x][k] : m;
        atomic_max_f64_bits(A.peak_bits + threadIdx.x, m);
    }
}

// ------------------------------------------------------------------------------------------------ sum of squares, long clips
// analysis.rs:338 adds s * s into ONE f32 accumulator, sample after sample. That sum is not associative - for minutes of
// audio it is off the true sum by up to a per cent (terms far below the accumulator's ulp), and `avg_loudness` is a
// truncating cast of its logarithm - so partial sums added afterwards do not reproduce it. But every term is >= 0: the
// accumulator S only grows, and while it stays inside one binade [2^e, 2^(e+1)) it is a multiple of u = 2^(e-23) and one
// addition is S <- S + u * rne(t / u): an INTEGER increment that depends on the term alone - unless t / u falls exactly
// half-way between two integers (the tie goes to the even neighbour, which depends on S). So:
//   1. chunks of 1024 samples are summed in double, and a prefix over the chunks predicts S at every chunk's start, hence
//      its binade e (an_sq_dsum_kernel, an_sq_prefix_kernel);
//   2. every chunk adds up its integer increments R = sum rne(t / u) for the predicted binade and its two neighbours
//      (the sequential sum drifts off the true one), for both parities of S / u at its start (ties: see
//      an_sq_terms_kernel), and notes non-finite terms and terms of 4 * 2^e and more (R < 2^35 is exact in a double);
//   3. one wave chains the chunks: when S sits in one of a chunk's three binades, the chunk has no oddity and S + u R
//      stays below 2^(e+1) - then no intermediate sum left the binade either, S being monotone - the chunk is ONE
//      addition; otherwise (the first chunk, binade crossings, NaN or infinite samples) it is walked sample by sample
//      (an_sq_chain_kernel).
// The result is the reference's sum bit for bit, at any length; a three-minute clip has ~30 walked chunks of 15 000.
constexpr int kSqChunk = 1024;
__device__ __forceinline__ int sq_binade(float S) {   // exponent e with S in [2^e, 2^(e+1)); far out of range for 0, tiny, inf, NaN
    if (!(S >= 1e-30f) || !(S < 1e30f)) return -100000;
    int e;
    frexpf(S, &e);
    return e - 1;
}
__global__ __launch_bounds__(64) void an_sq_dsum_kernel(AnalysisArgs A) {
    const unsigned long long c = blockIdx.x;
    const unsigned long long i0 = c * kSqChunk + 16ull * threadIdx.x;
    double d = 0.0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const float s = i0 + j < A.n ? A.pcm[i0 + j] : 0.f;
        const float t = s * s;
        d += (double)t;
    }
    for (int o = 32; o; o >>= 1) d += __shfl_xor(d, o);
    if (threadIdx.x == 0) A.sq_dsum[c] = d;
}
// exclusive prefix of the chunk sums, in place (one workgroup; a NaN or infinite chunk poisons what follows: those chunks
// are walked)
__global__ __launch_bounds__(256) void an_sq_prefix_kernel(AnalysisArgs A) {
    __shared__ double wsum[4];
    __shared__ double carry;
    if (threadIdx.x == 0) carry = 0.0;
    __syncthreads();
    for (unsigned long long base = 0; base < A.n_sq_chunks; base += 256) {
        const unsigned long long i = base + threadIdx.x;
        const double v = i < A.n_sq_chunks ? A.sq_dsum[i] : 0.0;
        double x = v;
        for (int o = 1; o < 64; o <<= 1) {
            const double y = __shfl_up(x, o);
            if ((int)(threadIdx.x & 63) >= o) x += y;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = x;
        __syncthreads();
        double off = carry;
        for (unsigned w = 0; w < (threadIdx.x >> 6); w++) off += wsum[w];
        if (i < A.n_sq_chunks) A.sq_dsum[i] = off + (x - v);
        __syncthreads();
        if (threadIdx.x == 255) carry = off + x;
        __syncthreads();
    }
}
// A tie - t / u exactly half-way between two integers - rounds to the EVEN neighbour, so its increment depends on the
// parity of S / u when it is added; 16-bit material is full of them (s = m / 32768 gives t / u = m^2 / 2^(7 + e): a tie
// whenever m^2 has exactly 6 + e trailing zero bits, one sample in a few hundred in every other binade). Parity is a
// two-state automaton: a term without tie flips it by its increment's low bit, a tie leaves it EVEN whatever it was. Those
// maps compose associatively, so a lane summarises its sixteen terms as (sum, parity out) for both parities in, one wave
// scan composes the maps of the lanes in front of each lane, and the chunk's increment comes out for both parities of
// S / u at the chunk's start - the chain picks the one S really has (the low bit of its mantissa).
__global__ __launch_bounds__(64) void an_sq_terms_kernel(AnalysisArgs A) {
    const unsigned long long c = blockIdx.x;
    const unsigned lane = threadIdx.x;
    const unsigned long long i0 = c * kSqChunk + 16ull * lane;
    const int eg = sq_binade((float)A.sq_dsum[c]);
    float t[16];
    unsigned flags = 0;   // bit 3 + k: a term of 4 * 2^e or more under candidate k | bit 6: non-finite term
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const float s = i0 + j < A.n ? A.pcm[i0 + j] : 0.f;
        t[j] = s * s;
        if (!(t[j] <= 3.0e38f)) flags |= 64u;
    }
    double Rk[3][2];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        // this lane's sixteen terms for parity q in: sum of increments and parity out
        double sum[2] = {0.0, 0.0};
        unsigned par[2] = {0u, 1u};
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const float sc = ldexpf(t[j], 23 - (eg + k - 1));   // t / u, exact (a power of two)
            if (!(sc < 33554432.0f)) flags |= 8u << k;
            const float fl = floorf(sc), fr = sc - fl;
            const unsigned fi = (unsigned)(int)fminf(fl, 33554432.0f);   // (an unsafe chunk is walked anyway)
            if (fr == 0.5f) {
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    sum[q] += (double)fl + (double)((par[q] + fi) & 1u);
                    par[q] = 0u;
                }
            } else {
                const unsigned up = fr > 0.5f ? 1u : 0u;
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    sum[q] += (double)fl + (double)up;
                    par[q] ^= (fi + up) & 1u;
                }
            }
        }
        // maps of the lanes in front: F = parity out for parity in 0 | for parity in 1 << 1; inclusive scan by composition
        unsigned F = par[0] | (par[1] << 1);
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned G = (unsigned)__shfl_up((int)F, o);   // the lanes further in front act first
            if ((int)lane >= o) F = ((F >> (G & 1u)) & 1u) | (((F >> ((G >> 1) & 1u)) & 1u) << 1);
        }
        unsigned E = (unsigned)__shfl_up((int)F, 1);   // exclusive: what reaches this lane
        if (lane == 0) E = 2u;                          // identity
#pragma unroll
        for (int P = 0; P < 2; P++) {
            double v = sum[(E >> P) & 1u];
            for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
            Rk[k][P] = v;
        }
    }
    for (int o = 32; o; o >>= 1) flags |= (unsigned)__shfl_xor((int)flags, o);
    if (lane == 0) {
        double *r = A.sq_rec + 8 * c;
#pragma unroll
        for (int k = 0; k < 3; k++) r[2 * k] = Rk[k][0], r[2 * k + 1] = Rk[k][1];
        r[6] = __longlong_as_double(((long long)(eg + 200000) << 32) | (long long)flags);
    }
}
// The chain keeps S as an integer mantissa M in [2^23, 2^24) and its binade e while it can (S = M 2^(e-23)): a chunk is then
// M += R. Sixty-four chunks go at once: each lane picks its chunk's R for both start parities, the parity maps of the
// chunks (out = in + R(in) mod 2) are composed by the same scan as inside a chunk, a prefix sum over the lanes gives
// every chunk's start - and if the last sum is still below 2^24 no chunk of the sixty-four left the binade (M only grows).
// Otherwise the sixty-four are taken one by one, and a chunk that cannot be ONE addition is walked sample by sample.
__global__ __launch_bounds__(64) void an_sq_chain_kernel(AnalysisArgs A) {
    __shared__ float xs[kSqChunk];
    const unsigned lane = threadIdx.x;
    float S = 0.f;        // every lane carries the same value: the chain is uniform
    unsigned walked = 0;
    double nx[7];   // the NEXT sixty-four chunks' records, one per lane, fetched while the current ones are chained
    {
        const unsigned long long m0 = lane < A.n_sq_chunks ? lane : A.n_sq_chunks - 1;
#pragma unroll
        for (int q = 0; q < 7; q++) nx[q] = A.sq_rec[8 * m0 + q];
    }
    for (unsigned long long c0 = 0; c0 < A.n_sq_chunks; c0 += 64) {
        const unsigned cnt = A.n_sq_chunks - c0 < 64 ? (unsigned)(A.n_sq_chunks - c0) : 64u;
        double rr[6];
#pragma unroll
        for (int q = 0; q < 6; q++) rr[q] = nx[q];
        const long long pk = __double_as_longlong(nx[6]);
        {
            const unsigned long long m1 = c0 + 64 + lane < A.n_sq_chunks ? c0 + 64 + lane : A.n_sq_chunks - 1;
#pragma unroll
            for (int q = 0; q < 7; q++) nx[q] = A.sq_rec[8 * m1 + q];
        }
        const int eg = (int)(pk >> 32) - 200000;
        const unsigned flags = (unsigned)pk;
        unsigned j0 = 0;
        {   // all at once?
            const int e = sq_binade(S);
            const int k = e - (eg - 1);
            const bool ok = lane >= cnt || (k >= 0 && k < 3 && !(flags & ((8u << (k & 3)) | 64u)));
            if (e > -1000 && __ballot(ok) == ~0ull) {   // uniform
                double r0 = rr[0], r1 = rr[1];
#pragma unroll
                for (int q = 1; q < 3; q++) {
                    r0 = k == q ? rr[2 * q] : r0;
                    r1 = k == q ? rr[2 * q + 1] : r1;
                }
                const long long i0 = lane < cnt ? (long long)r0 : 0ll, i1 = lane < cnt ? (long long)r1 : 0ll;
                unsigned F = ((unsigned)i0 & 1u) | ((((unsigned)i1 + 1u) & 1u) << 1);   // parity out for parity in 0 | in 1
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned G = (unsigned)__shfl_up((int)F, o);
                    if ((int)lane >= o) F = ((F >> (G & 1u)) & 1u) | (((F >> ((G >> 1) & 1u)) & 1u) << 1);
                }
                unsigned E = (unsigned)__shfl_up((int)F, 1);
                if (lane == 0) E = 2u;
                const unsigned M = (__float_as_uint(S) & 0x7FFFFFu) | 0x800000u;
                const unsigned pin = (E >> (M & 1u)) & 1u;
                long long x = pin ? i1 : i0;
                for (int o = 1; o < 64; o <<= 1) {
                    const long long y = __shfl_up(x, o);
                    if ((int)lane >= o) x += y;
                }
                const long long total = __shfl(x, 63);
                if ((long long)M + total < (1ll << 24)) {
                    S = __uint_as_float((__float_as_uint(S) & 0xFF800000u) | ((unsigned)((long long)M + total) & 0x7FFFFFu));
                    j0 = cnt;   // done
                }
            }
        }
        for (unsigned j = j0; j < cnt; j++) {   // one by one
            const long long pj = __shfl(pk, (int)j);
            const int egj = (int)(pj >> 32) - 200000;
            const unsigned fj = (unsigned)pj;
            const int e = sq_binade(S), k = e - (egj - 1);
            bool done = false;
            if (k >= 0 && k < 3 && !(fj & ((8u << k) | 64u))) {
                const unsigned P = __float_as_uint(S) & 1u;   // parity of S / u: the low bit of the mantissa
                const int idx = 2 * k + (int)P;
                double mineR = rr[0];
#pragma unroll
                for (int q = 1; q < 6; q++) mineR = idx == q ? rr[q] : mineR;
                const long long R = (long long)__shfl(mineR, (int)j);
                const long long M = (long long)((__float_as_uint(S) & 0x7FFFFFu) | 0x800000u);
                if (M + R < (1ll << 24)) {
                    S = __uint_as_float((__float_as_uint(S) & 0xFF800000u) | ((unsigned)(M + R) & 0x7FFFFFu));
                    done = true;
                }
            }
            if (!done) {   // uniform: the chunk is walked in the reference's own order
                walked++;
                const unsigned long long i0 = (c0 + j) * kSqChunk;
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const unsigned long long i = i0 + 64ull * q + lane;
                    xs[64 * q + lane] = i < A.n ? A.pcm[i] : 0.f;
                }
                __syncthreads();
                const unsigned m = A.n - i0 < (unsigned long long)kSqChunk ? (unsigned)(A.n - i0) : (unsigned)kSqChunk;
                for (unsigned i = 0; i < m; i++) {
                    const float s = xs[i];
                    S += s * s;
                }
            }
        }
    }
    if (lane == 0) {
        A.sumsq_part[0] = S;
        A.sumsq_part[1] = (float)walked;   // diagnostic (FLO_TRACE): chunks that took the sample-by-sample walk
    }
}

// f32 sum of s * s over the interleaved samples in order (analysis.rs:338), one wave per segment of A.sq_seg samples
__global__ __launch_bounds__(64) void an_sumsq_kernel(AnalysisArgs A) {
    __shared__ float xs[4096];
    const unsigned long long b0 = (unsigned long long)blockIdx.x * A.sq_seg;
    const unsigned long long b1 = b0 + A.sq_seg < A.n ? b0 + A.sq_seg : A.n;
    float acc = 0.f;
    for (unsigned long long t0 = b0; t0 < b1; t0 += 4096) {
        const unsigned cnt = b1 - t0 < 4096 ? (unsigned)(b1 - t0) : 4096u;
        for (unsigned i = threadIdx.x; i < cnt; i += 64) xs[i] = A.pcm[t0 + i];
        __syncthreads();
        if (threadIdx.x == 0)
            for (unsigned i = 0; i < cnt; i++) {
                const float s = xs[i];
                acc += s * s;
            }
        __syncthreads();
    }
    if (threadIdx.x == 0) A.sumsq_part[blockIdx.x] = acc;
}

// ------------------------------------------------------------------------------------------------ BLAKE3
__device__ __constant__ unsigned int kB3IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
__device__ __forceinline__ unsigned int rotr(unsigned int x, int n) { return (x >> n) | (x << (32 - n)); }
#define B3G(a, b, c, d, mx, my)        \
    do {                               \
        a = a + b + (mx);              \
        d = rotr(d ^ a, 16);           \
        c = c + d;                     \
        b = rotr(b ^ c, 12);           \
        a = a + b + (my);              \
        d = rotr(d ^ a, 8);            \
        c = c + d;                     \
        b = rotr(b ^ c, 7);            \
    } while (0)
// one compression: cv (8 words, updated in place to the new chaining value) with block m[16]
__device__ __forceinline__ void b3_compress(unsigned int (&cv)[8], const unsigned int (&mi)[16], unsigned long long counter,
                                            unsigned int block_len, unsigned int flags) {
    unsigned int v0 = cv[0], v1 = cv[1], v2 = cv[2], v3 = cv[3], v4 = cv[4], v5 = cv[5], v6 = cv[6], v7 = cv[7];
    unsigned int v8 = kB3IV[0], v9 = kB3IV[1], v10 = kB3IV[2], v11 = kB3IV[3];
    unsigned int v12 = (unsigned int)counter, v13 = (unsigned int)(counter >> 32), v14 = block_len, v15 = flags;
    unsigned int m[16];
#pragma unroll
    for (int i = 0; i < 16; i++) m[i] = mi[i];
#pragma unroll
    for (int r = 0; r < 7; r++) {
        B3G(v0, v4, v8, v12, m[0], m[1]);
        B3G(v1, v5, v9, v13, m[2], m[3]);
        B3G(v2, v6, v10, v14, m[4], m[5]);
        B3G(v3, v7, v11, v15, m[6], m[7]);
        B3G(v0, v5, v10, v15, m[8], m[9]);
        B3G(v1, v6, v11, v12, m[10], m[11]);
        B3G(v2, v7, v8, v13, m[12], m[13]);
        B3G(v3, v4, v9, v14, m[14], m[15]);
        const unsigned int t[16] = {m[2], m[6], m[3], m[10], m[7], m[0], m[4], m[13], m[1], m[11], m[12], m[5], m[9], m[14], m[15], m[8]};
#pragma unroll
        for (int i = 0; i < 16; i++) m[i] = t[i];
    }
    cv[0] = v0 ^ v8; cv[1] = v1 ^ v9; cv[2] = v2 ^ v10; cv[3] = v3 ^ v11;
    cv[4] = v4 ^ v12; cv[5] = v5 ^ v13; cv[6] = v6 ^ v14; cv[7] = v7 ^ v15;
}
// byte `pos` of the hashed message: 9 bytes of format information (analysis.rs:246-249), then the sample bytes
__device__ __forceinline__ unsigned int msg_byte(const AnalysisArgs &A, unsigned long long pos, unsigned long long total) {
    if (pos >= total) return 0u;
    if (pos == 0) return A.channels & 0xFFu;
    if (pos < 5) return (A.sample_rate >> (8 * (pos - 1))) & 0xFFu;
    if (pos < 9) return ((unsigned int)A.n >> (8 * (pos - 5))) & 0xFFu;
    return reinterpret_cast<const unsigned char *>(A.pcm)[pos - 9];
}
__global__ void an_blake3_chunks_kernel(AnalysisArgs A) {
    const unsigned long long c = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= A.n_chunks) return;
    const unsigned long long total = 9ull + 4ull * A.n;
    const unsigned long long off = c * 1024ull;
    const unsigned long long len = total - off < 1024ull ? total - off : 1024ull;
    const unsigned nblocks = len ? (unsigned)((len + 63) / 64) : 1u;
    unsigned int cv[8];
#pragma unroll
    for (int i = 0; i < 8; i++) cv[i] = kB3IV[i];
    const unsigned int *words = reinterpret_cast<const unsigned int *>(A.pcm);
    for (unsigned b = 0; b < nblocks; b++) {
        const unsigned long long bo = off + 64ull * b;
        const unsigned take = len - 64ull * b < 64ull ? (unsigned)(len - 64ull * b) : 64u;
        unsigned int m[16];
        if (bo >= 12 && bo + 64 <= total) {
            // message word j = sample bytes 4j - 9 .. 4j - 6: the top byte of sample word j - 3 and three of word j - 2
            const unsigned long long w0 = bo / 4 - 3;
            unsigned int prev = words[w0];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const unsigned int cur = words[w0 + 1 + i];
                m[i] = (prev >> 24) | (cur << 8);
                prev = cur;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const unsigned long long p = bo + 4ull * i;
                m[i] = msg_byte(A, p, total) | (msg_byte(A, p + 1, total) << 8) | (msg_byte(A, p + 2, total) << 16) | (msg_byte(A, p + 3, total) << 24);
            }
        }
        unsigned int flags = (b == 0 ? 1u : 0u) | (b == nblocks - 1 ? 2u : 0u);
        if (A.n_chunks == 1 && b == nblocks - 1) flags |= 8u;   // the only chunk is the root
        b3_compress(cv, m, c, take, flags);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) A.cvs[c * 8 + i] = cv[i];
}
// The tree: pairs are merged level by level, an odd last node is carried up unchanged, the last merge is the root.
// One workgroup walks all levels (a 10-second stereo clip has 3446 chunks: twelve levels).
__global__ __launch_bounds__(256) void an_blake3_tree_kernel(AnalysisArgs A) {
    unsigned long long n = A.n_chunks;
    unsigned int *src = A.cvs, *dst = A.cvs + A.n_chunks * 8;
    while (n > 1) {
        const unsigned long long pairs = n / 2;
        for (unsigned long long p = threadIdx.x; p < pairs; p += blockDim.x) {
            unsigned int cv[8], m[16];
#pragma unroll
            for (int i = 0; i < 8; i++) cv[i] = kB3IV[i];
#pragma unroll
            for (int i = 0; i < 16; i++) m[i] = src[2 * p * 8 + i];
            b3_compress(cv, m, 0ull, 64u, 4u | (n == 2 ? 8u : 0u));
#pragma unroll
            for (int i = 0; i < 8; i++) dst[p * 8 + i] = cv[i];
        }
        if ((n & 1) && threadIdx.x == 0)
            for (int i = 0; i < 8; i++) dst[pairs * 8 + i] = src[(n - 1) * 8 + i];
        __syncthreads();
        __threadfence_block();
        n = pairs + (n & 1);
        unsigned int *t = src;
        src = dst;
        dst = t;
        __syncthreads();
    }
    if (threadIdx.x < 8) A.cvs[2 * A.n_chunks * 8 + threadIdx.x] = src[threadIdx.x];   // the root, behind the two buffers
}

// ------------------------------------------------------------------------------------------------ FFT sections
// One workgroup of 128 threads per analysis point: mono mix-down, bit reversal, eight radix-2 stages (the butterfly
// arithmetic and the twiddle values of the oracle's FFT, two products and one sum per component, nothing fused), then the
// band energies summed bin by bin in ascending order and the per-band peak bins (analysis.rs:281-333).
__global__ __launch_bounds__(128) void an_fft_kernel(AnalysisArgs A) {
    __shared__ float zr[256], zi[256];
    const unsigned p = blockIdx.x, t = threadIdx.x;
    if (!A.point_ok[p]) return;
    const unsigned ch = A.channels;
    for (unsigned i = t; i < 256; i += 128) {
        float s = 0.f;
        for (unsigned c = 0; c < ch; c++) {
            const unsigned long long idx = (A.points[p] + i) * ch + c;
            if (idx < A.n) s += A.pcm[idx];
        }
        s /= (float)ch;
        const unsigned j = __brev(i) >> 24;
        zr[j] = s;
        zi[j] = 0.f;
    }
    __syncthreads();
    for (int s = 0; s < 8; s++) {
        const unsigned half = 1u << s, len = half << 1;
        const unsigned k = t & (half - 1), st = (t >> s) * len;
        const float wr = A.fft_tw[(s * 128 + k) * 2], wi = A.fft_tw[(s * 128 + k) * 2 + 1];
        const float ar = zr[st + k], ai = zi[st + k], br = zr[st + k + half], bi = zi[st + k + half];
        const float tr = br * wr - bi * wi;
        const float ti = br * wi + bi * wr;
        __syncthreads();
        zr[st + k] = ar + tr;
        zi[st + k] = ai + ti;
        zr[st + k + half] = ar - tr;
        zi[st + k + half] = ai - ti;
        __syncthreads();
    }
    if (t < 16) {
        const unsigned sb = t * 8, eb = (t + 1) * 8 < 128 ? (t + 1) * 8 : 128;
        float energy = 0.f;
        for (unsigned b = sb; b < eb; b++) energy += zr[b] * zr[b] + zi[b] * zi[b];
        A.band_sqrt[p * 16 + t] = __fsqrt_rn(energy);
    } else if (t >= 32 && t < 40) {
        const unsigned band = t - 32, sb = band * 16, eb = (band + 1) * 16 < 128 ? (band + 1) * 16 : 128;
        unsigned best = 0;
        float bestv = 0.f;
        bool have = false;
        for (unsigned b = sb; b < eb; b++) {   // Iterator::max_by: the last of several equal maxima
            const float v = __fsqrt_rn(zr[b] * zr[b] + zi[b] * zi[b]);
            if (!have || !(v < bestv)) {
                best = b;
                bestv = v;
                have = true;
            }
        }
        A.peak_bin[p * 8 + band] = best;
    }
}

int launch_analysis(const AnalysisArgs &A, hipStream_t s, const AnalysisSide *side) {
    // s0: peaks of the waveform + K-weighting; s1: true / sample peak; s2: sum of squares; s3: BLAKE3 + spectrum
    hipStream_t s1 = s, s2 = s, s3 = s;
    if (side && A.n) {
        if (hipEventRecord(side->fork, s) != hipSuccess) return -1;
        for (int i = 0; i < 3; i++)
            if (hipStreamWaitEvent(side->st[i], side->fork, 0) != hipSuccess) return -1;
        s1 = side->st[0], s2 = side->st[1], s3 = side->st[2];
    }
    if (A.n_peaks) {
        hipLaunchKernelGGL(an_peaks_kernel, dim3(A.n_peaks), dim3(64), 0, s, A);
        AN_LAUNCH_CHECK();
    }
    if (A.n) {
        if (A.fast) {
            hipLaunchKernelGGL(an_kw_pass_kernel<1>, dim3((A.n_kseg + 63) / 64, A.channels), dim3(64), 0, s, A);
            AN_LAUNCH_CHECK();
            hipLaunchKernelGGL(an_kw_scan_kernel, dim3(A.channels), dim3(64), 0, s, A);
            AN_LAUNCH_CHECK();
            hipLaunchKernelGGL(an_kw_pass_kernel<2>, dim3((A.n_kseg + 63) / 64, A.channels), dim3(64), 0, s, A);
            AN_LAUNCH_CHECK();
            const unsigned long long longest = (A.n + A.channels - 1) / A.channels;
            const unsigned tiles = (unsigned)((longest + kAnTile - 1) / kAnTile);
            hipLaunchKernelGGL(an_peak_kernel, dim3(tiles, A.channels), dim3(256), 0, s1, A);
            AN_LAUNCH_CHECK();
            hipLaunchKernelGGL(an_peak_reduce_kernel, dim3(1), dim3(256), 0, s1, A, (unsigned long long)tiles * A.channels);
            AN_LAUNCH_CHECK();
        } else {
            hipLaunchKernelGGL(an_loud_kernel, dim3(A.n_seg, A.channels), dim3(128), 0, s, A);
            AN_LAUNCH_CHECK();
        }
        if (A.sq_exact) {
            hipLaunchKernelGGL(an_sq_dsum_kernel, dim3((unsigned)A.n_sq_chunks), dim3(64), 0, s2, A);
            AN_LAUNCH_CHECK();
            hipLaunchKernelGGL(an_sq_prefix_kernel, dim3(1), dim3(256), 0, s2, A);
            AN_LAUNCH_CHECK();
            hipLaunchKernelGGL(an_sq_terms_kernel, dim3((unsigned)A.n_sq_chunks), dim3(64), 0, s2, A);
            AN_LAUNCH_CHECK();
            hipLaunchKernelGGL(an_sq_chain_kernel, dim3(1), dim3(64), 0, s2, A);
            AN_LAUNCH_CHECK();
        } else {
            hipLaunchKernelGGL(an_sumsq_kernel, dim3(A.n_sq_seg), dim3(64), 0, s2, A);
            AN_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(an_blake3_chunks_kernel, dim3((unsigned)((A.n_chunks + 127) / 128)), dim3(128), 0, s3, A);
        AN_LAUNCH_CHECK();
        hipLaunchKernelGGL(an_blake3_tree_kernel, dim3(1), dim3(256), 0, s3, A);
        AN_LAUNCH_CHECK();
        hipLaunchKernelGGL(an_fft_kernel, dim3(3), dim3(128), 0, s3, A);
        AN_LAUNCH_CHECK();
    }
    if (side && A.n)
        for (int i = 0; i < 3; i++)
            if (hipEventRecord(side->join[i], side->st[i]) != hipSuccess || hipStreamWaitEvent(s, side->join[i], 0) != hipSuccess) return -1;
    return 0;
}

}  // namespace flo
