// analysis_kernels.hpp — device side of the analysis metadata libflo's free encode functions add to the META chunk
// (lib.rs:219-283): waveform peaks, spectral fingerprint (BLAKE3 + 256-point FFT bands), EBU R128 block energies.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace flo {

struct AnalysisArgs {
    const float *pcm;            // one clip, interleaved
    unsigned long long n;        // interleaved samples
    unsigned int sample_rate;
    unsigned int channels;
    // waveform peaks (analysis.rs:38-115)
    double samples_per_peak;
    unsigned int n_peaks;
    float *peaks;                // [n_peaks] before normalisation
    // sequential scans: sum of squares (f32, in sample order) and K-weighted block sums per channel (ebu_r128.rs)
    double shelf[5], hp[5];      // b0 b1 b2 a1 a2
    unsigned int hop, n_blocks;
    // The order-bound scans run in SEGMENTS (see analysis_kernels.hip): a clip shorter than one segment is accumulated in
    // exactly the reference's order; a longer one restarts the filters `warm` frames ahead of every further segment.
    unsigned int seg_frames, warm_frames, n_seg;      // K-weighting: frames per segment, warm-up, segments per channel
    unsigned int sq_seg, n_sq_seg;                    // sum of squares: interleaved samples per segment, segments
    // Clips beyond one exact segment take the K-weighting in two passes over SHORT segments, one lane per segment (see
    // "K-weighting, long clips" in analysis_kernels.hip):
    unsigned int fast;           // 1: that path
    unsigned int kseg_frames, n_kseg, kq;   // frames per short segment, segments per channel, hop-quantum slots per segment
    double kpow[16];             // the filters' 4 x 4 state transition over kseg_frames steps (row-major)
    double *kstate;              // [channels][n_kseg][4]: pass 1 leaves zero-state END states, the scan turns them into START states
    double *kqpart;              // [channels][n_kseg][kq]: pass 2, the segment's share of the 100 ms quanta it overlaps
    // The f32 sum of squares of clips beyond one segment: chunks of 1024 samples summed in parallel and chained so that the
    // result IS the sequential f32 sum, bit for bit (see "sum of squares, long clips" in analysis_kernels.hip)
    unsigned int sq_exact;       // 1: that path (the result lands in sumsq_part[0])
    unsigned long long n_sq_chunks;
    double *sq_dsum;             // [n_sq_chunks + 1]: double-precision sum per chunk, then (in place) its exclusive prefix
    double *sq_rec;              // [n_sq_chunks][8]: R for the three candidate binades x both start parities | (guessed exponent, flags) packed in [6]
    float *sumsq_part;           // [n_sq_seg]: partial sums, added in order on the host
    double *block_part;          // [channels][n_blocks][2]: a 400 ms block's sum as written by the (at most two) segments it spans
    // sample peak and "true peak" (ebu_r128.rs:112-179, :211-217): largest |x| over whole frames, largest |FIR output|
    double tp_coef[49];
    unsigned long long *peak_bits;   // [2] bit patterns of non-negative doubles (atomicMax): sample peak, FIR peak
    double *peak_part;               // long clips: [tiles x channels][2] per-workgroup maxima (an_peak_kernel), reduced into peak_bits
    // BLAKE3 of (channels u8 | sample_rate u32 | len u32 | sample bytes)
    unsigned long long n_chunks;
    unsigned int *cvs;           // [2][n_chunks][8] ping-pong
    // FFT sections (analysis.rs:265-333)
    const float *fft_tw;         // [8][128][2]
    unsigned long long points[3];
    unsigned int point_ok[3];
    float *band_sqrt;            // [3][16]
    unsigned int *peak_bin;      // [3][8]
};

// The analysis is four independent chains of kernels - K-weighting, sum of squares, BLAKE3, peaks - each ending in a step
// that one wavefront walks alone (0.4 - 1 ms for a 3-minute clip). With `side` streams given they run side by side: fork
// behind what `s` holds at the call, join before the call returns (everything later on `s` sees all results).
struct AnalysisSide {
    hipStream_t st[3] = {nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr, join[3] = {nullptr, nullptr, nullptr};
};
int launch_analysis(const AnalysisArgs &A, hipStream_t s, const AnalysisSide *side = nullptr);

}  // namespace flo
