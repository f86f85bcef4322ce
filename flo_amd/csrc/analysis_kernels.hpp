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
    float *sumsq_part;           // [n_sq_seg]: partial sums, added in order on the host
    double *block_part;          // [channels][n_blocks][2]: a 400 ms block's sum as written by the (at most two) segments it spans
    // sample peak and "true peak" (ebu_r128.rs:112-179, :211-217): largest |x| over whole frames, largest |FIR output|
    double tp_coef[49];
    unsigned long long *peak_bits;   // [2] bit patterns of non-negative doubles (atomicMax): sample peak, FIR peak
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

int launch_analysis(const AnalysisArgs &A, hipStream_t s);

}  // namespace flo
