// analysis_kernels.hip — gfx950 kernels behind flo_analysis_metadata (see analysis_kernels.hpp).
//
// Reference behaviour replaced (files under /root/reference/libflo/src):
//   extract_waveform_peaks ............. core/analysis.rs:38-115      an_peaks_kernel (one wave per peak window)
//   extract_spectral_fingerprint ....... core/analysis.rs:223-357     an_blake3_chunks / an_blake3_tree (the hash; BLAKE3
//                                        is a tree of 1 KiB chunks, so chunks hash in parallel), an_fft_kernel (the three
//                                        256-point sections), the sequential f32 sum of squares in an_scan_kernel
//   compute_ebu_r128_loudness .......... core/ebu_r128.rs:182-266      an_scan_kernel: K-weighting (two biquads, f64) and the
//                                        400 ms block sums, one lane per channel walking the samples in order
// Everything the reference accumulates sequentially is accumulated in the same order here (the sums feed truncating
// casts to u8 and an f32 cast of the loudness, so the order matters for byte equality); only order-free work (maxima,
// the hash tree, butterflies) is spread over lanes. Compiled with -ffp-contract=off like the other kernels.
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

// ------------------------------------------------------------------------------------------------ sequential scans
// thread 0: sum of s*s over all samples, f32, in order (analysis.rs:338). thread 1 + c: channel c through the
// K-weighting filter, its squares summed into the (up to four) 400 ms blocks that contain the sample, each block's
// sum in sample order (ebu_r128.rs:219-262).
__global__ __launch_bounds__(64) void an_scan_kernel(AnalysisArgs A) {
    // one WAVE per scan (lane 0 works): scans in one wave would run one after the other
    if (threadIdx.x != 0) return;
    const unsigned t = blockIdx.x;
    const unsigned ch = A.channels;
    if (t == 0) {
        float acc = 0.f;
        for (unsigned long long i = 0; i < A.n; i++) {
            const float s = A.pcm[i];
            acc += s * s;
        }
        A.sumsq[0] = acc;
        return;
    }
    const unsigned c = t - 1;
    if (c >= ch) return;
    const unsigned long long frames = A.n / ch;
    const unsigned hop = A.hop;
    double s1 = 0, s2 = 0, h1 = 0, h2 = 0;
    double acc[4] = {0, 0, 0, 0};
    double *out = A.block_sums + (unsigned long long)c * A.n_blocks;
    // block k covers frames [k hop, min(k hop + 4 hop, frames)); the last block is the first one that reaches `frames`
    unsigned long long next_edge = hop;   // frame index at which a block starts (and, four hops later, one ends)
    unsigned k_start = 0;                 // blocks started so far - 1 = index of the newest block
    for (unsigned long long i = 0; i < frames; i++) {
        if (hop && i == next_edge) {
            // a new block starts here; the block that started four hops ago ended with the previous sample
            k_start++;
            if (k_start >= 4) {
                const unsigned done = k_start - 4;
                if (done < A.n_blocks) out[done] = acc[done & 3];
            }
            acc[k_start & 3] = 0.0;
            next_edge += hop;
        }
        const double x = (double)A.pcm[i * ch + c];
        const double y = A.shelf[0] * x + s1;
        s1 = A.shelf[1] * x - A.shelf[3] * y + s2;
        s2 = A.shelf[2] * x - A.shelf[4] * y;
        const double y2 = A.hp[0] * y + h1;
        h1 = A.hp[1] * y - A.hp[3] * y2 + h2;
        h2 = A.hp[2] * y - A.hp[4] * y2;
        const double e = y2 * y2;
        // the blocks alive at this sample: k_start, k_start - 1, .. down to max(0, k_start - 3)
        acc[k_start & 3] += e;
        if (k_start >= 1) acc[(k_start - 1) & 3] += e;
        if (k_start >= 2) acc[(k_start - 2) & 3] += e;
        if (k_start >= 3) acc[(k_start - 3) & 3] += e;
    }
    // blocks still open at the end: the reference stops at the first block whose end is `frames` (n_blocks counts them)
    for (unsigned k = (k_start >= 3 ? k_start - 3 : 0); k <= k_start; k++)
        if (k < A.n_blocks) out[k] = acc[k & 3];
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

int launch_analysis(const AnalysisArgs &A, hipStream_t s) {
    if (A.n_peaks) {
        hipLaunchKernelGGL(an_peaks_kernel, dim3(A.n_peaks), dim3(64), 0, s, A);
        AN_LAUNCH_CHECK();
    }
    if (A.n) {
        hipLaunchKernelGGL(an_scan_kernel, dim3(A.channels + 1), dim3(64), 0, s, A);
        AN_LAUNCH_CHECK();
        hipLaunchKernelGGL(an_blake3_chunks_kernel, dim3((unsigned)((A.n_chunks + 127) / 128)), dim3(128), 0, s, A);
        AN_LAUNCH_CHECK();
        hipLaunchKernelGGL(an_blake3_tree_kernel, dim3(1), dim3(256), 0, s, A);
        AN_LAUNCH_CHECK();
        hipLaunchKernelGGL(an_fft_kernel, dim3(3), dim3(128), 0, s, A);
        AN_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace flo
