// container_kernels.hip — see container_kernels.hpp. Replaces writer.rs:132-224 (header, TOC) and core/crc32.rs:2-30
// for DATA chunks that already sit in HBM.
//
// CRC32 (IEEE, reflected, init/xorout 0xFFFFFFFF) of a chunk is computed by all 256 threads of a workgroup at once
// from the linearity of the CRC register over GF(2) (see finish_files_kernel); products x^k * r mod P are 32-step
// shift-and-xor loops, the per-stripe skip is a table.
#include <stdlib.h>
#include "container_kernels.hpp"

#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace flo {

constexpr uint32_t kPoly = 0xEDB88320u;

__host__ __device__ inline uint32_t multmodp(uint32_t a, uint32_t b) {   // a(x) * b(x) mod P, reflected bit order
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ kPoly : b >> 1;
    }
    return p;
}
__host__ __device__ inline uint32_t x8n_modp(unsigned long long n) {   // x^(8 n) mod P
    uint32_t sq = 0x00800000u;   // x^8 in the reflected representation (x^0 = 0x80000000)
    uint32_t p = 0x80000000u;
    while (n) {
        if (n & 1ull) p = multmodp(sq, p);
        sq = multmodp(sq, sq);
        n >>= 1;
    }
    return p;
}
// x^(8 n) mod P from the table of x^(8 2^j): one product per set bit of n
__device__ __forceinline__ uint32_t x8n_tab(const unsigned int (&pow2)[40], unsigned long long n) {
    uint32_t p = 0x80000000u;
    for (int j = 0; n; j++, n >>= 1)
        if (n & 1ull) p = multmodp(pow2[j], p);
    return p;
}

// x^(8 n) mod P from three tables when n < 4 MiB (two products), else bit by bit
__device__ __forceinline__ uint32_t x8n_fast(const FinishArgs &A, unsigned long long n) {
    if (n >= (256ull << 14)) return x8n_tab(A.x8pow2, n);
    const uint32_t a = A.stripe_pow[n >> 14], b = A.blk_pow[(n >> 6) & 255u], c = A.byte_pow[n & 63u];
    return multmodp(multmodp(a, b), c);
}

__device__ __forceinline__ void put8(uint8_t *p, unsigned v) { *p = (uint8_t)v; }
__device__ __forceinline__ void put32(uint8_t *p, uint32_t v) {
    for (int i = 0; i < 4; i++) p[i] = (uint8_t)(v >> (8 * i));
}
__device__ __forceinline__ void put64(uint8_t *p, unsigned long long v) {
    for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i));
}

constexpr int kFinThreads = 256;
constexpr unsigned kBlk = 64;                       // bytes a thread consumes per stripe
constexpr unsigned kStripe = kFinThreads * kBlk;    // 16 KiB: one coalesced sweep of the workgroup

// The CRC register after a message M from initial value I is (I x^(8n) + M(x) x^32) mod P: linear in I and in M.
// So a slice of the DATA chunk is cut into 64-byte blocks dealt round-robin to the 256 threads (every load of the
// workgroup is 16 KiB contiguous); a thread carries one register across its blocks, multiplying by
// x^(8 (16384 - 64)) to skip the other threads' bytes (a 4 x 256 table, like the byte tables), and at the end each
// register is moved to the end of the slice by x^(8 tail) and all are xor-ed together. A clip is cut into `parts`
// slices (one workgroup each, so that a single long clip still fills the chip); finish_files_kernel moves the slice
// registers to the end of the message and adds the contribution of the initial value.
__device__ __forceinline__ unsigned long long slice_bytes(unsigned long long n, unsigned parts) {
    unsigned long long s = (n + parts - 1) / parts;
    return (s + kStripe - 1) / kStripe * kStripe;
}

__device__ __forceinline__ void crc_slice_body(const FinishArgs &A, const unsigned clip, const unsigned part) {
    __shared__ uint32_t tab[4][256];    // slicing-by-4 byte tables
    __shared__ uint32_t skip[4][256];   // multiplication by x^(8 (kStripe - kBlk))
    __shared__ uint32_t s_red[kFinThreads / 64];
    __shared__ uint32_t s_pw[2];
    if (clip >= (unsigned)A.n_clips) return;
    const unsigned t = threadIdx.x;
    const unsigned long long total = A.clip_bytes[clip];
    const unsigned long long S = slice_bytes(total, A.parts);
    const unsigned long long beg = (unsigned long long)part * S < total ? (unsigned long long)part * S : total;
    const unsigned long long n = beg + S < total ? S : total - beg;
    if (n == 0) {
        if (t == 0) A.part_reg[(unsigned long long)clip * A.parts + part] = 0;
        return;
    }
    const uint8_t *data = A.out + A.data_off[clip] + beg;

    // byte tables (crc32.rs:2-20 builds the first one the same way) and the skip table, made once on the host
    for (int k = 0; k < 4; k++) {
        tab[k][t] = A.tables[k * 256 + t];
        skip[k][t] = A.tables[1024 + k * 256 + t];
    }
    const unsigned long long full = n / kStripe;          // complete stripes
    const unsigned long long rem0 = full * kStripe;       // first byte behind them
    const unsigned rem = (unsigned)(n - rem0);
    const unsigned nb = rem / kBlk, last = rem % kBlk;
    // two per-slice powers from the tables: x^(8 rem) = x^(8 * 64 * nb) * x^(8 last), and x^(8 last)
    if (t == 0) s_pw[0] = multmodp(A.blk_pow[nb], A.byte_pow[last]);
    if (t == 64) s_pw[1] = A.byte_pow[last];
    __syncthreads();
    auto eat = [&](uint32_t reg, const uint8_t *p, unsigned bytes) {   // bytes is a multiple of 4, p 4-byte aligned
        for (unsigned i = 0; i < bytes; i += 4) {
            const uint32_t w = *reinterpret_cast<const uint32_t *>(p + i) ^ reg;
            reg = tab[3][w & 0xFFu] ^ tab[2][(w >> 8) & 0xFFu] ^ tab[1][(w >> 16) & 0xFFu] ^ tab[0][w >> 24];
        }
        return reg;
    };
    uint32_t acc = 0;
    if (full) {
        uint32_t reg = 0;
        const uint8_t *p = data + (unsigned long long)t * kBlk;
        for (unsigned long long sidx = 0; sidx < full; sidx++, p += kStripe) {
            reg = skip[0][reg & 0xFFu] ^ skip[1][(reg >> 8) & 0xFFu] ^ skip[2][(reg >> 16) & 0xFFu] ^ skip[3][reg >> 24];
            const uint4 a = *reinterpret_cast<const uint4 *>(p), b = *reinterpret_cast<const uint4 *>(p + 16),
                        c = *reinterpret_cast<const uint4 *>(p + 32), d = *reinterpret_cast<const uint4 *>(p + 48);
            const uint32_t w[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const uint32_t v = w[i] ^ reg;
                reg = tab[3][v & 0xFFu] ^ tab[2][(v >> 8) & 0xFFu] ^ tab[1][(v >> 16) & 0xFFu] ^ tab[0][v >> 24];
            }
        }
        // the register now stands behind this thread's block of the last complete stripe
        // ... and moves to the end of the slice: x^(8 (64 (255 - t) + rem))
        acc = multmodp(multmodp(A.blk_pow[kFinThreads - 1 - t], s_pw[0]), reg);
    }
    {   // the incomplete stripe: one 64-byte block per thread, then the last < 64 bytes on thread 0
        if (t < nb) {
            const uint32_t reg = eat(0, data + rem0 + (unsigned long long)t * kBlk, kBlk);
            acc ^= multmodp(multmodp(A.blk_pow[nb - 1 - t], s_pw[1]), reg);
        }
        if (t == 0) {
            uint32_t reg = 0;
            const uint8_t *p = data + rem0 + (unsigned long long)nb * kBlk;
            for (unsigned i = 0; i < last; i++) reg = tab[0][(reg ^ p[i]) & 0xFFu] ^ (reg >> 8);
            acc ^= reg;
        }
    }
    for (int d = 32; d > 0; d >>= 1) acc ^= __shfl_down(acc, d);
    if ((t & 63) == 0) s_red[t >> 6] = acc;
    __syncthreads();
    if (t == 0) {
        uint32_t r = 0;
        for (int k = 0; k < kFinThreads / 64; k++) r ^= s_red[k];
        A.part_reg[(unsigned long long)clip * A.parts + part] = r;
    }
}

__global__ __launch_bounds__(kFinThreads) void crc_slices_kernel(FinishArgs A) {
    crc_slice_body(A, blockIdx.x, blockIdx.y);   // clips in x: gridDim.y stops at 65535
}

// header (writer.rs:132-191); total_samples (bytes 14 .. 21) comes from the TOC's last chunk
__device__ __forceinline__ void write_header(const FinishArgs &A, uint8_t *file, const unsigned clip, const unsigned nf, const unsigned long long n,
                                             const uint32_t crc) {
    if (A.crc_out) A.crc_out[clip] = crc;
    uint8_t *p = file;
    p[0] = 'F'; p[1] = 'L'; p[2] = 'O'; p[3] = '!';
    put8(p + 4, 1);    // version 1.2 (core/types.rs:12-13)
    put8(p + 5, 2);
    put8(p + 6, A.flags & 0xFF);
    put8(p + 7, A.flags >> 8);
    put32(p + 8, A.sample_rate);
    put8(p + 12, A.channels);
    put8(p + 13, A.bit_depth);
    put8(p + 22, A.level);
    put8(p + 23, 0); put8(p + 24, 0); put8(p + 25, 0);
    put32(p + 26, crc);
    put64(p + 30, 66);
    put64(p + 38, 4ull + 20ull * nf);
    put64(p + 46, n);
    put64(p + 54, 0);
    put64(p + 62, 0);    // meta_size: patched by whoever appends a META chunk
    put32(p + 70, nf);
}

// THREADS: 256 for batches of many clips, 1024 for a few long ones. The TOC of a long clip is cut into chunks of
// A.toc_chunk frames, one workgroup each (blockIdx.y): a 3-minute clip has 7752 entries, and one workgroup writing all
// of them was 20-28 us of a 0.15 ms encode. A chunk's workgroup first adds up the sizes in front of its chunk (strided
// loads, a block reduction), then scans its own frames; chunk 0 also joins the CRC slices and writes the header, the
// last chunk the header's total_samples.
template <int THREADS>
__device__ __forceinline__ void finish_body(const FinishArgs &A, const unsigned clip, const unsigned chunk_idx) {
    __shared__ uint32_t s_red[THREADS / 64];
    __shared__ unsigned long long w_sum[THREADS / 64], w_smp[THREADS / 64], p_sum[THREADS / 64], p_smp[THREADS / 64];
    if (clip >= (unsigned)A.n_clips) return;
    const unsigned t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    const unsigned nf = A.clip_frames[clip];
    const unsigned chunk = A.toc_chunk ? A.toc_chunk : 0xFFFFFFFFu;
    const unsigned long long c0l = (unsigned long long)chunk_idx * chunk;
    if (c0l >= nf && chunk_idx != 0) return;   // (an empty clip still gets its header from chunk 0)
    const unsigned c0 = (unsigned)(c0l < nf ? c0l : nf), c1 = (unsigned)(c0l + chunk < nf ? c0l + chunk : nf);
    const bool first = chunk_idx == 0, last = c1 == nf;
    const bool do_crc = A.mode != 1u, do_toc = A.mode != 2u;
    const unsigned long long n = A.clip_bytes[clip];
    uint8_t *file = A.out + A.data_off[clip] - (74ull + 20ull * nf);
    const unsigned long long fb = A.clip_frame0[clip];

    // slice registers -> end of the message; thread `parts` adds the initial register carried through all n bytes
    uint32_t acc = 0;
    if (first && do_crc) {
        const unsigned long long S = slice_bytes(n, A.parts);
        if (t < A.parts) {
            const unsigned long long end = (unsigned long long)(t + 1) * S < n ? (unsigned long long)(t + 1) * S : n;
            const unsigned long long beg = (unsigned long long)t * S < n ? (unsigned long long)t * S : n;
            if (end > beg) acc = multmodp(x8n_fast(A, n - end), A.part_reg[(unsigned long long)clip * A.parts + t]);
        } else if (t == A.parts) {
            acc = multmodp(x8n_fast(A, n), 0xFFFFFFFFu);
        }
        for (int d = 32; d > 0; d >>= 1) acc ^= __shfl_down(acc, d);
        if (lane == 0) s_red[wv] = acc;
    }

    if (!do_toc) {   // (uniform) CRC and header only
        __syncthreads();
        if (first && t == 0) {
            uint32_t r = 0;
            for (int k = 0; k < THREADS / 64; k++) r ^= s_red[k];
            write_header(A, file, clip, nf, n, ~r);
        }
        return;
    }
    // bytes and samples in front of the chunk
    unsigned long long pb = 0, ps = 0;
    for (unsigned f = t; f < c0; f += 4 * THREADS) {
        uint32_t a4[4], b4[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned g = f + j * THREADS;
            a4[j] = g < c0 ? A.frame_size[fb + g] : 0u;
            b4[j] = g < c0 ? (A.frame_samples ? A.frame_samples[fb + g] : A.const_samples) : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) { pb += a4[j]; ps += b4[j]; }
    }
    // TOC: thread t owns a contiguous run of the chunk's frames; byte offsets and sample counts by a block scan of the run
    // sums (wave scans by shuffles, the wave totals by the first wave: two barriers instead of the twenty of a
    // Hillis-Steele scan). The run's sizes and sample counts are fetched eight independent loads at a time (a loop of
    // single loads pays one memory latency per frame); the first eight stay in registers for the entries written below.
    const unsigned per = (c1 - c0 + THREADS - 1) / THREADS;
    const unsigned f0 = c0 + t * per < c1 ? c0 + t * per : c1, f1 = f0 + per < c1 ? f0 + per : c1;
    unsigned long long bytes = 0, smp = 0;
    uint32_t kfs[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ksm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (unsigned fbb = f0; fbb < f1; fbb += 8) {
        uint32_t a8[8], b8[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            a8[j] = fbb + j < f1 ? A.frame_size[fb + fbb + j] : 0u;
            b8[j] = fbb + j < f1 ? (A.frame_samples ? A.frame_samples[fb + fbb + j] : A.const_samples) : 0u;
        }
#pragma unroll
        for (int j = 0; j < 8; j++) { bytes += a8[j]; smp += b8[j]; }
        if (fbb == f0) {
#pragma unroll
            for (int j = 0; j < 8; j++) { kfs[j] = a8[j]; ksm[j] = b8[j]; }
        }
    }
    unsigned long long sb = bytes, ss = smp;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long ub = __shfl_up(sb, d), us = __shfl_up(ss, d);
        if (lane >= (unsigned)d) { sb += ub; ss += us; }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { pb += __shfl_down(pb, d); ps += __shfl_down(ps, d); }
    if (lane == 63) { w_sum[wv] = sb; w_smp[wv] = ss; }
    if (lane == 0) { p_sum[wv] = pb; p_smp[wv] = ps; }
    __syncthreads();
    if (wv == 0) {
        unsigned long long xb = lane < THREADS / 64 ? w_sum[lane] : 0, xs = lane < THREADS / 64 ? w_smp[lane] : 0;
        unsigned long long qb = lane < THREADS / 64 ? p_sum[lane] : 0, qs = lane < THREADS / 64 ? p_smp[lane] : 0;
#pragma unroll
        for (int d = 1; d < THREADS / 64; d <<= 1) {
            const unsigned long long ub = __shfl_up(xb, d), us = __shfl_up(xs, d);
            if (lane >= (unsigned)d) { xb += ub; xs += us; }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { qb += __shfl_down(qb, d); qs += __shfl_down(qs, d); }
        if (lane < THREADS / 64) { w_sum[lane] = xb; w_smp[lane] = xs; }
        if (lane == 0) { p_sum[0] = qb; p_smp[0] = qs; }
    }
    __syncthreads();
    if (wv) { sb += w_sum[wv - 1]; ss += w_smp[wv - 1]; }
    sb += p_sum[0];
    ss += p_smp[0];
    {
        unsigned long long off = sb - bytes, cum = ss - smp;
        // The DATA chunk is 16-byte aligned and the TOC ends right in front of it, 20 bytes per entry: every entry is
        // 4-byte aligned, five dword stores. The timestamp floor(cum * 1000 / rate) is carried as quotient and
        // remainder from entry to entry (one 64-bit division per thread, 32-bit ones after it where they suffice).
        uint32_t *toc = reinterpret_cast<uint32_t *>(file + 70 + 4);
        const unsigned long long rate = A.sample_rate;
        unsigned long long q = cum * 1000ull / rate, r = cum * 1000ull - q * rate;
        auto entry = [&](const unsigned f, const unsigned fs, const unsigned nsmp) {   // writer.rs:193-224: index, byte offset, size, timestamp in ms
            uint32_t *e = toc + 5ull * f;
            e[0] = f;
            e[1] = (uint32_t)off;
            e[2] = (uint32_t)(off >> 32);
            e[3] = fs;
            e[4] = (uint32_t)q;
            off += fs;
            const unsigned long long add = (unsigned long long)nsmp * 1000ull + r;
            if (add < 0x100000000ull) {
                const uint32_t a32 = (uint32_t)add, r32 = (uint32_t)rate, dq = a32 / r32;
                q += dq;
                r = a32 - dq * r32;
            } else {
                const unsigned long long dq = add / rate;
                q += dq;
                r = add - dq * rate;
            }
        };
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (f0 + j < f1) entry(f0 + j, kfs[j], ksm[j]);
        for (unsigned f = f0 + 8; f < f1; f++) entry(f, A.frame_size[fb + f], A.frame_samples ? A.frame_samples[fb + f] : A.const_samples);
    }
    if (last && t == 0) put64(file + 14, w_smp[THREADS / 64 - 1] + p_smp[0]);   // total_samples = sum of frame_samples
    if (first && t == 0 && do_crc) {
        uint32_t r = 0;
        for (int k = 0; k < THREADS / 64; k++) r ^= s_red[k];
        write_header(A, file, clip, nf, n, ~r);
    }
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void finish_files_kernel(FinishArgs A) {
    finish_body<THREADS>(A, blockIdx.x, blockIdx.y);
}

// A few long clips: the CRC slices and the TOC chunks (256 frames each) are one launch - blockIdx.y below `parts` takes a
// slice, the rows behind a chunk of the TOC (neither needs the other) - and the CRC + header follow in a launch of one
// workgroup per clip. Two kernels one behind the other had each paid its own ramp and drain: 27.8 us for a 3-minute clip.
__global__ __launch_bounds__(kFinThreads) void crc_and_toc_kernel(FinishArgs A) {
    if (blockIdx.y < A.parts) crc_slice_body(A, blockIdx.x, blockIdx.y);
    else finish_body<kFinThreads>(A, blockIdx.x, blockIdx.y - A.parts);
}

int launch_finish_files(FinishArgs A, hipStream_t s) {
    if (!A.n_clips) return 0;
    if (A.parts < 1 || A.parts > (A.n_clips < 64 ? 512u : 128u) || !A.part_reg) return -1;
    static unsigned int pow2[40], blk[256], bytep[64], stripep[256];
    static std::vector<unsigned int> host_tables;
    static std::mutex mu;
    static std::map<int, unsigned int *> dev_tables;   // one copy per device, made on first use, kept for the process
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (host_tables.empty()) {
            uint32_t p = 0x00800000u;   // x^8
            for (int j = 0; j < 40; j++, p = multmodp(p, p)) pow2[j] = p;
            for (int i = 0; i < 256; i++) blk[i] = x8n_modp(64ull * i);
            for (int i = 0; i < 64; i++) bytep[i] = x8n_modp((unsigned long long)i);
            for (int i = 0; i < 256; i++) stripep[i] = x8n_modp((unsigned long long)i << 14);
            host_tables.resize(2048);
            for (uint32_t i = 0; i < 256; i++) {
                uint32_t c = i;
                for (int j = 0; j < 8; j++) c = (c & 1u) ? (c >> 1) ^ kPoly : c >> 1;
                host_tables[i] = c;
            }
            for (uint32_t i = 0; i < 256; i++) {
                uint32_t c = host_tables[i];
                for (int k = 1; k < 4; k++) {
                    c = host_tables[c & 0xFFu] ^ (c >> 8);
                    host_tables[k * 256 + i] = c;
                }
            }
            const uint32_t skipm = x8n_modp(kStripe - kBlk);
            for (int k = 0; k < 4; k++)
                for (uint32_t i = 0; i < 256; i++) host_tables[1024 + k * 256 + i] = multmodp(skipm, i << (8 * k));
        }
        if (!dev_tables.count(dev)) {
            unsigned int *d = nullptr;
            if (hipMalloc(&d, host_tables.size() * 4) != hipSuccess) return -1;
            if (hipMemcpy(d, host_tables.data(), host_tables.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return -1;
            dev_tables[dev] = d;
        }
        A.tables = dev_tables[dev];
    }
    memcpy(A.x8pow2, pow2, sizeof pow2);
    memcpy(A.blk_pow, blk, sizeof blk);
    memcpy(A.byte_pow, bytep, sizeof bytep);
    memcpy(A.stripe_pow, stripep, sizeof stripep);
    if (A.n_clips < 64 && A.max_frames && !getenv("FLO_FINISH_TWO_KERNELS")) {
        A.toc_chunk = kFinThreads;
        A.mode = 1;
        const unsigned chunks = (A.max_frames + kFinThreads - 1u) / kFinThreads;
        hipLaunchKernelGGL(crc_and_toc_kernel, dim3((unsigned)A.n_clips, A.parts + chunks), dim3(kFinThreads), 0, s, A);
        A.mode = 2;
        hipLaunchKernelGGL((finish_files_kernel<1024>), dim3((unsigned)A.n_clips, 1u), dim3(1024), 0, s, A);
        hipError_t e2 = hipGetLastError();
        return e2 == hipSuccess ? 0 : (int)e2;
    }
    A.mode = 0;
    hipLaunchKernelGGL(crc_slices_kernel, dim3((unsigned)A.n_clips, A.parts), dim3(kFinThreads), 0, s, A);
    if (A.n_clips < 64) {
        // a few long clips: the TOC in chunks of 1024 frames (one entry per thread)
        A.toc_chunk = 1024;
        const unsigned chunks = A.max_frames ? (A.max_frames + 1023u) / 1024u : 1u;
        if (!A.max_frames) A.toc_chunk = 0;
        hipLaunchKernelGGL((finish_files_kernel<1024>), dim3((unsigned)A.n_clips, chunks), dim3(1024), 0, s, A);
    } else {
        A.toc_chunk = 0;
        hipLaunchKernelGGL((finish_files_kernel<256>), dim3((unsigned)A.n_clips), dim3(256), 0, s, A);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

__global__ void table_crcs_kernel(const uint8_t *base, unsigned long long *row, unsigned long long n, unsigned long long max_clips) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *p = base + row[1 + max_clips + i] + 26;   // the header's data_crc32 (writer.rs:150), any alignment
    row[1 + 2 * max_clips + i] = (unsigned long long)p[0] | ((unsigned long long)p[1] << 8) | ((unsigned long long)p[2] << 16) | ((unsigned long long)p[3] << 24);
}
int launch_table_crcs(const uint8_t *base, unsigned long long *row, unsigned long long n, unsigned long long max_clips, hipStream_t s) {
    if (!n) return 0;
    hipLaunchKernelGGL(table_crcs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, base, row, n, max_clips);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

}  // namespace flo
