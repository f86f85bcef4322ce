// lldec_kernels.hip — parallel decode of ALPC channel wrappers for gfx950.
//
// Reference behaviour replaced (files under /root/reference/libflo/src):
//   core/rice.rs:123-159 (decode_i32) + :217-259 (BitReader), lossless/decoder.rs:92-150 (decode_channel_int),
//   :152-184 (reconstruct_lpc_int), :186-266 (reconstruct_fixed).
//
// A Rice stream is a chain: code i + 1 starts where code i ends. What breaks the chain is that, for a fixed k, the
// only thing a stretch of the stream needs to know about everything before it is *how it is entered*: in the middle
// of a unary run, or with 0..k remainder bits still to skip. So
//   1. rice_scan   : the stream is cut into tiles of kRiceTileBits (1024) bits; a lane walks one tile from one of the
//                    k + 2 possible entry states (k + 2 lanes per tile, 64 / (k + 2) tiles per wavefront; parses that have
//                    met continue as one, see the kernel) and records where it leaves the tile and how many codes
//                    started inside it;
//   2. rice_chain  : one wavefront per wrapper follows those tables from tile to tile (a few hundred dependent LDS
//                    reads) and notes, per tile, the real entry state and the index of its first code;
//   3. rice_decode : a lane per tile walks its tile once more from the now known entry and writes the residuals;
//   4. predict     : the LPC recurrence is the one truly serial piece (the shift rounds, so it is no linear scan).
//                    It runs in transposed form, four wrappers per wavefront (one per row of sixteen lanes): lane l
//                    of a 16-sample block accumulates the prediction of sample l, and a finished sample reaches the
//                    later accumulators of its row through the DPP operand of one v_fmac_f64 (row_newbcast) whose
//                    coefficient register is the tap vector rotated to that step: a sample costs floor + fmac,
//                    21 cycles (diag/dpp_step_rate.hip; the 64-lane form of rounds 2 - 3 paid floor + 2 readlane +
//                    fma, 30 cycles, for ONE wrapper per wavefront). See predict_rows.
//                    Doubles are exact here: the host only admits wrappers with sum |coef| < 2^21 and shift <= 20, so
//                    every partial sum (residual included) is a multiple of 2^-shift whose numerator stays below
//                    2^31 * 2^20 + 2^21 * 2^31 < 2^53. The reference's i32 wrap-around cannot be followed
//                    that way; a sample that leaves the i32 range flags the wrapper and the serial kernel redoes it
//                    (as it does wrappers with k > 14, a 256-ones escape, or larger coefficients).
//                    Fixed predictors of order 1..4 are `order` wrapping prefix sums (decoder.rs:186-266 read as
//                    difference equations, warm-up included), all taken in one sweep; raw and silent wrappers are copies.
// Zero padding stands in for the reader's end-of-stream rules: ones up to the end then a (virtual) 0, remainder bits
// past the end read as 0, and a value that starts past the end is 0 (rice_chain clears those samples).
#include "decode_kernels.hpp"

namespace flo {

namespace {
// ordering point between LDS accesses of ONE wavefront (its LDS instructions execute in order; this pins the compiler)
__device__ __forceinline__ void wave_sync_l() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}
constexpr int kTileWords = kRiceTileBits / 32;
constexpr int kTileWordsLog2 = kTileWords == 64 ? 6 : kTileWords == 32 ? 5 : 4;
static_assert((1 << kTileWordsLog2) == kTileWords, "tile of 512, 1024 or 2048 bits");
constexpr int kScanTiles = 4;                    // tiles per wavefront in rice_scan at k = 14 (16 entry states); 64 / (k + 2) in general
constexpr int kScanTilesMax = 32;                // ... at k = 0
constexpr int kScanStride = kTileWords + 2;      // a window read touches word w + 1
constexpr int kDecOver = 16;                     // words past the last tile a code may reach (256 ones + k bits)
constexpr int kChainChunk = 256;                 // tile tables staged per step of the chain walk

__device__ __forceinline__ uint32_t be_word(const uint8_t *p, uint32_t len, uint32_t w) {
    const uint32_t b = 4u * w;
    if (b + 4u <= len) return ((uint32_t)p[b] << 24) | ((uint32_t)p[b + 1] << 16) | ((uint32_t)p[b + 2] << 8) | (uint32_t)p[b + 3];
    uint32_t v = 0;
    if (b < len) v |= (uint32_t)p[b] << 24;
    if (b + 1u < len) v |= (uint32_t)p[b + 1] << 16;
    if (b + 2u < len) v |= (uint32_t)p[b + 2] << 8;
    return v;
}
// Stage big-endian words [w0, w0 + count) of a stream into LDS through `put(i, word)`, eight loads in flight per lane:
// one load at a time costs a memory round trip each, and that was most of what these kernels did.
template <class Put>
__device__ __forceinline__ void stage_words(const uint8_t *p, uint32_t len, uint32_t w0, int count, int lane, Put put) {
    constexpr int U = 8;
    for (int base = 0; base < count; base += 64 * U) {
        uint32_t v[U];
        const bool whole = base + 64 * U <= count && 4ull * (w0 + (uint32_t)base + 64u * U) + 4ull <= (unsigned long long)len;   // uniform
        if (whole) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                uint32_t x;
                __builtin_memcpy(&x, p + 4ull * (w0 + (uint32_t)(base + 64 * u + lane)), 4);
                v[u] = __builtin_bswap32(x);
            }
#pragma unroll
            for (int u = 0; u < U; u++) put(base + 64 * u + lane, v[u]);
        } else {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = base + 64 * u + lane;
                v[u] = i < count ? be_word(p, len, w0 + (uint32_t)i) : 0u;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = base + 64 * u + lane;
                if (i < count) put(i, v[u]);
            }
        }
    }
}

__device__ __forceinline__ uint32_t leading_ones(uint32_t x) { return x == 0xFFFFFFFFu ? 32u : (uint32_t)__clz((int)~x); }

}  // namespace

// ------------------------------------------------------------------------------------------------ 1. tile tables
// Walking a tile from every one of its k + 2 entry states repeats itself: the parses fall into step with each other
// after a few codes (two that reach the same bit position are one parse from there on). So the walk has two phases:
//   phase 1: k + 2 lanes per tile, as many tiles as fit a wavefront, eight wavefronts per workgroup - every entry state
//            is walked up to the first position at or behind bit kScanFrontier;
//   phase 2: of the lanes of a tile that stopped at the same position one (the lowest) is its leader; the leaders of
//            the whole workgroup - one or two per tile - are packed into as few wavefronts as they need and walk the
//            rest of their tiles; a lane's result is its own count up to the frontier plus its leader's behind it.
// (All k + 2 lanes to the end of the tile was 975 vector instructions per tile, the kernel at the issue limit.)
constexpr int kScanWaves = 8;
#ifndef FLO_SCAN_FRONTIER
#define FLO_SCAN_FRONTIER 192
#endif
constexpr uint32_t kScanFrontier = FLO_SCAN_FRONTIER;   // (>= the tile: one phase)

__global__ __launch_bounds__(64 * kScanWaves) void ll_rice_scan_kernel(LlParArgs A) {
    __shared__ uint32_t words[kScanWaves * kScanTilesMax * kScanStride];
    __shared__ uint32_t posv[64 * kScanWaves], slotv[64 * kScanWaves], list[64 * kScanWaves], res[64 * kScanWaves];
    __shared__ uint32_t count;
    const unsigned ch = blockIdx.x;
    const unsigned nt = A.tile0[ch + 1] - A.tile0[ch];
    const LlChannelDev c = A.ch[ch];
    const uint32_t k = c.rice_k;
    // The grid is sized for four tiles per wavefront (k = 14): with a smaller k the surplus workgroups find nothing to do.
    const uint32_t S = k + 2u, tpw = 64u / S;
    const unsigned t0g = blockIdx.y * (kScanWaves * tpw);
    if (t0g >= nt) return;   // (the whole workgroup)
    const int tid = (int)threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) count = 0;
    const unsigned t0 = t0g + (unsigned)wave * tpw;
    const uint8_t *p = A.bytes + c.off;
    uint32_t *wrows = words + wave * kScanTilesMax * kScanStride;
    // a wavefront's tiles are consecutive words of the stream: word i goes to tile i / kTileWords (the two spare words
    // of a tile's row repeat the next tile's first two)
    if (t0 < nt)
        stage_words(p, c.len, t0 * kTileWords, (int)(tpw * kTileWords + 2), lane, [&](int i, uint32_t v) {
            const int tile = i >> kTileWordsLog2, w = i & (kTileWords - 1);
            if (tile < (int)tpw) wrows[tile * kScanStride + w] = v;
            if (w < 2 && tile > 0) wrows[(tile - 1) * kScanStride + kTileWords + w] = v;
        });
    __syncthreads();
    // One step = one look at the 32 bits at `pos`, where the terminator of the current code (already counted) is being
    // searched: either the window is all ones (32 bits further, still inside the run) or it shows the terminating 0,
    // behind which k remainder bits are skipped and - if that is still inside the tile - the next code starts.
    // A lane that has left its tile keeps the way it left in `pos` itself: T + (0..k remainder bits still to skip in the
    // next tile), or T + k + 1 when it left inside a run - its exit state is pos - T (and what it reads there are the
    // row's two spare words). Straight-line, predicated by `go`: nested data-dependent loops cost more in exec-mask
    // bookkeeping than in arithmetic.
    constexpr uint32_t T = (uint32_t)kRiceTileBits;
    const uint32_t kp1 = k + 1u;
    uint32_t kp1v = kp1, endv = T + kp1;   // (in vector registers: as scalars they were copied into one at every use)
    asm volatile("" : "+v"(kp1v), "+v"(endv));
    auto step = [&](const uint32_t *w, const bool go, uint32_t &pos, uint32_t &n) {
        // (the two words arrive as one 64-bit register pair, high word first: ds_read2_b32 with its offsets crossed)
        unsigned long long two;
        asm volatile("ds_read2_b32 %0, %1 offset0:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(two) : "v"((uint32_t)(uintptr_t)(w + (pos >> 5))) : "memory");
        const uint32_t ones = (uint32_t)__clz((int)~(uint32_t)((two << (pos & 31u)) >> 32));   // (32 for a window of ones)
        const uint32_t z = pos + ones;                      // the terminating 0, or 32 bits on
        const bool in_tile = z < T;
        const bool term = in_tile && ones < 32u;
        const uint32_t npos = (in_tile ? z : endv) + (term ? kp1v : 0u);
        n += (go && term && npos < T) ? 1u : 0u;
        pos = go ? npos : pos;
    };
    const uint32_t tile = (uint32_t)lane / S, st = (uint32_t)lane - tile * S;
    const bool mine = tile < tpw && t0 + tile < nt;
    const uint32_t row = (uint32_t)wave * kScanTilesMax + (mine ? tile : 0u);
    uint32_t pos = mine ? (st <= k ? st : 0u) : T + kp1, n = st <= k ? 1u : 0u;
    while (__ballot(pos < kScanFrontier) != 0ull) step(words + row * kScanStride, pos < kScanFrontier, pos, n);
    // leaders: the lowest lane of the tile that stopped at this position
    const bool cand = pos < T;
    const uint32_t g0 = (uint32_t)wave * 64u + tile * S;
    uint32_t leader = st;
    if (kScanFrontier < T) {
        posv[tid] = pos;
        wave_sync_l();
        for (uint32_t j = 0; j + 1u < S; j++)   // (uniform bound)
            if (cand && j < st && leader == st && posv[g0 + j] == pos) leader = j;
        if (cand && leader == st) {
            const uint32_t slot = atomicAdd(&count, 1u);
            list[slot] = (row << 16) | pos;
            slotv[tid] = slot;
        }
        __syncthreads();
        const uint32_t cnt = count;
        const bool have = (uint32_t)tid < cnt;
        if (__ballot(have) != 0ull) {   // (whole wavefronts without work skip)
            const uint32_t e = have ? list[tid] : 0u;
            const uint32_t *w2 = words + (e >> 16) * kScanStride;
            uint32_t pos2 = have ? e & 0xFFFFu : T + kp1, n2 = 0;
            while (__ballot(pos2 < T) != 0ull) step(w2, pos2 < T, pos2, n2);
            if (have) res[tid] = (pos2 - T) | (n2 << 5);
        }
        __syncthreads();
    }
    if (mine) {
        uint32_t o = (pos - T) | (n << 5);
        if (cand) {
            const uint32_t r = res[slotv[g0 + leader]];
            o = (r & 31u) | ((n + (r >> 5)) << 5);
        }
        A.tabs[(size_t)(A.tile0[ch] + t0 + tile) * kRiceStates + st] = o;
    }
}

// ------------------------------------------------------------------------------------------------ 2. the chain
__global__ __launch_bounds__(64) void ll_rice_chain_kernel(LlParArgs A) {
    __shared__ uint32_t tab[kChainChunk * kRiceStates];
    __shared__ uint2 ent[kChainChunk];
    __shared__ uint32_t carry[2];
    const unsigned ch = blockIdx.x;
    const unsigned first = A.tile0[ch], nt = A.tile0[ch + 1] - first;
    if (!nt) return;
    const int lane = (int)threadIdx.x;
    if (lane == 0) {
        carry[0] = 0;   // entry state of the next tile
        carry[1] = 0;   // codes started so far
    }
    for (unsigned base = 0; base < nt; base += kChainChunk) {
        const unsigned m = nt - base < (unsigned)kChainChunk ? nt - base : (unsigned)kChainChunk;
        for (unsigned i = lane; i < m * kRiceStates; i += 64) tab[i] = A.tabs[(size_t)(first + base) * kRiceStates + i];
        __syncthreads();
        if (lane == 0) {
            uint32_t st = carry[0], idx = carry[1];
            for (unsigned t = 0; t < m; t++) {
                ent[t] = make_uint2(idx, st);
                const uint32_t e = tab[t * kRiceStates + st];
                st = e & 31u;
                idx += e >> 5;
            }
            carry[0] = st;
            carry[1] = idx;
        }
        __syncthreads();
        for (unsigned t = lane; t < m; t += 64) A.tile_entry[first + base + t] = ent[t];
        __syncthreads();
    }
    // a value whose code would start behind the end of the stream is 0 (rice.rs:129-133): the scratch is not
    // zero-filled beforehand, so the samples behind the last code that starts inside the stream are cleared here
    {
        const LlChannelDev c = A.ch[ch];
        const uint32_t total = carry[1];
        int *out = A.scratch + c.out_off;
        for (uint32_t i = total + (uint32_t)lane; i < c.samples; i += 64u) out[i] = 0;
    }
}

// ------------------------------------------------------------------------------------------------ 3. residuals
__global__ __launch_bounds__(64) void ll_rice_decode_kernel(LlParArgs A) {
    // 64 tiles of 64 words; one pad word per tile so that lanes at the same offset of their tiles hit 64 banks
    __shared__ uint32_t words[64 * (kTileWords + 1) + kDecOver + kDecOver / kTileWords + 4];
    __shared__ uint32_t stg[64 * 17];   // 16 pending values per lane (one pad word per row),
    __shared__ uint2 s_cb[64];          // their count and place
    const unsigned ch = blockIdx.x;
    const unsigned first = A.tile0[ch], nt = A.tile0[ch + 1] - first;
    const unsigned t0 = blockIdx.y * 64u;
    if (t0 >= nt) return;
    const int lane = (int)threadIdx.x;
    const LlChannelDev c = A.ch[ch];
    const uint8_t *p = A.bytes + c.off;
    stage_words(p, c.len, t0 * kTileWords, 64 * kTileWords + kDecOver, lane, [&](int i, uint32_t v) { words[i + (i >> kTileWordsLog2)] = v; });
    __syncthreads();
    const unsigned t = t0 + lane;
    const bool have_tile = t < nt;   // (lanes without a tile stay for the cooperative stores)
    const uint2 e = have_tile ? A.tile_entry[first + t] : make_uint2(0u, 0u);
    const uint32_t k = c.rice_k, n = c.samples;
    int *out = A.scratch + c.out_off;
    // bit positions are relative to the wavefront's first tile; word i lives at words[i + i / 64]
    // the 64 bits at `pos` (three words, one LDS round trip): a code's unary part is looked at 32 bits at a time, and
    // its k <= 14 remainder bits lie inside the same 64 whenever the terminator does (a second, dependent window read
    // for them was a third of an iteration's latency)
    auto win64 = [&](uint32_t pos, uint32_t &hi, uint32_t &lo) {
        const uint32_t i = pos >> 5, sh = pos & 31u, i1 = i + 1u, i2 = i + 2u;
        const uint32_t w0 = words[i + (i >> kTileWordsLog2)], w1 = words[i1 + (i1 >> kTileWordsLog2)], w2 = words[i2 + (i2 >> kTileWordsLog2)];
        hi = (uint32_t)(((((unsigned long long)w0) << 32 | w1) << sh) >> 32);
        lo = (uint32_t)(((((unsigned long long)w1) << 32 | w2) << sh) >> 32);
    };
    const uint32_t tile_lo = (uint32_t)lane * kRiceTileBits, tile_hi = tile_lo + kRiceTileBits;
    const uint32_t limit = 64u * kRiceTileBits + 32u * (kDecOver - 2);   // staged bits a run may be followed through
    // One predicated loop, one look at a 32-bit window per iteration (see ll_rice_scan_kernel). `skip`: the run under
    // way when the tile was entered belongs to a code of an earlier tile - it is followed to its terminator inside this
    // tile (or the tile holds no code start at all) and produces no value here.
    bool skip = e.y == k + 1u, active = true, escape = false;
    uint32_t pos = tile_lo + (skip ? 0u : e.y), idx = e.x, q = 0;
    active = have_tile && (skip || (pos < tile_hi && idx < n));
    // A lane's values go to consecutive places of ITS tile's run: written one at a time, a store instruction touched 64
    // different lines (4 bytes each) and the kernel ran at the pace of the store path. They are collected in LDS - at
    // most one per lane and iteration, sixteen iterations at a time - and written out four rows per instruction, a row
    // = up to 16 consecutive values of one lane.
    uint32_t pend = 0, pbase = idx, it = 0;
    auto flush = [&]() {
        s_cb[lane] = make_uint2(pend, pbase);
        wave_sync_l();
        // (all the reads first: one LDS round trip for the sixteen rows instead of one per row)
        uint2 cb[16];
        uint32_t val[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int row = 4 * j + (lane >> 4), col = lane & 15;
            cb[j] = s_cb[row];
            val[j] = stg[row * 17 + col];
        }
#pragma unroll
        for (int j = 0; j < 16; j++)
            if ((uint32_t)(lane & 15) < cb[j].x) out[cb[j].y + (uint32_t)(lane & 15)] = (int)val[j];
        wave_sync_l();
        pend = 0;
        pbase = idx;
    };
    while (__ballot(active) != 0ull) {
        const uint32_t rp = active ? pos : tile_lo;
        uint32_t whi, wlo;
        win64(rp, whi, wlo);
        const uint32_t ones = leading_ones(whi);
        const uint32_t z = rp + ones, q2 = q + ones;
        const bool term = ones < 32u;
        // our own code: the 256-ones escape (rice.rs:134-139), or a run past what is staged -> the serial reader
        const bool esc = !skip && (q2 >= 256u || z >= limit);
        // an inherited run must end inside the tile
        const bool lost = skip && (z >= tile_hi);
        // the 32 bits behind the terminator: ({whi, wlo} << (ones + 1)) >> 32 (v_alignbit; ones <= 31 when it matters)
        const uint32_t rem = (uint32_t)((unsigned long long)__builtin_amdgcn_alignbit(whi, wlo, (31u - ones) & 31u) >> (32u - k));   // (k = 0: nothing)
        const bool emit = active && term && !skip && !esc;
        if (emit) {
            const uint32_t u = (q2 << k) | rem;
            stg[lane * 17 + (int)pend] = (uint32_t)((int)(u >> 1) ^ -(int)(u & 1u));
            pend++;
        }
        const uint32_t npos = term ? z + 1u + k : z;
        const uint32_t nidx = idx + (emit ? 1u : 0u);
        escape = escape || (active && esc);
        const bool go_on = !esc && !lost && (term ? (npos < tile_hi && nidx < n) : true);
        pos = active ? npos : pos;
        idx = active ? nidx : idx;
        q = term ? 0u : q2;
        skip = skip && !term;
        active = active && go_on;
        if ((++it & 15u) == 0u) flush();   // (uniform)
    }
    flush();
    if (escape) A.serial[ch] = 1;
}

// ------------------------------------------------------------------------------------------------ 4. predictors
// inclusive wrapping sum over the 64 lanes, in the DPP network: four shifts inside the rows of sixteen, then lane 15 of
// row 0 / 2 into rows 1 / 3 and lane 31 into the upper half
__device__ __forceinline__ unsigned wave_scan_add_dpp(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1, 3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2, 3
    return v;
}

// One wrapper by the whole wavefront: fixed predictors, raw and silent wrappers, LPC wrappers without a recurrence to run
// (ll_predict_kernel takes the others four at a time in rows, see predict_rows).
__device__ __forceinline__ void predict_one(const LlParArgs &A, const unsigned chi, const int lane, double *cs) {
    if (A.serial[chi]) return;
    const LlChannelDev c = A.ch[chi];
    int *r = A.scratch + c.out_off;
    const uint32_t n = c.samples;
    const bool has_coeffs = c.n_coeffs > 0, has_res = c.len > 0;

    if (!has_coeffs && has_res && c.shift_bits >= 128) {
        // reconstruct_fixed: sample i < order uses order i, so with D^m the m-th difference, D^m_m = r_m and
        // D^m_i = D^m_(i-1) + D^(m+1)_i: for m = order-1 .. 0 an inclusive wrapping prefix sum over a[m..]
        const int order = c.shift_bits - 128;
        if (order < 1 || order > 4) return;   // order 0, and unknown orders, copy the residuals
        // All `order` sums in one sweep: a tile of 256 samples (four consecutive ones per lane) takes sum m = order - 1
        // down to 0 in registers - a lane's four serially, the lanes' totals by a DPP scan - with one carry per sum
        // from tile to tile; the plane is read and written once (it was `order` sweeps of shuffle scans: 2 ms for a
        // one-second frame at 96 kHz, the longest thing the decode did).
        int carry[4] = {0, 0, 0, 0};
        for (uint32_t base4 = 0; base4 < n; base4 += 1024u) {   // four tiles' loads in flight together
            int v[4][4];
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t i = base4 + 256u * (uint32_t)q + 4u * (uint32_t)lane + (uint32_t)j;
                    v[q][j] = i < n ? r[i] : 0;
                }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t i0 = base4 + 256u * (uint32_t)q + 4u * (uint32_t)lane;
#pragma unroll
                for (int m = 3; m >= 0; m--) {
                    if (m < order) {   // (uniform)
                        unsigned a[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) a[j] = i0 + (uint32_t)j >= (uint32_t)m ? (unsigned)v[q][j] : 0u;   // sum m starts at sample m
                        a[1] += a[0], a[2] += a[1], a[3] += a[2];
                        const unsigned tot = wave_scan_add_dpp(a[3]);
                        const unsigned ex = tot - a[3] + (unsigned)carry[m];
                        carry[m] = (int)((unsigned)__builtin_amdgcn_readlane((int)tot, 63) + (unsigned)carry[m]);
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (i0 + (uint32_t)j >= (uint32_t)m) v[q][j] = (int)(a[j] + ex);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t i = base4 + 256u * (uint32_t)q + 4u * (uint32_t)lane + (uint32_t)j;
                    if (i < n) r[i] = v[q][j];
                }
        }
        return;
    }
    if (!has_coeffs) {
        // raw PCM: complete i16 pairs, zeros behind them; silence: zeros
        const uint8_t *p = A.bytes + c.off;
        const uint32_t pairs = has_res ? c.len >> 1 : 0u;
        for (uint32_t i = lane; i < n; i += 64)
            r[i] = i < pairs ? (int)(short)((uint32_t)p[2 * i] | ((uint32_t)p[2 * i + 1] << 8)) : 0;
        return;
    }

    // reconstruct_lpc_int
    if (!has_res) {   // no residual bytes: every residual, hence every sample, is 0
        for (uint32_t i = lane; i < n; i += 64) r[i] = 0;
        return;
    }
    // (n <= order: the residuals are the samples; everything longer is predict_rows' business)
}

// Four LPC wrappers per wavefront, one per row of sixteen lanes - the same transposed recurrence, but a finished sample
// reaches the row's other accumulators inside the multiply-add itself: v_fmac_f64 takes its first operand through DPP
// (row_newbcast:t = lane t of every row), so a step is v_floor_f64 + v_fmac_f64_dpp, no v_readlane and no scalar
// registers in the loop (the 64-lane form spends floor + 2 readlane + fma on ONE wrapper's step).
// TWO accumulators per lane: `cur` holds the prediction of sample l of the block being stepped, `nxt` that of sample l of
// the block behind it. Step t adds sample t's contribution to the later lanes of its own block (taps l - t - 1, l > t,
// into `cur` - the dependent chain) and, in the last MAXO steps, to the first lanes of the next block (taps
// 16 + l - t - 1 into `nxt` - a second multiply-add off the chain). At the block's end every lane's `cur` is its sample
// (one floor for all sixteen), `nxt` becomes `cur`, and the old `cur` register is read out and armed with the residuals of
// the block after next during the first steps of the next block.
// MAXO = 8 or 12 (the format's maximum): how many steps need the second multiply-add. Residuals are fetched a super-block
// of 16 x 16 samples ahead, so no step waits for memory.
template <int MAXO>
__device__ __forceinline__ void predict_rows(const LlParArgs &A, const unsigned chi, const int lane, double *cs, const bool act) {
    // (a row without an LPC wrapper of its own - `act` false - runs along on zeros and touches no memory)
    const int rl = lane & 15, row0 = lane & 48;
    const LlChannelDev *cd = A.ch + (act ? chi : 0u);
    int *r = A.scratch + cd->out_off;
    const uint32_t n = act ? cd->samples : 0u;
    const int order = act ? (int)cd->n_coeffs : 0;
    const uint32_t sh = cd->shift_bits & 63u;
    wave_sync_l();
    cs[lane] = rl < order ? ldexp((double)cd->coeffs[rl], -(int)sh) : 0.0;   // (entries order .. 15 are 0: order <= 12)
    wave_sync_l();
    // C1[t][lane]: what sample t adds to sample `lane` of its own block (tap lane - t - 1 for lane > t);
    // C2[t][lane]: ... to sample `lane` of the next block (tap 16 + lane - t - 1, only for the last MAXO steps)
    double C1[16], C2[MAXO];
#pragma unroll
    for (int t = 0; t < 16; t++) C1[t] = rl > t ? cs[row0 + (rl - t - 1)] : 0.0;
#pragma unroll
    for (int u = 0; u < MAXO; u++) {
        const int t = 16 - MAXO + u, k = 16 + rl - t - 1;   // k >= rl
        C2[u] = k < 16 ? cs[row0 + k] : 0.0;
    }
    double accA, accB = 0.0;
    {
        long long v = (uint32_t)rl < n ? (long long)r[rl] : 0;
        if (rl < order) {   // the first `order` samples are their residuals: take the loop's prediction off beforehand
            long long pred = 0;
            for (int j = 0; j < rl; j++) pred += (long long)cd->coeffs[j] * (long long)r[rl - 1 - j];
            v -= pred >> sh;
        }
        accA = (double)v;
    }
    // wave-uniform block count: the longest of the four wrappers (a shorter one runs on over zero residuals, unstored)
    uint32_t nmax = n;
#pragma unroll
    for (int k = 1; k < 4; k++) {
        const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)n, 16 * k);
        const uint32_t f = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmax);
        nmax = o > f ? o : f;
    }
    nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmax);
    const uint32_t nb = (nmax + 15u) >> 4;
    uint32_t nmin = 0xFFFFFFFFu;   // the shortest of the wrappers the wavefront really has
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)(act ? n : 0xFFFFFFFFu), 16 * k);
        nmin = o < nmin ? o : nmin;
    }
    // Memory is touched at ONE point per super-block of kSb blocks: wait for what was issued a super-block ago (the
    // residuals of this one, the samples of the one before - both long complete), then issue the next residual loads
    // and the finished samples' stores, then run kSb x 16 steps on registers. (Loads and stores inside the stepping
    // loop made the compiler wait for all outstanding memory operations at every block: the loop-carried count is
    // unknown to it, and a block then cost a memory round trip.)
    constexpr int kSb = 16;
    static_assert(kSb % 2 == 0, "the two accumulators swap roles block by block");
    auto fetch = [&](int (&buf)[kSb], const uint32_t b0) {
#pragma unroll
        for (int j = 0; j < kSb; j++) {
            const uint32_t i = 16u * (b0 + (uint32_t)j) + (uint32_t)rl;
            buf[j] = i < n ? r[i] : 0;
        }
    };
#ifdef FLO_PRED_DBG
    const unsigned long long dbg_t0 = __builtin_readcyclecounter(), dbg_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    int pf[kSb], outs[kSb];
    fetch(pf, 1u);   // (while block b is stepped, the other accumulator is armed with the residuals of block b + 1)
#pragma unroll
    for (int j = 0; j < kSb; j++) outs[j] = 0;
    double worst = 0.0;
    // One block: `cur` is stepped; `nxt` still holds the finished block before this one - its samples are read out and
    // the register is re-armed with `rn` (the residuals of the block behind this one) in the first four steps, one
    // instruction per step in the two wait states a DPP read needs behind the floor anyway (in program order behind
    // the block they stalled the wave: floor -> conversion -> maximum are a dependent chain of their own, and a wave
    // issues in order). The largest magnitude is checked once, behind the loop (v_cvt saturates meanwhile; a wrapper
    // that leaves the i32 range is decoded again by the serial kernel, so what is stored for it does not matter). Lanes
    // behind a shorter wrapper's end run on over zero residuals and are counted too: at worst a needless serial decode.
    // (A NaN can only follow a finite value beyond the range, which is recorded.)
    auto block = [&](double &cur, double &nxt, const int rn, int &out_prev) {
        double x, sv;
        asm("v_floor_f64 %1, %0\n\tv_floor_f64 %2, %3\n\ts_nop 0\n\tv_fmac_f64_dpp %0, %1, %4 row_newbcast:0 row_mask:0xf bank_mask:0xf"
            : "+v"(cur), "=&v"(x), "=&v"(sv)
            : "v"(nxt), "v"(C1[0]));
        asm("v_floor_f64 %1, %0\n\tv_cvt_i32_f64 %2, %3\n\ts_nop 0\n\tv_fmac_f64_dpp %0, %1, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf"
            : "+v"(cur), "=&v"(x), "=&v"(out_prev)
            : "v"(sv), "v"(C1[1]));
        asm("v_floor_f64 %1, %0\n\tv_max_f64 %2, %2, |%3|\n\ts_nop 0\n\tv_fmac_f64_dpp %0, %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf"
            : "+v"(cur), "=&v"(x), "+v"(worst)
            : "v"(sv), "v"(C1[2]));
        asm("v_floor_f64 %1, %0\n\tv_cvt_f64_i32 %2, %3\n\ts_nop 0\n\tv_fmac_f64_dpp %0, %1, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf"
            : "+v"(cur), "=&v"(x), "=&v"(nxt)
            : "v"(rn), "v"(C1[3]));
#pragma unroll
        for (int t = 4; t < 16; t++) {
            // (two wait states between a VALU write and a DPP read of the same register)
            if (t < 16 - MAXO) {
                asm("v_floor_f64 %1, %0\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                    : "+v"(cur), "=&v"(x)
                    : "v"(C1[t]), "n"(t));
            } else {
                asm("v_floor_f64 %2, %0\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %2, %3 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
                    "v_fmac_f64_dpp %1, %2, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
                    : "+v"(cur), "+v"(nxt), "=&v"(x)
                    : "v"(C1[t]), "v"(C2[t < 16 - MAXO ? 0 : t - (16 - MAXO)]), "n"(t));
            }
        }
    };
    static_assert(MAXO <= 12, "the other accumulator is re-armed in step 3: its first contribution comes in step 16 - MAXO");
    // blocks 0 .. nb: block nb is a dummy whose first steps read block nb - 1 out. outs[j] of super-block sb belongs to
    // block kSb sb + j - 1.
    for (uint32_t sb = 0; (uint32_t)kSb * sb < nb + 1u + (uint32_t)kSb; sb++) {   // (one extra round stores the last super-block)
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
        int cur[kSb];
#pragma unroll
        for (int j = 0; j < kSb; j++) cur[j] = pf[j];
        // (inside every wrapper of the wavefront - all super-blocks but the first two and the last - the sixteen loads and
        // stores are plain accesses at constant offsets from one address each, under one exec mask for the rows that have a
        // wrapper; with an index test and a branch per element they were 500 instructions per super-block, a third of the loop)
        if (sb >= 2u && 16u * ((uint32_t)kSb * (sb + 1u) + (uint32_t)kSb + 1u) <= nmin) {   // (uniform)
            if (act) {
                const int *rp = r + 16u * ((uint32_t)kSb * (sb + 1u) + 1u) + (uint32_t)rl;
                int *wp = r + 16u * ((uint32_t)kSb * (sb - 1u) - 1u) + (uint32_t)rl;
#pragma unroll
                for (int j = 0; j < kSb; j++) pf[j] = rp[16 * j];
#pragma unroll
                for (int j = 0; j < kSb; j++) wp[16 * j] = outs[j];
            }
        } else {
            fetch(pf, (uint32_t)kSb * (sb + 1u) + 1u);
            if (sb > 0) {
#pragma unroll
                for (int j = 0; j < kSb; j++) {
                    const uint32_t i = 16u * ((uint32_t)kSb * (sb - 1u) + (uint32_t)j - 1u) + (uint32_t)rl;   // (block -1: beyond n, nothing stored)
                    if (i < n) r[i] = outs[j];
                }
            }
        }
        if ((uint32_t)kSb * sb >= nb + 1u) break;
#pragma unroll
        for (int j = 0; j < kSb; j += 2) {
            block(accA, accB, cur[j], outs[j]);
            block(accB, accA, cur[j + 1], outs[j + 1]);
        }
    }
    const bool bad = !(worst < 2147483648.0);
#ifdef FLO_PRED_DBG
    if (lane == 0 && (blockIdx.x == 7 || (__builtin_amdgcn_s_memrealtime() - dbg_r0) > 110000ull)) printf("rows wg %u: nb %u cycles %llu realtime(100MHz) %llu\n", blockIdx.x, nb, (unsigned long long)(__builtin_readcyclecounter() - dbg_t0), (unsigned long long)(__builtin_amdgcn_s_memrealtime() - dbg_r0));
#endif
    if (bad) A.serial[chi] = 1;
}

// Does wrapper `chi` take the row form? (LPC with residuals, more samples than taps, not handed to the serial kernel)
__device__ __forceinline__ bool takes_rows(const LlParArgs &A, const unsigned chi, int &order) {
    order = 0;
    if (chi >= A.n_ch || A.serial[chi]) return false;
    const LlChannelDev *cd = A.ch + chi;
    order = cd->n_coeffs;
    return order > 0 && order <= 12 && cd->len > 0 && cd->samples > (uint32_t)order;
}

// Workgroups [0, ceil(n_ch / 4)): the LPC wrappers, four per wavefront; workgroups behind them: one per wrapper of the
// host's list of everything else (fixed predictors, raw, silent), so a frame's fixed-predictor channel never waits behind
// a recurrence.
__global__ __launch_bounds__(64) void ll_predict_kernel(LlParArgs A) {
    __shared__ double cs[64];
    const int lane = (int)threadIdx.x;
    const unsigned groups = (A.n_ch + 3u) / 4u;
    int order;
    if (blockIdx.x < groups) {
        const unsigned chi = 4u * blockIdx.x + ((unsigned)lane >> 4);
        const bool rows = takes_rows(A, chi, order);
        if (__ballot(rows) == 0ull) return;
        if (__ballot(rows && order > 8) == 0ull) predict_rows<8>(A, chi, lane, cs, rows);
        else predict_rows<12>(A, chi, lane, cs, rows);
        return;
    }
    const unsigned oi = blockIdx.x - groups;   // (uniform)
    if (oi >= A.n_others) return;
    const unsigned chi = A.others[oi];
#ifdef FLO_PRED_DBG
    const unsigned long long dbg_f0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (!takes_rows(A, chi, order)) predict_one(A, chi, lane, cs);
#ifdef FLO_PRED_DBG
    if (lane == 0 && (chi == 5 || (__builtin_amdgcn_s_memrealtime() - dbg_f0) > 60000ull)) printf("other wg %u chi %u realtime(100MHz) %llu\n", blockIdx.x, chi, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - dbg_f0));
#endif
}

// ------------------------------------------------------------------------------------------------ launcher
int launch_ll_decode_parallel(const LlParArgs &A, unsigned max_tiles, hipStream_t s) {
    if (!A.n_ch) return 0;
    if (max_tiles) {
        hipLaunchKernelGGL(ll_rice_scan_kernel, dim3(A.n_ch, (max_tiles + kScanWaves * kScanTiles - 1) / (kScanWaves * kScanTiles)), dim3(64 * kScanWaves), 0, s, A);
        hipLaunchKernelGGL(ll_rice_chain_kernel, dim3(A.n_ch), dim3(64), 0, s, A);
        hipLaunchKernelGGL(ll_rice_decode_kernel, dim3(A.n_ch, (max_tiles + 63) / 64), dim3(64), 0, s, A);
    }
    hipLaunchKernelGGL(ll_predict_kernel, dim3((A.n_ch + 3u) / 4u + A.n_others), dim3(64), 0, s, A);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    return 0;
}

}  // namespace flo
