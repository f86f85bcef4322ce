// decode_kernels.hip — device decode of .flo payloads for gfx950.
//
// Reference behaviour replaced (files under /root/reference/libflo/src):
//   transform frames : lossy/decoder.rs:29-52 (dequantise), :61-131 (deserialize_frame), :134-188 (sparse + varint),
//                      lossy/mdct.rs:231-290 (inverse MDCT), :437-468 (overlap-add), lib.rs:325-352 (first frame dropped)
//   ALPC frames      : core/rice.rs:123-159 (decode_i32) + :217-259 (BitReader), lossless/decoder.rs:92-273
//                      (decode_channel_int, reconstruct_lpc_int, reconstruct_fixed), :21-72 (mid/side, interleave)
// The inverse transform reuses the in-wave FFT-512 of the encoder (lossy_device.hpp). Rice decoding and the LPC
// recurrence are serial per channel of a frame, so the lossless side runs one thread per channel wrapper: slow per
// clip, parallel over a batch.
#include "decode_kernels.hpp"

#include <type_traits>

namespace flo {

// ------------------------------------------------------------------------------------------------ transform frames

__device__ __forceinline__ uint32_t rd_u16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ uint32_t rd_u32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

// One wavefront decodes a run of D.run (8 or 16) consecutive output blocks of one clip: output block b (1024 sample-frames) is
// the first half of frame b + 1 plus the second half of frame b (mdct.rs:449-456; the first frame's own block is
// dropped, lib.rs:338-341), so the run needs one frame more than blocks and keeps the previous frame's second half in LDS.
// Every output sample is written exactly once, by plain stores.
constexpr int kDecRunShort = 8, kDecRunLong = 16;   // output blocks per wavefront: short runs when there is little to decode
constexpr int kBlobStage = 2304;

// FLO_DEC_STAMPS (diagnostic builds only): s_memtime at the phase boundaries of a channel-frame, summed per wave and added
// to D.dbg at the end (launch_lossy_decode prints the shares).
#ifdef FLO_DEC_STAMPS
#define DSTAMP(i)                                                                   \
    do {                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                          \
        unsigned long long t_;                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                          \
        dst_sum[i] += t_ - dst_last;                                                \
        dst_last = t_;                                                              \
    } while (0)
#else
#define DSTAMP(i) do {} while (0)
#endif
__global__ __launch_bounds__(64, 3) void lossy_decode_kernel(LossyDecArgs D) {
    // One 8 KiB buffer serves three phases of a channel-frame in turn: the parse table of the record headers, then the
    // integers + the FFT exchange buffer, then the windowed output. Everything that does not change from frame to frame
    // lives in REGISTERS for the whole run (window x 2/1024 at the 32 positions the lane writes, rotation and FFT
    // twiddles, the band of each of its 16 coefficients: 64 global loads per channel-frame before), and so does the
    // second half of the previous frame (16 values per lane instead of a 4 KiB LDS buffer written and read back every
    // frame). 12 KiB of LDS per wave: twelve waves per CU (29.6 KiB / five in round 1, 18.8 KiB / eight in round 2).
    __shared__ __attribute__((aligned(16))) float u1[2048];
    short *const q = reinterpret_cast<short *>(u1);                                               // [1024]
    float (*const xch)[kXchFloats] = reinterpret_cast<float (*)[kXchFloats]>(u1 + 512);          // behind q
    float *const recon = u1;                                                                      // [2048]
    static_assert(512 * 4 + kXchFloats * 4 <= 2048 * 4, "q and the exchange buffer share the 8 KiB with room to spare");
    __shared__ float sf[32];
    // the frame's bytes, staged whole (header, scale words, both channels' blobs: ~0.5 KB, 2.2 KB at most for what the
    // encoder makes); a frame that does not fit is parsed in global memory and only its blob comes here
    __shared__ __attribute__((aligned(16))) uint8_t sblob[kBlobStage + 16];
    __shared__ unsigned long long s_foff[kDecRunLong + 1], s_boff[kDecRunLong + 1];   // where a frame of the run lies, and the bytes of the channel being walked
    __shared__ uint32_t s_blen[kDecRunLong + 1], s_flag[kDecRunLong + 1];
    // The workgroup is ONE wavefront: its LDS instructions execute in order, so the ordering points between the phases are
    // wave_sync() (a compiler fence), not __syncthreads() - whose s_waitcnt vmcnt(0) made every phase boundary wait for
    // the next frame's prefetch and for the previous block's output stores (several exposed HBM round trips per
    // channel-frame: most of what the kernel spent its time on).
    const int lane = (int)threadIdx.x;
    // One wavefront = one CHANNEL of one run of one clip. The channels of a run write the same cache lines of the
    // interleaved output (4 of every 4 x channels bytes each): their wavefronts must be on the same XCD - one L2 - and run
    // side by side, so that a line is complete before it leaves the L2. Workgroups are dealt round-robin to the eight
    // XCDs, so workgroups L and L + 8 are neighbours in time on one XCD: they are the two channels of one run. (One
    // wavefront walking the run once per channel wrote every line twice, a run's length apart in time: half-filled
    // lines went to memory twice - stereo decoded at 264 Gsamples/s where the same channel-frames as mono clips made 316.)
    const unsigned nc = (unsigned)D.channels;
    const unsigned L = blockIdx.x;
    const uint32_t c = (L >> 3) % nc;
    const unsigned long long P = (unsigned long long)(L / (8u * nc)) * 8u + (L & 7u);
    if (P >= (unsigned long long)D.n_clips * D.n_runs) return;
    const unsigned clip = (unsigned)(P % (unsigned)D.n_clips);   // (clips fastest)
    const unsigned nframes = D.clip_frames[clip];
    const unsigned run = (unsigned)D.run;
    const unsigned h0 = (unsigned)(P / (unsigned)D.n_clips) * run;   // first frame of the run = first output block
    if (nframes < 2 || h0 + 1 >= nframes) return;
    const unsigned h1 = h0 + run < nframes - 1 ? h0 + run : nframes - 1;   // last frame of the run
    float *out = D.out + D.clip_out[clip];
    const float scale = 2.0f / 1024.0f;
    // loop invariants of the run, in registers
    float wn[8][2];        // window x 2/1024 (a power of two: exact) at the positions row r of this lane writes: four positions,
                           // two values - the other two are their mirror images n <-> 2047 - n, and the window is symmetric
                           // (the table's two halves agree to the last ulp or two of f32: 1e-7 of the 2e-6 the decode is held to)
    uint32_t bnd[8];       // 4 x band of coefficient 2 i (low half) and of 1023 - 2 i (high half), i = lane + 64 r
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int idx = lane + 64 * r;
        const int ke = 2 * idx, ko = 1023 - 2 * idx;
        bnd[r] = 4u * (uint32_t)D.T.band[ke] | (4u * (uint32_t)D.T.band[ko]) << 16;
        const float *win = D.window;
        if (idx < 256) {
            const int fi = 2 * idx, ri = 511 - 2 * idx;
            wn[r][0] = scale * win[ri]; wn[r][1] = scale * win[512 + fi];      // = win[1536 + fi], win[1024 + ri]
        } else {
            const int i2 = idx - 256;
            const int fi = 2 * i2, ri = 511 - 2 * i2;
            wn[r][0] = scale * win[fi]; wn[r][1] = scale * win[512 + ri];      // = win[1536 + ri], win[1024 + fi]
        }
    }
    float pv[16];          // second half of the previous frame of the channel being walked: positions lane + 64 k
#ifdef FLO_DEC_STAMPS
    unsigned long long dst_sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, dst_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dst_last)::"memory");
#endif

    // Where the run's frames are, and inside each frame where the sparse bytes of channel c are (deserialize_frame:
    // [block_size][channels][25 x u16 per channel][per channel: u32 len, sparse bytes]): one lane per frame reads the few
    // header bytes once per run and channel. The walk over the frames then stages nothing but the bytes of the channel
    // it is decoding (a whole frame - both channels' bytes - was staged and its header walked again for every channel:
    // a fifth of a channel-frame's time). False: a frame of the run is malformed.
    auto index_run = [&](const uint32_t c) -> bool {
        wave_sync();
        if ((unsigned)lane <= h1 - h0) {
            const unsigned long long f = D.clip_frame0[clip] + h0 + (unsigned)lane;
            const unsigned long long foff = D.blob_off[f];
            const uint32_t flen = D.blob_len[f];
            const uint8_t *g = D.bytes + foff;
            bool bad = flen < 2;
            uint32_t nch = 0, pos = 0, blen = 0;
            if (!bad) {
                nch = g[1];
                bad = g[0] != 0 /* only Long blocks are produced or accepted */ || nch > (uint32_t)D.channels;
                pos = 2 + 50 * nch;
                bad = bad || pos > flen;
            }
            for (uint32_t k = 0; k <= c && k < nch && !bad; k++) {   // walk to channel c's sparse bytes
                if (pos + 4 > flen) {
                    bad = true;
                    break;
                }
                blen = rd_u32(g + pos);
                pos += 4;
                if (pos + blen > flen || pos + blen < pos) {
                    bad = true;
                    break;
                }
                if (k < c) pos += blen;
            }
            s_foff[lane] = foff;
            s_boff[lane] = foff + pos;
            s_blen[lane] = blen;
            s_flag[lane] = bad ? 2u : (c < nch ? 1u : 0u);   // 1: the frame carries channel c (one with fewer channels leaves the others silent)
        }
        wave_sync();
        bool any_bad = false;
        for (unsigned i = 0; i <= h1 - h0; i++) any_bad = any_bad || s_flag[i] == 2u;   // (uniform)
        if (any_bad && lane == 0) atomicExch(D.error, 1);
        return !any_bad;
    };
    // A channel's bytes travel global memory -> registers -> LDS, and the registers are filled one frame ahead: the walk
    // over a frame used to start with three dependent global round trips (offset, channel length, blob), which was most
    // of what a wave spent its time on.
    constexpr int kPre = (kBlobStage + 255) / 256;   // dwords per lane
    uint32_t pre[kPre], pre_sfw = 0;
    auto fetch = [&](const uint32_t c, const unsigned h) {
        const uint32_t blen = s_blen[h - h0];
        if (!s_flag[h - h0]) return;   // the frame does not carry this channel
        if (lane < 25) {
            const uint8_t *w = D.bytes + s_foff[h - h0] + 2u + 50u * c + 2u * (uint32_t)lane;
            pre_sfw = (uint32_t)w[0] | ((uint32_t)w[1] << 8);
        }
        if (blen > (uint32_t)kBlobStage) return;
        const uint8_t *g = D.bytes + s_boff[h - h0];
#pragma unroll
        for (int j = 0; j < kPre; j++) {
            if (256u * (uint32_t)j < blen) {   // uniform: a channel's bytes are about 0.25 KB, one of the nine rows
                const uint32_t i = 4u * ((uint32_t)lane + 64u * (uint32_t)j);
                uint32_t w4 = 0;
                if (i < blen) __builtin_memcpy(&w4, g + i, 4);   // up to 3 bytes past the blob: inside the file + slack
                pre[j] = w4;
            }
        }
    };
    // one channel of one frame: `staged` - its sparse bytes are in sblob (from its first byte); else they are walked where
    // they lie. `sfw`: the scale word of band `lane`.
    auto frame_ch = [&](const uint32_t c, const unsigned h, const bool present, const uint32_t blen, const bool staged, const uint32_t sfw) -> bool {
        // (the lane index behind an optimisation barrier: inside the frame loop every lane-derived address is recomputed - a
        // few integer operations - instead of being hoisted into dozens of loop-invariant registers)
        const int ln = lane_id_opaque();
        {
            DSTAMP(1);
            if (present) {
                // the record walk below is a chain of dependent byte reads: from LDS it costs a tenth of what it
                // costs from global memory. Valid blobs are at most ~2.1 KB; anything longer is walked in place.
                // Staged bytes are read through `sblob` itself: the compiler then knows they are in LDS (ds_read). Through
                // a generic pointer every byte is a FLAT load, and the wait behind a flat load is s_waitcnt vmcnt(0)
                // lgkmcnt(0) - it also waits for the next frame's prefetch and the previous block's output stores.
                const uint8_t *sp = D.bytes + s_boff[h - h0];   // (generic: only the in-place walk reads through it)
                const uint8_t *lsp = sblob;
                const bool sp_lds = staged;
                // scale factors: 2^((word - 32768) / 256), 0 when the word is 0 (decoder.rs:91-99)
                if (ln < 25) {
                    const uint32_t wv = sfw;
                    // kept as the reciprocal: one IEEE division per band here instead of sixteen per ln below (q / sf
                    // becomes q * (1 / sf): one rounding more, 6e-8 relative, far inside the 2e-6 the transform allows)
                    // (exp2f: the exponent is a multiple of 2^-8 in [-128, 128); one ulp from powf(2, .) at most, 1e-7 relative)
                    const float pw = wv > 0 ? exp2f(((float)wv - 32768.0f) / 256.0f) : 0.0f;
                    sf[ln] = pw > 0.0f ? __fdiv_rn(1.0f, pw) : 0.0f;   // (2^-128 .. 2^128: the reciprocal is finite)
                }
                // deserialize_sparse (decoder.rs:134-167). The record headers form a chain (a record starts where the
                // previous one ends) - but what is found at a byte position does not depend on how it was reached. So every
                // ln parses "a record starting here" for its share of the byte positions into a jump table (next record's
                // position, output positions covered), the table is squared six times (pointer doubling: entry p of level
                // k jumps 2^k records), and ln i reaches record i - and i + 64 - by the binary digits of i: a dozen
                // dependent LDS round trips for up to 128 records where one ln used to follow the chain record by record
                // (9.2 k of a channel-frame's 24 k ticks). Each ln then places its own records' values. Blobs of more
                // than 1024 bytes or 128 records (dense frames of high-quality encodes), and blobs walked in place, take the
                // byte-wise walk of one ln.
                constexpr uint32_t kEnd = 0xFFFu;
                // the four bytes at blob position p0 (the blob is in LDS: two aligned dwords and a byte shift; up to seven bytes
                // past p0 are touched, inside the staging buffer's slack)
                auto bytes_at = [&](const uint32_t p0) -> uint32_t {
                    const uint32_t a = (uint32_t)(uintptr_t)(lsp + p0);
                    const uint32_t *dw = reinterpret_cast<const uint32_t *>(lsp + p0 - (a & 3u));
                    return __builtin_amdgcn_alignbyte(dw[1], dw[0], a & 3u);
                };
                // "a record starting at p0", from its first bytes w: the usual header is [zero run < 128][count] or [two-byte
                // zero run][count]; anything else (longer varints, the blob's last bytes) takes the byte loop. True: count byte present.
                auto parse_eval = [&](const uint32_t p0, const uint32_t w, uint32_t &zr, uint32_t &cnt_a, uint32_t &adv) -> bool {
                    uint32_t p, value, nz;
                    if (!(w & 0x80u) && p0 + 2 <= blen) {
                        value = w & 0x7Fu;
                        nz = (w >> 8) & 0xFFu;
                        p = p0 + 1;
                    } else if ((w & 0x8080u) == 0x0080u && p0 + 3 <= blen) {
                        value = (w & 0x7Fu) | ((w >> 1) & 0x3F80u);
                        nz = (w >> 16) & 0xFFu;
                        p = p0 + 2;
                    } else if ((w & 0x8080u) == 0x8080u && (w & 0x7800u) != 0u && p0 + 2 <= blen) {
                        // three or more bytes and already >= 1024 after two: the walk ends at this record whatever follows
                        // (most byte positions are not record starts, and the high byte of a negative value looks like this)
                        zr = 2047u;
                        cnt_a = 0;
                        adv = 2;
                        return false;
                    } else {
                        p = p0, value = 0;
                        uint32_t shift = 0;
                        while (p < blen) {   // decode_varint (:170-188)
                            const uint32_t b = lsp[p++];
                            value |= (b & 0x7Fu) << shift;
                            if (!(b & 0x80u)) break;
                            shift += 7;
                            if (shift >= 32) break;
                        }
                        nz = p < blen ? (uint32_t)lsp[p] : 0u;
                    }
                    zr = value < 2047u ? value : 2047u;   // (anything >= 1024 ends the walk)
                    cnt_a = 0;
                    adv = p - p0;
                    if (p >= blen) return false;
                    p++;
                    const uint32_t avail = (blen - p) >> 1;
                    cnt_a = nz < avail ? nz : avail;
                    adv = p + 2u * cnt_a - p0;
                    return true;
                };
                uint32_t *ta = reinterpret_cast<uint32_t *>(recon), *tb = ta + 1024;
                bool fast = sp_lds && blen <= 1024u;
                // rotation and FFT twiddles of this ln: twelve 16-byte loads (12 KB per CU, resident in its L1) issued here,
                // a phase ahead of their use, and dead again after the transform: held across frames they were 48 registers
                // that cost the kernel a third of its resident waves
                float4 wtw[4], wf1[4], wf2[4];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    wtw[kk] = D.T.pack[(kRowTw + kk) * 64 + ln];
                    wf1[kk] = D.T.pack[(kRowF1 + kk) * 64 + ln];
                    wf2[kk] = D.T.pack[(kRowF2 + kk) * 64 + ln];
                }
                DSTAMP(2);
                uint32_t cur0 = blen ? 0u : kEnd, acc0 = 0, cur1 = kEnd, acc1 = 0;   // (every stored jump is kEnd or < blen)
                if (fast) {
                    // (four positions per ln at a time: the loads of a batch are in flight together)
                    for (uint32_t pb = (uint32_t)ln; pb < blen; pb += 256u) {
                        uint32_t w[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) w[u] = pb + 64u * u < blen ? bytes_at(pb + 64u * u) : 0u;
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const uint32_t p0 = pb + 64u * u;
                            if (p0 < blen) {
                                uint32_t zr, cnt_a, adv;
                                const bool has = parse_eval(p0, w[u], zr, cnt_a, adv);
                                const uint32_t nx = has && p0 + adv < blen ? p0 + adv : kEnd;
                                const uint32_t st = zr + cnt_a < 2047u ? zr + cnt_a : 2047u;
                                ta[p0] = nx | (st << 12);
                            }
                        }
                    }
                    wave_sync();
                    DSTAMP(3);
                    bool ended = false;   // the chain ended before record 2^k: no later level is needed
#pragma unroll 1
                    for (int k = 0; k < 6; k++) {
                        // one round trip: the level-k jump from the blob's start (is there a record 2^k at all?), the lane's own
                        // jump, and the entries to be squared
                        const uint32_t e0 = ta[0];
                        const uint32_t eu = cur0 != kEnd ? ta[cur0] : kEnd;
                        uint32_t e1[4], e2[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) e1[u] = (uint32_t)ln + 64u * u < blen ? ta[(uint32_t)ln + 64u * u] : kEnd;
                        if (((uint32_t)__builtin_amdgcn_readfirstlane((int)e0) & 0xFFFu) == kEnd) {   // uniform
                            if (((uint32_t)ln >> k) != 0u) cur0 = kEnd;   // records 2^k and up do not exist
                            ended = true;
                            break;
                        }
                        if ((((uint32_t)ln >> k) & 1u) && cur0 != kEnd) {
                            const uint32_t a = acc0 + (eu >> 12);
                            acc0 = a < 2047u ? a : 2047u;
                            cur0 = eu & 0xFFFu;
                        }
#pragma unroll
                        for (int u = 0; u < 4; u++) e2[u] = (e1[u] & 0xFFFu) != kEnd ? ta[e1[u] & 0xFFFu] : kEnd;
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const uint32_t a = (e1[u] >> 12) + (e2[u] >> 12);
                            if ((uint32_t)ln + 64u * u < blen) tb[(uint32_t)ln + 64u * u] = (e2[u] & 0xFFFu) | ((a < 2047u ? a : 2047u) << 12);
                        }
                        for (uint32_t pb = (uint32_t)ln + 256u; pb < blen; pb += 256u) {   // blobs of more than 256 bytes
#pragma unroll
                            for (int u = 0; u < 4; u++) e1[u] = pb + 64u * u < blen ? ta[pb + 64u * u] : kEnd;
#pragma unroll
                            for (int u = 0; u < 4; u++) e2[u] = (e1[u] & 0xFFFu) != kEnd ? ta[e1[u] & 0xFFFu] : kEnd;
#pragma unroll
                            for (int u = 0; u < 4; u++) {
                                const uint32_t a = (e1[u] >> 12) + (e2[u] >> 12);
                                if (pb + 64u * u < blen) tb[pb + 64u * u] = (e2[u] & 0xFFFu) | ((a < 2047u ? a : 2047u) << 12);
                            }
                        }
                        wave_sync();
                        uint32_t *t = ta;
                        ta = tb;
                        tb = t;
                    }
                    // ta = level 6 (jumps of 64 records): the lane's second record, and whether a 129th record matters
                    if (!ended) {
                        if (cur0 != kEnd) {
                            const uint32_t e = ta[cur0];
                            const uint32_t a = acc0 + (e >> 12);
                            acc1 = a < 2047u ? a : 2047u;
                            cur1 = e & 0xFFFu;
                        }
                        const uint32_t c1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur1), a1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)acc1);
                        if (c1 != kEnd) {   // uniform (lane 0 = records 0 and 64)
                            const uint32_t e = ta[c1];
                            if ((e & 0xFFFu) != kEnd && a1 + (e >> 12) < 1024u) fast = false;
                        }
                    }
                }
                DSTAMP(4);
                if (fast) {
                    // the ln's records: output index, count, where the values lie (parsed again: the tables are about to be q)
                    uint32_t o[2] = {1024u, 1024u}, n[2] = {0u, 0u}, vp[2] = {0u, 0u};
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const uint32_t cu = u ? cur1 : cur0, ac = u ? acc1 : acc0;
                        if (cu != kEnd && ac < 1024u) {
                            uint32_t zr, cnt_a, adv;
                            const bool has = parse_eval(cu, bytes_at(cu), zr, cnt_a, adv);
                            const uint32_t oi = ac + zr;
                            if (has && oi < 1024u) {
                                const uint32_t room = 1024u - oi;
                                o[u] = oi;
                                n[u] = cnt_a < room ? cnt_a : room;
                                vp[u] = cu + adv - 2u * cnt_a;
                            }
                        }
                    }
                    wave_sync();
                    {
                        uint4 *qz = reinterpret_cast<uint4 *>(q);
                        qz[ln] = make_uint4(0u, 0u, 0u, 0u);
                        qz[64 + ln] = make_uint4(0u, 0u, 0u, 0u);
                    }
                    wave_sync();
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const uint8_t *v = lsp + vp[u];
                        for (uint32_t i = 0; i < n[u]; i++) q[o[u] + i] = (short)rd_u16(v + 2 * i);
                    }
                } else {
                    // a blob too long for the tables, with more than 128 records, or walked in place: dense frames of
                    // high-quality encodes at most. One ln walks it byte by byte and places the values itself.
                    wave_sync();
                    for (int i = ln; i < 1024; i += 64) q[i] = 0;
                    wave_sync();
                    if (ln == 0) {
                        uint32_t p = 0;
                        unsigned long long oi = 0;
                        while (p < blen && oi < 1024) {
                            uint32_t value = 0, shift = 0;
                            while (p < blen) {   // decode_varint (:170-188)
                                const uint32_t b = sp[p++];
                                value |= (b & 0x7Fu) << shift;
                                if (!(b & 0x80u)) break;
                                shift += 7;
                                if (shift >= 32) break;
                            }
                            oi += value;
                            if (p >= blen) break;
                            const uint32_t nz = sp[p++];
                            const uint32_t avail = (blen - p) >> 1;
                            const uint32_t room = oi < 1024 ? (uint32_t)(1024 - oi) : 0u;
                            uint32_t cnt = nz < avail ? nz : avail;
                            cnt = cnt < room ? cnt : room;
                            for (uint32_t i = 0; i < cnt; i++) q[(uint32_t)oi + i] = (short)rd_u16(sp + p + 2 * i);
                            p += 2 * cnt;
                            oi += cnt;
                        }
                    }
                }
                wave_sync();
                DSTAMP(5);
                // dequantise (decoder.rs:35-48) straight into the inverse transform's pre-rotation (mdct.rs:238-247)
                float zr[1][8], zi[1][8];
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int i = ln + 64 * r;
                    const int ke = 2 * i, ko = 1023 - 2 * i;
                    const float se = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(sf) + (bnd[r] & 0xFFFFu));
                    const float so = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(sf) + (bnd[r] >> 16));
                    const float even = (float)q[ke] * se;   // se, so: 1 / scale factor, 0 for a band without one
                    const float odd = -((float)q[ko] * so);
                    const float2 w = (r & 1) ? make_float2(wtw[r >> 1].z, wtw[r >> 1].w) : make_float2(wtw[r >> 1].x, wtw[r >> 1].y);
                    zr[0][r] = odd * w.y - even * w.x;
                    zi[0][r] = odd * w.x + even * w.y;
                }
                DSTAMP(6);
                fft512_w(ln, zr[0], zi[0], xch[0], wf1, wf2);
                DSTAMP(7);
                // post-rotation, scale 2 / 1024 and window (mdct.rs:252-287); every position is written exactly once
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int idx = ln + 64 * r;
                    const float2 w = (r & 1) ? make_float2(wtw[r >> 1].z, wtw[r >> 1].w) : make_float2(wtw[r >> 1].x, wtw[r >> 1].y);
                    const float val_re = w.x * zr[0][r] + w.y * zi[0][r];
                    const float val_im = w.y * zr[0][r] - w.x * zi[0][r];
                    if (r < 4) {   // idx < 256
                        const int fi = 2 * idx, ri = 511 - 2 * idx;
                        recon[ri] = -val_im * wn[r][0];
                        recon[512 + fi] = val_im * wn[r][1];
                        recon[1024 + ri] = val_re * wn[r][1];
                        recon[1536 + fi] = val_re * wn[r][0];
                    } else {
                        const int i2 = idx - 256;
                        const int fi = 2 * i2, ri = 511 - 2 * i2;
                        recon[fi] = -val_re * wn[r][0];
                        recon[512 + ri] = val_re * wn[r][1];
                        recon[1024 + fi] = val_im * wn[r][1];
                        recon[1536 + ri] = val_im * wn[r][0];
                    }
                }
                wave_sync();
            }
            DSTAMP(8);
            // overlap-add (mdct.rs:449-456): block h - 1 = first half of this frame + second half of the previous one
            // (carried in registers). A channel the frame does not carry gives a silent block and leaves the overlap
            // alone (the reference would fail on such a frame; the oracle behaves like this).
            if (h > h0) {
                // (a uniform block base and a 32-bit lane offset: the stores take the scalar-base form, no 64-bit address
                // arithmetic per store)
                float *ob = out + ((unsigned long long)(h - 1) * 1024) * D.channels + c;
                const uint32_t nchu = (uint32_t)D.channels;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const uint32_t j = (uint32_t)ln + 64u * (uint32_t)k;
                    const float a = present ? recon[j] + pv[k] : 0.0f;
                    ob[j * nchu] = a;
                }
            }
            if (present) {
#pragma unroll
                for (int k = 0; k < 16; k++) pv[k] = recon[1024 + ln + 64 * k];
            }
            wave_sync();
            DSTAMP(9);
        }
        return true;
    };

    {
#pragma unroll
        for (int k = 0; k < 16; k++) pv[k] = 0.0f;
        if (!index_run(c)) return;
        fetch(c, h0);
        for (unsigned h = h0; h <= h1; h++) {
            const uint32_t blen = s_blen[h - h0];
            const bool present = s_flag[h - h0] != 0u;
            const bool staged = present && blen <= (uint32_t)kBlobStage;
            const uint32_t sfw = pre_sfw;
            if (staged) {
#pragma unroll
                for (int j = 0; j < kPre; j++) {
                    if (256u * (uint32_t)j < blen) {   // uniform
                        const uint32_t i = 4u * ((uint32_t)lane + 64u * (uint32_t)j);
                        if (i < blen) *reinterpret_cast<uint32_t *>(sblob + i) = pre[j];
                    }
                }
                wave_sync();
            }
            if (h < h1) fetch(c, h + 1);
            DSTAMP(0);
            if (!frame_ch(c, h, present, blen, staged, sfw)) return;
        }
    }
#ifdef FLO_DEC_STAMPS
    if (D.dbg && lane == 0)
        for (int i = 0; i < 10; i++) atomicAdd(D.dbg + i, dst_sum[i]);
#endif
}

// ------------------------------------------------------------------------------------------------ ALPC frames
// MSB-first bit reader with the semantics of rice.rs:217-259 (a bit past the end reads as 0 and does not advance;
// "exhausted" = every real bit consumed), buffered: up to 64 bits sit in a register, refilled a byte at a time, so the
// unary part of a Rice code is one count-leading-ones instead of a loop over bits. (A deeper variant that prefetched
// aligned 8-byte words one ahead was measured and was no faster: the per-sample control flow, not the loads, is
// what a single serial lane spends its time on.)
struct BitRd {
    const uint8_t *p;
    uint32_t len, pos;          // payload length, next byte to load
    unsigned long long buf;     // next bit = bit 63; bits below the valid ones are 0
    int cnt;                    // valid (real) bits in buf
    __device__ __forceinline__ void init(const uint8_t *ptr, uint32_t n) {
        p = ptr;
        len = n;
        pos = 0;
        buf = 0;
        cnt = 0;
    }
    __device__ __forceinline__ void refill() {
        if (cnt <= 32 && pos + 4 <= len) {   // four bytes at once (any alignment), big-endian into the queue
            uint32_t w;
            __builtin_memcpy(&w, p + pos, 4);
            buf |= (unsigned long long)__builtin_bswap32(w) << (32 - cnt);
            pos += 4;
            cnt += 32;
        }
        while (cnt <= 56 && pos < len && pos + 4 > len) {   // the last three bytes
            buf |= (unsigned long long)p[pos++] << (56 - cnt);
            cnt += 8;
        }
    }
    __device__ __forceinline__ void consume(int n) {
        buf = n >= 64 ? 0ull : buf << n;
        cnt -= n;
    }
    __device__ __forceinline__ bool exhausted() const { return cnt == 0 && pos >= len; }
};

__device__ __forceinline__ int rice_next(BitRd &r, uint32_t k) {   // one value of decode_i32 (rice.rs:127-155)
    if (r.exhausted()) return 0;
    // unary quotient: ones until a zero (which is consumed), at most 256 of them (the 256th ends the run without a
    // terminator), or until the stream runs out
    uint32_t quotient = 0;
    for (;;) {
        r.refill();
        if (r.cnt == 0) break;
        const unsigned long long inv = ~r.buf;
        int ones = inv ? __clzll((long long)inv) : 64;
        if (ones > r.cnt) ones = r.cnt;
        const int room = 256 - (int)quotient;
        const int take = ones < room ? ones : room;
        const int before = r.cnt;
        r.consume(take);
        quotient += (uint32_t)take;
        if (quotient == 256u) break;
        if (take < before) {   // the next real bit is the terminating zero
            r.consume(1);
            break;
        }
    }
    // k remainder bits, zeros past the end
    uint32_t rem = 0;
    if (k <= 32u) {
        r.refill();
        if (k) {
            rem = (uint32_t)(r.buf >> (64 - k));
            r.consume((int)k < r.cnt ? (int)k : r.cnt);
        }
    } else {
        for (uint32_t i = 0; i < k; i++) {
            r.refill();
            uint32_t bit = 0;
            if (r.cnt) {
                bit = (uint32_t)(r.buf >> 63);
                r.consume(1);
            }
            rem = (rem << 1) | bit;
        }
    }
    const uint32_t u = (quotient << (k & 31u)) | rem;   // release-mode Rust masks the shift amount
    return (int)(u >> 1) ^ -(int)(u & 1u);
}

// reconstruct_lpc_int (decoder.rs:152-184): the first ORDER values are the residuals themselves, then
// s[i] = ((sum_j coef[j] * s[i - 1 - j]) >> shift) + r[i] with i64 accumulation and a wrapping add
template <int ORDER>
__device__ __forceinline__ void lpc_decode(const LlChannelDev &c, const uint8_t *res, int *out) {
    BitRd r;
    r.init(res, c.len);
    int hist[ORDER];   // hist[j] = s[i - 1 - j]
#pragma unroll
    for (int j = 0; j < ORDER; j++) hist[j] = 0;
    const uint32_t sh = c.shift_bits & 63u;
    const uint32_t n = c.samples;
    for (uint32_t i = 0; i < n; i++) {
        const int rv = rice_next(r, c.rice_k);
        int v = rv;
        if (i >= (uint32_t)ORDER) {
            long long pred = 0;
#pragma unroll
            for (int j = 0; j < ORDER; j++) pred += (long long)c.coeffs[j] * (long long)hist[j];
            v = (int)((unsigned)(int)(pred >> sh) + (unsigned)rv);
        }
        out[i] = v;
#pragma unroll
        for (int j = ORDER - 1; j > 0; j--) hist[j] = hist[j - 1];
        hist[0] = v;
    }
}

// SPREAD = 1: one channel wrapper per wavefront (lane 0 only). The walk is serial and data-dependent, so lanes of one
// wave that decode different streams execute the union of their paths (measured: 380 instructions per sample with 20
// lanes against ~150 alone); with few wrappers (a single file) a wave each is faster, with many the lanes are needed.
template <int SPREAD>
__global__ __launch_bounds__(64) void ll_decode_kernel(LlDecArgs A) {
    const unsigned t = SPREAD ? blockIdx.x : blockIdx.x * blockDim.x + threadIdx.x;
    if (SPREAD && threadIdx.x != 0) return;
    if (t >= A.n_ch) return;
    if (A.only && !A.only[t]) return;
    const LlChannelDev c = A.ch[t];
    int *out = A.scratch + c.out_off;
    const uint8_t *res = A.bytes + c.off;
    const uint32_t n = c.samples;
    const bool has_coeffs = c.n_coeffs > 0, has_res = c.len > 0;
    if (!has_coeffs && has_res && c.shift_bits >= 128) {
        // fixed predictor (decoder.rs:186-266): warm-up with lower orders, then binomial recurrences, wrapping adds
        const int order = c.shift_bits - 128;
        BitRd r;
        r.init(res, c.len);
        int s1 = 0, s2 = 0, s3 = 0, s4 = 0;   // s[i-1] .. s[i-4]
        for (uint32_t i = 0; i < n; i++) {
            const int rv = rice_next(r, c.rice_k);
            int eff = order > 4 ? 0 : order;   // unknown orders copy the residuals
            if ((int)i < eff) eff = (int)i;
            long long pred = 0;
            if (eff == 1) pred = s1;
            else if (eff == 2) pred = 2ll * s1 - s2;
            else if (eff == 3) pred = 3ll * s1 - 3ll * s2 + s3;
            else if (eff == 4) pred = 4ll * s1 - 6ll * s2 + 4ll * s3 - s4;
            const int v = (int)((unsigned)rv + (unsigned)(int)pred);
            out[i] = v;
            s4 = s3; s3 = s2; s2 = s1; s1 = v;
        }
        return;
    }
    if (has_coeffs) {
        // reconstruct_lpc_int (decoder.rs:152-184), one instantiation per order so that the history shift and the
        // multiply-adds are exactly `order` long
        switch (c.n_coeffs) {
            case 1: lpc_decode<1>(c, res, out); break;
            case 2: lpc_decode<2>(c, res, out); break;
            case 3: lpc_decode<3>(c, res, out); break;
            case 4: lpc_decode<4>(c, res, out); break;
            case 5: lpc_decode<5>(c, res, out); break;
            case 6: lpc_decode<6>(c, res, out); break;
            case 7: lpc_decode<7>(c, res, out); break;
            case 8: lpc_decode<8>(c, res, out); break;
            case 9: lpc_decode<9>(c, res, out); break;
            case 10: lpc_decode<10>(c, res, out); break;
            case 11: lpc_decode<11>(c, res, out); break;
            default: lpc_decode<12>(c, res, out); break;
        }
        return;
    }
    if (has_res) {   // raw PCM: complete i16 pairs, zero padding
        uint32_t k = 0;
        for (uint32_t i = 0; i + 1 < c.len && k < n; i += 2) out[k++] = (int)(short)rd_u16(res + i);
        while (k < n) out[k++] = 0;
        return;
    }
    for (uint32_t i = 0; i < n; i++) out[i] = 0;   // silence
}

// mid/side (decoder.rs:76-90: truncating halves of wrapping sums), interleave, i32 -> f32 (audio_constants.rs:23-26)
__global__ __launch_bounds__(256) void ll_finish_kernel(LlFinishArgs A) {
    const unsigned f = blockIdx.x;   // frames in x: gridDim.y stops at 65535
    if (f >= A.n_frames) return;
    const LlFrameDev fr = A.fr[f];
    const float scale = 1.0f / 32767.0f;
    // stereo frames whose planes and output are 16-byte aligned: four sample frames per thread, two 16-byte loads and two
    // 16-byte stores per output (one sample frame per thread, 4-byte accesses, ran at 40 % of this)
    if (A.channels == 2 && fr.n_channels == 2 && A.out && !A.out_i32) {
        const unsigned long long o0 = fr.scratch_off[0], o1 = fr.mid_side ? fr.scratch_off[1] : A.ch[fr.first_channel + 1].out_off;
        const unsigned long long oa = fr.mid_side ? o0 : A.ch[fr.first_channel].out_off;
        if (((oa | o1) & 3ull) == 0ull && (fr.out_off & 1ull) == 0ull && ((uintptr_t)A.out & 15u) == 0u) {
            const int4 *pa = reinterpret_cast<const int4 *>(A.scratch + oa), *pb = reinterpret_cast<const int4 *>(A.scratch + o1);
            float4 *po = reinterpret_cast<float4 *>(A.out + fr.out_off * 2ull);
            const unsigned quads = fr.samples >> 2;
            const bool ms = fr.mid_side != 0;
            for (unsigned q = blockIdx.y * blockDim.x + threadIdx.x; q < quads; q += gridDim.y * blockDim.x) {
                const int4 a = pa[q], b = pb[q];
                const int av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
                float l[4], r[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int li = ms ? (int)((unsigned)av[j] + (unsigned)bv[j]) / 2 : av[j];
                    const int ri = ms ? (int)((unsigned)av[j] - (unsigned)bv[j]) / 2 : bv[j];
                    l[j] = (float)li * scale;
                    r[j] = (float)ri * scale;
                }
                po[2u * q] = make_float4(l[0], r[0], l[1], r[1]);
                po[2u * q + 1u] = make_float4(l[2], r[2], l[3], r[3]);
            }
            for (unsigned i = 4u * quads + blockIdx.y * blockDim.x + threadIdx.x; i < fr.samples; i += gridDim.y * blockDim.x) {   // (up to three)
                const int av = A.scratch[oa + i], bv = A.scratch[o1 + i];
                const int li = ms ? (int)((unsigned)av + (unsigned)bv) / 2 : av, ri = ms ? (int)((unsigned)av - (unsigned)bv) / 2 : bv;
                A.out[(fr.out_off + i) * 2ull] = (float)li * scale;
                A.out[(fr.out_off + i) * 2ull + 1ull] = (float)ri * scale;
            }
            return;
        }
    }
    for (unsigned i = blockIdx.y * blockDim.x + threadIdx.x; i < fr.samples; i += gridDim.y * blockDim.x) {
        if (fr.mid_side && fr.n_channels == 2) {
            const int m = A.scratch[fr.scratch_off[0] + i], s = A.scratch[fr.scratch_off[1] + i];
            const int l = (int)((unsigned)m + (unsigned)s) / 2, r = (int)((unsigned)m - (unsigned)s) / 2;
            const unsigned long long o = (fr.out_off + i) * 2;
            if (A.out) { A.out[o] = (float)l * scale; A.out[o + 1] = (float)r * scale; }
            if (A.out_i32) { A.out_i32[o] = l; A.out_i32[o + 1] = r; }
        } else {
            for (unsigned c = 0; c < fr.n_channels && c < (unsigned)A.channels; c++) {
                const int v = A.scratch[A.ch[fr.first_channel + c].out_off + i];
                const unsigned long long o = (fr.out_off + i) * A.channels + c;
                if (A.out) A.out[o] = (float)v * scale;
                if (A.out_i32) A.out_i32[o] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ launchers
#define FLO_LAUNCH_CHECK()                      \
    do {                                        \
        hipError_t e_ = hipGetLastError();      \
        if (e_ != hipSuccess) return (int)e_;   \
    } while (0)

int launch_lossy_decode(const LossyDecArgs &A0, unsigned max_frames, hipStream_t s) {
    if (max_frames < 2 || !A0.n_clips || A0.channels < 1) return 0;
    LossyDecArgs A = A0;
    // a run of R blocks decodes R + 1 frames: long runs waste less, short ones give a single file enough wavefronts
    const unsigned long long blocks = (unsigned long long)A.n_clips * (max_frames - 1);
    A.run = blocks >= 8ull * 4096ull ? kDecRunLong : kDecRunShort;
    const unsigned runs = (max_frames - 1 + (unsigned)A.run - 1) / (unsigned)A.run;
    A.n_runs = runs;
    const unsigned long long pairs = ((unsigned long long)A.n_clips * runs + 7ull) / 8ull * 8ull, wgs = pairs * (unsigned)(A.channels > 0 ? A.channels : 1);
    if (wgs > 0x7FFFFFFFull) return -1;
#ifdef FLO_DEC_STAMPS
    unsigned long long *d_dbg = nullptr;
    if (hipMalloc(&d_dbg, 80) != hipSuccess || hipMemset(d_dbg, 0, 80) != hipSuccess) return -1;
    A.dbg = d_dbg;
#endif
    hipLaunchKernelGGL(lossy_decode_kernel, dim3((unsigned)wgs), dim3(64), 0, s, A);
    FLO_LAUNCH_CHECK();
#ifdef FLO_DEC_STAMPS
    {
        unsigned long long h[10];
        hipStreamSynchronize(s);
        hipMemcpy(h, d_dbg, 80, hipMemcpyDeviceToHost);
        hipFree(d_dbg);
        static const char *nm[10] = {"stage+prefetch", "header", "stage-blob+scale", "parse-table", "walk", "zero+place", "dequant+prerot", "fft", "postrot+window", "overlap-add+store"};
        double tot = 0;
        for (int i = 0; i < 10; i++) tot += (double)h[i];
        const double fc = (double)blocks * A.channels * ((double)(A.run + 1) / A.run);
        fprintf(stderr, "[dec stamps] ticks per channel-frame:");
        for (int i = 0; i < 10; i++) fprintf(stderr, " %s=%.0f", nm[i], (double)h[i] / fc);
        fprintf(stderr, " total=%.0f\n", tot / fc);
    }
#endif
    return 0;
}
int launch_ll_decode(const LlDecArgs &A, hipStream_t s) {
    if (!A.n_ch) return 0;
    if (A.n_ch <= 8192 || A.only) hipLaunchKernelGGL(ll_decode_kernel<1>, dim3(A.n_ch), dim3(64), 0, s, A);
    else hipLaunchKernelGGL(ll_decode_kernel<0>, dim3((A.n_ch + 63) / 64), dim3(64), 0, s, A);
    FLO_LAUNCH_CHECK();
    return 0;
}
int launch_ll_finish(const LlFinishArgs &A, unsigned max_samples, hipStream_t s) {
    if (!A.n_frames || !max_samples) return 0;
    unsigned bx = (max_samples + 1023) / 1024;   // (a thread of the stereo path takes four sample frames)
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(ll_finish_kernel, dim3(A.n_frames, bx), dim3(256), 0, s, A);
    FLO_LAUNCH_CHECK();
    return 0;
}

}  // namespace flo
