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
constexpr int kMaxRecords = 1056;   // a record takes >= 2 bytes and a sparse blob is at most ~2.1 KB

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

__global__ __launch_bounds__(64) void lossy_decode_kernel(LossyDecArgs D) {
    // One 8 KiB buffer serves three phases of a channel-frame in turn: the parse table of the record headers, then the
    // integers + the FFT exchange buffer, then the windowed output. (Apart they were 29.6 KiB per wave: five waves per
    // CU; now 18.8 KiB: eight.)
    __shared__ __attribute__((aligned(16))) float u1[2048];
    short *const q = reinterpret_cast<short *>(u1);                                               // [1024]
    float (*const xch)[kXchFloats] = reinterpret_cast<float (*)[kXchFloats]>(u1 + 512);          // behind q
    float *const recon = u1;                                                                      // [2048]
    static_assert(512 * 4 + kXchFloats * 4 <= 2048 * 4, "q and the exchange buffer share the 8 KiB with room to spare");
    __shared__ float prev[1024];                // second half of the previous frame of the channel being walked
    __shared__ uint32_t rec[kMaxRecords];       // output index | count << 10 | byte position of the first value << 18
    __shared__ float sf[32];
    __shared__ int s_nrec;
    // the frame's bytes, staged whole (header, scale words, both channels' blobs: ~0.5 KB, 2.2 KB at most for what the
    // encoder makes); a frame that does not fit is parsed in global memory and only its blob comes here
    __shared__ __attribute__((aligned(16))) uint8_t sblob[kBlobStage + 16];
    __shared__ unsigned long long s_foff[kDecRunLong + 1];
    __shared__ uint32_t s_flen[kDecRunLong + 1];
    const int lane = (int)threadIdx.x;
    const unsigned clip = blockIdx.x;   // clips in x: gridDim.y stops at 65535
    if (clip >= (unsigned)D.n_clips) return;
    const unsigned nframes = D.clip_frames[clip];
    const unsigned run = (unsigned)D.run;
    const unsigned h0 = blockIdx.y * run;              // first frame of the run = first output block
    if (nframes < 2 || h0 + 1 >= nframes) return;
    const unsigned h1 = h0 + run < nframes - 1 ? h0 + run : nframes - 1;   // last frame of the run
    float *out = D.out + D.clip_out[clip];
    const float scale = 2.0f / 1024.0f;
    const float *win = D.window;

    // where the run's frames are: one load per lane instead of one dependent load per frame
    if ((unsigned)lane <= h1 - h0) {
        const unsigned long long f = D.clip_frame0[clip] + h0 + (unsigned)lane;
        s_foff[lane] = D.blob_off[f];
        s_flen[lane] = D.blob_len[f];
    }
    __syncthreads();
    // A frame travels global memory -> registers -> LDS, and the registers are filled one frame ahead: the walk over a
    // frame used to start with three dependent global round trips (offset, channel length, blob), which was most of
    // what a wave spent its time on.
    constexpr int kPre = (kBlobStage + 255) / 256;   // dwords per lane
    uint32_t pre[kPre];
    auto fetch = [&](unsigned h) {
        const uint8_t *g = D.bytes + s_foff[h - h0];
        const uint32_t len = s_flen[h - h0];
#pragma unroll
        for (int j = 0; j < kPre; j++) {
            const uint32_t i = 4u * ((uint32_t)lane + 64u * (uint32_t)j);
            uint32_t w4 = 0;
            if (len <= (uint32_t)kBlobStage && i < len) __builtin_memcpy(&w4, g + i, 4);   // up to 3 bytes past the frame: inside the file + slack
            pre[j] = w4;
        }
    };
    auto frame_ch = [&](const uint32_t c, const unsigned h, const uint8_t *data, const uint32_t len, auto IN_LDS) -> bool {
        constexpr bool in_lds = decltype(IN_LDS)::value;
        {
            // deserialize_frame: [block_size][channels][25 x u16 per channel][per channel: u32 len, sparse bytes]
            if (len < 2 || data[0] != 0 /* only Long blocks are produced or accepted */ || data[1] > D.channels) {
                if (lane == 0) atomicExch(D.error, 1);
                return false;
            }
            const uint32_t nch = data[1];
            uint32_t pos = 2 + 50 * nch;
            bool bad = pos > len;
            uint32_t blen = 0;
            for (uint32_t k = 0; k <= c && k < nch && !bad; k++) {   // walk to channel c's sparse blob
                if (pos + 4 > len) {
                    bad = true;
                    break;
                }
                blen = rd_u32(data + pos);
                pos += 4;
                if (pos + blen > len || pos + blen < pos) {
                    bad = true;
                    break;
                }
                if (k < c) pos += blen;
            }
            if (bad) {
                if (lane == 0) atomicExch(D.error, 1);
                return false;
            }
            const bool present = c < nch;      // a frame with fewer channels leaves the others silent
            if (present) {
                // the record walk below is a chain of dependent byte reads: from LDS it costs a tenth of what it
                // costs from global memory. Valid blobs are at most ~2.1 KB; anything longer is walked in place.
                const uint8_t *sp = data + pos;
                bool sp_lds = in_lds;
                if (!in_lds && blen <= (uint32_t)kBlobStage) {
                    for (uint32_t i = 4u * lane; i < blen; i += 256u) {
                        uint32_t w4;
                        __builtin_memcpy(&w4, sp + i, 4);   // up to 3 bytes past the blob: inside the file + slack
                        *reinterpret_cast<uint32_t *>(sblob + i) = w4;
                    }
                    __syncthreads();
                    sp = sblob;
                    sp_lds = true;
                }
                // scale factors: 2^((word - 32768) / 256), 0 when the word is 0 (decoder.rs:91-99)
                if (lane < 25) {
                    const uint32_t wv = rd_u16(data + 2 + 50 * c + 2 * lane);
                    // kept as the reciprocal: one IEEE division per band here instead of sixteen per lane below (q / sf
                    // becomes q * (1 / sf): one rounding more, 6e-8 relative, far inside the 2e-6 the transform allows)
                    const float pw = wv > 0 ? powf(2.0f, ((float)wv - 32768.0f) / 256.0f) : 0.0f;
                    sf[lane] = pw > 0.0f ? __fdiv_rn(1.0f, pw) : 0.0f;   // (2^-128 .. 2^128: the reciprocal is finite)
                }
                // deserialize_sparse (decoder.rs:134-167). The record headers form a chain (a record starts where the
                // previous one ends), so one lane has to follow it - but what it finds at a position does not depend on
                // how it got there: every lane first parses "a record starting here" for its share of the byte
                // positions (varint, count, length) into a table, and the chain walk is then ONE dependent LDS read per
                // record instead of three to five byte reads. The table borrows `recon`, which is idle until the inverse
                // transform; blobs beyond its 2048 entries (or walked in place) keep the byte-wise walk.
                const bool tabled = sp_lds && blen <= 2048u;
                uint32_t *ptab = reinterpret_cast<uint32_t *>(recon);
                if (tabled) {
                    for (uint32_t p0 = (uint32_t)lane; p0 < blen; p0 += 64u) {
                        uint32_t p = p0, value = 0, shift = 0;
                        while (p < blen) {   // decode_varint (:170-188)
                            const uint32_t b = sp[p++];
                            value |= (b & 0x7Fu) << shift;
                            if (!(b & 0x80u)) break;
                            shift += 7;
                            if (shift >= 32) break;
                        }
                        // entry: zero run (clamped: anything >= 1024 ends the walk) | count << 11 | bytes to the next
                        // record << 19 | "no count byte" << 30
                        uint32_t e = value < 2047u ? value : 2047u;
                        if (p >= blen) {
                            e |= 1u << 30;
                        } else {
                            const uint32_t nz = sp[p++];
                            const uint32_t avail = (blen - p) >> 1;
                            const uint32_t cnt = nz < avail ? nz : avail;
                            e |= (cnt << 11) | ((p + 2u * cnt - p0) << 19);
                        }
                        ptab[p0] = e;
                    }
                    __syncthreads();
                }
#ifdef FLO_DEC_ABLATE   // diagnostic: timing without the record walk (results invalid)
                if (lane == 0) s_nrec = 0;
#else
                if (lane == 0 && tabled) {
                    uint32_t p = 0, nrec = 0, oi = 0;
                    while (p < blen && oi < 1024u) {
                        const uint32_t e = ptab[p];
                        oi += e & 2047u;
                        if (e >> 30) break;
                        const uint32_t cnt_a = (e >> 11) & 255u, adv = (e >> 19) & 2047u;
                        const uint32_t room = oi < 1024u ? 1024u - oi : 0u;
                        const uint32_t cnt = cnt_a < room ? cnt_a : room;
                        if (cnt && nrec < kMaxRecords) {
                            rec[nrec] = oi | (cnt << 10) | ((p + adv - 2u * cnt_a) << 18);
                            nrec++;
                        }
                        p += adv;      // (a count cut by `room` ends the walk: oi reaches 1024)
                        oi += cnt;
                    }
                    s_nrec = (int)nrec;
                }
#endif
                if (tabled) {
                    __syncthreads();                                   // the table is dead: its memory becomes q
                    for (int i = lane; i < 1024; i += 64) q[i] = 0;
                    __syncthreads();
                    for (int r = lane; r < s_nrec; r += 64) {
                        const uint32_t o = rec[r] & 1023u, cnt = (rec[r] >> 10) & 255u;
                        const uint8_t *v = sp + (rec[r] >> 18);
                        for (uint32_t i = 0; i < cnt; i++) q[o + i] = (short)rd_u16(v + 2 * i);
                    }
                } else {
                    // a blob too long for the table (or walked in place): outside what the encoder produces. One lane
                    // walks it byte by byte and places the values itself.
                    for (int i = lane; i < 1024; i += 64) q[i] = 0;
                    __syncthreads();
                    if (lane == 0) {
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
                __syncthreads();
                // dequantise (decoder.rs:35-48) straight into the inverse transform's pre-rotation (mdct.rs:238-247)
                float zr[1][8], zi[1][8];
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int i = lane + 64 * r;
                    const int ke = 2 * i, ko = 1023 - 2 * i;
                    const float se = sf[D.T.band[ke]], so = sf[D.T.band[ko]];
                    const float even = (float)q[ke] * se;   // se, so: 1 / scale factor, 0 for a band without one
                    const float odd = -((float)q[ko] * so);
                    const float4 t4 = D.T.pack[(kRowTw + (r >> 1)) * 64 + lane];
                    const float2 w = (r & 1) ? make_float2(t4.z, t4.w) : make_float2(t4.x, t4.y);
                    zr[0][r] = odd * w.y - even * w.x;
                    zi[0][r] = odd * w.x + even * w.y;
                }
                fft512<1>(lane, zr, zi, xch, D.T);
                // post-rotation, scale 2 / 1024 and window (mdct.rs:252-287); every position is written exactly once
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int idx = lane + 64 * r;
                    const float4 t4 = D.T.pack[(kRowTw + (r >> 1)) * 64 + lane];
                    const float2 w = (r & 1) ? make_float2(t4.z, t4.w) : make_float2(t4.x, t4.y);
                    const float val_re = w.x * zr[0][r] + w.y * zi[0][r];
                    const float val_im = w.y * zr[0][r] - w.x * zi[0][r];
                    if (idx < 256) {
                        const int fi = 2 * idx, ri = 511 - 2 * idx;
                        recon[ri] = -val_im * scale * win[ri];
                        recon[512 + fi] = val_im * scale * win[512 + fi];
                        recon[1024 + ri] = val_re * scale * win[1024 + ri];
                        recon[1536 + fi] = val_re * scale * win[1536 + fi];
                    } else {
                        const int i2 = idx - 256;
                        const int fi = 2 * i2, ri = 511 - 2 * i2;
                        recon[fi] = -val_re * scale * win[fi];
                        recon[512 + ri] = val_re * scale * win[512 + ri];
                        recon[1024 + fi] = val_im * scale * win[1024 + fi];
                        recon[1536 + ri] = val_im * scale * win[1536 + ri];
                    }
                }
                __syncthreads();
            }
            // overlap-add (mdct.rs:449-456): block h - 1 = first half of this frame + second half of the previous one.
            // A channel the frame does not carry gives a silent block and leaves its overlap buffer alone (the
            // reference would fail on such a frame; the oracle behaves like this).
            if (h > h0) {
                for (int j = lane; j < 1024; j += 64) {
                    const float a = present ? recon[j] + prev[j] : 0.0f;
                    out[((unsigned long long)(h - 1) * 1024 + j) * D.channels + c] = a;
                }
            }
            if (present)
                for (int j = lane; j < 1024; j += 64) prev[j] = recon[1024 + j];
            __syncthreads();
        }
        return true;
    };

    for (uint32_t c = 0; c < (uint32_t)D.channels; c++) {
        for (int j = lane; j < 1024; j += 64) prev[j] = 0.0f;
        fetch(h0);
        for (unsigned h = h0; h <= h1; h++) {
            const uint32_t len = s_flen[h - h0];
            bool ok;
            if (len <= (uint32_t)kBlobStage) {
#pragma unroll
                for (int j = 0; j < kPre; j++) {
                    const uint32_t i = 4u * ((uint32_t)lane + 64u * (uint32_t)j);
                    if (i < len) *reinterpret_cast<uint32_t *>(sblob + i) = pre[j];
                }
                __syncthreads();
                if (h < h1) fetch(h + 1);
                ok = frame_ch(c, h, sblob, len, std::true_type{});
            } else {
                if (h < h1) fetch(h + 1);
                ok = frame_ch(c, h, D.bytes + s_foff[h - h0], len, std::false_type{});
            }
            if (!ok) return;
        }
    }
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
    if (max_frames < 2 || !A0.n_clips) return 0;
    LossyDecArgs A = A0;
    // a run of R blocks decodes R + 1 frames: long runs waste less, short ones give a single file enough wavefronts
    const unsigned long long blocks = (unsigned long long)A.n_clips * (max_frames - 1);
    A.run = blocks >= 8ull * 4096ull ? kDecRunLong : kDecRunShort;
    const unsigned runs = (max_frames - 1 + (unsigned)A.run - 1) / (unsigned)A.run;
    hipLaunchKernelGGL(lossy_decode_kernel, dim3((unsigned)A.n_clips, runs), dim3(64), 0, s, A);
    FLO_LAUNCH_CHECK();
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
    unsigned bx = (max_samples + 255) / 256;
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(ll_finish_kernel, dim3(A.n_frames, bx), dim3(256), 0, s, A);
    FLO_LAUNCH_CHECK();
    return 0;
}

}  // namespace flo
