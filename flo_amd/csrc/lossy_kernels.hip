// lossy_kernels.hip — gfx950 kernels of the lossy encode path and their launchers.
//
//   lossy_chain2q_kernel     stereo batches (the default and the benchmarked kernel): per clip one TRANSFORM wave that carries
//                            both channels in lock-step (packed f32) through fold, FFT, post-rotation, band statistics and
//                            masking pass, and one PACKER wave that quantises, serialises and flushes; persistent
//                            workgroups deal the clips dynamically.
//   lossy_chain_kernel<NW>   one wavefront per (clip, channel) walks the clip's frames in order (the psychoacoustic
//                            model's 25-float temporal state, psychoacoustic.rs:198-203, lives in registers),
//                            reads every PCM sample once and appends finished frame bytes to the clip's DATA
//                            chunk: replaces the hot loop of TransformEncoder::encode_to_flo (encoder.rs:200-225).
//                            Mono batches, and the independently written cross-check of the stereo form.
//   lossy_frame_kernel<CH,P> frame-parallel form for few/long clips: P=1 computes only the per-band masking
//                            level before temporal masking, lossy_scan_kernel resolves the recurrence, P=2
//                            re-runs the transform and finishes each frame into a fixed slot; compact_kernel
//                            packs the slots. lossy_frame2x_kernel is its stereo form built from the lock-step device
//                            functions, lossy_frame_n_kernel<P> the same for 3 to 8 channels.
//   All forms produce identical bytes (tests compare them file by file).
#include <stdlib.h>

#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

#include "lossy_device.hpp"
#include "lossy_kernels.hpp"
#include "../../include/flo_synth.h"

// Diagnostic builds only (never shipped): FLO_ABLATE=n cuts the chain kernel's frame body short so that phase costs can
// be read from timing differences; intermediate values are kept alive so nothing upstream is optimised away.
//   0 full | 1 no flush/barriers | 2 no emit | 3 no sparse plan | 4 no quantise | 5 no band stats/psy | 6 loads+fold only
#ifndef FLO_ABLATE
#define FLO_ABLATE 0
#endif
// FLO_ABLATE3=n (three-wave form): 1 packer does nothing | 2 + channel waves stop after the transform |
//   3 + no post-rotation/transposition | 4 + no FFT (loads and fold only)
#ifndef FLO_SKIP
#define FLO_SKIP 0
#endif
#ifndef FLO_ABLATE3
#define FLO_ABLATE3 0
#endif
#define FLO_KEEP(x) asm volatile("" ::"v"(x))
// FLO_STAMPS (diagnostic builds only): s_memtime at phase boundaries of the chain kernel, summed per wave.
#ifdef FLO_STAMPS
#define STAMP(i)                                                                                  \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        unsigned long long t_;                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        st_sum[i] += t_ - st_last;                                                                \
        st_last = t_;                                                                             \
    } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

namespace flo {

// ---------------------------------------------------------------------------------------------- frame body
template <int CH>
struct FrameState {
    float prev[CH];  // lanes 0..24: temporal masking state of band `lane`
};

// Everything between the MDCT and the byte stream for CH channels held by this wave: band statistics, masking
// level, temporal masking, scale factors, quantiser, sparse-RLE plan. ch0 = index of c[0] among the clip's channels.
#ifdef FLO_STAMPS
#define ASTAMP(i)                                                                                 \
    do {                                                                                          \
        if (stamps) {                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                    \
            unsigned long long t_;                                                                \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
            __builtin_amdgcn_sched_barrier(0);                                                    \
            stamps[i] += t_ - stamps[15];                                                         \
            stamps[15] = t_;                                                                      \
        }                                                                                         \
    } while (0)
#else
#define ASTAMP(i) do {} while (0)
#endif
template <int CH, bool BANDS_ONLY, bool EXACT, bool PLAN = true>
__device__ __forceinline__ void analyse_frame(const int lane, float (&c)[CH][16], WaveLds<CH> &lds, const LaneConst &L,
                                              const LossyArgs &A, const LossyDevTables &T, int ch0, FrameState<CH> &st,
                                              unsigned long long gframe, int (&q)[CH][16], uint32_t (&sfw)[CH],
                                              SparsePlan (&P)[CH], unsigned long long *stamps = nullptr) {
    float energy[CH], bmax[CH];
    band_stats<CH>(lane, c, lds.slots, T, energy, bmax);
    ASTAMP(10);
#pragma unroll
    for (int ch = 0; ch < CH; ch++) {
        float a = spread_threshold(lane, energy[ch], L.rcount, T);
        if (BANDS_ONLY) {
            if (lane < 25) A.a_t[(gframe * A.nch + ch0 + ch) * 32 + lane] = a;
            continue;
        }
        // temporal masking (psychoacoustic.rs:196-203)
        float s = max_raw(a, st.prev[ch] * 0.7f);
        st.prev[ch] = s;
        const float tlin = masking_amplitude(s, T.smr_thr);
        // scale factor (encoder.rs:121-127): IEEE division, as the reference
        const float sf = bmax[ch] > 1e-10f ? __fdiv_rn(30000.0f, bmax[ch]) : 1.0f;
        sfw[ch] = sf_word(sf);
        if (lane < 25) {
            lds.bandv[ch][lane] = make_float2(tlin, sf);
            if (EXACT) lds.band_s[ch][lane] = s;
        }
    }
    if (BANDS_ONLY) return;
    wave_sync();
    ASTAMP(11);
#if FLO_ABLATE >= 4
    for (int ch = 0; ch < CH; ch++) { FLO_KEEP(sfw[ch]); for (int e = 0; e < 16; e++) q[ch][e] = 0; P[ch].total = 3; P[ch].off0 = 0; P[ch].M = 0; }
    return;
#endif
    quantise<CH, EXACT>(lane, c, lds, L, T, q);
    ASTAMP(12);
#if FLO_ABLATE >= 3
    for (int ch = 0; ch < CH; ch++) { P[ch].total = 3; P[ch].off0 = 0; P[ch].M = 0; }
    return;
#endif
    if (A.dbg_q) {
#pragma unroll
        for (int ch = 0; ch < CH; ch++) {
            short *dq = A.dbg_q + (gframe * A.nch + ch0 + ch) * 1024 + 16 * lane;
#pragma unroll
            for (int e = 0; e < 16; e++) dq[e] = (short)q[ch][e];
        }
    }
    if (A.dbg_sfw && lane < 25) {
#pragma unroll
        for (int ch = 0; ch < CH; ch++) A.dbg_sfw[(gframe * A.nch + ch0 + ch) * 25 + lane] = (unsigned short)sfw[ch];
    }
    if (!PLAN) return;
#pragma unroll
    for (int ch = 0; ch < CH; ch++) sparse_plan(lane, q[ch], P[ch]);
}

// Write this wave's share of the frame bytes (writer.rs:236-254 + encoder.rs:243-280) into the staging buffer.
// tot[c] = sparse bytes of channel c for ALL nch channels of the frame (uniform); the wave that holds channel 0
// also writes the frame and blob headers. Returns the frame length.
template <int CH>
__device__ __forceinline__ uint32_t emit_frame(const int lane, uint8_t *f, int nch, int ch0, const uint32_t *tot,
                                               const uint32_t (&sfw)[CH], const SparsePlan (&P)[CH],
                                               const int (&q)[CH][16]) {
    uint32_t pos = 12 + 50 * (uint32_t)nch;
    uint32_t chpos[CH];   // positions of this call's channels ch0 .. ch0 + CH - 1
    for (int c = 0; c < nch; c++) {
        if (c >= ch0 && c < ch0 + CH) chpos[c - ch0] = pos;
        pos += 4 + tot[c];
    }
    const uint32_t flen = pos, blob_len = flen - 10;
    if (ch0 == 0 && lane == 0) {
        f[0] = 253;
        f[1] = 0x00; f[2] = 0x04; f[3] = 0; f[4] = 0;  // frame_samples = 1024
        f[5] = 0;
        f[6] = (uint8_t)blob_len; f[7] = (uint8_t)(blob_len >> 8); f[8] = (uint8_t)(blob_len >> 16); f[9] = (uint8_t)(blob_len >> 24);
        f[10] = 0;  // BlockSize::Long
        f[11] = (uint8_t)nch;
    }
#pragma unroll
    for (int ch = 0; ch < CH; ch++) {
        const int c = ch0 + ch;
        if (lane < 25) {
            uint8_t *p = f + 12 + 50 * c + 2 * lane;
            lds_st8<0>(p, sfw[ch]);
            lds_st8<1>(p, sfw[ch] >> 8);
        }
        if (lane == 32) {
            const uint32_t l = tot[c];
            uint8_t *p = f + chpos[ch];
            p[0] = (uint8_t)l; p[1] = (uint8_t)(l >> 8); p[2] = (uint8_t)(l >> 16); p[3] = (uint8_t)(l >> 24);
        }
    }
    // trash bytes: past the end of the whole frame, two per lane and channel (the staging buffer has the slack)
    uint8_t *dsts[CH];
    uint32_t trash[CH];
#pragma unroll
    for (int ch = 0; ch < CH; ch++) {
        dsts[ch] = f + chpos[ch] + 4;
        trash[ch] = (flen - (chpos[ch] + 4)) + 2u * (uint32_t)lane + 128u * (uint32_t)ch;
    }
    sparse_emit_n<CH>(lane, q, P, dsts, trash);
    return flen;
}

// MDCT of one frame of CH channels: halves (ae,ao) + (be,bo) -> c (contiguous layout)
template <int CH>
__device__ __forceinline__ void mdct_frame(const int lane, const float (&ae)[CH][8], const float (&ao)[CH][8],
                                           const float (&be)[CH][8], const float (&bo)[CH][8], WaveLds<CH> &lds,
                                           const LossyDevTables &T, float (&c)[CH][16]) {
    float zr[CH][8], zi[CH][8];
    fold<CH>(lane, ae, ao, be, bo, zr, zi, T);
    fft512<CH>(lane, zr, zi, lds.u.xch, T);
    post_rotate_transpose<CH>(lane, zr, zi, lds.u.coef, c, T);
}

template <int CH>
__device__ __forceinline__ void store_coeffs_dbg(const int lane, const float (&c)[CH][16], const LossyArgs &A,
                                                 unsigned long long gframe, int ch0) {
    if (!A.dbg_coeffs) return;
#pragma unroll
    for (int ch = 0; ch < CH; ch++) {
        float4 *d = reinterpret_cast<float4 *>(A.dbg_coeffs + (gframe * A.nch + ch0 + ch) * 1024 + 16 * lane);
#pragma unroll
        for (int q = 0; q < 4; q++) d[q] = make_float4(c[ch][4 * q], c[ch][4 * q + 1], c[ch][4 * q + 2], c[ch][4 * q + 3]);
    }
}

template <int CH>
__device__ __forceinline__ void load_coeffs(const int lane, float (&c)[CH][16], const LossyArgs &A, unsigned long long gframe, int ch0) {
#pragma unroll
    for (int ch = 0; ch < CH; ch++) {
        const float4 *s = reinterpret_cast<const float4 *>(A.in_coeffs + (gframe * A.nch + ch0 + ch) * 1024 + 16 * lane);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float4 v = s[q];
            c[ch][4 * q] = v.x; c[ch][4 * q + 1] = v.y; c[ch][4 * q + 2] = v.z; c[ch][4 * q + 3] = v.w;
        }
    }
}

// ---------------------------------------------------------------------------------------------- chain kernel
// One wavefront per (clip, channel); a workgroup hosts G clips (NW = channels = 1 or 2 waves each) that share ONE
// copy of the constant pack in LDS (46 KiB: window, twiddles, ATH thresholds, band bookkeeping), so the only
// global traffic of the frame loop is the PCM stream in and the bitstream out. Each wave walks its channel's
// frames in order: the raw samples of the overlapping half-frame and the 25-float masking state stay in registers,
// so every PCM sample is read from HBM once. The two waves of a stereo clip meet twice per frame (LDS flag
// hand-shakes, no workgroup barrier: other clips of the workgroup never wait) to assemble and flush the frame.
constexpr int kPackBytes = kPackRows * 64 * 16;
struct ClipLds {
    WaveLds<1> wl[2];
    __attribute__((aligned(16))) uint8_t stage[kFrameCap + 64 + 128];
    uint32_t tot[2];
    uint32_t cnt[2];
};
static_assert(sizeof(ClipLds) % 16 == 0, "clip LDS block keeps 16-byte alignment");

// Rendezvous of the two waves of one clip: publish my step, wait for the partner's. Both waves are resident in the
// same workgroup, so the wait cannot deadlock; LDS is one unit per CU, so a wave's earlier LDS writes are visible
// to whoever observes its counter.
__device__ __forceinline__ void pair_sync(uint32_t *cnt, int w, uint32_t step) {
    // explicit DS instructions: a volatile access through a generic pointer would become a system-scope flat load
    const uint32_t mine = (uint32_t)(uintptr_t)(cnt + w), theirs = (uint32_t)(uintptr_t)(cnt + (w ^ 1));
    asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" ::"v"(mine), "v"(step) : "memory");
    uint32_t seen;
    do {
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(theirs) : "memory");
        if (seen >= step) break;
        __builtin_amdgcn_s_sleep(1);
    } while (true);
}

#ifndef FLO_CHAIN_WAVES_PER_SIMD
#define FLO_CHAIN_WAVES_PER_SIMD 3
#endif
template <int NW, bool EXACT>
__global__ __launch_bounds__(768, FLO_CHAIN_WAVES_PER_SIMD) void lossy_chain_kernel(LossyArgs A, int clips_per_wg) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    // constant pack -> LDS (all waves of the workgroup, once)
    {
        float4 *dstp = reinterpret_cast<float4 *>(lds_raw);
        for (int i = tid; i < kPackRows * 64; i += (int)blockDim.x) dstp[i] = A.T.pack[i];
        // rendezvous counters must be zero before any wave can look at its partner's
        for (int i = tid; i < clips_per_wg; i += (int)blockDim.x) {
            ClipLds &c0 = *reinterpret_cast<ClipLds *>(lds_raw + kPackBytes + (size_t)i * sizeof(ClipLds));
            c0.cnt[0] = 0;
            c0.cnt[1] = 0;
        }
    }
    __syncthreads();
    // wave-uniform values are made provably uniform: the clip's pointers, sizes and the frame loop then live in
    // SGPRs and on the scalar unit (64-bit address arithmetic and compares cost 4-cycle VALU slots otherwise)
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cl = wv / NW;                 // clip slot inside the workgroup
    const int w = NW == 1 ? 0 : wv % NW;    // channel of this wave
    const unsigned clip = blockIdx.x * (unsigned)clips_per_wg + (unsigned)cl;
    if (clip >= (unsigned)A.n_clips) return;
    ClipLds &cs = *reinterpret_cast<ClipLds *>(lds_raw + kPackBytes + (size_t)cl * sizeof(ClipLds));
    WaveLds<1> &lds = cs.wl[w];
    uint8_t *stage = cs.stage;
    if (lane == 0) lds.slots[0][kZeroSlot] = make_float2(0.f, 0.f);
    LossyDevTables T = A.T;
    T.pack = reinterpret_cast<const float4 *>(lds_raw);

    const float *pcm = A.pcm + A.clip_off[clip];
    const unsigned hops = A.clip_hops[clip];
    const unsigned long long frame0 = A.clip_frame0[clip];
    uint8_t *gout = A.out + A.out_off[clip];
    const int ptid = NW == 1 ? lane : (w * 64 + lane);  // thread index inside the clip's wave pair

    FrameState<1> st;
    st.prev[0] = 0.f;
    float ae[1][8], ao[1][8], be[1][8], bo[1][8];
#pragma unroll
    for (int r = 0; r < 8; r++) ae[0][r] = ao[0][r] = 0.f;  // pre-roll: 1024 zeros (encoder.rs:177)
    if (!A.in_coeffs) {
        load_half_fast<1>(lane, pcm, NW, w, 0, be, bo);   // the batch pads every clip with zeros to hops * 1024 sample-frames
    }
    unsigned long long written = 0;
    uint32_t pend = 0, step = 0, tailb = 0;
#ifdef FLO_STAMPS
    unsigned long long st_sum[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
    // One frame. (pe, po) hold the frame's first half, (ce, co) its second half = the next frame's first half.
    auto frame_body = [&](const unsigned h, float (&pe)[1][8], float (&po)[1][8], float (&ce)[1][8],
                          float (&co)[1][8]) __attribute__((always_inline)) {
        const int ln = lane_id_opaque();
        {
            unsigned zero = 0;  // keep the (rarely used) global tables out of loop-invariant registers
            asm volatile("" : "+s"(zero));
            T.ath_db += zero;
        }
        float c[1][16];
        if (A.in_coeffs) {
            load_coeffs<1>(ln, c, A, frame0 + h, w);
        } else {
            float zr[1][8], zi[1][8];
            fold<1>(ln, pe, po, ce, co, zr, zi, T);
            STAMP(0);
            // the older half is dead after the fold: the half-frame after next is loaded into its registers now and
            // consumed at the top of the next call, where the two register sets have swapped roles
            // unconditional, also behind the last frame (the batch allocates one spare half-frame per clip): a
            // conditional load would merge "loaded" and "kept" registers and cost 16 copies per frame
            load_half_fast<1>(ln, pcm, NW, w, (long long)(h + 1) * 1024, pe, po);
#if FLO_ABLATE >= 6
            for (int r = 0; r < 8; r++) { FLO_KEEP(zr[0][r]); FLO_KEEP(zi[0][r]); }
            return;
#endif
            STAMP(1);
            fft512<1>(ln, zr, zi, lds.u.xch, T);
            STAMP(2);
            post_rotate_transpose<1>(ln, zr, zi, lds.u.coef, c, T);
            STAMP(3);
            store_coeffs_dbg<1>(ln, c, A, frame0 + h, w);
        }
#if FLO_ABLATE >= 5
        for (int e = 0; e < 16; e++) FLO_KEEP(c[0][e]);
        return;
#endif
        int q[1][16];
        uint32_t sfw[1];
        SparsePlan P[1];
        {
            LaneConst L;
            load_lane_const(ln, L, T);
#ifdef FLO_STAMPS
            st_sum[15] = st_last;
            analyse_frame<1, false, EXACT>(ln, c, lds, L, A, T, w, st, frame0 + h, q, sfw, P, st_sum);
            st_last = st_sum[15];
#else
            analyse_frame<1, false, EXACT>(ln, c, lds, L, A, T, w, st, frame0 + h, q, sfw, P);
#endif
        }
        STAMP(4);
#if FLO_ABLATE >= 2
        FLO_KEEP(P[0].total); FLO_KEEP(P[0].off0); FLO_KEEP(P[0].M); FLO_KEEP(sfw[0]);
        for (int e = 0; e < 16; e++) FLO_KEEP(q[0][e]);
        return;
#endif
        uint32_t tot[2];
        if (NW > 1) {
            if (ln == 0) cs.tot[w] = P[0].total;
            pair_sync(cs.cnt, w, ++step);   // both plans are known; the previous frame's flush has been read out
            STAMP(5);
            tot[0] = cs.tot[0];
            tot[1] = cs.tot[1];
        } else {
            wave_sync();
            tot[0] = P[0].total;
            tot[1] = 0u;
        }
        // the < 16 bytes the previous frame left unflushed go back to the front of the staging buffer
        if (ptid < (int)pend) stage[ptid] = (uint8_t)tailb;
        const uint32_t flen = emit_frame<1>(ln, stage + pend, NW, w, tot, sfw, P, q);
        STAMP(6);
#if FLO_ABLATE >= 1
        FLO_KEEP(flen);
        return;
#endif
        if (NW > 1) pair_sync(cs.cnt, w, ++step); else wave_sync();
        STAMP(7);
        if (ptid == 0) A.frame_size[frame0 + h] = flen;
        // flush complete 16-byte chunks; the rest is carried in a register until the next frame's rendezvous
        const uint32_t have = pend + flen;
        const uint32_t n16 = have >> 4;
        const uint4 *src = reinterpret_cast<const uint4 *>(stage);
        uint4 *dst = reinterpret_cast<uint4 *>(gout + written);
        for (uint32_t i = ptid; i < n16; i += 64 * NW) dst[i] = src[i];
        pend = have & 15u;
        tailb = (ptid < (int)pend) ? stage[(n16 << 4) + ptid] : 0u;
        written += (unsigned long long)n16 << 4;
        STAMP(8);
    };
    // two frames per trip so that the two half-frame register sets alternate roles without copies
    for (unsigned h = 0; h < hops; h += 2) {
        frame_body(h, ae, ao, be, bo);
        if (h + 1 < hops) frame_body(h + 1, be, bo, ae, ao);
    }
    if (ptid < (int)pend) gout[written + ptid] = (uint8_t)tailb;
    if (ptid == 0) A.clip_bytes[clip] = written + pend;
#ifdef FLO_STAMPS
    if (A.dbg_stamps && lane == 0)
        for (int i = 0; i < 14; i++) A.dbg_stamps[((unsigned long long)clip * NW + w) * 16 + i] = st_sum[i];
#endif
}

// ---------------------------------------------------------------------------------------------- counters between the waves of a clip
// wait until the LDS counter at `p` reaches `want` (written by another wave of this workgroup)
template <int SLEEP = 1>
__device__ __forceinline__ void wait_counter(const uint32_t *p, uint32_t want) {
    const uint32_t a = (uint32_t)(uintptr_t)p;
    uint32_t seen;
    do {
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(a) : "memory");
        if (seen >= want) break;
        __builtin_amdgcn_s_sleep(SLEEP);
    } while (true);
}
// publish: every LDS access this wave issued before is performed first (a wave's LDS instructions execute in order)
// The counter's value now, through a ds_read the compiler tracks (it waits where the value is first used): issued well
// ahead of the hand-over, it turns the usual case - the partner is already there - into no wait at all.
__device__ __forceinline__ uint32_t peek_counter(const uint32_t *p) {
    typedef volatile __attribute__((address_space(3))) uint32_t lds_vu32;
    return *reinterpret_cast<lds_vu32 *>((uintptr_t)(uint32_t)(uintptr_t)p);
}
__device__ __forceinline__ void set_counter(uint32_t *p, uint32_t v) {
    const uint32_t a = (uint32_t)(uintptr_t)p;
    asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory");
}

#ifndef FLO_C2X_THREADS
#define FLO_C2X_THREADS 768
#endif
// DIRTY (template parameter of the lock-step chain kernel): the element positions (bit e of 16) at which some lane of the band
// table closes a segment; the other positions skip the slot store and the restart multiplication of band_stats_2. 0xFFFF
// serves every table; the launcher picks the instantiation made for 44.1 kHz when the table agrees.
constexpr uint32_t kDirty44k = 0xBDBEu;
#ifdef FLO_MARKS   // diagnostic builds: section markers in the assembly listing (they pin the schedule: never shipped)
#define FLO_MARK(x) asm volatile("; MARK " x ::: "memory")
#else
#define FLO_MARK(x)
#endif

// ---------------------------------------------------------------------------------------------- two waves per clip
// Stereo clips. One TRANSFORM wave per clip carries both channels in lock-step: every constant row (window, twiddles, band
// tables) is read from LDS once per frame for both channels, the PCM comes in as float2 loads (both channels of a
// sample-frame), every arithmetic instruction of fold, FFT and post-rotation is a packed one, and the two channels are
// two independent dependency chains inside one instruction stream. It stops after the masking pass (fold, FFT,
// post-rotation, band statistics, masking level, scale factors); the PACKER wave quantises, serialises and flushes.
// Twelve waves land round-robin on four SIMDs, so two SIMDs hold (transform, transform, packer) and two hold
// (transform, packer, packer): with the quantiser in the transform wave (rounds 2 and 3) the first kind carried 1941
// vector instructions per frame round against 1350 and set the launch time. What crosses the waves:
//   * the f32 coefficients, which are NOT copied: the packer reads them (one 16-byte read per block of 128 positions, in
//     the layout its sparse packer wants: no re-dealing of integers either) out of the transposition buffer the
//     post-rotation left them in, as soon as `coef_ready` says so, and answers `consumed`; the transform wave needs the
//     buffer again for the first FFT exchange of the NEXT frame;
//   * 25 x (threshold amplitude, scale factor) per channel, the scale words and the band-alive ballot, in tables of
//     their own chosen by frame parity (`ts_ready`): the packer may still be reading frame h - 1's while frame h's are
//     written.
// The quantiser's per-position constants (ATH thresholds, band offsets) live in the packer's registers for the whole
// launch (rows kRowAthN / kRowBoN, read from global memory once): they are not in the LDS pack, and there is no i16
// hand-over buffer. Same device functions where the work is the same, same bytes as the other forms.
struct Clip2qLds {
    union {
        float4 xch4[kXch4];      // FFT exchanges
        float2 coef2[kCoef2];    // coefficient k of both channels at element k + 2 (k >> 4): written by the post-rotation, read by both waves
    } u;
    float4 slot[kSlots];         // band-statistics slots (see band_stats_2)
    float4 ts[2][32];            // [frame parity][band]: amplitude thresholds (left, right), scale factors (left, right)
    __attribute__((aligned(16))) uint8_t stage[kFrameCap + 64 + 256];
    uint16_t sfwh[2][2][32];     // [frame parity][channel][band] scale words
    uint32_t alive[2][2];        // [frame parity][channel]: bit b = band b holds a coefficient above its masking amplitude
    uint32_t pad2[4];
    uint32_t packtab[kPackTabDwords];     // item list + record table of the item-form packer / run table of the block form
    uint32_t coef_ready;         // frames whose coefficients are in coef2
    uint32_t ts_ready;           // frames whose band tables are published
    uint32_t consumed;           // frames whose coefficients the packer has taken
    uint32_t clip_seq;           // clips handed to the transform wave so far ...
    uint32_t clip_cur;           // ... and the latest one
    uint32_t pad[3];
};
static_assert(sizeof(Clip2qLds) % 16 == 0, "clip LDS block keeps 16-byte alignment");
constexpr int kPackBytesHotT = kPackRowsHotT * 64 * 16;

#ifndef FLO_PSLEEP
#define FLO_PSLEEP 1
#endif
template <bool COEFFS, uint32_t DIRTY, bool DBG>
__global__ __launch_bounds__(FLO_C2X_THREADS) void lossy_chain2q_kernel(LossyArgs A, int clips_per_wg) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    {
        float4 *dstp = reinterpret_cast<float4 *>(lds_raw);
        for (int i = tid; i < kPackRowsHotT * 64; i += (int)blockDim.x) dstp[i] = A.T.pack[i];
        for (int i = tid; i < clips_per_wg; i += (int)blockDim.x) {
            Clip2qLds &c0 = *reinterpret_cast<Clip2qLds *>(lds_raw + kPackBytesHotT + (size_t)i * sizeof(Clip2qLds));
            c0.coef_ready = 0;
            c0.ts_ready = 0;
            c0.consumed = 0;
            c0.clip_seq = 0;
        }
    }
    __syncthreads();
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // waves 0 .. g-1 are the transform waves, waves g .. 2g-1 the packers
    const int cl = wv % clips_per_wg;
    Clip2qLds &cs = *reinterpret_cast<Clip2qLds *>(lds_raw + kPackBytesHotT + (size_t)cl * sizeof(Clip2qLds));
    // Clips are dealt dynamically: the workgroups are persistent (one per CU, the LDS holds no second one) and every
    // (transform wave, packer wave) pair takes the next unclaimed clip of the batch when it has finished one, so CUs
    // stay full until the batch runs out whatever the clip lengths. The packer claims (one atomic per clip) and tells
    // its transform wave through LDS; frame counters run on across clips (fbase), so nothing is ever reset. The two
    // roles are two separate loops so that neither's registers are live in the other's code.
    uint32_t fbase = 0, seq = 0;
    if (wv >= clips_per_wg) {
        // ------------------------------------------------------------------ packer: quantise, serialise, flush
        // its clip slot waits for this wave (the transform wave has a third of a frame to spare): it goes first on its SIMD
#ifndef FLO_PRIO_P
#define FLO_PRIO_P 1
#endif
        __builtin_amdgcn_s_setprio(FLO_PRIO_P);
        typedef __attribute__((address_space(3))) v4f lds_v4f;
        float athn[16];
        uint32_t ts_a[16];
        {
            const uint32_t ts0 = (uint32_t)(uintptr_t)cs.ts[0];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const float4 a4 = A.T.pack[(kRowAthN + q) * 64 + lane];
                const float4 b4 = A.T.pack[(kRowBoN + q) * 64 + lane];
                athn[4 * q + 0] = a4.x, athn[4 * q + 1] = a4.y, athn[4 * q + 2] = a4.z, athn[4 * q + 3] = a4.w;
                ts_a[4 * q + 0] = ts0 + __float_as_uint(b4.x), ts_a[4 * q + 1] = ts0 + __float_as_uint(b4.y);
                ts_a[4 * q + 2] = ts0 + __float_as_uint(b4.z), ts_a[4 * q + 3] = ts0 + __float_as_uint(b4.w);
            }
        }
        const uint32_t *const blk_g = reinterpret_cast<const uint32_t *>(A.T.pack + kRowBlk * 64);
        // the lane's 16 bytes of block k of the coefficient buffer: element 144 k + 2 lane + 2 (lane >> 3)
        const uint32_t cf_a = (uint32_t)(uintptr_t)cs.u.coef2 + 16u * (uint32_t)lane + 16u * ((uint32_t)lane >> 3);
        uint8_t *const stage = cs.stage;
        const uint32_t tab_a = (uint32_t)(uintptr_t)cs.packtab;
        v4f cf[8];
        bool have = false;   // cf holds the next frame's coefficients (taken before the previous frame's flush)
        for (;;) {
            unsigned got = 0;
            if (lane == 0) got = atomicAdd(A.next_clip, 1u);
            const unsigned clip = (unsigned)__builtin_amdgcn_readfirstlane((int)got);
            if (lane == 0) cs.clip_cur = clip;
            set_counter(&cs.clip_seq, ++seq);
            if (clip >= (unsigned)A.n_clips) return;
            const unsigned hops = A.clip_hops[clip];
            const unsigned long long frame0 = A.clip_frame0[clip];
            uint8_t *gout = A.out + A.out_off[clip];
            unsigned long long written = 0;
            uint32_t pend = 0, tailb = 0;
#ifdef FLO_STAMPS
            unsigned long long st_sum[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
            for (unsigned h = 0; h < hops; h++) {
                const int ln = lane_id_opaque();
                const uint32_t g = fbase + h;
                // The next frame's coefficients are taken as soon as the transform wave has them and this wave's registers
                // are free (behind the quantiser, between the channels' blobs, in front of the flush): the sooner the
                // transform wave has its buffer back, the less it waits in front of its next FFT exchange.
                auto take_next = [&]() __attribute__((always_inline)) {
                    if (!have && (uint32_t)__builtin_amdgcn_readfirstlane((int)peek_counter(&cs.coef_ready)) >= g + 2) {
#pragma unroll
                        for (int k = 0; k < 8; k++) cf[k] = *reinterpret_cast<const lds_v4f *>((uintptr_t)(cf_a + 1152u * (uint32_t)k));
                        set_counter(&cs.consumed, g + 2);
                        have = true;
                    }
                };
                // bands present in each block of 128 positions: eight scalars fetched per frame (one s_load, answered by the
                // scalar cache while this wave waits for its partner) rather than held across the packer's scalar-heavy code
                uint32_t blk[8];
                {
                    const uint32_t *bp = blk_g;
                    asm volatile("" : "+s"(bp));
#pragma unroll
                    for (int q = 0; q < 8; q++) blk[q] = bp[q];
                }
                if (!have) {
                    wait_counter<FLO_PSLEEP>(&cs.coef_ready, g + 1);
#pragma unroll
                    for (int k = 0; k < 8; k++) cf[k] = *reinterpret_cast<const lds_v4f *>((uintptr_t)(cf_a + 1152u * (uint32_t)k));
                    set_counter(&cs.consumed, g + 1);   // (behind the reads: a wave's LDS instructions execute in order)
                }
                STAMP(0);
#ifdef FLO_STAMPS
                const unsigned long long st_frame0 = st_last;
#endif
                wait_counter<FLO_PSLEEP>(&cs.ts_ready, g + 1);
                STAMP(5);
                const uint32_t par = g & 1u;
                const uint32_t sfw_both = cs.sfwh[par][ln >> 5][ln & 31];   // scale words: lanes 0..24 left, 32..56 right
                const uint32_t alive = (uint32_t)__builtin_amdgcn_readfirstlane((int)(cs.alive[par][0] | cs.alive[par][1]));
                uint32_t xd[2][8];   // xd[ch][k] = positions 128 k + 2 lane (low half) and + 1
#if (FLO_SKIP & 64) == 0
                if (par) quantise_nat<512>(cf, ts_a, athn, alive, blk, xd);
                else quantise_nat<0>(cf, ts_a, athn, alive, blk, xd);
#else
#pragma unroll
                for (int q = 0; q < 8; q++) { xd[0][q] = __float_as_uint(cf[q].x) & alive & 0x00010001u; xd[1][q] = __float_as_uint(cf[q].y) & blk[q] & 0x00010001u; }
#endif
                have = false;   // cf is free again
                take_next();
                STAMP(1);
                if (DBG && A.dbg_q) {
#pragma unroll
                    for (int ch = 0; ch < 2; ch++) {
                        uint32_t *dq = reinterpret_cast<uint32_t *>(A.dbg_q + ((frame0 + h) * 2 + ch) * 1024);
#pragma unroll
                        for (int k = 0; k < 8; k++) dq[64 * k + ln] = xd[ch][k];
                    }
                }
                // the < 16 bytes the previous frame left unflushed go back to the front of the staging buffer
                if (ln < (int)pend) stage[ln] = (uint8_t)tailb;
                uint8_t *f = stage + pend;
                const uint32_t f_a = (uint32_t)(uintptr_t)f;
                if ((ln & 31) < 25) {
                    uint8_t *p = f + 12 + 50 * (ln >> 5) + 2 * (ln & 31);
                    lds_st8<0>(p, sfw_both);
                    lds_st8<1>(p, sfw_both >> 8);
                }
                uint32_t tot[2];
                uint32_t pos = 112;   // 12 + 50 * 2: length word of channel 0
                // each channel's blob: the item form (up to 128 non-zeros), else the block form, else - a dense frame with many
                // runs or a run longer than 255, q >= 0.99 in practice - the general form; same bytes (tests compare them)
#pragma unroll
                for (int ch = 0; ch < 2; ch++) {
#if (FLO_SKIP & 32) == 0
                    uint32_t t = sparse_item_pack(ln, xd[ch], f_a + pos + 4u, tab_a);
#ifdef FLO_STAMPS
                    if (t == kSparseFallback) st_sum[6] += 1;   // channel-frames the item form declined
#endif
                    if (t == kSparseFallback) t = sparse_block_pack(ln, xd[ch], f_a + pos + 4u, tab_a);
#else
                    uint32_t t = 3u + ((xd[ch][0] | xd[ch][3]) & 1u);
                    FLO_KEEP(xd[ch][1]); FLO_KEEP(xd[ch][2]); FLO_KEEP(xd[ch][4]); FLO_KEEP(xd[ch][5]); FLO_KEEP(xd[ch][6]); FLO_KEEP(xd[ch][7]);
#endif
                    if (t == kSparseFallback) {   // uniform: dense frame (many runs, a run longer than 255)
                        // the general form wants the lane's 16 contiguous values: one trip through the (still unused) tail of
                        // the staging buffer re-deals the dwords
                        uint32_t *scr = reinterpret_cast<uint32_t *>(stage + 2560);
#pragma unroll
                        for (int k = 0; k < 8; k++) scr[64 * k + ln] = xd[ch][k];
                        wave_sync();
                        const uint4 x0 = reinterpret_cast<const uint4 *>(scr)[2 * ln], x1 = reinterpret_cast<const uint4 *>(scr)[2 * ln + 1];
                        wave_sync();
                        const uint32_t xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                        int q[1][16];
                        uint32_t hi[8];
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            hi[k] = xs[k] >> 16;
                            q[0][2 * k] = (int)xs[k];
                            q[0][2 * k + 1] = (int)hi[k];
                        }
                        SparsePlan P[1];
                        sparse_plan_m(ln, nonzero_mask16_packed(xs, hi), P[0]);
                        uint8_t *const dsts[1] = {f + pos + 4};
                        // trash bytes of the general form: the tail of the staging buffer, two per lane
                        const uint32_t trash[1] = {(uint32_t)((stage + kFrameCap + 64 + 2 * ln) - dsts[0])};
                        sparse_emit_n<1>(ln, q, P, dsts, trash);
                        t = P[0].total;
                    }
                    tot[ch] = t;
                    pos += 4u + t;
                    if (ch == 0) take_next();
                    STAMP(2 + ch);
                }
                const uint32_t flen = pos, blob_len = flen - 10;
                {
                    // the 12 header bytes [253][frame_samples = 1024 u32][0][blob_len u32][BlockSize::Long = 0][2 channels] and the
                    // two channel length words as ONE byte store: lane i < 12 holds header byte i, lanes 12..15 / 16..19 the bytes
                    // of the first / second length word
                    const uint32_t i = (uint32_t)ln;
                    const uint32_t w0 = 0x000400FDu, w1 = blob_len << 16, w2 = (blob_len >> 16) | 0x02000000u;
                    uint32_t wv4 = i < 4u ? w0 : (i < 8u ? w1 : w2);
                    wv4 = i < 12u ? wv4 : (i < 16u ? tot[0] : tot[1]);
                    const uint32_t off = i < 12u ? i : (i < 16u ? 112u - 12u + i : 116u - 16u + tot[0] + i);
                    const uint32_t bv = wv4 >> (8u * (i & 3u));
                    if (i < 20u) lds_st8<0>(f + off, bv);
                    if (ln == 0) A.frame_size[frame0 + h] = flen;
                }
                wave_sync();
                // the next frame's coefficients, if the transform wave already has them: taken in front of the flush, so
                // that it has its buffer back a flush earlier and the reads are answered while the stores go out
                take_next();
                const uint32_t haveb = pend + flen;
                const uint32_t n16 = haveb >> 4;
                const uint4 *src = reinterpret_cast<const uint4 *>(stage);
                uint4 *dst = reinterpret_cast<uint4 *>(gout + written);
                for (uint32_t i = ln; i < n16; i += 64) dst[i] = src[i];
                pend = haveb & 15u;
                tailb = (ln < (int)pend) ? stage[(n16 << 4) + ln] : 0u;
                written += (unsigned long long)n16 << 4;
                wave_sync();
                STAMP(4);
#ifdef FLO_STAMPS
                {
                    const unsigned long long busy = st_last - st_frame0;
                    if (busy > 12000) { st_sum[9] += busy; st_sum[10] += 1; }
                    if (busy > 20000) { st_sum[11] += busy; st_sum[12] += 1; }
                }
#endif
            }
            if (lane < (int)pend) gout[written + lane] = (uint8_t)tailb;
            if (lane == 0) A.clip_bytes[clip] = written + pend;
#ifdef FLO_STAMPS
            if (A.dbg_stamps && lane == 0) {
                st_sum[13] = (unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) |
                             ((unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) << 32) | ((unsigned long long)cl << 40);
                for (int i = 0; i < 14; i++) A.dbg_stamps[((unsigned long long)clip * 2 + 1) * 16 + i] = st_sum[i];
                A.dbg_stamps[((unsigned long long)clip * 2 + 1) * 16 + 14] = __builtin_amdgcn_s_memrealtime();   // when the clip's bytes were out
            }
#endif
            fbase += hops;
        }
    }

    // ---------------------------------------------------------------------- transform wave: both channels
#ifdef FLO_PRIO_T
    __builtin_amdgcn_s_setprio(FLO_PRIO_T);
#endif
    LossyDevTables T = A.T;
    T.pack = reinterpret_cast<const float4 *>(lds_raw);
#ifndef FLO_SO_LDS
    // the slot area belongs to band statistics alone in this form: its zero slot is written once and the lane's twelve
    // gather addresses stay in registers (the wave has them to spare now that the quantiser lives in the packer)
    uint32_t so_pre[12];
    band_stats_2_prepare(lane, cs.slot, A.T.pack, so_pre);
    constexpr bool kSoPre = true;
#else
    const uint32_t *so_pre = nullptr;
    constexpr bool kSoPre = false;
#endif
    for (;;) {
        wait_counter(&cs.clip_seq, ++seq);
        const unsigned clip = (unsigned)__builtin_amdgcn_readfirstlane((int)cs.clip_cur);
        if (clip >= (unsigned)A.n_clips) return;
        const unsigned hops = A.clip_hops[clip];
        const unsigned long long frame0 = A.clip_frame0[clip];
        const float *pcm = A.pcm + A.clip_off[clip];

        float prev = 0.f;   // temporal masking state: channel 0's band b on lane b, channel 1's on lane 32 + b
        v2f ae[8], ao[8], be[8], bo[8];
#pragma unroll
        for (int r = 0; r < 8; r++) ae[r] = ao[r] = splat2(0.f);  // pre-roll: 1024 zeros (encoder.rs:177)
        if (!COEFFS) load_half_fast_2(lane_id_opaque(), pcm, 0, be, bo);   // (opaque: keeps 16 address pairs out of loop-invariant registers)
#ifdef FLO_STAMPS
        unsigned long long st_sum[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
        // The frame loop is software-pipelined by two phases. (1) The fold of frame h + 1, which only needs PCM that is already
        // in registers, runs at the end of frame h: entering frame h, (zr, zi) hold its folded input, and the half-frame
        // after next is loaded at the TOP of the frame into the registers the last fold freed. (2) The masking pass of
        // frame h - one long dependent chain on 50 lanes (logarithm, DPP maxima, exponential, division), 6.6 % of the launch
        // when it runs by itself - is deferred into frame h + 1's FFT, behind the issue of the first exchange's reads: it
        // fills the LDS round trip the wave would otherwise sit out (a wave issues in order). The packer learns a frame's
        // band tables a third of a frame later and takes its coefficients correspondingly early (take_next).
        v2f zr[8], zi[8];
        if (!COEFFS) fold_2(lane_id_opaque(), ae, ao, be, bo, zr, zi, T);
        float pend_e = 0.f, pend_m = 0.f;   // band energies / maxima of the frame whose masking pass is pending
        // masking level, temporal masking, scale factors of frame gp (clip frame hp) from its band statistics; publishes them
        auto mask_publish = [&](const int ln, const uint32_t gp, const unsigned hp, const float energy1, const float bmax1,
                                const float rcount, const float4 sd0, const float4 sd1) __attribute__((always_inline)) {
            const int bnd = ln & 31, up = ln >> 5;
#if (FLO_SKIP & 16) == 0
            // (the scale factors first: their division and logarithm are a chain of their own, and in front of the masking
            // pass - which ends in a branch - they share its basic block and fill its dependent-issue gaps)
            const float bm = bmax1;
            const float sfv1 = bm > 1e-10f ? __fdiv_rn(30000.0f, bm) : 1.0f;   // encoder.rs:121-127
            const uint32_t sfw1 = sf_word(sfv1);
            const float a = spread_threshold_2r(ln, energy1, rcount, sd0, sd1, T);
            const float sl = max_raw(a, prev * 0.7f);   // temporal masking (psychoacoustic.rs:196-203)
            prev = sl;
            const float tl1 = masking_amplitude(sl, T.smr_thr);
#else   // diagnostic (results invalid): what the masking pass costs
            const float bm = bmax1, tl1 = energy1 * 1e-3f + rcount + sd0.x + sd1.y, sfv1 = bm * 100.f + 1.f;
            const uint32_t sfw1 = __float_as_uint(tl1) >> 16;
#endif
            // bands with anything above their masking amplitude: channel 0's on bits 0..24, channel 1's on bits 32..56
            const unsigned long long al = __ballot(bnd < 25 && bm > tl1);
            if (bnd < 25) {
                typedef __attribute__((address_space(3))) float lds_f32;
                const uint32_t ts_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)cs.ts[gp & 1u]);
                lds_f32 *tp = reinterpret_cast<lds_f32 *>((uintptr_t)(ts_s + 16u * (uint32_t)bnd + 4u * (uint32_t)up));
                tp[0] = tl1;
                tp[2] = sfv1;
                cs.sfwh[gp & 1u][up][bnd] = (uint16_t)sfw1;
            }
            if (ln == 0) {
                cs.alive[gp & 1u][0] = (uint32_t)al;
                cs.alive[gp & 1u][1] = (uint32_t)(al >> 32);
            }
            if (DBG && A.dbg_sfw && bnd < 25) A.dbg_sfw[((frame0 + hp) * 2 + up) * 25 + bnd] = (unsigned short)sfw1;
            set_counter(&cs.ts_ready, gp + 1);
        };
        auto frame_body = [&](const unsigned h, v2f (&ne)[8], v2f (&no)[8], v2f (&fe)[8], v2f (&fo)[8]) __attribute__((always_inline)) {
            const int ln = lane_id_opaque();
            const uint32_t g = fbase + h;
            FLO_MARK("frame_begin");
            // has the packer taken the previous frame's coefficients out of the buffer? (asked now, needed at the first exchange)
            const uint32_t consumed_early = peek_counter(&cs.consumed);
            // the masking pass's constants, fetched ahead of the exchange it runs behind (it must not wait for LDS there)
            const float4 sd0 = T.pack[kRowS10 * 64], sd1 = T.pack[kRowS10 * 64 + 1];
            const float rcount = T.pack[kRowLane * 64 + ln].z;   // (the row holds band (lane & 31)'s value on every lane)
            v2f c[16];
            if (COEFFS) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const float4 v0 = reinterpret_cast<const float4 *>(A.in_coeffs + ((frame0 + h) * 2 + 0) * 1024 + 16 * ln)[q];
                    const float4 v1 = reinterpret_cast<const float4 *>(A.in_coeffs + ((frame0 + h) * 2 + 1) * 1024 + 16 * ln)[q];
                    c[4 * q] = (v2f){v0.x, v1.x}; c[4 * q + 1] = (v2f){v0.y, v1.y};
                    c[4 * q + 2] = (v2f){v0.z, v1.z}; c[4 * q + 3] = (v2f){v0.w, v1.w};
                }
                if ((uint32_t)__builtin_amdgcn_readfirstlane((int)consumed_early) < g) wait_counter(&cs.consumed, g);
                float4 *p = reinterpret_cast<float4 *>(&cs.u.coef2[18 * ln]);   // where post_rotate_transpose_2 leaves them
#pragma unroll
                for (int q = 0; q < 8; q++) p[q] = make_float4(c[2 * q].x, c[2 * q].y, c[2 * q + 1].x, c[2 * q + 1].y);
                wave_sync();
            } else {
                // the half-frame after next, into the registers the last fold freed; consumed at the end of this frame.
                // Unconditional, also behind the last frame (the batch allocates one spare half-frame per clip)
                load_half_fast_2(ln, pcm, (long long)(h + 1) * 1024, ne, no);
                FLO_MARK("prefetch_done");
                STAMP(1);
                if ((uint32_t)__builtin_amdgcn_readfirstlane((int)consumed_early) < g) wait_counter(&cs.consumed, g);
                STAMP(7);
#if (FLO_SKIP & 8) == 0   // FLO_SKIP (diagnostic builds, results invalid): leave a phase out to see what it costs
                fft512_2(ln, zr, zi, cs.u.xch4, T, [&]() __attribute__((always_inline)) {
                    if (h > 0) mask_publish(ln, g - 1u, h - 1u, pend_e, pend_m, rcount, sd0, sd1);   // uniform
                });
#else
                if (h > 0) mask_publish(ln, g - 1u, h - 1u, pend_e, pend_m, rcount, sd0, sd1);
#endif
                FLO_MARK("fft_done");
                STAMP(2);
#if (FLO_SKIP & 4) == 0
                post_rotate_transpose_2(ln, zr, zi, cs.u.coef2, c, T);
#else
#pragma unroll
                for (int e = 0; e < 8; e++) { c[2 * e] = zr[e]; c[2 * e + 1] = zi[e]; }
#endif
                FLO_MARK("postrot_done");
                STAMP(3);
                if (DBG && A.dbg_coeffs) {
#pragma unroll
                    for (int ch = 0; ch < 2; ch++) {
                        float *d = A.dbg_coeffs + ((frame0 + h) * 2 + ch) * 1024 + 16 * ln;
#pragma unroll
                        for (int e = 0; e < 16; e++) d[e] = ch ? c[e].y : c[e].x;
                    }
                }
            }
            set_counter(&cs.coef_ready, g + 1);
            // band statistics (both channels): channel 0's band b on lane b, channel 1's on lane 32 + b
            float energy1, bmax1;
#if (FLO_SKIP & 1) == 0
            band_stats_2<DIRTY, kSoPre>(ln, c, cs.slot, T, energy1, bmax1, so_pre);
#else
            energy1 = c[0].x + c[5].y;
            bmax1 = c[1].x + c[7].y;
#endif
            FLO_MARK("bandstats_done");
            STAMP(4);
            if (COEFFS) {
                mask_publish(ln, g, h, energy1, bmax1, rcount, sd0, sd1);
            } else {
                pend_e = energy1, pend_m = bmax1;
                // the next frame's fold
#if (FLO_SKIP & 128) == 0
                fold_2(ln, fe, fo, ne, no, zr, zi, T);
#else
#pragma unroll
                for (int r = 0; r < 8; r++) { zr[r] = fe[r] + ne[r]; zi[r] = fo[r] - no[r]; }
#endif
            }
            FLO_MARK("frame_end");
            STAMP(5);
        };
        for (unsigned h = 0; h < hops; h += 2) {
            frame_body(h, ae, ao, be, bo);
            if (h + 1 < hops) frame_body(h + 1, be, bo, ae, ao);
        }
        if (!COEFFS && hops) {   // the last frame's masking pass has no FFT to hide behind
            const int ln = lane_id_opaque();
            const float4 sd0 = T.pack[kRowS10 * 64], sd1 = T.pack[kRowS10 * 64 + 1];
            mask_publish(ln, fbase + hops - 1u, hops - 1u, pend_e, pend_m, T.pack[kRowLane * 64 + ln].z, sd0, sd1);
        }
#ifdef FLO_STAMPS
        if (A.dbg_stamps && lane == 0) {   // [13]: where the wave ran (HW_ID, XCC_ID, clip slot): diag/stamps_clips.py groups the records by it
            st_sum[13] = (unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)) |
                         ((unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) << 32) | ((unsigned long long)cl << 40);
            for (int i = 0; i < 14; i++) A.dbg_stamps[((unsigned long long)clip * 2) * 16 + i] = st_sum[i];
            A.dbg_stamps[((unsigned long long)clip * 2) * 16 + 14] = __builtin_amdgcn_s_memrealtime();   // when the clip's last frame left the transform wave (100 MHz)
        }
#endif
        fbase += hops;
    }
}

// ---------------------------------------------------------------------------------------------- frame-parallel
// PASS 1: a_t only. PASS 2: full frame into slot gframe. One wave per frame, CH = all channels in lock-step.
template <int CH, int PASS, bool EXACT>
__global__ __launch_bounds__(64) void lossy_frame_kernel(LossyArgs A) {
    __shared__ WaveLds<CH> lds;
    __shared__ __attribute__((aligned(16))) uint8_t stage[kFrameCap + 64 + 256];
    const int lane = lane_id();
    const unsigned long long gframe = blockIdx.x;
    if (gframe >= A.total_frames) return;
    // locate clip by binary search over clip_frame0
    int lo = 0, hi = A.n_clips - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (A.clip_frame0[mid] <= gframe) lo = mid; else hi = mid - 1;
    }
    const unsigned clip = (unsigned)lo;
    const unsigned h = (unsigned)(gframe - A.clip_frame0[clip]);
    const float *pcm = A.pcm + A.clip_off[clip];
    const long long n_sf = (long long)A.clip_nsf[clip];

    LaneConst L;
    load_lane_const(lane, L, A.T);
    if (lane < CH) lds.slots[lane][kZeroSlot] = make_float2(0.f, 0.f);
    float c[CH][16];
    if (A.in_coeffs) {
        load_coeffs<CH>(lane, c, A, gframe, 0);
    } else {
        float ae[CH][8], ao[CH][8], be[CH][8], bo[CH][8];
        load_half<CH>(lane, pcm, n_sf, A.nch, 0, (long long)h * 1024 - 1024, ae, ao);
        load_half<CH>(lane, pcm, n_sf, A.nch, 0, (long long)h * 1024, be, bo);
        mdct_frame<CH>(lane, ae, ao, be, bo, lds, A.T, c);
    }
    FrameState<CH> st;
    int q[CH][16];
    uint32_t sfw[CH];
    SparsePlan P[CH];
    if (PASS == 1) {
#pragma unroll
        for (int ch = 0; ch < CH; ch++) st.prev[ch] = 0.f;
        analyse_frame<CH, true, EXACT>(lane, c, lds, L, A, A.T, 0, st, gframe, q, sfw, P);
        return;
    }
    store_coeffs_dbg<CH>(lane, c, A, gframe, 0);
#pragma unroll
    for (int ch = 0; ch < CH; ch++) st.prev[ch] = lane < 25 ? A.s_prev[(gframe * A.nch + ch) * 32 + lane] : 0.f;
    analyse_frame<CH, false, EXACT>(lane, c, lds, L, A, A.T, 0, st, gframe, q, sfw, P);
    uint32_t tot[2];
    tot[0] = P[0].total;
    tot[1] = P[CH - 1].total;
    const uint32_t flen = emit_frame<CH>(lane, stage, CH, 0, tot, sfw, P, q);
    wave_sync();
    if (lane == 0) A.frame_size[gframe] = flen;
    const uint32_t n16 = (flen + 15) >> 4;
    const uint4 *src = reinterpret_cast<const uint4 *>(stage);
    uint4 *dst = reinterpret_cast<uint4 *>(A.slots + gframe * (unsigned long long)A.slot_bytes);
    for (uint32_t i = lane; i < n16; i += 64) dst[i] = src[i];
}

// Stereo frames, shipped quantiser, PCM input: the frame-parallel passes built from the lock-step stereo device
// functions of lossy_chain2q_kernel (packed f32 transform of both channels, both channels' masking in one pass, ballot
// packer) - one wave does a frame's transform AND packing here. Same bytes as every other form (tests compare them).
constexpr int kScanBlock = 64;   // frames of history the temporal chain is warmed up over (lossy_scan_kernel, and pass 2 below)
// FROMCOEF (pass 2 only): the coefficients come from pass 1's hand-over buffer (A.coef_t). That pass needs no FFT exchange
// buffer: 9.7 KB of LDS instead of 14.4 and 128 registers - sixteen frames per CU instead of eleven, two rounds of
// workgroups for a 3-minute clip instead of three.
struct QuantOnlyLds {
    float4 ts[32];
    uint32_t qh[2][512];
};
template <int PASS, bool FROMCOEF = false>
__global__ __launch_bounds__(64) void lossy_frame2x_kernel(LossyArgs A) {
    constexpr bool kSmall = PASS == 2 && FROMCOEF;
    __shared__ typename std::conditional<kSmall, QuantOnlyLds, StereoLds>::type lds;
    __shared__ __attribute__((aligned(16))) uint8_t stage[PASS == 2 ? kFrameCap + 64 + 256 : 16];
    // the re-dealing buffer of the integers lies over the analysis buffers, which are dead once the quantiser has read the
    // band table (a wave's LDS instructions execute in order): 14.4 KB per frame instead of 18.5, eleven frames per CU
    static_assert(sizeof(StereoLds) >= 2 * 512 * 4, "the integers of both channels fit the analysis buffers");
    uint32_t (*qh)[512];
    float4 *ts_tab;
    if constexpr (kSmall) {
        qh = lds.qh;
        ts_tab = lds.ts;
    } else {
        qh = reinterpret_cast<uint32_t (*)[512]>(&lds);
        ts_tab = lds.u.a.ts;
    }
    __shared__ uint32_t runtab[PASS == 2 ? kRunTabEntries : 1];
    const int lane = lane_id();
    const unsigned long long gframe = blockIdx.x;
    if (gframe >= A.total_frames) return;
    int lo = 0, hi = A.n_clips - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (A.clip_frame0[mid] <= gframe) lo = mid; else hi = mid - 1;
    }
    const unsigned clip = (unsigned)lo;
    const unsigned h = (unsigned)(gframe - A.clip_frame0[clip]);
    const float *pcm = A.pcm + A.clip_off[clip];
    const LossyDevTables &T = A.T;   // constant rows straight from global memory: one frame per wave reads each once

    v2f c[16];
    if constexpr (kSmall) {   // pass 1 left this frame's coefficients as its lanes held them
        const float4 *src = A.coef_t + gframe * 512ull + (unsigned)lane;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float4 v = src[64 * k];
            c[2 * k] = (v2f){v.x, v.y};
            c[2 * k + 1] = (v2f){v.z, v.w};
        }
    } else {
        v2f ae[8], ao[8], be[8], bo[8];
        if (h == 0) {   // pre-roll: 1024 zeros (encoder.rs:177)
#pragma unroll
            for (int r = 0; r < 8; r++) ae[r] = ao[r] = splat2(0.f);
        } else {
            load_half_fast_2(lane, pcm, (long long)h * 1024 - 1024, ae, ao);
        }
        load_half_fast_2(lane, pcm, (long long)h * 1024, be, bo);   // the batch pads every clip: no bounds to check
        v2f zr[8], zi[8];
        fold_2(lane, ae, ao, be, bo, zr, zi, T);
        fft512_2(lane, zr, zi, lds.u.xch4, T);
        post_rotate_transpose_2(lane, zr, zi, lds.u.coef2, c, T);
        if (PASS == 1 && A.coef_t) {
            float4 *dst = A.coef_t + gframe * 512ull + (unsigned)lane;
#pragma unroll
            for (int k = 0; k < 8; k++) dst[64 * k] = make_float4(c[2 * k].x, c[2 * k].y, c[2 * k + 1].x, c[2 * k + 1].y);
        }
    }
    if (PASS == 2 && A.dbg_coeffs) {
#pragma unroll
        for (int ch = 0; ch < 2; ch++) {
            float *d = A.dbg_coeffs + (gframe * 2 + ch) * 1024 + 16 * lane;
#pragma unroll
            for (int e = 0; e < 16; e++) d[e] = ch ? c[e].y : c[e].x;
        }
    }
    // channel 0's band b on lane b, channel 1's on lane 32 + b. Pass 1 makes the band statistics and the masking level
    // before temporal masking and leaves both for pass 2, which only transforms again (the coefficients are what it cannot
    // keep): statistics and spreading are a sixth of a frame's instructions.
    const int bnd = lane & 31, up = lane >> 5;
    float a, bmax1;
    if constexpr (PASS == 1) {
        float energy1;
        band_stats_2(lane, c, lds.u.a.slot, T, energy1, bmax1);
        const float rcount = T.pack[kRowLane * 64 + bnd].z;
        a = spread_threshold_2(lane, energy1, rcount, T);
        if (bnd < 25) {
            A.a_t[(gframe * 2 + up) * 32 + bnd] = a;
            A.bmax_t[(gframe * 2 + up) * 32 + bnd] = bmax1;
        }
        return;
    }
    a = bnd < 25 ? A.a_t[(gframe * 2 + up) * 32 + bnd] : 0.f;
    bmax1 = bnd < 25 ? A.bmax_t[(gframe * 2 + up) * 32 + bnd] : 0.f;
    float prev = 0.f;
    if constexpr (kSmall) {
        // The temporal chain s_t = max(a_t, 0.7 s_(t-1)) over the 64 frames before this one, from 0: what lossy_scan_kernel
        // computes for the first frame of each of its blocks (history older than 64 frames cannot reach a threshold, see
        // there), here for every frame by the workgroup that needs it - 64 independent 4-byte loads per lane and a chain
        // of 128 instructions instead of a launch of its own (10.6 us for a 3-minute clip) between the passes.
        const float *at = A.a_t + ((gframe - h) * 2 + (unsigned)up) * 32 + (unsigned)(bnd < 25 ? bnd : 0);
        float s = 0.f;
        if (h >= (unsigned)kScanBlock) {   // (uniform)
            float av[kScanBlock];
#pragma unroll
            for (int j = 0; j < kScanBlock; j++) av[j] = at[(unsigned long long)(h - kScanBlock + j) * 64];
#pragma unroll
            for (int j = 0; j < kScanBlock; j++) s = fmaxf(av[j], s * 0.7f);
        } else {
            for (unsigned hb = 0; hb < h; hb += 16) {
                float av[16];
#pragma unroll
                for (int j = 0; j < 16; j++) av[j] = hb + j < h ? at[(unsigned long long)(hb + j) * 64] : 0.f;
#pragma unroll
                for (int j = 0; j < 16; j++)
                    if (hb + j < h) s = fmaxf(av[j], s * 0.7f);
            }
        }
        prev = s;
    } else {
        prev = bnd < 25 ? A.s_prev[(gframe * 2 + up) * 32 + bnd] : 0.f;
    }
    const float sl = max_raw(a, prev * 0.7f);   // temporal masking (psychoacoustic.rs:196-203)
    const float tl1 = masking_amplitude(sl, T.smr_thr);
    const float bm = bmax1;
    const float sfv1 = bm > 1e-10f ? __fdiv_rn(30000.0f, bm) : 1.0f;   // encoder.rs:121-127
    const uint32_t sfw1 = sf_word(sfv1);
    if (bnd < 25) {
        reinterpret_cast<float *>(&ts_tab[bnd])[up] = tl1;
        reinterpret_cast<float *>(&ts_tab[bnd])[2 + up] = sfv1;
    }
    wave_sync();
    uint32_t xs[2][8];
    {
        QuantRows qrows;
        quant_rows_load(lane, T, qrows);
        quantise_2(lane, c, ts_tab, T, qrows, xs);
    }
    if (A.dbg_q) {
#pragma unroll
        for (int ch = 0; ch < 2; ch++) {
            uint32_t *dq = reinterpret_cast<uint32_t *>(A.dbg_q + (gframe * 2 + ch) * 1024 + 16 * lane);
#pragma unroll
            for (int k = 0; k < 8; k++) dq[k] = xs[ch][k];
        }
    }
    if (A.dbg_sfw && bnd < 25) A.dbg_sfw[(gframe * 2 + up) * 25 + bnd] = (unsigned short)sfw1;

    // ---- packing: the block form wants the dword with positions 128 k + 2 lane, + 1 in register k: one trip through LDS
    // re-deals the integers
#pragma unroll
    for (int ch = 0; ch < 2; ch++) {
        uint4 *dq = reinterpret_cast<uint4 *>(qh[ch]);
        dq[2 * lane] = make_uint4(xs[ch][0], xs[ch][1], xs[ch][2], xs[ch][3]);
        dq[2 * lane + 1] = make_uint4(xs[ch][4], xs[ch][5], xs[ch][6], xs[ch][7]);
    }
    wave_sync();
    uint32_t xd[2][8];
#pragma unroll
    for (int ch = 0; ch < 2; ch++) {
#pragma unroll
        for (int k = 0; k < 8; k++) xd[ch][k] = qh[ch][64 * k + lane];
    }
    uint8_t *f = stage;
    const uint32_t f_a = (uint32_t)(uintptr_t)f, tab_a = (uint32_t)(uintptr_t)runtab;
    if (bnd < 25) {
        uint8_t *p = f + 12 + 50 * up + 2 * bnd;
        lds_st8<0>(p, sfw1);
        lds_st8<1>(p, sfw1 >> 8);
    }
    uint32_t pos = 112;   // 12 + 50 * 2: length word of channel 0
#pragma unroll
    for (int ch = 0; ch < 2; ch++) {
        uint32_t t = sparse_block_pack(lane, xd[ch], f_a + pos + 4u, tab_a);
        if (t == kSparseFallback) {   // uniform: dense frame (a run longer than 255, or more than 126 runs)
            int q[1][16];
            uint32_t hi16[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                hi16[k] = xs[ch][k] >> 16;
                q[0][2 * k] = (int)xs[ch][k];
                q[0][2 * k + 1] = (int)hi16[k];
            }
            SparsePlan P[1];
            sparse_plan_m(lane, nonzero_mask16_packed(xs[ch], hi16), P[0]);
            uint8_t *const dsts[1] = {f + pos + 4};
            const uint32_t trash[1] = {(uint32_t)((stage + kFrameCap + 64 + 2 * lane) - dsts[0])};
            sparse_emit_n<1>(lane, q, P, dsts, trash);
            t = P[0].total;
        }
        if (lane == 32 + ch) {
            uint8_t *p = f + pos;
            p[0] = (uint8_t)t; p[1] = (uint8_t)(t >> 8); p[2] = (uint8_t)(t >> 16); p[3] = (uint8_t)(t >> 24);
        }
        pos += 4u + t;
    }
    const uint32_t flen = pos, blob_len = flen - 10;
    if (lane == 0) {
        f[0] = 253;
        f[1] = 0x00; f[2] = 0x04; f[3] = 0; f[4] = 0;  // frame_samples = 1024
        f[5] = 0;
        f[6] = (uint8_t)blob_len; f[7] = (uint8_t)(blob_len >> 8); f[8] = (uint8_t)(blob_len >> 16); f[9] = (uint8_t)(blob_len >> 24);
        f[10] = 0;  // BlockSize::Long
        f[11] = 2;
        A.frame_size[gframe] = flen;
    }
    wave_sync();
    const uint32_t n16 = (flen + 15) >> 4;
    const uint4 *src = reinterpret_cast<const uint4 *>(stage);
    uint4 *dst = reinterpret_cast<uint4 *>(A.slots + gframe * (unsigned long long)A.slot_bytes);
    for (uint32_t i = lane; i < n16; i += 64) dst[i] = src[i];
}

// Any channel count up to kMaxLossyChannels: one wave per frame walks the channels one after the other with the
// single-channel device functions. Pass 2 parks every channel's integers (i16) in LDS because the byte position of a
// channel's sparse blob depends on the sizes of the channels before it, then plans again and emits in channel order.
template <int PASS, bool EXACT>
__global__ __launch_bounds__(64) void lossy_frame_n_kernel(LossyArgs A) {
    extern __shared__ __attribute__((aligned(16))) uint8_t dyn[];   // stage[slot_bytes + 256] | park[nch][1024] i16
    __shared__ WaveLds<1> lds;
    __shared__ uint32_t s_tot[kMaxLossyChannels], s_sfw[kMaxLossyChannels][32];
    const int lane = lane_id();
    const unsigned long long gframe = blockIdx.x;
    if (gframe >= A.total_frames) return;
    int lo = 0, hi = A.n_clips - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (A.clip_frame0[mid] <= gframe) lo = mid; else hi = mid - 1;
    }
    const unsigned clip = (unsigned)lo;
    const unsigned h = (unsigned)(gframe - A.clip_frame0[clip]);
    const float *pcm = A.pcm + A.clip_off[clip];
    const long long n_sf = (long long)A.clip_nsf[clip];
    const int nch = A.nch;
    uint8_t *stage = dyn;
    short *park = reinterpret_cast<short *>(dyn + A.slot_bytes + 256);

    LaneConst L;
    load_lane_const(lane, L, A.T);
    if (lane == 0) lds.slots[0][kZeroSlot] = make_float2(0.f, 0.f);
    for (int ch = 0; ch < nch; ch++) {
        float c[1][16];
        if (A.in_coeffs) {
            load_coeffs<1>(lane, c, A, gframe, ch);
        } else {
            float ae[1][8], ao[1][8], be[1][8], bo[1][8];
            load_half<1>(lane, pcm, n_sf, nch, ch, (long long)h * 1024 - 1024, ae, ao);
            load_half<1>(lane, pcm, n_sf, nch, ch, (long long)h * 1024, be, bo);
            mdct_frame<1>(lane, ae, ao, be, bo, lds, A.T, c);
        }
        FrameState<1> st;
        int q[1][16];
        uint32_t sfw[1];
        SparsePlan P[1];
        if (PASS == 1) {
            st.prev[0] = 0.f;
            analyse_frame<1, true, EXACT>(lane, c, lds, L, A, A.T, ch, st, gframe, q, sfw, P);
            continue;
        }
        store_coeffs_dbg<1>(lane, c, A, gframe, ch);
        st.prev[0] = lane < 25 ? A.s_prev[(gframe * nch + ch) * 32 + lane] : 0.f;
        analyse_frame<1, false, EXACT>(lane, c, lds, L, A, A.T, ch, st, gframe, q, sfw, P);
#pragma unroll
        for (int e = 0; e < 16; e++) park[ch * 1024 + 16 * lane + e] = (short)q[0][e];
        if (lane < 32) s_sfw[ch][lane] = sfw[0];
        if (lane == 0) s_tot[ch] = P[0].total;
        wave_sync();
    }
    if (PASS == 1) return;
    uint32_t tot[kMaxLossyChannels];
    for (int ch = 0; ch < kMaxLossyChannels; ch++) tot[ch] = ch < nch ? s_tot[ch] : 0u;
    uint32_t flen = 0;
    for (int ch = 0; ch < nch; ch++) {
        int q[1][16];
#pragma unroll
        for (int e = 0; e < 16; e++) q[0][e] = park[ch * 1024 + 16 * lane + e];
        uint32_t sfw[1] = {s_sfw[ch][lane & 31]};
        SparsePlan P[1];
        sparse_plan(lane, q[0], P[0]);
        flen = emit_frame<1>(lane, stage, nch, ch, tot, sfw, P, q);
        wave_sync();
    }
    if (lane == 0) A.frame_size[gframe] = flen;
    const uint32_t n16 = (flen + 15) >> 4;
    const uint4 *src = reinterpret_cast<const uint4 *>(stage);
    uint4 *dst = reinterpret_cast<uint4 *>(A.slots + gframe * (unsigned long long)A.slot_bytes);
    for (uint32_t i = lane; i < n16; i += 64) dst[i] = src[i];
}

// temporal recurrence s_t = max(a_t, 0.7 s_{t-1}), s_{-1} = 0 (psychoacoustic.rs:198-203) for the frame-parallel
// form: blocks of 64 frames, one thread per (clip, block, channel, band), each warmed up over the 64 frames before its
// block. History older than 64 frames can only contribute 0.7^64 (1e-10) of its level; levels below 4.77e-7 (half an
// ulp of 10) vanish in fl(s - 10) and in max(s, ath) - 10, so every threshold derived from the blocked scan is
// bit-identical to the sequential chain's (the chain kernel keeps the true sequential state).
// Writes the state seen BEFORE each frame.
__global__ void lossy_scan_kernel(LossyArgs A) {
    const unsigned clip = blockIdx.x;   // clips in x: gridDim.y stops at 65535
    if (clip >= (unsigned)A.n_clips) return;
    const unsigned hops = A.clip_hops[clip];
    const unsigned blk = blockIdx.y;
    const unsigned fs = blk * kScanBlock;
    if (fs >= hops) return;
    const unsigned ch = threadIdx.x >> 5, band = threadIdx.x & 31u;
    if (ch >= (unsigned)A.nch || band >= 25) return;
    const unsigned long long f0 = A.clip_frame0[clip];
    const unsigned fw = fs >= kScanBlock ? fs - kScanBlock : 0;
    const unsigned fe = fs + kScanBlock < hops ? fs + kScanBlock : hops;
    // The recurrence is a serial chain, the loads are not: fetch sixteen levels at a time, then run the chain on
    // registers (one thread would otherwise pay a full memory latency per frame, 128 times in a row).
    float s = 0.f;
    const unsigned long long stride = (unsigned long long)A.nch * 32;
    const float *at = A.a_t + (f0 * A.nch + ch) * 32 + band;
    float *sp = A.s_prev_out + (f0 * A.nch + ch) * 32 + band;
    for (unsigned h0 = fw; h0 < fe; h0 += 16) {
        float a[16];
#pragma unroll
        for (int j = 0; j < 16; j++) a[j] = h0 + j < fe ? at[(unsigned long long)(h0 + j) * stride] : 0.f;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const unsigned h = h0 + j;
            if (h < fe) {
                if (h >= fs) sp[(unsigned long long)h * stride] = s;
                s = fmaxf(a[j], s * 0.7f);
            }
        }
    }
}

// pack the fixed-size slots of one clip into its DATA chunk: one workgroup per frame
__global__ void lossy_compact_kernel(LossyArgs A) {
    const unsigned long long gframe = blockIdx.x;
    if (gframe >= A.total_frames) return;
    int lo = 0, hi = A.n_clips - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (A.clip_frame0[mid] <= gframe) lo = mid; else hi = mid - 1;
    }
    const unsigned clip = (unsigned)lo;
    const unsigned long long off = A.frame_off[gframe];  // byte offset inside the clip's DATA chunk
    const uint32_t len = A.frame_size[gframe];
    const uint8_t *src = A.slots + gframe * (unsigned long long)A.slot_bytes;
    uint8_t *dst = A.out + A.out_off[clip] + off;
    for (uint32_t i = threadIdx.x; i < len; i += blockDim.x) dst[i] = src[i];
}

// exclusive scan of frame sizes inside each clip: one workgroup per clip. Thread t owns a contiguous run of frames; the run
// sums are scanned by wave shuffles and the wave totals by the first wave (two barriers; the Hillis-Steele form it
// replaces needed twenty at 1024 threads and took 13.6 us for a 3-minute clip).
template <int THREADS>   // 256, or 1024 for a few long clips
__global__ __launch_bounds__(THREADS) void lossy_frame_offsets_kernel(LossyArgs A) {
    __shared__ unsigned long long wtot[THREADS / 64];
    const unsigned clip = blockIdx.x;
    if (clip >= (unsigned)A.n_clips) return;
    const unsigned long long f0 = A.clip_frame0[clip];
    const unsigned hops = A.clip_hops[clip];
    const unsigned t = threadIdx.x, lane = t & 63u, w = t >> 6;
    const unsigned per = (hops + THREADS - 1) / THREADS;
    const unsigned h0 = t * per < hops ? t * per : hops, h1 = h0 + per < hops ? h0 + per : hops;
    // the run's sizes, eight independent loads at a time (a loop of dependent single loads paid one memory latency per
    // frame: 12 us of this kernel's 13 for a 3-minute clip); short runs (<= 8 frames) stay in registers for the second pass
    unsigned long long sum = 0;
    uint32_t keep[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (unsigned hb = h0; hb < h1; hb += 8) {
        uint32_t v8[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v8[j] = hb + j < h1 ? A.frame_size[f0 + hb + j] : 0u;
#pragma unroll
        for (int j = 0; j < 8; j++) sum += v8[j];
        if (hb == h0) {
#pragma unroll
            for (int j = 0; j < 8; j++) keep[j] = v8[j];
        }
    }
    unsigned long long v = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long u = __shfl_up(v, d);
        if (lane >= (unsigned)d) v += u;
    }
    if (lane == 63) wtot[w] = v;
    __syncthreads();
    if (w == 0) {
        unsigned long long x = lane < THREADS / 64 ? wtot[lane] : 0;
#pragma unroll
        for (int d = 1; d < THREADS / 64; d <<= 1) {
            const unsigned long long u = __shfl_up(x, d);
            if (lane >= (unsigned)d) x += u;
        }
        if (lane < THREADS / 64) wtot[lane] = x;
    }
    __syncthreads();
    unsigned long long off = v - sum + (w ? wtot[w - 1] : 0);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (h0 + j < h1) A.frame_off[f0 + h0 + j] = off;
        off += keep[j];
    }
    for (unsigned h = h0 + 8; h < h1; h++) {
        A.frame_off[f0 + h] = off;
        off += A.frame_size[f0 + h];
    }
    if (t == THREADS - 1) A.clip_bytes[clip] = wtot[THREADS / 64 - 1];
}

// Few clips (configs[1]: one of three minutes): frame offsets and packing in ONE launch. A workgroup owns a chunk of
// kCompactChunk consecutive frames of a clip; it adds up the sizes of the clip's frames in front of its chunk itself
// (at most a few thousand 4-byte loads, spread over its threads: 31 KB out of the L2 for the last chunk of a 3-minute
// clip), scans its own frames' sizes and copies their slots to their places. One kernel instead of a one-workgroup scan
// followed by a copy kernel that waited for it: of the two launches' 17.4 us for a 3-minute clip each had been mostly launch,
// ramp and drain.
constexpr int kCompactChunk = 32;
__global__ __launch_bounds__(256) void lossy_offsets_compact_kernel(LossyArgs A) {
    __shared__ unsigned long long wsum[4];
    __shared__ unsigned long long foff[kCompactChunk + 1];
    const unsigned clip = blockIdx.y;
    const unsigned hops = A.clip_hops[clip];
    const unsigned h0 = blockIdx.x * (unsigned)kCompactChunk;
    if (h0 >= hops && !(hops == 0 && blockIdx.x == 0)) return;   // (uniform)
    const unsigned long long f0 = A.clip_frame0[clip];
    const unsigned t = threadIdx.x, lane = t & 63u, w = t >> 6;
    // bytes in front of the chunk
    unsigned long long sum = 0;
    for (unsigned hb = t; hb < h0; hb += 256u * 8u) {
        uint32_t v8[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v8[j] = hb + 256u * (unsigned)j < h0 ? A.frame_size[f0 + hb + 256u * (unsigned)j] : 0u;
#pragma unroll
        for (int j = 0; j < 8; j++) sum += v8[j];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
    if (lane == 0) wsum[w] = sum;
    // the chunk's own sizes: an exclusive scan by the first wave
    const unsigned nf = hops - h0 < (unsigned)kCompactChunk ? hops - h0 : (unsigned)kCompactChunk;
    __syncthreads();
    if (w == 0) {
        const unsigned long long before = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        const uint32_t mine = lane < nf ? A.frame_size[f0 + h0 + lane] : 0u;
        unsigned long long v = mine;
#pragma unroll
        for (int d = 1; d < kCompactChunk; d <<= 1) {
            const unsigned long long u = __shfl_up(v, d);
            if (lane >= (unsigned)d) v += u;
        }
        if (lane < nf) {
            foff[lane] = before + v - mine;
            A.frame_off[f0 + h0 + lane] = before + v - mine;
        }
        if (lane == (nf ? nf - 1 : 0)) {
            foff[nf] = before + (nf ? v : 0ull);
            if (h0 + nf >= hops) A.clip_bytes[clip] = before + (nf ? v : 0ull);   // the clip's last chunk knows the total
        }
    }
    __syncthreads();
    // copy: a frame is a run of bytes at any alignment in the clip's DATA chunk; its slot is 16-byte aligned. Four bytes per
    // thread and step (global memory takes unaligned dwords), the last one to three byte by byte.
    // Eight threads per frame, all frames of the chunk at once, four independent dwords per thread and round (one frame
    // after the other, a round trip per frame, the chunk took 32 memory latencies).
    uint8_t *const out = A.out + A.out_off[clip];
    const unsigned j = t >> 3, sub = t & 7u;
    if (j < nf) {
        const uint8_t *src = A.slots + (f0 + h0 + j) * (unsigned long long)A.slot_bytes;
        uint8_t *dst = out + foff[j];
        const uint32_t len = (uint32_t)(foff[j + 1] - foff[j]);
        const uint32_t n4 = len >> 2;
        for (uint32_t i = sub; i < n4; i += 32u) {
            uint32_t v[4];
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (i + 8u * (uint32_t)u < n4) __builtin_memcpy(&v[u], src + 4u * (i + 8u * (uint32_t)u), 4);
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (i + 8u * (uint32_t)u < n4) __builtin_memcpy(dst + 4u * (i + 8u * (uint32_t)u), &v[u], 4);
        }
        if (sub < (len & 3u)) dst[4u * n4 + sub] = src[4u * n4 + sub];
    }
}

// ---------------------------------------------------------------------------------------------- stage kernels
// forward MDCT of independent 2048-sample mono windows (flo_mdct_forward)
__global__ __launch_bounds__(64) void mdct_only_kernel(LossyDevTables T, const float *frames, unsigned long long n,
                                                       float *out) {
    __shared__ WaveLds<1> lds;
    const unsigned long long w = blockIdx.x;
    if (w >= n) return;
    const int lane = lane_id();
    float ae[1][8], ao[1][8], be[1][8], bo[1][8];
    const float *p = frames + w * 2048;
    load_half<1>(lane, p, 2048, 1, 0, 0, ae, ao);
    load_half<1>(lane, p, 2048, 1, 0, 1024, be, bo);
    float c[1][16];
    mdct_frame<1>(lane, ae, ao, be, bo, lds, T, c);
    float4 *d = reinterpret_cast<float4 *>(out + w * 1024 + 16 * lane);
#pragma unroll
    for (int q = 0; q < 4; q++) d[q] = make_float4(c[0][4 * q], c[0][4 * q + 1], c[0][4 * q + 2], c[0][4 * q + 3]);
}

// TransformEncoder::quantize_coefficients (encoder.rs:109-154) as the reference exposes it: the caller brings the per-coefficient
// signal-to-mask ratios, the kernel makes the band scale factors (30000 / band maximum, IEEE division) and quantises every
// coefficient whose ratio exceeds the quality's threshold. One wave per vector of 1024 coefficients; band maxima through
// LDS atomics on the bit patterns (non-negative floats order like their bits). smr == nullptr: scale factors only.
__global__ __launch_bounds__(64) void quantise_smr_kernel(LossyDevTables T, const float *coeffs, const float *smr, unsigned long long n,
                                                          short *q, float *sf_out) {
    __shared__ uint32_t bmax[32];
    const unsigned long long w = blockIdx.x;
    if (w >= n) return;
    const int lane = lane_id();
    if (lane < 32) bmax[lane] = 0u;
    __syncthreads();
    const float *c = coeffs + w * 1024;
    for (int k = lane; k < 1024; k += 64) {
        const float a = fabsf(c[k]);
        if (a == a) atomicMax(&bmax[T.band[k]], __float_as_uint(a));   // f32::max ignores a NaN operand
    }
    __syncthreads();
    float sf = 1.0f;
    if (lane < 25) {
        const float m = __uint_as_float(bmax[lane]);
        if (m > 1e-10f) sf = __fdiv_rn(30000.0f, m);
        sf_out[w * 25 + lane] = sf;
    }
    if (!smr) return;
    __shared__ float sfs[32];
    if (lane < 25) sfs[lane] = sf;
    __syncthreads();
    for (int k = lane; k < 1024; k += 64) {
        short v = 0;
        if (smr[w * 1024 + k] > T.smr_thr) {
            const float r = round_away(c[k] * sfs[T.band[k]]);
            // .clamp(I16_MIN_F32, I16_MAX_F32) as i16: f32::clamp passes a NaN through and `as i16` maps it to 0
            v = (r == r) ? (short)fminf(fmaxf(r, -32768.0f), 32767.0f) : (short)0;
        }
        q[w * 1024 + k] = v;
    }
}

// serialize_sparse of independent 1024-value vectors into fixed slots (flo_sparse_pack). form 0: the packer wave's own
// routine - the item form, behind it the block form for vectors with more than 128 non-zeros, behind that the general
// form for the vectors the block form declines; form 2 starts at the block form, form 1 forces the general form for
// every vector (tests compare the three).
__global__ __launch_bounds__(64) void sparse_only_kernel(const short *q, unsigned long long n, uint8_t *slots,
                                                         uint32_t *sizes, int form) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[2080 + 128];
    __shared__ uint32_t runtab[kPackTabDwords];
    const unsigned long long w = blockIdx.x;
    if (w >= n) return;
    const int lane = lane_id();
    const unsigned short *qv = reinterpret_cast<const unsigned short *>(q) + w * 1024;
    uint32_t total = kSparseFallback;
    if (form == 0 || form == 2) {
        uint32_t xd[8];
#pragma unroll
        for (int k = 0; k < 8; k++) xd[k] = reinterpret_cast<const uint32_t *>(qv)[64 * k + lane];
        if (form == 0) total = sparse_item_pack(lane, xd, (uint32_t)(uintptr_t)stage, (uint32_t)(uintptr_t)runtab);
        if (total == kSparseFallback) total = sparse_block_pack(lane, xd, (uint32_t)(uintptr_t)stage, (uint32_t)(uintptr_t)runtab);
    }
    if (total == kSparseFallback) {
        int v[16];
#pragma unroll
        for (int e = 0; e < 16; e++) v[e] = q[w * 1024 + 16 * lane + e];
        SparsePlan P;
        sparse_plan(lane, v, P);
        sparse_emit(lane, v, P, stage, 2080u + 2u * (uint32_t)lane);
        total = P.total;
    }
    wave_sync();
    for (uint32_t i = lane; i < total; i += 64) slots[w * 2080 + i] = stage[i];
    if (lane == 0) sizes[w] = total;
}

// Copy every clip's DATA chunk or finished file into one caller-provided buffer, back to back at 16-byte aligned
// offsets (the packed form handed to the RCCL gather). Destinations are 16-byte aligned; sources need not be (a
// finished file starts 74 + 20 frames bytes in front of its aligned DATA chunk): each 16-byte unit is assembled from the
// two aligned units that cover it, and the bytes behind a clip's last byte are written as zeros. The aligned reads stay
// inside the batch's output allocation (it starts aligned and ends with slack).
__global__ void pack_streams_kernel(const uint8_t *src, const unsigned long long *src_off, const unsigned long long *dst_off,
                                    const unsigned long long *sizes, int n_clips, uint8_t *dst) {
    const int clip = blockIdx.x;   // clips in x: gridDim.y stops at 65535
    if (clip >= n_clips) return;
    const unsigned long long size = sizes[clip];
    const unsigned long long n16 = (size + 15) >> 4;
    const uint8_t *sp = src + src_off[clip];
    const unsigned mis = (unsigned)((uintptr_t)sp & 15u);
    const uint4 *s = reinterpret_cast<const uint4 *>(sp - mis);
    const unsigned wq = mis >> 2, sh = (mis & 3u) * 8u;
    uint4 *d = reinterpret_cast<uint4 *>(dst + dst_off[clip]);
    for (unsigned long long i = (unsigned long long)blockIdx.y * blockDim.x + threadIdx.x; i < n16;
         i += (unsigned long long)gridDim.y * blockDim.x) {
        uint4 v = s[i];
        if (mis) {
            const uint4 hi = s[i + 1];
            const uint32_t w[8] = {v.x, v.y, v.z, v.w, hi.x, hi.y, hi.z, hi.w};
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // words wq + j and wq + j + 1, selected without run-time register indexing (wq is uniform)
                uint32_t lo = 0, up = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    lo = wq == (unsigned)q ? w[q + j] : lo;
                    up = wq == (unsigned)q ? w[q + j + 1 < 8 ? q + j + 1 : 7] : up;
                }
                o[j] = sh ? (lo >> sh) | (up << (32u - sh)) : lo;
            }
            v = make_uint4(o[0], o[1], o[2], o[3]);
        }
        if (i == n16 - 1 && (size & 15u)) {   // zero the padding behind the clip's last byte
            const unsigned keep = (unsigned)(size & 15u);
            uint32_t o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int kb = (int)keep - 4 * j;   // bytes of word j that belong to the clip
                o[j] = kb >= 4 ? o[j] : (kb <= 0 ? 0u : (o[j] & ((1u << (8 * kb)) - 1u)));
            }
            v = make_uint4(o[0], o[1], o[2], o[3]);
        }
        d[i] = v;
    }
}

// integer-exact synthetic PCM (include/flo_synth.h), one thread per 4 interleaved samples
__global__ void synth_fill_kernel(float *pcm, const unsigned long long *clip_off, const unsigned long long *clip_nsf,
                                  int n_clips, int nch, uint32_t seed, unsigned long long clip_id0) {
    const unsigned clip = blockIdx.x;   // clips in x: gridDim.y stops at 65535
    if (clip >= (unsigned)n_clips) return;
    const unsigned long long n = clip_nsf[clip] * (unsigned long long)nch;
    float *p = pcm + clip_off[clip];
    for (unsigned long long i = (unsigned long long)blockIdx.y * blockDim.x + threadIdx.x; i < n;
         i += (unsigned long long)gridDim.y * blockDim.x) {
        const unsigned long long sf = i / (unsigned)nch;
        const unsigned ch = (unsigned)(i % (unsigned)nch);
        p[i] = flo_synth_sample(seed, clip_id0 + clip, ch, sf);
    }
}

// ---------------------------------------------------------------------------------------------- launchers
#define FLO_LAUNCH_CHECK()                     \
    do {                                       \
        hipError_t e_ = hipGetLastError();     \
        if (e_ != hipSuccess) return (int)e_;  \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: set once per (device, kernel), from
// whichever thread launches first on that device (contexts on several GPUs may live in one process).
static int allow_big_lds(const void *fn) {
    static std::mutex mu;
    static std::set<std::pair<int, const void *>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    std::lock_guard<std::mutex> g(mu);
    if (done.count({dev, fn})) return 0;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    done.insert({dev, fn});
    return 0;
}

// clips per workgroup: as few as fill the chip once (256 CUs), at most what 160 KiB of LDS holds
int chain_clips_per_wg(int n_clips) {
    int g = (n_clips + 255) / 256;
    const int gmax = (int)((160 * 1024 - kPackBytes) / sizeof(ClipLds));
    if (g > gmax) g = gmax;
    if (g * 2 * 64 > 768) g = 6;   // the kernel is built for at most 768 threads (12 waves = 3 per SIMD)
    return g < 1 ? 1 : g;
}

template <int NW, bool EXACT>
static int launch_chain_t(const LossyArgs &A, hipStream_t s) {
    const int g = chain_clips_per_wg(A.n_clips);
    const size_t lds = kPackBytes + (size_t)g * sizeof(ClipLds);
    if (int rc = allow_big_lds(reinterpret_cast<const void *>(&lossy_chain_kernel<NW, EXACT>))) return rc;
    const unsigned wgs = (unsigned)((A.n_clips + g - 1) / g);
    hipLaunchKernelGGL((lossy_chain_kernel<NW, EXACT>), dim3(wgs), dim3(64 * NW * g), lds, s, A, g);
    FLO_LAUNCH_CHECK();
    return 0;
}

// clips per workgroup of the two-wave form whose packer quantises
int chain2q_clips_per_wg(int n_clips) {
    int g = (n_clips + 255) / 256;
    const int gmax = (int)((160 * 1024 - kPackBytesHotT) / sizeof(Clip2qLds));
    if (g > gmax) g = gmax;
    if (g > FLO_C2X_THREADS / 128) g = FLO_C2X_THREADS / 128;   // twelve waves: three per SIMD (up to 168 registers each)
    return g < 1 ? 1 : g;
}
template <bool COEFFS, uint32_t DIRTY, bool DBG>
static int launch_chain2q_t(const LossyArgs &A, hipStream_t s) {
    int g = chain2q_clips_per_wg(A.n_clips);
    if (const char *e = getenv("FLO_CHAIN2X_CLIPS")) {   // diagnostic: clips per workgroup
        const int v = atoi(e);
        if (v >= 1 && v <= FLO_C2X_THREADS / 128) g = v;
    }
    const size_t lds = kPackBytesHotT + (size_t)g * sizeof(Clip2qLds);
    if (int rc = allow_big_lds(reinterpret_cast<const void *>(&lossy_chain2q_kernel<COEFFS, DIRTY, DBG>))) return rc;
    unsigned wgs = (unsigned)((A.n_clips + g - 1) / g);
    if (A.n_cus > 0 && wgs > (unsigned)A.n_cus) wgs = (unsigned)A.n_cus;   // persistent: one workgroup per CU, clips dealt dynamically
    hipLaunchKernelGGL((lossy_chain2q_kernel<COEFFS, DIRTY, DBG>), dim3(wgs), dim3(128 * g), lds, s, A, g);
    FLO_LAUNCH_CHECK();
    return 0;
}
int launch_lossy_chain2q(const LossyArgs &A, hipStream_t s) {
    if (A.nch != 2 || A.exact) return -1;   // the exact-threshold test yardstick lives in the other forms
    if (A.in_coeffs) return launch_chain2q_t<true, 0xFFFFu, true>(A, s);
    if (A.dbg_coeffs || A.dbg_q || A.dbg_sfw) return launch_chain2q_t<false, 0xFFFFu, true>(A, s);
    return (A.T.dirty | 0x8000u) == kDirty44k ? launch_chain2q_t<false, kDirty44k, false>(A, s) : launch_chain2q_t<false, 0xFFFFu, false>(A, s);
}
int launch_lossy_chain(const LossyArgs &A, hipStream_t s) {
    if (A.nch == 1) return A.exact ? launch_chain_t<1, true>(A, s) : launch_chain_t<1, false>(A, s);
    if (A.nch == 2) return A.exact ? launch_chain_t<2, true>(A, s) : launch_chain_t<2, false>(A, s);
    return -1;
}
int launch_lossy_frames_pass(const LossyArgs &A, int pass, hipStream_t s) {
    dim3 g((unsigned)A.total_frames), b(64);
    if (A.nch == 1) {
        if (pass == 1) hipLaunchKernelGGL((lossy_frame_kernel<1, 1, false>), g, b, 0, s, A);
        else if (A.exact) hipLaunchKernelGGL((lossy_frame_kernel<1, 2, true>), g, b, 0, s, A);
        else hipLaunchKernelGGL((lossy_frame_kernel<1, 2, false>), g, b, 0, s, A);
    } else if (A.nch == 2 && !A.exact && !A.in_coeffs && !getenv("FLO_FRAME_OLD")) {
        if (pass == 1) hipLaunchKernelGGL((lossy_frame2x_kernel<1>), g, b, 0, s, A);
        else if (A.coef_t) hipLaunchKernelGGL((lossy_frame2x_kernel<2, true>), g, b, 0, s, A);
        else hipLaunchKernelGGL((lossy_frame2x_kernel<2>), g, b, 0, s, A);
    } else if (A.nch == 2) {
        if (pass == 1) hipLaunchKernelGGL((lossy_frame_kernel<2, 1, false>), g, b, 0, s, A);
        else if (A.exact) hipLaunchKernelGGL((lossy_frame_kernel<2, 2, true>), g, b, 0, s, A);
        else hipLaunchKernelGGL((lossy_frame_kernel<2, 2, false>), g, b, 0, s, A);
    } else if (A.nch <= kMaxLossyChannels) {
        const size_t dynb = (size_t)A.slot_bytes + 256 + (size_t)A.nch * 2048;
        if (pass == 1) hipLaunchKernelGGL((lossy_frame_n_kernel<1, false>), g, b, dynb, s, A);
        else if (A.exact) hipLaunchKernelGGL((lossy_frame_n_kernel<2, true>), g, b, dynb, s, A);
        else hipLaunchKernelGGL((lossy_frame_n_kernel<2, false>), g, b, dynb, s, A);
    } else return -1;
    FLO_LAUNCH_CHECK();
    return 0;
}
// the stereo frame-parallel form with handed-over coefficients walks the temporal chain inside pass 2: no scan launch
bool lossy_pass2_scans_itself(const LossyArgs &A) { return A.nch == 2 && !A.exact && !A.in_coeffs && A.coef_t && !getenv("FLO_FRAME_OLD"); }
int launch_lossy_scan(const LossyArgs &A, hipStream_t s) {
    if (lossy_pass2_scans_itself(A)) return 0;
    unsigned max_hops = (unsigned)A.max_hops;
    hipLaunchKernelGGL(lossy_scan_kernel, dim3(A.n_clips, (max_hops + kScanBlock - 1) / kScanBlock), dim3(32 * A.nch), 0, s, A);
    FLO_LAUNCH_CHECK();
    return 0;
}
int launch_lossy_compact(const LossyArgs &A, hipStream_t s) {
    if (A.n_clips <= 16 && !getenv("FLO_COMPACT_TWO_KERNELS")) {   // few clips: the fused form (a chunk's workgroup sums the sizes in front of it itself)
        const unsigned chunks = ((unsigned)A.max_hops + kCompactChunk - 1) / kCompactChunk;
        hipLaunchKernelGGL(lossy_offsets_compact_kernel, dim3(chunks ? chunks : 1u, (unsigned)A.n_clips), dim3(256), 0, s, A);
        FLO_LAUNCH_CHECK();
        return 0;
    }
    if (A.n_clips < 64) hipLaunchKernelGGL((lossy_frame_offsets_kernel<1024>), dim3(A.n_clips), dim3(1024), 0, s, A);
    else hipLaunchKernelGGL((lossy_frame_offsets_kernel<256>), dim3(A.n_clips), dim3(256), 0, s, A);
    FLO_LAUNCH_CHECK();
    hipLaunchKernelGGL(lossy_compact_kernel, dim3((unsigned)A.total_frames), dim3(256), 0, s, A);
    FLO_LAUNCH_CHECK();
    return 0;
}
int launch_mdct_only(const LossyDevTables &T, const float *frames, unsigned long long n, float *out, hipStream_t s) {
    hipLaunchKernelGGL(mdct_only_kernel, dim3((unsigned)n), dim3(64), 0, s, T, frames, n, out);
    FLO_LAUNCH_CHECK();
    return 0;
}
int launch_quantise_smr(const LossyDevTables &T, const float *coeffs, const float *smr, unsigned long long n, short *q, float *sf,
                        hipStream_t s) {
    hipLaunchKernelGGL(quantise_smr_kernel, dim3((unsigned)n), dim3(64), 0, s, T, coeffs, smr, n, q, sf);
    FLO_LAUNCH_CHECK();
    return 0;
}
int launch_sparse_only(const short *q, unsigned long long n, uint8_t *slots, uint32_t *sizes, int form, hipStream_t s) {
    hipLaunchKernelGGL(sparse_only_kernel, dim3((unsigned)n), dim3(64), 0, s, q, n, slots, sizes, form);
    FLO_LAUNCH_CHECK();
    return 0;
}
int launch_pack_streams(const uint8_t *src, const unsigned long long *src_off, const unsigned long long *dst_off,
                        const unsigned long long *sizes, int n_clips, uint8_t *dst, hipStream_t s) {
    hipLaunchKernelGGL(pack_streams_kernel, dim3(n_clips, 8), dim3(256), 0, s, src, src_off, dst_off, sizes, n_clips, dst);
    FLO_LAUNCH_CHECK();
    return 0;
}
int launch_synth_fill(float *pcm, const unsigned long long *clip_off, const unsigned long long *clip_nsf, int n_clips,
                      int nch, uint32_t seed, unsigned long long clip_id0, hipStream_t s) {
    hipLaunchKernelGGL(synth_fill_kernel, dim3(n_clips, 64), dim3(256), 0, s, pcm, clip_off, clip_nsf, n_clips, nch,
                       seed, clip_id0);
    FLO_LAUNCH_CHECK();
    return 0;
}

}  // namespace flo
