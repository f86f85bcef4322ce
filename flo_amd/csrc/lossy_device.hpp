// lossy_device.hpp — device functions of the lossy encode path, written for gfx950 (wave64, one wave per
// stereo/mono frame stream, no workgroup barriers: every workgroup is exactly one wavefront).
//
// Reference behaviour replaced (files under /root/reference/libflo/src):
//   fold + pre-rotation + FFT-512 + post-rotation .... lossy/mdct.rs:166-226 (MdctTransform::forward)
//   band energies, spreading, temporal masking ....... lossy/psychoacoustic.rs:151-214
//   SMR keep/drop test ................................ lossy/psychoacoustic.rs:218-235, lossy/encoder.rs:129-151
//   scale factors + quantiser ......................... lossy/encoder.rs:109-154
//   scale-factor words, sparse RLE + varint, blob ..... lossy/encoder.rs:243-329
//   frame header ...................................... writer.rs:236-254 (type 253, one channel wrapper)
//
// Data layout in a wave ("lane" = 0..63, CH = 1 or 2 channels processed in lock-step):
//   FFT input/output : lane l, register r  <->  z[l + 64 r]           (stride-64, three radix-8 passes)
//   coefficients     : lane j, register e  <->  c[16 j + e]           (contiguous, after an LDS transpose)
// The 1024-sample overlap between consecutive frames never goes through memory twice: the raw second-half
// samples a lane folds in frame h are exactly the first-half samples the same lane folds in frame h+1, so
// they are carried in registers (see DESIGN.md "fold symmetry").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pack_rows.h"

namespace flo {

// Per-lane constant pack, laid out [row][lane] as float4 so that one wave-wide load is 1 KiB contiguous; the rows are
// listed in pack_rows.h (kRowWin ... kRowLstCold).

struct LossyDevTables {
    const float4 *pack;      // [kPackRows][64]; kernels that copy the pack to LDS point this at their copy ...
    const float4 *pack_g;    // ... and keep the global copy here (rows from kPackRowsHot on may not be in LDS)
    const float4 *pack_ext;  // [2][64] slot-list groups 6, 7 (global memory; sample rates from 128 kHz up)
    const float *ath_db;     // [1024]
    const uint8_t *band;     // [1024]
    const float *band_count; // [25]
    const float *s10d;       // [25]
    const uint32_t *lane_bnd;    // [64]
    const uint32_t *lane_slot0;  // [64]
    const uint32_t *band_slot0;  // [26]
    int max_band_slots;
    uint32_t dirty;          // bit e: some lane closes a band segment at element e of its 16 (the others skip the slot store)
    float smr_thr;
    int q_transparent;
};

constexpr int kXchStride = 9;            // complex elements per exchange row (8 + 1 pad)
constexpr int kXchFloats = 64 * kXchStride * 2;  // 1152 floats per channel
#ifdef FLO_COEF_PAD16
constexpr int kCoefFloats = 1280;
#else
constexpr int kCoefFloats = 1152;        // 1024 coefficients, 4 floats of padding per 32 (see post_rotate_transpose)
#endif
constexpr int kSlotCap = 96;
constexpr int kFrameCap = 4352;          // >= 16 + 6+4+2+100+2*(4+2064), multiple of 16

// LDS owned by ONE wavefront (CH = channels it processes in lock-step). The three big arrays are live at
// different times of a frame and share storage.
template <int CH>
struct WaveLds {
    union {
        float xch[CH][kXchFloats];     // FFT exchanges (float2 pairs)
        float coef[CH][kCoefFloats];   // transposition to the contiguous layout
        int16_t qbuf[CH][1024];        // quantised values, fetched by run-time position while emitting
    } u;
    float2 slots[CH][kSlotCap + 64 + 1];  // (sum c^2, max |c|) per lane segment | 64 per-lane trash slots | one zero slot
    float2 bandv[CH][32];              // per band: (amplitude threshold, scale factor)
    float band_s[CH][32];              // per band: masking level s in dB (exact re-check only)
};

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
// The same value behind an optimisation barrier: inside the frame loop every lane-derived address is then
// recomputed per frame (a few integer ops) instead of being hoisted into dozens of loop-invariant registers.
__device__ __forceinline__ int lane_id_opaque() {
    int l = (int)(threadIdx.x & 63);
    asm volatile("" : "+v"(l));
    return l;
}

// Ordering point between LDS accesses of ONE wavefront (lanes exchange data through LDS). The hardware executes a
// wave's LDS instructions in order, so no s_barrier is needed; this only pins the compiler.
__device__ __forceinline__ void wave_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}
// Workgroup barrier that orders LDS traffic only: outstanding global loads (the next half-frame's prefetch) and
// stores (the previous frame's flush) stay in flight across it, unlike __syncthreads() which drains vmcnt(0).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ------------------------------------------------------------------------------------------------ scalar helpers
// max / |x| max without the sNaN-quieting v_max(v, v) pair the compiler adds in front of every fmaxf: the hardware
// instruction already returns the non-NaN operand, which is all f32::max needs.
__device__ __forceinline__ float max_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float max_abs_raw(float a, float b) {  // max(a, |b|)
    float r;
    asm("v_max_f32 %0, %1, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// three-way maxima (exact, like max_raw: the hardware returns the non-NaN operand): one instruction where two follow each other
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max3_abs_raw(float a, float b, float c) {  // max(a, |b|, |c|)
    float r;
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// truncating float -> int conversion as ONE instruction (saturates, NaN -> 0); the C cast would add a v_trunc
__device__ __forceinline__ int cvt_rz(float x) {
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// Byte store into LDS as its own instruction. The frame bytes sit at arbitrary alignment; left to the optimiser, two
// neighbouring byte stores can be merged into one misaligned 16-bit store, which gfx950 executes correctly but far
// slower than two byte stores (measured on the value stores: chain kernel 3.1 ms -> 5.0 ms). p must point into LDS
// (the low half of a generic LDS address is the LDS offset).
template <int OFF = 0>
__device__ __forceinline__ void lds_st8(uint8_t *p, uint32_t v) {
    const uint32_t a = (uint32_t)(uintptr_t)p;
    asm volatile("ds_write_b8 %0, %1 offset:%2" ::"v"(a), "v"(v), "n"(OFF) : "memory");
}

// ------------------------------------------------------------------------------------------------ cross-lane
// DPP moves (no LDS round trip). Lanes whose source lane does not exist keep `old`.
//   0x111..0x11F row_shr:n   0x101..0x10F row_shl:n   0x138 wave_shr:1   0x130 wave_shl:1
//   0x142 row_bcast:15       0x143 row_bcast:31
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ int dpp_i(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, BANK_MASK, false);
}
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float old, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROW_MASK, BANK_MASK, false));
}
// value of lane + 32 (lanes 0..31); lanes 32..63 receive lane - 32
__device__ __forceinline__ float from_other_half(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
    return __int_as_float(lane_id() < 32 ? r[1] : r[0]);
}

// ------------------------------------------------------------------------------------------------ DFT-8
// forward (e^{-2 pi i nk/8}), natural order in and out
__device__ __forceinline__ void dft8(float *xr, float *xi) {
    const float s = 0.70710678118654752440f;
    float a0r = xr[0] + xr[4], a0i = xi[0] + xi[4];
    float a1r = xr[0] - xr[4], a1i = xi[0] - xi[4];
    float a2r = xr[2] + xr[6], a2i = xi[2] + xi[6];
    float a3r = xr[2] - xr[6], a3i = xi[2] - xi[6];
    float a4r = xr[1] + xr[5], a4i = xi[1] + xi[5];
    float a5r = xr[1] - xr[5], a5i = xi[1] - xi[5];
    float a6r = xr[3] + xr[7], a6i = xi[3] + xi[7];
    float a7r = xr[3] - xr[7], a7i = xi[3] - xi[7];
    float b0r = a0r + a2r, b0i = a0i + a2i;
    float b2r = a0r - a2r, b2i = a0i - a2i;
    float b1r = a1r + a3i, b1i = a1i - a3r;  // a1 - i a3
    float b3r = a1r - a3i, b3i = a1i + a3r;  // a1 + i a3
    float c0r = a4r + a6r, c0i = a4i + a6i;
    float c2r = a4r - a6r, c2i = a4i - a6i;
    float c1r = a5r + a7i, c1i = a5i - a7r;
    float c3r = a5r - a7i, c3i = a5i + a7r;
    // W8^1 c1 = ((c1r + c1i) s, (c1i - c1r) s);  W8^3 c3 = ((c3i - c3r) s, -(c3r + c3i) s);  W8^2 c2 = (c2i, -c2r)
    const float e1r = c1r + c1i, e1i = c1i - c1r;
    const float e3r = c3i - c3r, e3i = c3r + c3i;
    xr[0] = b0r + c0r; xi[0] = b0i + c0i;
    xr[4] = b0r - c0r; xi[4] = b0i - c0i;
    xr[1] = fmaf(e1r, s, b1r); xi[1] = fmaf(e1i, s, b1i);
    xr[5] = fmaf(-e1r, s, b1r); xi[5] = fmaf(-e1i, s, b1i);
    xr[2] = b2r + c2i; xi[2] = b2i - c2r;
    xr[6] = b2r - c2i; xi[6] = b2i + c2r;
    xr[3] = fmaf(e3r, s, b3r); xi[3] = fmaf(-e3i, s, b3i);
    xr[7] = fmaf(-e3r, s, b3r); xi[7] = fmaf(e3i, s, b3i);
}

__device__ __forceinline__ void cmul(float &xr, float &xi, float wr, float wi) {
    float r = fmaf(xr, wr, -(xi * wi));
    float i = fmaf(xr, wi, xi * wr);
    xr = r;
    xi = i;
}

// ------------------------------------------------------------------------------------------------ FFT-512
// One 512-point complex FFT per channel per wave. In: lane l register r = z[l + 64 r]; out: same layout of Z.
//   z index n = 64 na + 8 nb + nc, Z index k = ka + 8 kb + 64 kc.
//   pass 1: DFT over na (registers) -> ka, twiddle W512^(lane*ka); LDS exchange nb <-> ka
//   pass 2: DFT over nb -> kb, twiddle W64^(nc*kb);             LDS exchange nc <-> kb
//   pass 3: DFT over nc -> kc.
template <int CH>
__device__ __forceinline__ void fft512(const int lane, float (&zr)[CH][8], float (&zi)[CH][8], float (*xch)[kXchFloats],
                                       const LossyDevTables &T) {
    // pass 1
#pragma unroll
    for (int c = 0; c < CH; c++) dft8(zr[c], zi[c]);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        const float4 w = T.pack[(kRowF1 + kk) * 64 + lane];
#pragma unroll
        for (int c = 0; c < CH; c++) {
            cmul(zr[c][2 * kk + 1], zi[c][2 * kk + 1], w.x, w.y);
            if (kk < 3) cmul(zr[c][2 * kk + 2], zi[c][2 * kk + 2], w.z, w.w);
        }
    }
    {
        const int nb = lane >> 3, nc = lane & 7;
#pragma unroll
        for (int c = 0; c < CH; c++) {
            float2 *x = reinterpret_cast<float2 *>(xch[c]);
#pragma unroll
            for (int ka = 0; ka < 8; ka++) x[(8 * ka + nc) * kXchStride + nb] = make_float2(zr[c][ka], zi[c][ka]);
        }
        wave_sync();
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const float2 *x = reinterpret_cast<const float2 *>(xch[c]);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                float2 v = x[lane * kXchStride + r];
                zr[c][r] = v.x;
                zi[c][r] = v.y;
            }
        }
        wave_sync();
    }
    // pass 2 (lane = 8 ka + nc, register = nb)
#pragma unroll
    for (int c = 0; c < CH; c++) dft8(zr[c], zi[c]);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        const float4 w = T.pack[(kRowF2 + kk) * 64 + lane];
#pragma unroll
        for (int c = 0; c < CH; c++) {
            cmul(zr[c][2 * kk + 1], zi[c][2 * kk + 1], w.x, w.y);
            if (kk < 3) cmul(zr[c][2 * kk + 2], zi[c][2 * kk + 2], w.z, w.w);
        }
    }
    {
        const int ka = lane >> 3, nc = lane & 7;
#pragma unroll
        for (int c = 0; c < CH; c++) {
            float2 *x = reinterpret_cast<float2 *>(xch[c]);
#pragma unroll
            for (int kb = 0; kb < 8; kb++) x[(ka + 8 * kb) * kXchStride + nc] = make_float2(zr[c][kb], zi[c][kb]);
        }
        wave_sync();
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const float2 *x = reinterpret_cast<const float2 *>(xch[c]);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                float2 v = x[lane * kXchStride + r];
                zr[c][r] = v.x;
                zi[c][r] = v.y;
            }
        }
        wave_sync();
    }
    // pass 3 (lane = ka + 8 kb, register = nc)
#pragma unroll
    for (int c = 0; c < CH; c++) dft8(zr[c], zi[c]);
}

// The same transform, one channel, with the inter-stage twiddles handed in (the decoder keeps them in registers for a
// whole run of frames instead of fetching eight rows per frame): identical operations, identical bits.
__device__ __forceinline__ void fft512_w(const int lane, float (&zr)[8], float (&zi)[8], float *xch1, const float4 (&wf1)[4],
                                         const float4 (&wf2)[4]) {
    dft8(zr, zi);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        cmul(zr[2 * kk + 1], zi[2 * kk + 1], wf1[kk].x, wf1[kk].y);
        if (kk < 3) cmul(zr[2 * kk + 2], zi[2 * kk + 2], wf1[kk].z, wf1[kk].w);
    }
    {
        const int nb = lane >> 3, nc = lane & 7;
        float2 *x = reinterpret_cast<float2 *>(xch1);
#pragma unroll
        for (int ka = 0; ka < 8; ka++) x[(8 * ka + nc) * kXchStride + nb] = make_float2(zr[ka], zi[ka]);
        wave_sync();
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const float2 v = x[lane * kXchStride + r];
            zr[r] = v.x;
            zi[r] = v.y;
        }
        wave_sync();
    }
    dft8(zr, zi);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        cmul(zr[2 * kk + 1], zi[2 * kk + 1], wf2[kk].x, wf2[kk].y);
        if (kk < 3) cmul(zr[2 * kk + 2], zi[2 * kk + 2], wf2[kk].z, wf2[kk].w);
    }
    {
        const int ka = lane >> 3, nc = lane & 7;
        float2 *x = reinterpret_cast<float2 *>(xch1);
#pragma unroll
        for (int kb = 0; kb < 8; kb++) x[(ka + 8 * kb) * kXchStride + nc] = make_float2(zr[kb], zi[kb]);
        wave_sync();
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const float2 v = x[lane * kXchStride + r];
            zr[r] = v.x;
            zi[r] = v.y;
        }
        wave_sync();
    }
    dft8(zr, zi);
}

// ------------------------------------------------------------------------------------------------ input
// Half-frame offsets a lane folds: row r < 4: i = lane + 64 r  -> even 512 + 2i, odd 511 - 2i
//                                  row r >= 4: i = lane + 64(r-4) -> even 2i, odd 1023 - 2i
__device__ __forceinline__ void half_offsets(int lane, int r, int &eo, int &oo) {
    if (r < 4) {
        int i = lane + 64 * r;
        eo = 512 + 2 * i;
        oo = 511 - 2 * i;
    } else {
        int i = lane + 64 * (r - 4);
        eo = 2 * i;
        oo = 1023 - 2 * i;
    }
}

// Load the raw samples of one half-frame (1024 sample-frames starting at real sample s0; s0 may be negative
// = pre-roll) that this lane folds: even-offset and odd-offset sample of each of its 8 rows, CH channels
// starting at channel c0 of an nch-channel interleaved clip of n_frames_total sample-frames.
template <int CH>
__device__ __forceinline__ void load_half(const int lane, const float *__restrict__ pcm, long long n_sf, int nch, int c0,
                                          long long s0, float (&he)[CH][8], float (&ho)[CH][8]) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
        int eo, oo;
        half_offsets(lane, r, eo, oo);
        long long se = s0 + eo, so = s0 + oo;
        bool ve = se >= 0 && se < n_sf, vo = so >= 0 && so < n_sf;
        if (CH == 2 && nch == 2) {
            float2 a = ve ? *reinterpret_cast<const float2 *>(pcm + 2 * se) : make_float2(0.f, 0.f);
            float2 b = vo ? *reinterpret_cast<const float2 *>(pcm + 2 * so) : make_float2(0.f, 0.f);
            he[0][r] = a.x;
            he[CH - 1][r] = a.y;
            ho[0][r] = b.x;
            ho[CH - 1][r] = b.y;
        } else {
#pragma unroll
            for (int c = 0; c < CH; c++) {
                he[c][r] = ve ? pcm[se * nch + c0 + c] : 0.f;
                ho[c][r] = vo ? pcm[so * nch + c0 + c] : 0.f;
            }
        }
    }
}

// Same, when every one of the 1024 sample-frames starting at s0 exists (all but the clip's last frame or two): no
// per-lane predicates, the 16 loads issue back to back.
template <int CH>
__device__ __forceinline__ void load_half_fast(const int lane, const float *__restrict__ pcm, int nch, int c0, long long s0,
                                               float (&he)[CH][8], float (&ho)[CH][8]) {
    // wave-uniform base (SGPR pair) + 32-bit per-lane BYTE offsets: selects global_load ... v_off, s[base] and keeps
    // 64-bit address arithmetic off the vector ALU
    const char *base = reinterpret_cast<const char *>(pcm + s0 * nch + c0);
    const unsigned stride = 4u * (unsigned)nch;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        int eo, oo;
        half_offsets(lane, r, eo, oo);
        const unsigned be = (unsigned)eo * stride, bo = (unsigned)oo * stride;
        if (CH == 2) {
            float2 a = *reinterpret_cast<const float2 *>(base + be);
            float2 b = *reinterpret_cast<const float2 *>(base + bo);
            he[0][r] = a.x;
            he[CH - 1][r] = a.y;
            ho[0][r] = b.x;
            ho[CH - 1][r] = b.y;
        } else {
            he[0][r] = *reinterpret_cast<const float *>(base + be);
            ho[0][r] = *reinterpret_cast<const float *>(base + bo);
        }
    }
}

// Fold first half (ae, ao) and second half (be, bo) into the 8 complex FFT inputs of this lane
// (window, butterflies and pre-rotation of mdct.rs:174-197).
template <int CH>
__device__ __forceinline__ void fold(const int lane, const float (&ae)[CH][8], const float (&ao)[CH][8], const float (&be)[CH][8],
                                     const float (&bo)[CH][8], float (&zr)[CH][8], float (&zi)[CH][8],
                                     const LossyDevTables &T) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const float4 ww = T.pack[(kRowWin + r) * 64 + lane];
        const float wae = ww.x, wao = ww.y, wbe = ww.z, wbo = ww.w;
        const float4 t4 = T.pack[(kRowTw + (r >> 1)) * 64 + lane];
        const float2 w = (r & 1) ? make_float2(t4.z, t4.w) : make_float2(t4.x, t4.y);
#pragma unroll
        for (int c = 0; c < CH; c++) {
            // One product and one fused multiply-add per value (the reference rounds both products; like the FFT's
            // own arithmetic this differs from it at the 1e-7 level, far inside the transform's parity tolerance).
            float re, im;
            if (r < 4) {
                // re = -x[2i+n3] - x[n3-1-2i];  im = -x[n4+2i] + x[n4-1-2i]
                re = fmaf(-bo[c][r], wbo, -(be[c][r] * wbe));
                im = fmaf(ao[c][r], wao, -(ae[c][r] * wae));
            } else {
                // re2 = x[2i] - x[n2-1-2i];  im2 = -x[n2+2i] - x[n-1-2i]
                re = fmaf(-ao[c][r], wao, ae[c][r] * wae);
                im = fmaf(-bo[c][r], wbo, -(be[c][r] * wbe));
            }
            zr[c][r] = fmaf(-im, w.y, -(re * w.x));
            zi[c][r] = fmaf(re, w.y, -(im * w.x));
        }
    }
}

// Post-rotation (mdct.rs:203-223) + transpose to the contiguous layout through LDS:
//   out[2m] = -Z.re w.re - Z.im w.im,  out[1023 - 2m] = -Z.re w.im + Z.im w.re,   m = lane + 64 r
// Coefficient k sits at dword k + 4 (k >> 5). A store instruction writes every other coefficient (2m, or 1023 - 2m), so
// its 32 lanes of a bank group touch four blocks of 16 coefficients, 8 lanes each at a stride of two dwords: with the
// blocks at bank offsets 0, 16, 4, 20 no bank is hit more than twice (free for ds_write_b32), where 4 floats of padding
// per 16 put three lanes on the same bank. The 16-byte alignment the ds_read_b128 of the read side needs is kept.
template <int CH>
__device__ __forceinline__ void post_rotate_transpose(const int lane, const float (&zr)[CH][8], const float (&zi)[CH][8],
                                                      float (*coef)[kCoefFloats], float (&c)[CH][16],
                                                      const LossyDevTables &T) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int m = lane + 64 * r;
        const float4 t4 = T.pack[(kRowTw + (r >> 1)) * 64 + lane];
        const float2 w = (r & 1) ? make_float2(t4.z, t4.w) : make_float2(t4.x, t4.y);
        const int k0 = 2 * m, k1 = 1023 - 2 * m;
#ifdef FLO_COEF_PAD16
        const int p0 = k0 + 4 * (k0 >> 4), p1 = k1 + 4 * (k1 >> 4);
#else
        const int p0 = k0 + 4 * (k0 >> 5), p1 = k1 + 4 * (k1 >> 5);
#endif
#pragma unroll
        for (int ch = 0; ch < CH; ch++) {
            float R = fmaf(-zi[ch][r], w.y, -(zr[ch][r] * w.x));
            float I = fmaf(zi[ch][r], w.x, -(zr[ch][r] * w.y));
            coef[ch][p0] = R;
            coef[ch][p1] = I;
        }
    }
    wave_sync();
#pragma unroll
    for (int ch = 0; ch < CH; ch++) {
#ifdef FLO_COEF_PAD16
        const float4 *p = reinterpret_cast<const float4 *>(&coef[ch][20 * lane]);
#else
        const float4 *p = reinterpret_cast<const float4 *>(&coef[ch][16 * lane + 4 * (lane >> 1)]);
#endif
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float4 v = p[q];
            c[ch][4 * q + 0] = v.x;
            c[ch][4 * q + 1] = v.y;
            c[ch][4 * q + 2] = v.z;
            c[ch][4 * q + 3] = v.w;
        }
    }
    wave_sync();
}

// ------------------------------------------------------------------------------------------------ stereo in lock-step
// The transform of BOTH channels of a stereo frame in one wave, the two channels as the halves of a float2: every
// arithmetic instruction is a packed one (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, one issue slot for two channels;
// constants are broadcast to both halves by the instruction's op_sel bits, negations are source modifiers), and the LDS
// exchanges move 16 bytes per element (re0, re1, im0, im1). Each component goes through exactly the operations of the
// single-channel functions above - same products, same fused multiply-adds, same order - so the coefficients are
// bit-identical to theirs (the kernel forms are compared byte for byte by the tests).
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
// (-a) * w.y + (-c) and a * w.y + (-c) on both halves, w.y broadcast by op_sel: the same fused operation as fma2 with a splat
// of w.y - for the SECOND pair of a 16-byte constant row the compiler copies the component into a fresh register pair
// first (one v_mov per use site: a dozen per frame in the fold alone), for the first pair it uses op_sel as here.
__device__ __forceinline__ v2f fma2_na_hi_nc(v2f a, v2f w, v2f c) {
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,1] neg_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(c));
    return r;
}
__device__ __forceinline__ v2f fma2_a_hi_c(v2f a, v2f w, v2f c) {   // a * w.y + c
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "v"(w), "v"(c));
    return r;
}
__device__ __forceinline__ v2f mul2_hi(v2f a, v2f w) {   // a * w.y on both halves
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "v"(w));
    return r;
}
__device__ __forceinline__ v2f fma2_a_hi_nc(v2f a, v2f w, v2f c) {
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(w), "v"(c));
    return r;
}
__device__ __forceinline__ v2f splat2(float x) { return (v2f){x, x}; }

constexpr int kXch4 = 64 * kXchStride;        // float4 elements of the stereo exchange buffer
constexpr int kCoef2 = 1024 + 2 * 64;         // float2 elements: 16 coefficient pairs per lane, 2 elements of padding each

__device__ __forceinline__ void dft8_2(v2f *xr, v2f *xi) {
    const v2f s = splat2(0.70710678118654752440f);
    v2f a0r = xr[0] + xr[4], a0i = xi[0] + xi[4];
    v2f a1r = xr[0] - xr[4], a1i = xi[0] - xi[4];
    v2f a2r = xr[2] + xr[6], a2i = xi[2] + xi[6];
    v2f a3r = xr[2] - xr[6], a3i = xi[2] - xi[6];
    v2f a4r = xr[1] + xr[5], a4i = xi[1] + xi[5];
    v2f a5r = xr[1] - xr[5], a5i = xi[1] - xi[5];
    v2f a6r = xr[3] + xr[7], a6i = xi[3] + xi[7];
    v2f a7r = xr[3] - xr[7], a7i = xi[3] - xi[7];
    v2f b0r = a0r + a2r, b0i = a0i + a2i;
    v2f b2r = a0r - a2r, b2i = a0i - a2i;
    v2f b1r = a1r + a3i, b1i = a1i - a3r;
    v2f b3r = a1r - a3i, b3i = a1i + a3r;
    v2f c0r = a4r + a6r, c0i = a4i + a6i;
    v2f c2r = a4r - a6r, c2i = a4i - a6i;
    v2f c1r = a5r + a7i, c1i = a5i - a7r;
    v2f c3r = a5r - a7i, c3i = a5i + a7r;
    const v2f e1r = c1r + c1i, e1i = c1i - c1r;
    const v2f e3r = c3i - c3r, e3i = c3r + c3i;
    xr[0] = b0r + c0r; xi[0] = b0i + c0i;
    xr[4] = b0r - c0r; xi[4] = b0i - c0i;
    xr[1] = fma2(e1r, s, b1r); xi[1] = fma2(e1i, s, b1i);
    xr[5] = fma2(-e1r, s, b1r); xi[5] = fma2(-e1i, s, b1i);
    xr[2] = b2r + c2i; xi[2] = b2i - c2r;
    xr[6] = b2r - c2i; xi[6] = b2i + c2r;
    xr[3] = fma2(e3r, s, b3r); xi[3] = fma2(-e3i, s, b3i);
    xr[7] = fma2(-e3r, s, b3r); xi[7] = fma2(e3i, s, b3i);
}

__device__ __forceinline__ void cmul_2(v2f &xr, v2f &xi, float wr, float wi) {
    const v2f WR = splat2(wr), WI = splat2(wi);
    v2f r = fma2(xr, WR, -(xi * WI));
    v2f i = fma2(xr, WI, xi * WR);
    xr = r;
    xi = i;
}
// the same with the twiddle as the SECOND pair (z, w) of a 16-byte row: w.y through the op_sel forms (no copy), identical operations
__device__ __forceinline__ void cmul_2_pair(v2f &xr, v2f &xi, v2f w) {
    const v2f WR = splat2(w.x);
    v2f r = fma2(xr, WR, -mul2_hi(xi, w));
    v2f i = fma2_a_hi_c(xr, w, xi * WR);
    xr = r;
    xi = i;
}

// fft512 for both channels; x4: kXch4 float4 elements (re0, re1, im0, im1), row stride kXchStride elements.
// shadow(): code of the caller's that does not touch (zr, zi) nor wait for LDS, placed behind the issue of the first
// exchange's reads: it runs while they are answered (a wave issues in order: what follows the reads in program order is what
// fills their latency).
struct NoShadow { __device__ __forceinline__ void operator()() const {} };
template <typename F = NoShadow>
__device__ __forceinline__ void fft512_2(const int lane, v2f (&zr)[8], v2f (&zi)[8], float4 *x4, const LossyDevTables &T, F &&shadow = F()) {
    dft8_2(zr, zi);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        const float4 w = T.pack[(kRowF1 + kk) * 64 + lane];
        cmul_2(zr[2 * kk + 1], zi[2 * kk + 1], w.x, w.y);
        if (kk < 3) cmul_2_pair(zr[2 * kk + 2], zi[2 * kk + 2], (v2f){w.z, w.w});
    }
    {
        const int nb = lane >> 3, nc = lane & 7;
#pragma unroll
        for (int ka = 0; ka < 8; ka++) x4[(8 * ka + nc) * kXchStride + nb] = make_float4(zr[ka].x, zr[ka].y, zi[ka].x, zi[ka].y);
        wave_sync();
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const float4 v = x4[lane * kXchStride + r];
            zr[r] = (v2f){v.x, v.y};
            zi[r] = (v2f){v.z, v.w};
        }
        shadow();
        wave_sync();
    }
    dft8_2(zr, zi);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        const float4 w = T.pack[(kRowF2 + kk) * 64 + lane];
        cmul_2(zr[2 * kk + 1], zi[2 * kk + 1], w.x, w.y);
        if (kk < 3) cmul_2_pair(zr[2 * kk + 2], zi[2 * kk + 2], (v2f){w.z, w.w});
    }
    {
        const int ka = lane >> 3, nc = lane & 7;
#pragma unroll
        for (int kb = 0; kb < 8; kb++) x4[(ka + 8 * kb) * kXchStride + nc] = make_float4(zr[kb].x, zr[kb].y, zi[kb].x, zi[kb].y);
        wave_sync();
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const float4 v = x4[lane * kXchStride + r];
            zr[r] = (v2f){v.x, v.y};
            zi[r] = (v2f){v.z, v.w};
        }
        wave_sync();
    }
    dft8_2(zr, zi);
}

// half-frame of a stereo clip as float2 loads: he[r] / ho[r] = (left, right) of the lane's even / odd sample of row r
// Byte offsets (half_offsets x 8): rows r < 4 even 4096 + 16 lane + 1024 r, odd 4088 - 16 lane - 1024 r; rows r >= 4 even
// 16 lane + 1024 (r - 4), odd 8184 - 16 lane - 1024 (r - 4). With va = 16 lane and vb = 16 (63 - lane) every load is
// (uniform base or base + 4096) + (va or vb) + an instruction offset below 4096: two address registers, no arithmetic.
__device__ __forceinline__ void load_half_fast_2(const int lane, const float *__restrict__ pcm, long long s0, v2f (&he)[8], v2f (&ho)[8]) {
    const char *base0 = reinterpret_cast<const char *>(pcm + s0 * 2);
    const char *base1 = base0 + 4096;
    const unsigned va = 16u * (unsigned)lane, vb = 16u * (unsigned)(63 - lane);
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int rr = r & 3;
        const char *pe = (r < 4 ? base1 : base0) + va + 1024 * rr;
        const char *po = (r < 4 ? base0 : base1) + vb + (3080 - 1024 * rr);
        const float2 a = *reinterpret_cast<const float2 *>(pe);
        const float2 b = *reinterpret_cast<const float2 *>(po);
        he[r] = (v2f){a.x, a.y};
        ho[r] = (v2f){b.x, b.y};
    }
}

__device__ __forceinline__ void fold_2(const int lane, const v2f (&ae)[8], const v2f (&ao)[8], const v2f (&be)[8], const v2f (&bo)[8],
                                       v2f (&zr)[8], v2f (&zi)[8], const LossyDevTables &T) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const float4 ww = T.pack[(kRowWin + r) * 64 + lane];
        const v2f wae = splat2(ww.x), wao = splat2(ww.y), wbe = splat2(ww.z);
        const v2f whi = {ww.z, ww.w};   // (ww.w enters through fma2_na_hi_nc)
        const float4 t4 = T.pack[(kRowTw + (r >> 1)) * 64 + lane];
        const v2f wx = splat2((r & 1) ? t4.z : t4.x);
        const v2f thi = {t4.z, t4.w};
        v2f re, im;
        if (r < 4) {
            re = fma2_na_hi_nc(bo[r], whi, be[r] * wbe);
            im = fma2(ao[r], wao, -(ae[r] * wae));
        } else {
            re = fma2(-ao[r], wao, ae[r] * wae);
            im = fma2_na_hi_nc(bo[r], whi, be[r] * wbe);
        }
        if (r & 1) {
            zr[r] = fma2_na_hi_nc(im, thi, re * wx);
            zi[r] = fma2_a_hi_nc(re, thi, im * wx);
        } else {
            const v2f wy = splat2(t4.y);
            zr[r] = fma2(-im, wy, -(re * wx));
            zi[r] = fma2(re, wy, -(im * wx));
        }
    }
}

// post-rotation + transposition; c2: kCoef2 float2 elements, coefficient k of both channels at element k + 2 (k >> 4)
__device__ __forceinline__ void post_rotate_transpose_2(const int lane, const v2f (&zr)[8], const v2f (&zi)[8], float2 *c2,
                                                        v2f (&c)[16], const LossyDevTables &T) {
    // element of coefficient 2 m (m = lane + 64 r): 2 lane + 2 (lane >> 3) + 144 r; of coefficient 1023 - 2 m: 1149 minus
    // that. One address register per direction, the row term is an instruction offset.
    const int pl = 2 * lane + 2 * (lane >> 3);
    float2 *const w0 = c2 + pl;
    float2 *const w1 = c2 + (1149 - 7 * 144) - pl;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const float4 t4 = T.pack[(kRowTw + (r >> 1)) * 64 + lane];
        const v2f wx = splat2((r & 1) ? t4.z : t4.x);
        v2f R, I;
        if (r & 1) {   // (the row's second pair: its .w through the op_sel forms, see fma2_na_hi_nc)
            const v2f thi = {t4.z, t4.w};
            R = fma2_na_hi_nc(zi[r], thi, zr[r] * wx);
            I = fma2(zi[r], wx, -mul2_hi(zr[r], thi));
        } else {
            const v2f wy = splat2(t4.y);
            R = fma2(-zi[r], wy, -(zr[r] * wx));
            I = fma2(zi[r], wx, -(zr[r] * wy));
        }
        w0[144 * r] = make_float2(R.x, R.y);
        w1[144 * (7 - r)] = make_float2(I.x, I.y);
    }
    wave_sync();
    const float4 *p = reinterpret_cast<const float4 *>(&c2[18 * lane]);
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const float4 v = p[q];
        c[2 * q] = (v2f){v.x, v.y};
        c[2 * q + 1] = (v2f){v.z, v.w};
    }
    wave_sync();
}

// ------------------------------------------------------------------------------------------------ bands
// Per-lane constants of the contiguous layout (lane j owns coefficients 16 j .. 16 j + 15)
struct LaneConst {
    float rcount;      // band lanes: 1 / bins in the band (0 for an empty band)
};
// Band b is reduced by lane b (even slots of the band) and lane 32 + b (odd slots); lane b adds the two halves.

__device__ __forceinline__ void load_lane_const(const int lane, LaneConst &L, const LossyDevTables &T) {
    L.rcount = T.pack[kRowLane * 64 + lane].z;
}

constexpr int kZeroSlot = kSlotCap + 64;

// Band energy (sum of c^2) and band maximum |c| (psychoacoustic.rs:155-163, encoder.rs:111-118).
// Each lane accumulates its 16 coefficients in ascending order; after every element the running (sum, max) is
// stored to the lane's next segment slot when the element closes a band segment, else to the lane's trash slot
// (destinations and the 0/1 restart multipliers come from the pack: no compares, selects or branches).
// Lanes b and 32+b then add the even / odd slots of band b in ascending order and lane b adds the two halves:
// a fixed summation tree, independent of run and of grid shape. Result in lanes 0..24.
template <int CH>
__device__ __forceinline__ void band_stats(const int lane, const float (&c)[CH][16], float2 (*slots)[kSlotCap + 64 + 1],
                                           const LossyDevTables &T, float (&energy)[CH], float (&bmax)[CH]) {
    // LDS round trips are what this function costs, so they are batched: every constant row first, then the running
    // sums with their slot writes, then all slot reads of the first three list groups (24 slots per band: enough for
    // every band at 44.1 / 48 kHz), then arithmetic. Further groups (96 kHz) take the generic loop.
    float kp[16];
    uint32_t dv[16], so[12];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const float4 keep = T.pack[(kRowKeep + g) * 64 + lane];
        const float4 dsto = T.pack[(kRowDst + g) * 64 + lane];
        kp[4 * g + 0] = keep.x, kp[4 * g + 1] = keep.y, kp[4 * g + 2] = keep.z, kp[4 * g + 3] = keep.w;
        dv[4 * g + 0] = __float_as_uint(dsto.x), dv[4 * g + 1] = __float_as_uint(dsto.y);
        dv[4 * g + 2] = __float_as_uint(dsto.z), dv[4 * g + 3] = __float_as_uint(dsto.w);
    }
#pragma unroll
    for (int g = 0; g < 3; g++) {
        const float4 lst = T.pack[(kRowLst + g) * 64 + lane];
        so[4 * g + 0] = __float_as_uint(lst.x), so[4 * g + 1] = __float_as_uint(lst.y);
        so[4 * g + 2] = __float_as_uint(lst.z), so[4 * g + 3] = __float_as_uint(lst.w);
    }
    __builtin_amdgcn_sched_barrier(0);
    float acc[CH], mx[CH];
#pragma unroll
    for (int ch = 0; ch < CH; ch++) {
        acc[ch] = 0.f;
        mx[ch] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 16; e++) {
#pragma unroll
        for (int ch = 0; ch < CH; ch++) {
            acc[ch] = fmaf(c[ch][e], c[ch][e], acc[ch]);
            mx[ch] = max_abs_raw(mx[ch], c[ch][e]);
            *reinterpret_cast<float2 *>(reinterpret_cast<char *>(slots[ch]) + dv[e]) = make_float2(acc[ch], mx[ch]);
            acc[ch] *= kp[e];
            mx[ch] *= kp[e];
        }
    }
    wave_sync();
    // lanes b / 32+b: even / odd slots of band b in ascending order; the list is padded with the zero slot
    float2 v[CH][12];
#pragma unroll
    for (int u = 0; u < 12; u++)
#pragma unroll
        for (int ch = 0; ch < CH; ch++)
            v[ch][u] = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(slots[ch]) + so[u]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ch = 0; ch < CH; ch++) {
        energy[ch] = 0.f;
        bmax[ch] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < 12; u++)
#pragma unroll
        for (int ch = 0; ch < CH; ch++) {
            energy[ch] += v[ch][u].x;
            bmax[ch] = max_raw(bmax[ch], v[ch][u].y);
        }
    const int groups = (T.max_band_slots + 7) >> 3;  // 4 list entries per group, each lane takes every other slot
    for (int g = 3; g < groups; g++) {
        const float4 lst = g < 6 ? T.pack_g[(kRowLstCold + g - 3) * 64 + lane] : T.pack_ext[(g - 6) * 64 + lane];
        const uint32_t sx[4] = {__float_as_uint(lst.x), __float_as_uint(lst.y), __float_as_uint(lst.z), __float_as_uint(lst.w)};
        float2 vx[CH][4];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int ch = 0; ch < CH; ch++)
                vx[ch][u] = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(slots[ch]) + sx[u]);
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int ch = 0; ch < CH; ch++) {
                energy[ch] += vx[ch][u].x;
                bmax[ch] = max_raw(bmax[ch], vx[ch][u].y);
            }
    }
#pragma unroll
    for (int ch = 0; ch < CH; ch++) {
        energy[ch] += from_other_half(energy[ch]);   // lanes 0..24: even-slot half + odd-slot half
        bmax[ch] = max_raw(bmax[ch], from_other_half(bmax[ch]));
    }
    wave_sync();
}

// Spreading + masking offset (psychoacoustic.rs:166-194): lanes 0..24 in, a[band] out (before temporal masking).
// 10 log10(e / n) is evaluated as (10 log10 2) * log2(e * (1/n)) with the hardware log2 (1 ulp): the thresholds it
// feeds are compared at the 1e-6 level by both implementations.
__device__ __forceinline__ float spread_threshold(const int lane, float energy, float rcount, const LossyDevTables &T) {
    const float4 sd0 = T.pack[kRowS10 * 64], sd1 = T.pack[kRowS10 * 64 + 1];
    const bool is_band = lane < 25;
    float band_db = -100.0f;
    if (is_band && rcount > 0.f && energy > 1e-10f) band_db = 3.01029995663981195f * __builtin_amdgcn_logf(energy * rcount);
    if (!is_band) band_db = -__builtin_inff();
    // suffix maximum: bands j >= i mask band i at full strength (spreading[j][i] = 1 for j >= i). Within each row of
    // 16 lanes by row_shl steps; row 0 then takes the maximum of bands 16..24 from lane 16.
    const float ninf = -__builtin_inff();
    float sm = band_db;
    sm = max_raw(sm, dpp_f<0x101>(ninf, sm));
    sm = max_raw(sm, dpp_f<0x102>(ninf, sm));
    sm = max_raw(sm, dpp_f<0x104>(ninf, sm));
    sm = max_raw(sm, dpp_f<0x108>(ninf, sm));
    const float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sm), 16));
    if (lane < 16) sm = max_raw(sm, hi);
    // bands j < i: band_db[j] + s10d[i-j]. Deltas 1..8 always (their constants sit in pack row 45, read before the
    // logarithm so the LDS latency is hidden); larger deltas only when band_db_max + s10d[d] can exceed -100 dB,
    // which takes levels above 99 dB (input far outside [-1, 1]).
    float m = max_raw(-100.0f, sm);
    float cur = band_db;
    const float sd[8] = {sd0.x, sd0.y, sd0.z, sd0.w, sd1.x, sd1.y, sd1.z, sd1.w};
#pragma unroll
    for (int d = 1; d <= 8; d++) {
        cur = dpp_f<0x138>(ninf, cur);  // wave_shr:1 -> band_db[lane - d]
        m = max_raw(m, cur + sd[d - 1]);
    }
    const float gmax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sm), 0));
    if (gmax >= 99.0f) {
        int dmax = 24;
        if (gmax < 500.f) {
            int d = (int)((gmax + 100.0f) * (1.0f / 24.9f)) + 1;
            dmax = d < 1 ? 1 : (d > 24 ? 24 : d);
        }
        const float *srow = reinterpret_cast<const float *>(T.pack + kRowS10 * 64);
        for (int d = 9; d <= dmax; d++) {
            cur = dpp_f<0x138>(ninf, cur);
            m = max_raw(m, cur + srow[d - 1]);
        }
    }
    return m + (-6.0f);
}

// x = max(x, x of the lane CTRL names); lanes without a source keep x. One instruction: the maximum itself carries the
// DPP modifier (the generic dpp_f + max_raw pair costs a constant load, a move and the maximum). The s_nop covers the
// two wait states a DPP read needs behind the vector write of its source.
template <int CTRL>
__device__ __forceinline__ float max_self_dpp(float x) {
    static_assert(CTRL == 0x101 || CTRL == 0x102 || CTRL == 0x104 || CTRL == 0x108, "row_shl:1/2/4/8");
    if (CTRL == 0x101) asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(x));
    if (CTRL == 0x102) asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shl:2 row_mask:0xf bank_mask:0xf" : "+v"(x));
    if (CTRL == 0x104) asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0xf" : "+v"(x));
    if (CTRL == 0x108) asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shl:8 row_mask:0xf bank_mask:0xf" : "+v"(x));
    return x;
}
// value of lane - 1, lane 0 receives lane 63's (wave_ror:1): every lane has a source, so no `old` operand to prepare
__device__ __forceinline__ float ror1(float x) {
    float r;
    asm("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
    return r;
}

// The same for both channels of a stereo frame in ONE pass: channel 0's bands on lanes 0..24, channel 1's on lanes
// 32..56 (band = lane & 31). Every lane goes through exactly the operations of spread_threshold; what differs is the
// bookkeeping between the halves: the row-0 / row-2 lanes take bands 16..24 from lane 16 / 48, and the shifted copy of
// channel 0's band 24 is kept from reaching channel 1's band 0 (it arrives at lane 32 after exactly eight shifts).
// (sd0, sd1 = the first two float4 of row kRowS10, handed in: a caller that runs the pass in the shadow of other LDS traffic
// fetches them ahead of it)
__device__ __forceinline__ float spread_threshold_2r(const int lane, float energy, float rcount, const float4 sd0, const float4 sd1,
                                                     const LossyDevTables &T) {
    const int b = lane & 31;
    const bool is_band = b < 25;
    float band_db = -100.0f;
    if (is_band && rcount > 0.f && energy > 1e-10f) band_db = 3.01029995663981195f * __builtin_amdgcn_logf(energy * rcount);
    if (!is_band) band_db = -__builtin_inff();
    const float ninf = -__builtin_inff();
    // Two chains over the bands of both channels, written out as one instruction sequence so that each fills the other's
    // DPP wait states (a DPP operand may not be read within two instructions of its write; as separate helpers the
    // chains cost an s_nop per step):
    //   sm  = suffix maximum of band_db within each row of 16 lanes (row_shl 1, 2, 4, 8), then bands 0..15 of either
    //         channel also take bands 16..24 (lane 16 / 48) by two maxima under exec masks;
    //   acc = max over d = 1..8 of band_db[lane - d] + s10d[d]: the copies are ROTATED by one lane per step; what enters
    //         lane 0 (channel 0's band 0) comes from lanes 63, 62, ..., idle lanes holding -inf, exactly like the lanes
    //         31, 30, ... that feed channel 1's band 0 on lane 32. At the eighth step the values that started on lanes 24
    //         and 56 (band 24 of either channel) would arrive: they are cut off one step earlier, on lanes 31 and 63.
    // m = max(-100, sm, acc): the same set of maxima as the step-by-step form (no NaN can occur: band_db is a number or
    // -inf, the s10d are finite), hence the same value.
    float sm, cur, m, tt;
    {
        float h0, h1;
        unsigned long long sv;
        asm("v_mov_b32 %[sm], %[b]\n\t"
            "s_nop 0\n\t"
            "v_mov_b32_dpp %[cur], %[b] wave_ror:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32 %[t], %[cur], %[s1]\n\t"
            "v_max_f32_dpp %[sm], %[sm], %[sm] row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_mov_b32 %[acc], %[t]\n\t"
            "v_mov_b32_dpp %[cur], %[cur] wave_ror:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32 %[t], %[cur], %[s2]\n\t"
            "v_max_f32_dpp %[sm], %[sm], %[sm] row_shl:2 row_mask:0xf bank_mask:0xf\n\t"
            "v_max_f32 %[acc], %[acc], %[t]\n\t"
            "v_mov_b32_dpp %[cur], %[cur] wave_ror:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32 %[t], %[cur], %[s3]\n\t"
            "v_max_f32_dpp %[sm], %[sm], %[sm] row_shl:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_max_f32 %[acc], %[acc], %[t]\n\t"
            "v_mov_b32_dpp %[cur], %[cur] wave_ror:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32 %[t], %[cur], %[s4]\n\t"
            "v_max_f32_dpp %[sm], %[sm], %[sm] row_shl:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_max_f32 %[acc], %[acc], %[t]\n\t"
            "v_mov_b32_dpp %[cur], %[cur] wave_ror:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32 %[t], %[cur], %[s5]\n\t"
            "v_max_f32 %[acc], %[acc], %[t]\n\t"
            "v_mov_b32_dpp %[cur], %[cur] wave_ror:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32 %[t], %[cur], %[s6]\n\t"
            "v_max_f32 %[acc], %[acc], %[t]\n\t"
            "v_mov_b32_dpp %[cur], %[cur] wave_ror:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32 %[t], %[cur], %[s7]\n\t"
            "s_mov_b64 vcc, %[cut]\n\t"
            "v_cndmask_b32_e32 %[cur], %[cur], %[ninf], vcc\n\t"
            "v_max_f32 %[acc], %[acc], %[t]\n\t"
            "v_readlane_b32 %[h0], %[sm], 16\n\t"
            "v_mov_b32_dpp %[cur], %[cur] wave_ror:1 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_f32 %[t], %[cur], %[s8]\n\t"
            "v_readlane_b32 %[h1], %[sm], 48\n\t"
            "v_max_f32 %[acc], %[acc], %[t]\n\t"
            "s_mov_b64 %[sv], exec\n\t"
            "s_mov_b64 exec, 0xffff\n\t"
            "v_max_f32 %[sm], %[h0], %[sm]\n\t"
            "s_mov_b64 exec, %[m2]\n\t"
            "v_max_f32 %[sm], %[h1], %[sm]\n\t"
            "s_mov_b64 exec, %[sv]\n\t"
            "v_max3_f32 %[acc], %[sm], %[acc], %[lo]"
            : [sm] "=&v"(sm), [cur] "=&v"(cur), [acc] "=&v"(m), [t] "=&v"(tt), [h0] "=&s"(h0), [h1] "=&s"(h1), [sv] "=&s"(sv)
            : [b] "v"(band_db), [s1] "v"(sd0.x), [s2] "v"(sd0.y), [s3] "v"(sd0.z), [s4] "v"(sd0.w), [s5] "v"(sd1.x), [s6] "v"(sd1.y),
              [s7] "v"(sd1.z), [s8] "v"(sd1.w), [ninf] "v"(ninf), [cut] "s"(0x8000000080000000ull), [m2] "s"(0x0000ffff00000000ull),
              [lo] "s"(-100.0f)
            : "vcc");
    }
    const float g0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sm), 0));
    const float g1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sm), 32));
    // "some band is at 99 dB or more" on the scalar unit: a float >= 99.0f is positive, and positive floats order like their
    // bit patterns as signed integers (a negative one, sign bit set, compares below)
    // (and a NaN, exponent all ones with a mantissa, is not >= 99: 0x7f800000 = +inf is the last pattern that is)
    const auto loud = [](float g) { const int u = __float_as_int(g); return u >= 0x42c60000 && u <= 0x7f800000; };
    if (loud(g0) || loud(g1)) {
        const float gmax = g0 > g1 ? g0 : g1;
        int dmax = 24;
        if (gmax < 500.f) {
            int d = (int)((gmax + 100.0f) * (1.0f / 24.9f)) + 1;
            dmax = d < 1 ? 1 : (d > 24 ? 24 : d);
        }
        const float *srow = reinterpret_cast<const float *>(T.pack + kRowS10 * 64);
        for (int d = 9; d <= dmax; d++) {
            cur = dpp_f<0x138>(ninf, cur);
            if (b < d) cur = ninf;   // came from below band 0 of this channel (the other channel's bands, or nothing)
            m = max_raw(m, cur + srow[d - 1]);
        }
    }
    return m + (-6.0f);
}
__device__ __forceinline__ float spread_threshold_2(const int lane, float energy, float rcount, const LossyDevTables &T) {
    const float4 sd0 = T.pack[kRowS10 * 64], sd1 = T.pack[kRowS10 * 64 + 1];
    return spread_threshold_2r(lane, energy, rcount, sd0, sd1, T);
}

// amplitude-domain threshold of a masking level s (dB): 10^((smr_thr + fl(s - 10)) / 20), hardware exp2
__device__ __forceinline__ float masking_amplitude(float s, float smr_thr) {
    const float thr_db = s - 10.0f;
    return __builtin_amdgcn_exp2f((smr_thr + thr_db) * 0.16609640474436813f);  // log2(10) / 20
}

// scale-factor word (encoder.rs:262-266)
__device__ __forceinline__ uint32_t sf_word(float sf) {
    if (sf > 1e-10f) {
        // (sf > 1e-10 here: a normal number, so the library log2f's subnormal pre-scaling - five instructions: compare,
        // select, ldexp by 0, select, subtract 0 - never acts, and the bare v_log_f32 it ends in gives the same bits)
        float v = __builtin_amdgcn_logf(sf) * 256.0f + 32768.0f;
        v = __builtin_amdgcn_fmed3f(v, 0.0f, 65535.0f);   // clamp (v is a number here: one instruction instead of max + min)
        return (uint32_t)v;
    }
    return 0u;
}

// round half away from zero, exactly (f32::round)
__device__ __forceinline__ float round_away(float x) {
    float t = truncf(x);
    float d = x - t;  // exact
    return t + truncf(d + d);
}

// ------------------------------------------------------------------------------------------------ quantise
// Keep/drop + quantise 16 contiguous coefficients per channel (psychoacoustic.rs:205-234, encoder.rs:138-151).
// bandv[band] = (amplitude threshold from the masking level, scale factor). The keep test |c| > max(T_band, T_ath)
// is the reference's dB-domain test 20 log10|c| - (max(s, ath) - 10) > thr moved to the amplitude domain (both
// carry f32 rounding of a few 1e-7 relative). With EXACT, coefficients within 1e-5 (relative) of the
// threshold are re-decided with the reference's own f32 expression (used by the stage tests).
// No clamp is needed after rounding: |c * 30000/band_max| <= 30000 (1 + 2^-23), and band_max <= 1e-10 gives sf = 1.
template <int CH, bool EXACT>
__device__ __forceinline__ void quantise(const int lane, const float (&c)[CH][16], const WaveLds<CH> &lds, const LaneConst &L,
                                         const LossyDevTables &T, int (&q)[CH][16]) {
    // Two LDS round trips for the whole frame: all constant rows first, then all (threshold, scale) gathers, then
    // arithmetic only. (Left to itself the compiler interleaves them group by group: eight dependent round trips.)
    float al[16];
    uint32_t bo[16];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const float4 al4 = T.pack[(kRowAth + g) * 64 + lane];
        const float4 bo4 = T.pack[(kRowBo + g) * 64 + lane];
        al[4 * g + 0] = al4.x, al[4 * g + 1] = al4.y, al[4 * g + 2] = al4.z, al[4 * g + 3] = al4.w;
        bo[4 * g + 0] = __float_as_uint(bo4.x), bo[4 * g + 1] = __float_as_uint(bo4.y);
        bo[4 * g + 2] = __float_as_uint(bo4.z), bo[4 * g + 3] = __float_as_uint(bo4.w);
    }
    __builtin_amdgcn_sched_barrier(0);
    float2 bv[CH][16];
#pragma unroll
    for (int e = 0; e < 16; e++)
#pragma unroll
        for (int ch = 0; ch < CH; ch++)
            bv[ch][e] = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(lds.bandv[ch]) + bo[e]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 16; e++) {
#pragma unroll
        for (int ch = 0; ch < CH; ch++) {
            const float x = c[ch][e];
            const float ax = fabsf(x);
            const float thr = max_raw(bv[ch][e].x, al[e]);
            // round half away from zero == truncate(x + copysign(pred(0.5), x)) for every f32 (verified exhaustively
            // on [0.25, 4) and at all half-integers); the truncating conversion saturates and maps NaN to 0
            const float xs = x * bv[ch][e].y;
            const float half = __uint_as_float((__float_as_uint(xs) & 0x80000000u) | 0x3EFFFFFFu);
            const int v = cvt_rz(xs + half);
            if (EXACT) {
                bool keep = ax > thr;
                // |c| <= 1e-10 takes the reference's "-100 dB" branch; it can only be kept at quality >= 0.99
                const bool near = fabsf(ax - thr) <= 1e-5f * thr || (T.q_transparent && !(ax > 1e-10f));
                if (near) {
                    float signal_db = ax > 1e-10f ? 20.0f * log10f(ax) : -100.0f;
                    float sdb = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(lds.band_s[ch]) + (bo[e] >> 1));
                    float t = fmaxf(sdb, T.ath_db[16 * lane + e]) - 10.0f;
                    keep = (signal_db - t) > T.smr_thr;
                }
                q[ch][e] = keep ? v : 0;
            } else {
                // keep iff |c| > thr: all-ones mask from the sign of (thr - |c|); NaN compares false like the reference
                const int mask = __float_as_int(thr - ax) >> 31;
                q[ch][e] = v & mask;   // a NaN coefficient already gave v = 0
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ stereo analysis
// LDS of one lock-step transform wave. The exchange buffer, the transposition buffer and the band tables are live at
// different times of a frame and share storage (a wave's LDS instructions execute in order, so the next use may be
// issued as soon as the previous one's reads have been issued).
constexpr int kSlots = kSlotCap + 64 + 1;
struct StereoLds {
    union {
        float4 xch4[kXch4];
        float2 coef2[kCoef2];
        struct {
            float4 slot[kSlots];  // per lane segment: (sum c^2 left, right, max |c| left, right) | 64 trash slots | zero slot
            float4 ts[32];        // per band: amplitude thresholds (left, right), scale factors (left, right)
        } a;
    } u;
};

// band_stats for both channels: the sums are packed (one fused multiply-add per coefficient pair), the maxima are not
// (there is no packed maximum). Same slots, same order of additions per channel as band_stats<CH>: bit-identical.
// What differs is the schedule: the slot lists are fetched (and rebased) before the running sums, so that the gathers
// follow the slot writes without a table round trip in between; element positions at which no lane of the table closes
// a segment (T.dirty, a uniform bitmap) skip the store and the restart multiplication (x 1.0 for every lane); and the
// two halves of a band meet through ONE swap per quantity that also sorts the channels: the result has channel 0's
// band b on lane b and channel 1's on lane 32 + b, which is what the merged masking pass consumes.
// SO_PRE: the twelve gather addresses (so_pre) and the zero slot were prepared once by the caller (band_stats_2_prepare), whose
// slot area is not shared with anything else: fifteen instructions per frame less.
template <uint32_t DIRTY = 0xFFFFu, bool SO_PRE = false>
__device__ __forceinline__ void band_stats_2(const int lane, const v2f (&c)[16], float4 *slot, const LossyDevTables &T,
                                             float &energy1, float &bmax1, const uint32_t *so_pre = nullptr) {
    // Slot addresses as 32-bit LDS offsets from ONE scalar base (as the sum of the clip's LDS base and the member offset
    // the compiler re-added both terms for every access); the accesses go through address-space-3 pointers so that they
    // stay ds_* instructions. A slot is 16 bytes (sums of both channels, maxima of both channels): ONE ds_write_b128 per
    // stored element and ONE ds_read_b128 (4 LDS cycles) per gathered slot - a two-address ds_read2_b64 takes 8. The
    // table rows hold 8-byte-slot offsets (shared with band_stats<CH>), hence the factor 2.
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4f lds_v4f;
    const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)slot);
    // the zero slot shares storage with the exchange buffer: written every frame (lane l writes dword l & 3: no branch, one
    // register of zeros, a 4-byte store instead of a 16-byte one)
    if (!SO_PRE) {
        typedef __attribute__((address_space(3))) float lds_f32;
        float z = 0.f;
        asm volatile("" : "+v"(z));   // not a loop invariant: four registers of zeros held across the frame loop were spilled
        *reinterpret_cast<lds_f32 *>((uintptr_t)(s0 + 16u * (uint32_t)kZeroSlot + 4u * ((uint32_t)lane & 3u))) = z;
    }
    uint32_t so[12];
#pragma unroll
    for (int g = 0; g < 3; g++) {
        if (SO_PRE) {
            so[4 * g + 0] = so_pre[4 * g + 0], so[4 * g + 1] = so_pre[4 * g + 1], so[4 * g + 2] = so_pre[4 * g + 2], so[4 * g + 3] = so_pre[4 * g + 3];
        } else {
            const float4 lst = T.pack[(kRowLst + g) * 64 + lane];
            so[4 * g + 0] = s0 + 2u * __float_as_uint(lst.x), so[4 * g + 1] = s0 + 2u * __float_as_uint(lst.y);
            so[4 * g + 2] = s0 + 2u * __float_as_uint(lst.z), so[4 * g + 3] = s0 + 2u * __float_as_uint(lst.w);
        }
    }
    const uint32_t dirty = DIRTY;   // compile-time: straight-line code
    v4f am = {0.f, 0.f, 0.f, 0.f};   // running (sum left, sum right, max left, max right): one register quad, stored as it is
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const float4 keep = T.pack[(kRowKeep + g) * 64 + lane];
        const float4 dsto = T.pack[(kRowDst + g) * 64 + lane];
        const float kp[4] = {keep.x, keep.y, keep.z, keep.w};
        const uint32_t dv[4] = {__float_as_uint(dsto.x), __float_as_uint(dsto.y), __float_as_uint(dsto.z), __float_as_uint(dsto.w)};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int e = 4 * g + k;
            am.xy = fma2(c[e], c[e], am.xy);
            // (where no lane closes a segment at element e - 1, its maximum was left for this step: one three-way maximum)
            const bool prev_open = e > 0 && !((dirty >> (e - 1)) & 1u), this_open = e < 15 && !((dirty >> e) & 1u);
            if (prev_open) {
                am.z = max3_abs_raw(am.z, c[e - 1].x, c[e].x);
                am.w = max3_abs_raw(am.w, c[e - 1].y, c[e].y);
            } else if (!this_open) {
                am.z = max_abs_raw(am.z, c[e].x);
                am.w = max_abs_raw(am.w, c[e].y);
            }
            if ((dirty >> e) & 1u) {   // compile-time
#ifdef FLO_MASKED_SLOTS   // diagnostic: only the lanes that close a segment here store (the others' stores went to per-lane trash slots)
                if (kp[k] == 0.0f || e == 15) *reinterpret_cast<lds_v4f *>((uintptr_t)(s0 + 2u * dv[k])) = am;
#else
                *reinterpret_cast<lds_v4f *>((uintptr_t)(s0 + 2u * dv[k])) = am;
#endif
                if (e < 15) {
                    am.xy = am.xy * splat2(kp[k]);
                    am.zw = am.zw * splat2(kp[k]);
                }
            }
        }
    }
    wave_sync();
    v2f energy = splat2(0.f), bmax = splat2(0.f);
#pragma unroll
    for (int g = 0; g < 3; g++) {   // four slots in flight at a time: twelve would not fit the register budget
        v4f v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const lds_v4f *>((uintptr_t)so[4 * g + u]);
#pragma unroll
        for (int u = 0; u < 4; u++) energy = energy + (v2f){v[u].x, v[u].y};
#pragma unroll
        for (int u = 0; u < 4; u += 2) {
            bmax.x = max3_raw(bmax.x, v[u].z, v[u + 1].z);
            bmax.y = max3_raw(bmax.y, v[u].w, v[u + 1].w);
        }
    }
    const int groups = (T.max_band_slots + 7) >> 3;
    for (int g = 3; g < groups; g++) {
        const float4 lst = g < 6 ? T.pack_g[(kRowLstCold + g - 3) * 64 + lane] : T.pack_ext[(g - 6) * 64 + lane];
        const uint32_t sx[4] = {__float_as_uint(lst.x), __float_as_uint(lst.y), __float_as_uint(lst.z), __float_as_uint(lst.w)};
        v4f w[4];
#pragma unroll
        for (int u = 0; u < 4; u++) w[u] = *reinterpret_cast<const lds_v4f *>((uintptr_t)(s0 + 2u * sx[u]));
#pragma unroll
        for (int u = 0; u < 4; u++) {
            energy = energy + (v2f){w[u].x, w[u].y};
            bmax.x = max_raw(bmax.x, w[u].z);
            bmax.y = max_raw(bmax.y, w[u].w);
        }
    }
    // lanes b / 32 + b hold the even-slot / odd-slot halves of band b for both channels. After the swap the first result
    // has (channel 0 even | channel 1 even) and the second (channel 0 odd | channel 1 odd) on lanes (0..31 | 32..63).
    {
        const auto e2 = __builtin_amdgcn_permlane32_swap(__float_as_int(energy.x), __float_as_int(energy.y), false, false);
        const auto m2 = __builtin_amdgcn_permlane32_swap(__float_as_int(bmax.x), __float_as_int(bmax.y), false, false);
        energy1 = __int_as_float(e2[0]) + __int_as_float(e2[1]);
        bmax1 = max_raw(__int_as_float(m2[0]), __int_as_float(m2[1]));
    }
    wave_sync();
}

// once per launch, for band_stats_2<., true>: the zero slot and the gather addresses of this lane
__device__ __forceinline__ void band_stats_2_prepare(const int lane, float4 *slot, const float4 *pack, uint32_t (&so)[12]) {
    const uint32_t s0 = (uint32_t)(uintptr_t)slot;
    if (lane < 4) reinterpret_cast<float *>(slot + kZeroSlot)[lane] = 0.f;
#pragma unroll
    for (int g = 0; g < 3; g++) {
        const float4 lst = pack[(kRowLst + g) * 64 + lane];
        so[4 * g + 0] = s0 + 2u * __float_as_uint(lst.x), so[4 * g + 1] = s0 + 2u * __float_as_uint(lst.y);
        so[4 * g + 2] = s0 + 2u * __float_as_uint(lst.z), so[4 * g + 3] = s0 + 2u * __float_as_uint(lst.w);
    }
}

// The band-offset rows of the quantiser (what its gathers wait for). The chain kernel fetches them BEFORE the masking
// pass, whose long dependent chain then hides their latency.
struct QuantRows {
    float4 bo4[4];   // (the thresholds' rows are fetched next to the gathers: sixteen more registers would not fit)
};
__device__ __forceinline__ void quant_rows_load(const int lane, const LossyDevTables &T, QuantRows &R) {
#pragma unroll
    for (int g = 0; g < 4; g++) {
        R.bo4[g] = T.pack[(kRowBo + g) * 64 + lane];
    }
}
// (ts: the per-band table in LDS - amplitude thresholds (left, right), scale factors (left, right))
__device__ __forceinline__ void quantise_2(const int lane, const v2f (&c)[16], const float4 *ts, const LossyDevTables &T,
                                           const QuantRows &R, uint32_t (&xs)[2][8]) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) v4f lds_f4;
    const uint32_t ts0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)ts);
    const uint32_t sgn_mask = 0x7FFFFFFFu;
    uint32_t phalf = 0x3EFFFFFFu;
    asm volatile("" : "+v"(phalf));   // kept in a register: a VOP3 operand cannot be a literal
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const float4 al4 = T.pack[(kRowAth + g) * 64 + lane];
        const float4 bo4 = R.bo4[g];
        const float al[4] = {al4.x, al4.y, al4.z, al4.w};
        const uint32_t bo[4] = {__float_as_uint(bo4.x), __float_as_uint(bo4.y), __float_as_uint(bo4.z), __float_as_uint(bo4.w)};
        v4f tb[4];
#pragma unroll
        for (int k = 0; k < 4; k++) tb[k] = *reinterpret_cast<const lds_f4 *>((uintptr_t)(ts0 + 2u * bo[k]));   // bo = 8 * band
        int v[2][4];
        float t[2][4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const v2f x = c[4 * g + k];
            const v2f xsc = x * (v2f){tb[k].z, tb[k].w};
            uint32_t h0, h1;
            // sign of the product = sign of the coefficient (scale factors are positive): no wait for the multiplication
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(h0) : "s"(sgn_mask), "v"(phalf), "v"(x.x));
            asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(h1) : "s"(sgn_mask), "v"(phalf), "v"(x.y));
            const v2f rs = xsc + (v2f){__uint_as_float(h0), __uint_as_float(h1)};
            v[0][k] = cvt_rz(rs.x);
            v[1][k] = cvt_rz(rs.y);
            t[0][k] = max_raw(tb[k].x, al[k]);
            t[1][k] = max_raw(tb[k].y, al[k]);
        }
#pragma unroll
        for (int ch = 0; ch < 2; ch++)
#pragma unroll
            for (int k2 = 0; k2 < 2; k2++) {
                const float xe = ch ? c[4 * g + 2 * k2].y : c[4 * g + 2 * k2].x;
                const float xo = ch ? c[4 * g + 2 * k2 + 1].y : c[4 * g + 2 * k2 + 1].x;
                uint32_t r;
                // keep iff |c| > thr (a NaN coefficient compares false like the reference, and converted to 0 anyway)
                asm("v_cmp_gt_f32_e64 vcc, |%1|, %2\n\t"
                    "v_cndmask_b32_e32 %0, 0, %3, vcc\n\t"
                    "v_cmp_gt_f32_e64 vcc, |%4|, %5\n\t"
                    "v_cndmask_b32_sdwa %0, 0, %6, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
                    : "=&v"(r)
                    : "v"(xe), "v"(t[ch][2 * k2]), "v"(v[ch][2 * k2]), "v"(xo), "v"(t[ch][2 * k2 + 1]), "v"(v[ch][2 * k2 + 1])
                    : "vcc");
                xs[ch][2 * g + k2] = r;
            }
    }
}

// The quantiser in the packer wave's NATURAL layout (lossy_chain2q_kernel). cf[k] = (left, right) of position 128 k + 2 lane,
// (left, right) of the position behind it: one 16-byte read per block straight out of the transform wave's coefficient
// buffer. ts_a[e] = LDS byte address of the float4 (threshold left, right | scale factor left, right) of the band of entry
// e = 2 k + j (TSOFF selects the frame parity's table through the instruction offset), athn[e] = its ATH amplitude threshold.
// Every coefficient goes through exactly the operations of quantise_2 - same product, same rounding constant, same
// truncating conversion, same comparison - so the integers are identical. xd[ch][k] = the i16 of the two positions (even
// position in the low half): the dword sparse_block_pack takes, no re-dealing.
typedef float v4f __attribute__((ext_vector_type(4)));
// alive: bit b set when band b of either channel holds a coefficient above its masking amplitude (the transform wave's
// ballot of band maximum > threshold); blk[k]: the bands block k has bins of. A block none of whose bands is alive keeps
// nothing - the keep test is |c| > max(band threshold, ATH) - and is not quantised at all: its dwords are zero.
template <int TSOFF>
__device__ __forceinline__ void quantise_nat(const v4f (&cf)[8], const uint32_t (&ts_a)[16], const float (&athn)[16],
                                             const uint32_t alive, const uint32_t (&blk)[8], uint32_t (&xd)[2][8]) {
    typedef __attribute__((address_space(3))) v4f lds_f4;
    const uint32_t sgn_mask = 0x7FFFFFFFu;
    uint32_t phalf = 0x3EFFFFFFu;
    asm volatile("" : "+v"(phalf));   // kept in a register: a VOP3 operand cannot be a literal
    // one block: positions 128 k + 2 lane and + 1 of both channels against the tables t0 / t1 of their bands
    auto block = [&](const int k, const v4f t0, const v4f t1, uint32_t &out0, uint32_t &out1) __attribute__((always_inline)) {
        const v2f x0 = {cf[k].x, cf[k].y}, x1 = {cf[k].z, cf[k].w};
        const v2f s0 = x0 * (v2f){t0.z, t0.w}, s1 = x1 * (v2f){t1.z, t1.w};
        uint32_t h00, h01, h10, h11;
        // sign of the product = sign of the coefficient (scale factors are positive): no wait for the multiplication
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(h00) : "s"(sgn_mask), "v"(phalf), "v"(x0.x));
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(h01) : "s"(sgn_mask), "v"(phalf), "v"(x0.y));
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(h10) : "s"(sgn_mask), "v"(phalf), "v"(x1.x));
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(h11) : "s"(sgn_mask), "v"(phalf), "v"(x1.y));
        const v2f r0 = s0 + (v2f){__uint_as_float(h00), __uint_as_float(h01)};
        const v2f r1 = s1 + (v2f){__uint_as_float(h10), __uint_as_float(h11)};
        const int v0[2] = {cvt_rz(r0.x), cvt_rz(r0.y)}, v1[2] = {cvt_rz(r1.x), cvt_rz(r1.y)};
        const float a0 = athn[2 * k], a1 = athn[2 * k + 1];
        const float th0[2] = {max_raw(t0.x, a0), max_raw(t0.y, a0)}, th1[2] = {max_raw(t1.x, a1), max_raw(t1.y, a1)};
        uint32_t r[2];
#pragma unroll
        for (int ch = 0; ch < 2; ch++) {
            const float xe = ch ? x0.y : x0.x, xo = ch ? x1.y : x1.x;
            // keep iff |c| > thr (a NaN coefficient compares false like the reference, and converted to 0 anyway)
            asm("v_cmp_gt_f32_e64 vcc, |%1|, %2\n\t"
                "v_cndmask_b32_e32 %0, 0, %3, vcc\n\t"
                "v_cmp_gt_f32_e64 vcc, |%4|, %5\n\t"
                "v_cndmask_b32_sdwa %0, 0, %6, vcc dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
                : "=&v"(r[ch])
                : "v"(xe), "v"(th0[ch]), "v"(v0[ch]), "v"(xo), "v"(th1[ch]), "v"(v1[ch])
                : "vcc");
        }
        out0 = r[0];
        out1 = r[1];
    };
#pragma unroll
    for (int g = 0; g < 4; g++) {   // two blocks, four gathers in flight
        uint32_t o00 = 0u, o01 = 0u, o10 = 0u, o11 = 0u;   // [block of the pair][channel]
        if (alive & (blk[2 * g] | blk[2 * g + 1])) {   // uniform
            v4f tb[4];
#pragma unroll
            for (int j = 0; j < 4; j++) tb[j] = *reinterpret_cast<const lds_f4 *>((uintptr_t)(ts_a[4 * g + j] + (uint32_t)TSOFF));
            if (alive & blk[2 * g]) block(2 * g, tb[0], tb[1], o00, o01);
            if (alive & blk[2 * g + 1]) block(2 * g + 1, tb[2], tb[3], o10, o11);
        }
        xd[0][2 * g] = o00, xd[1][2 * g] = o01;
        xd[0][2 * g + 1] = o10, xd[1][2 * g + 1] = o11;
    }
}

// ------------------------------------------------------------------------------------------------ sparse RLE
// Non-zero mask of the lane's 16 values (bit e = value e). An add-with-carry chain (compare sets the carry, v_addc
// shifts it in: two instructions per value instead of three) was tried and lost: sixteen dependent steps.
__device__ __forceinline__ uint32_t nonzero_mask16(const int (&q)[16]) {
    uint32_t m = 0;
#pragma unroll
    for (int e = 0; e < 16; e++) m |= (q[e] != 0 ? 1u : 0u) << e;
    return m;
}
// The same for 16 values held as i16 pairs in 8 dwords (value 2k in the low half of x[k]); hi[k] = x[k] >> 16.
__device__ __forceinline__ uint32_t nonzero_mask16_packed(const uint32_t (&x)[8], const uint32_t (&hi)[8]) {
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        m |= ((x[k] & 0xFFFFu) != 0u ? 1u : 0u) << (2 * k);
        m |= (hi[k] != 0u ? 1u : 0u) << (2 * k + 1);
    }
    return m;
}
// inclusive prefix sum over the wave: Hillis-Steele inside each row of 16 lanes, then two row broadcasts
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t x) {
    int v = (int)x;
    v += dpp_i<0x111>(0, v);
    v += dpp_i<0x112>(0, v);
    v += dpp_i<0x114>(0, v);
    v += dpp_i<0x118>(0, v);
    v += dpp_i<0x142, 0xA>(0, v);  // row_bcast:15 into rows 1 and 3
    v += dpp_i<0x143, 0xC>(0, v);  // row_bcast:31 into rows 2 and 3
    return (uint32_t)v;
}

// serialize_sparse (encoder.rs:284-314) of the 1024 values of one channel held 16 per lane.
// Records: [varint zero_run][u8 n <= 255][n x i16]; a trailing zero run closes with [varint][0]; a non-zero run
// longer than 255 continues with [0][n] records; a walk that starts on a non-zero starts with [0][n].
// Every value and every record header has a closed-form byte offset:
//   offset(position p of lane) = off0 + 2 * popcount(M & K_p),  M = non-zero mask | header mask << 16
// where the header mask marks positions that start a record (zero-run starts, 255-cap continuations, the start
// record); only the last zero run of a lane can be >= 128 long (3-byte header) and nothing of that lane follows it.
struct SparsePlan {
    uint32_t M;        // bits 0..15 non-zero mask, bits 16..31 record-start mask
    int nn;            // position of the first non-zero after this lane (1024 if none)
    int nz_end;        // position of the first zero after this lane (1024 if none)
    uint32_t cross_cnt;  // count byte of the zero-run record that leaves this lane (if any)
    uint32_t off0;     // byte offset (within the channel's sparse blob) of this lane's first byte
    uint32_t total;    // total sparse bytes (uniform)
};

// m = non-zero mask of the lane's 16 values (bit e = value e)
__device__ __forceinline__ void sparse_plan_m(const int lane, const uint32_t m, SparsePlan &P) {
    const uint32_t prev_m = (uint32_t)dpp_i<0x138>(0, (int)m);  // mask of lane - 1 (0 for lane 0)
    const uint32_t prev_nz = (prev_m >> 15) & 1u;
    uint32_t zs = ~m & ((m << 1) | prev_nz) & 0xFFFFu;
    if (lane == 0 && !(m & 1u)) zs |= 1u;  // the walk starts with a zero run
    const int base = 16 * lane;
    // look-ahead / look-behind over the other lanes: find the neighbouring lane from ballots, then fetch its answer
    const uint32_t inv = ~m & 0xFFFFu;
    const int first_nz = m ? base + __builtin_ctz(m) : 1024;
    const int first_z = inv ? base + __builtin_ctz(inv) : 1024;
    const int last_z = inv ? base + 31 - __builtin_clz(inv) : -1;
    const unsigned long long has_nz = __ballot(m != 0u), has_z = __ballot(inv != 0u);
    const unsigned long long nz_above = (has_nz >> 1) >> lane, z_above = (has_z >> 1) >> lane;
    const unsigned long long z_below = lane ? has_z << (64 - lane) : 0ull;
    const int ln_nz = nz_above ? lane + 1 + __builtin_ctzll(nz_above) : lane;
    const int ln_z = z_above ? lane + 1 + __builtin_ctzll(z_above) : lane;
    const int ln_zb = z_below ? lane - 1 - __builtin_clzll(z_below) : lane;
    const int f_nz = __shfl(first_nz, ln_nz), f_z = __shfl(first_z, ln_z), l_z = __shfl(last_z, ln_zb);
    P.nn = nz_above ? f_nz : 1024;
    P.nz_end = z_above ? f_z : 1024;
    const int lz_before = z_below ? l_z : -1;
    const int t_in = base - 1 - lz_before;  // length of the non-zero run ending just before this lane
    // 255-cap continuation: a position i in the leading non-zeros with (t_in + i) % 255 == 0 and t_in + i > 0
    const int ln = inv ? __builtin_ctz(inv) : 16;
    uint32_t hmask = zs;
    if (t_in > 0) {
        int x = (255 - (t_in % 255)) % 255;
        if (x < ln) hmask |= 1u << x;
    }
    if (lane == 0 && (m & 1u)) hmask |= 1u;  // record that starts the walk on a non-zero
    P.M = m | (hmask << 16);
    // the zero run that leaves this lane (only the last one can): its length decides the varint size, and its count
    // byte needs the length of the non-zero run that follows in another lane
    uint32_t bytes = 2u * (uint32_t)__builtin_popcount(P.M);
    const int s_last = zs ? 31 - __builtin_clz(zs) : 0;
    const bool crossing = zs && ((m >> s_last) == 0u);
    const int end = crossing ? P.nn : base;
    if (crossing && end - (base + s_last) >= 128) bytes += 1u;
    {
        const int src = end < 1024 ? (end >> 4) : lane;
        const uint32_t m2 = __shfl(m, src);
        const int nzend2 = __shfl(P.nz_end, src);
        const int e2 = end & 15;
        const int run2 = __builtin_ctz(~(m2 >> e2));
        const int len = (e2 + run2 >= 16) ? nzend2 - end : run2;
        P.cross_cnt = end >= 1024 ? 0u : (uint32_t)(len < 255 ? len : 255);
    }
    const uint32_t incl = wave_incl_sum(bytes);
    P.off0 = incl - bytes;
    P.total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
}
__device__ __forceinline__ void sparse_plan(const int lane, const int (&q)[16], SparsePlan &P) {
    sparse_plan_m(lane, nonzero_mask16(q), P);
}

// Emit this lane's part of the sparse blobs of CH channels (LDS bytes; blob c starts at dst[c][0]). Zero values and
// the unused third header byte go to the lane's two private trash bytes of that channel (dst[c] + trash_off[c]).
// With CH = 2 the two channels' record loops run merged (one trip handles one record of each) and so do the value
// stores: two independent instruction streams for the price of the longer one.
template <int CH>
__device__ __forceinline__ void sparse_emit_n(const int lane, const int (&q)[CH][16], const SparsePlan (&P)[CH],
                                              uint8_t *const (&dst)[CH], const uint32_t (&trash_off)[CH]) {
    const int base = 16 * lane;
    uint32_t m[CH], hm[CH];
    int nn_rel[CH], nzend_rel[CH];
    uint32_t any = 0;
#pragma unroll
    for (int c = 0; c < CH; c++) {
        m[c] = P[c].M & 0xFFFFu;
        hm[c] = P[c].M >> 16;
#ifdef FLO_EMIT_NOHDR
        hm[c] = 0;
#endif
        nn_rel[c] = P[c].nn - base;
        nzend_rel[c] = P[c].nz_end - base;
        any |= hm[c];
    }
    // One record (per channel) per trip, branch-free: the trip count is the largest number of records any lane starts.
    while (any) {
        any = 0;
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const bool active = CH == 1 || hm[c] != 0u;
            const int s = __builtin_ctz(hm[c] | 0x10000u);
            hm[c] &= hm[c] - 1;
            any |= hm[c];
            const uint32_t below = (1u << s) - 1u;
            const uint32_t off = P[c].off0 + 2u * (uint32_t)__builtin_popcount(P[c].M & (below | (below << 16)));
            const uint32_t above = m[c] >> s;                   // bit 0 = position s
            const bool leaves = above == 0u;                    // the zero run leaves the lane
            const int zin = (int)__builtin_ctz(above | 0x10000u);   // zeros before the record's non-zero run (0 for [0][n])
            const uint32_t zc = leaves ? (uint32_t)(nn_rel[c] - s) : (uint32_t)zin;
            const int e = s + zin;                              // local start of the non-zero run (>= 16 if it leaves)
            const int run = __builtin_ctz(~(m[c] >> (e & 15)));  // m has 16 bits: always ends by position 16
            const int len = (e + run >= 16) ? nzend_rel[c] - e : run;
            uint32_t cnt = (uint32_t)(len < 255 ? len : 255);
            cnt = leaves ? P[c].cross_cnt : cnt;
            // zc >= 128 (two varint bytes) only happens when the run leaves the lane
            const bool wide = zc >= 128u;
            const uint32_t o = active ? off : trash_off[c];
            dst[c][o] = (uint8_t)(wide ? ((zc & 0x7Fu) | 0x80u) : zc);
            dst[c][o + 1] = (uint8_t)(wide ? (zc >> 7) : cnt);
            dst[c][(wide && active) ? off + 2 : trash_off[c]] = (uint8_t)cnt;
        }
    }
    // values: every position stores two bytes, zeros go to the trash bytes
#ifndef FLO_EMIT_NOVAL
#pragma unroll
    for (int i = 0; i < 16; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            const uint32_t K = ((1u << i) - 1u) | (((2u << i) - 1u) << 16);
            const uint32_t off = P[c].off0 + 2u * (uint32_t)__builtin_popcount(P[c].M & K);
            const uint32_t o = ((m[c] >> i) & 1u) ? off : trash_off[c];
            const uint32_t v = (uint32_t)q[c][i];
            dst[c][o] = (uint8_t)v;
            dst[c][o + 1] = (uint8_t)(v >> 8);
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------ sparse RLE, block form
constexpr int kRunTabEntries = 128;   // slot 0 and slot R + 1 are sentinels: up to 126 runs
constexpr uint32_t kSparseFallback = 0xFFFFFFFFu;

__device__ __forceinline__ void lds_st8_at(uint32_t a, uint32_t v, const int off) {
    if (off == 0) asm volatile("ds_write_b8 %0, %1" ::"v"(a), "v"(v) : "memory");
    else if (off == 1) asm volatile("ds_write_b8 %0, %1 offset:1" ::"v"(a), "v"(v) : "memory");
    else if (off == 2) asm volatile("ds_write_b8 %0, %1 offset:2" ::"v"(a), "v"(v) : "memory");
    else asm volatile("ds_write_b8 %0, %1 offset:3" ::"v"(a), "v"(v) : "memory");
}
__device__ __forceinline__ uint32_t mbcnt64(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// serialize_sparse (encoder.rs:284-314) from the natural-order hand-over read one dword per lane: xd[k] holds positions
// 128 k + 2 lane (low half) and + 1 (high half). The record structure of the sparse frames that make up nearly all of a
// q <= 0.8 encode (about 60 of 1024 values non-zero, 20 records) is computed in the BALLOT domain: two compares per
// block of 128 positions give the non-zero ballots E (even positions) and O (odd positions), everything that describes the record
// structure is scalar arithmetic on them (run starts S_lo = E & ~(O << 1 | carry), S_hi = O & ~E; a zero run of 128 or
// more can only sit in front of a block's FIRST non-zero), and a value's byte offset is
//   2 (non-zeros + run starts in front of it) + (wide records so far)
// from eight v_mbcnt. Two facts keep the odd positions free: the value of an odd position always sits exactly two bytes
// behind where its lane's even position goes (either the even value or the odd value's own record header fills those
// two bytes), and a lane starts at most one run, whose table slot and rank do not depend on which half starts it.
// Cost per non-empty block is independent of how many non-zeros it holds: dense frames cost what sparse ones do.
// Not handled (the caller takes the general form, which rewrites the blob): more runs than the run table holds, a
// non-zero run longer than 255 (continuation records).
__device__ __forceinline__ uint32_t sparse_block_pack(const int lane, const uint32_t (&xd)[8], const uint32_t blob, const uint32_t tab) {
    uint32_t nM = 0, nS = 0, W = 0, cz = 0;   // non-zeros, run starts, wide records so far; zeros since the last non-zero
    unsigned long long carry = 0;              // bit 0: the position in front of the block is a non-zero
    const uint32_t tmax = tab + 4u * (uint32_t)(kRunTabEntries - 1);
    const uint32_t pos_l = (uint32_t)(2 * lane);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        // ballots of the even (low halfword) and odd (high halfword) positions: a 16-bit compare reads the low half only
        unsigned long long E;
        asm("v_cmp_ne_u16_e64 %0, 0, %1" : "=s"(E) : "v"(xd[k]));
        const unsigned long long O = __ballot(xd[k] > 0xFFFFu);
        const unsigned long long any = E | O;
        if (any == 0ull) {   // uniform: 128 zeros
            cz += 128u;
            carry = 0;
            continue;
        }
        {   // zeros in front of the block's first non-zero, f = 2 t + (even position of pair t non-zero ? 0 : 1): a two-byte
            // varint (cz + f >= 128) shifts everything from here on by one byte
            uint32_t t, nf;
            asm("s_ff1_i32_b64 %0, %2\n\ts_bitcmp1_b64 %3, %0\n\ts_cselect_b32 %1, 0, 1" : "=&s"(t), "=s"(nf) : "s"(any), "s"(E) : "scc");
            W += cz + 2u * t + nf >= 128u ? 1u : 0u;
        }
        const unsigned long long S_lo = E & ~((O << 1) | carry), S_hi = O & ~E, S = S_lo | S_hi;
        carry = O >> 63;
        {   // zeros behind the block's last non-zero: 2 clz(any) + (odd position of the top pair non-zero ? 0 : 1)
            uint32_t fl, nb;
            asm("s_flbit_i32_b64 %0, %2\n\ts_sub_u32 %1, 63, %0\n\ts_bitcmp1_b64 %3, %1\n\ts_cselect_b32 %1, 0, 1"
                : "=&s"(fl), "=&s"(nb) : "s"(any), "s"(O) : "scc");
            cz = 2u * fl + nb;
        }
        const uint32_t nzb = __builtin_amdgcn_mbcnt_hi((uint32_t)(O >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)O, mbcnt64(E)));
        // a lane starts at most one run (S_lo needs E, S_hi needs ~E): one count over S serves both halves
        const uint32_t stb = mbcnt64(S);
        uint32_t slo, shi;
        asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(slo) : "s"(S_lo));
        asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(shi) : "s"(S_hi));
        // byte address of the lane's even-position value: 2 (items in front of it) + blob + W; the header of a run that
        // starts on the even position is one of those items
        const uint32_t a = ((nzb + stb + slo) << 1) + (blob + W + 2u * (nM + nS));
        // run table: slot 1 + (runs in front), entry (rank of the run's first non-zero) << 16 | position
        uint32_t ta = (stb << 2) + (tab + 4u * (nS + 1u));
        ta = ta < tmax ? ta : tmax;
        const uint32_t ent = (nzb << 16) + ((nM << 16) + (uint32_t)(128 * k)) + (pos_l + shi);
        {
            // bytes: low halfword from bits 0..7 of the dword and of the dword >> 8, high halfword from bits 16..23 of the same
            // two registers (ds_write_b8_d16_hi)
            const uint32_t x8 = xd[k] >> 8;
            unsigned long long sv;
            asm volatile("s_mov_b64 %0, exec\n\t"
                         "s_mov_b64 exec, %1\n\tds_write_b8 %4, %5\n\tds_write_b8 %4, %6 offset:1\n\t"
                         "s_mov_b64 exec, %2\n\tds_write_b8_d16_hi %4, %5 offset:2\n\tds_write_b8_d16_hi %4, %6 offset:3\n\t"
                         "s_mov_b64 exec, %3\n\tds_write_b32 %7, %8\n\t"
                         "s_mov_b64 exec, %0"
                         : "=&s"(sv)
                         : "s"(E), "s"(O), "s"(S), "v"(a), "v"(xd[k]), "v"(x8), "v"(ta), "v"(ent));
        }
        nM += (uint32_t)__builtin_popcountll(E) + (uint32_t)__builtin_popcountll(O);
        nS += (uint32_t)__builtin_popcountll(S);
    }
    const uint32_t N = nM, R = nS;
    if (N == 0u) {   // 1024 zeros: [varint 1024][0] = 80 08 00
        if (lane == 0) {
            lds_st8_at(blob, 0x80u, 0);
            lds_st8_at(blob, 0x08u, 1);
            lds_st8_at(blob, 0u, 2);
        }
        return 3u;
    }
    if (R > (uint32_t)(kRunTabEntries - 2)) return kSparseFallback;
    // sentinels: slot 0 = (position 0, rank 0); slot R + 1 = (rank N)
    if (lane == 0) {
        asm volatile("ds_write_b32 %0, %1" ::"v"(tab), "v"(0u) : "memory");
        asm volatile("ds_write_b32 %0, %1" ::"v"(tab + 4u * (R + 1u)), "v"(N << 16) : "memory");
    }
    // header pass: one lane per run
    uint32_t wdone = 0;   // wide records of the runs handled so far
    bool too_long = false;
    for (uint32_t j0 = 0; j0 < R; j0 += 64u) {
        const uint32_t j = j0 + (uint32_t)lane;
        const bool act = j < R;
        const uint32_t ja = tab + 4u * (act ? j : 0u);
        uint32_t e0, e1, e2;
        asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %3 offset:4\n\tds_read_b32 %2, %3 offset:8\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(e0), "=&v"(e1), "=&v"(e2)
                     : "v"(ja)
                     : "memory");
        const uint32_t p0 = e0 & 0xFFFFu, r0 = e0 >> 16, p1 = e1 & 0xFFFFu, r1 = e1 >> 16, r2 = e2 >> 16;
        const uint32_t cnt = r2 - r1;
        const uint32_t zrun = p1 - (p0 + (r1 - r0));
        const bool wide = act && zrun >= 128u;
        const unsigned long long wb = __ballot(wide);
        too_long |= __ballot(act && cnt > 255u) != 0ull;
        uint32_t wex = wdone;
        if (wb != 0ull) {   // uniform
            wex += mbcnt64(wb);
            wdone += (uint32_t)__builtin_popcountll(wb);
        }
        const uint32_t ha = ((r1 + j) << 1) + (blob + wex);
        if (act) {
            lds_st8_at(ha, wide ? ((zrun & 0x7Fu) | 0x80u) : zrun, 0);
            lds_st8_at(ha, wide ? (zrun >> 7) : cnt, 1);
            if (wide) lds_st8_at(ha, cnt, 2);
        }
    }
    if (too_long) return kSparseFallback;
    // closing record of a trailing zero run: [varint zeros][0]
    uint32_t tot = 2u * (N + R) + W;
    if (cz != 0u) {
        const uint32_t ta = blob + tot;
        if (lane == 0) {
            if (cz >= 128u) {
                lds_st8_at(ta, (cz & 0x7Fu) | 0x80u, 0);
                lds_st8_at(ta, cz >> 7, 1);
                lds_st8_at(ta, 0u, 2);
            } else {
                lds_st8_at(ta, cz, 0);
                lds_st8_at(ta, 0u, 1);
            }
        }
        tot += cz >= 128u ? 3u : 2u;
    }
    return tot;
}

// ------------------------------------------------------------------------------------------------ sparse RLE, item form
// serialize_sparse (encoder.rs:284-314) for the frames a q <= 0.8 encode is made of: a few dozen non-zeros in 1024 values.
// The block form above does per block of 128 POSITIONS what this form does per NON-ZERO:
//   1. compaction: per non-empty block two ballots and one rank (v_mbcnt) put every non-zero, in position order, into a
//      list in LDS as (position + 1) << 16 | value - one dword store per half, nothing else;
//   2. item pass: lane i takes items i and i + 64 and their predecessors (entry 0 of the list is a sentinel at "position
//      -1"). The zero run in front of an item is a subtraction; "starts a record" and "record with a two-byte varint" are
//      two compares whose ballots give, by v_mbcnt, the records and wide records up to the item, hence its byte offset
//      2 i + 2 records + wide; a record's count byte is the distance to the next record start, looked up through a
//      table "first item of record r" the starting lanes fill. No loop over positions, blocks or records.
// Up to kItemCap non-zeros (more: kSparseFallback, nothing usable written - the caller takes the block form); a run of
// non-zeros therefore never reaches 255 and no continuation records exist here.
// tabp: LDS byte address of kPackTabDwords dwords (item list, then the record table).
constexpr int kItemCap = 128;
constexpr int kItemTbl = 136;          // dword offset of the record table inside the area
constexpr int kPackTabDwords = 272;
__device__ __forceinline__ uint32_t sparse_item_pack(const int lane, const uint32_t (&xd)[8], const uint32_t blob, const uint32_t tabp) {
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    const uint32_t list1 = tabp + 4u;   // item r at list1 + 4 r
    const uint32_t pk0 = (uint32_t)(2 * lane + 1) << 16;   // (position + 1) << 16 of the lane's even position in block 0
    if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(tabp), "v"(0u) : "memory");   // the sentinel
    uint32_t nM = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        unsigned long long E;
        asm("v_cmp_ne_u16_e64 %0, 0, %1" : "=s"(E) : "v"(xd[k]));
        const unsigned long long O = __ballot(xd[k] > 0xFFFFu);
        if ((E | O) == 0ull) continue;   // uniform: 128 zeros
        // rank of the lane's even position: non-zeros so far + those of lower lanes (both halves); the odd position follows it.
        // Clamped to the list's capacity: a vector with more non-zeros than that is declined behind the loop (one exit
        // instead of a test per block), and what its late blocks wrote is never read.
        uint32_t re = __builtin_amdgcn_mbcnt_hi((uint32_t)(O >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)O,
                      __builtin_amdgcn_mbcnt_hi((uint32_t)(E >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)E, nM))));
        re = re < (uint32_t)kItemCap ? re : (uint32_t)kItemCap;
        const uint32_t a_e = (re << 2) + list1;
        uint32_t e4;
        asm("v_cndmask_b32_e64 %0, 0, 4, %1" : "=v"(e4) : "s"(E));
        const uint32_t a_o = a_e + e4;
        const uint32_t pk_e = pk0 + ((uint32_t)k << 23);
        uint32_t it_e;
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(it_e) : "s"(0xFFFFu), "v"(xd[k]), "v"(pk_e));
        const uint32_t it_o = __builtin_amdgcn_perm(pk_e + 0x10000u, xd[k], 0x07060302u);   // (pos + 2) << 16 | high halfword
        unsigned long long sv;
        asm volatile("s_mov_b64 %0, exec\n\t"
                     "s_mov_b64 exec, %1\n\tds_write_b32 %3, %4\n\t"
                     "s_mov_b64 exec, %2\n\tds_write_b32 %5, %6\n\t"
                     "s_mov_b64 exec, %0"
                     : "=&s"(sv)
                     : "s"(E), "s"(O), "v"(a_e), "v"(it_e), "v"(a_o), "v"(it_o)
                     : "memory");
        nM += (uint32_t)__builtin_popcountll(E) + (uint32_t)__builtin_popcountll(O);
    }
    const uint32_t N = nM;
    if (N > (uint32_t)kItemCap) return kSparseFallback;
    wave_sync();
    const uint32_t vaddr = tabp + 4u * (uint32_t)lane;
    const uint32_t tblp = tabp + 4u * (uint32_t)kItemTbl;
    const uint32_t lane2 = 2u * (uint32_t)lane;
    const uint32_t b3 = blob - 3u;
    const bool two = N > 64u;   // uniform
    // ---- first 64 items
    const uint32_t prevA = *reinterpret_cast<const lds_u32 *>((uintptr_t)vaddr), itA = *reinterpret_cast<const lds_u32 *>((uintptr_t)(vaddr + 4u));
    uint32_t diffA;   // (position of the item) - (position of its predecessor) = zero run + 1
    asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(diffA) : "v"(itA), "v"(prevA));
    const unsigned long long maskA = N >= 64u ? ~0ull : ((1ull << N) - 1ull);
    const unsigned long long SmA = (__ballot(diffA != 1u) | 1ull) & maskA;   // the walk's first record starts on item 0 whatever precedes it
    const unsigned long long WmA = __ballot(diffA >= 129u) & SmA;
    uint32_t ownA, ownWA;
    asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(ownA) : "s"(SmA));
    asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(ownWA) : "s"(WmA));
    const uint32_t rA = __builtin_amdgcn_mbcnt_hi((uint32_t)(SmA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)SmA, ownA));
    const uint32_t wA = __builtin_amdgcn_mbcnt_hi((uint32_t)(WmA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)WmA, ownWA));
    const uint32_t baseA = ((rA << 1) + wA) + (lane2 + b3);
    const uint32_t RA = (uint32_t)__builtin_popcountll(SmA), WA = (uint32_t)__builtin_popcountll(WmA);
    {
        unsigned long long sv;
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tds_write_b32 %2, %3\n\ts_mov_b64 exec, %0"
                     : "=&s"(sv) : "s"(SmA), "v"((rA << 2) + tblp), "v"((uint32_t)lane) : "memory");
    }
    // ---- items 64..127
    uint32_t itB = 0, diffB = 1, rB = 0, baseB = 0, RB = 0, WB = 0;
    unsigned long long maskB = 0, SmB = 0, WmB = 0;
    if (two) {
        const uint32_t prevB = *reinterpret_cast<const lds_u32 *>((uintptr_t)(vaddr + 256u));
        itB = *reinterpret_cast<const lds_u32 *>((uintptr_t)(vaddr + 260u));
        asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "=v"(diffB) : "v"(itB), "v"(prevB));
        maskB = N >= 128u ? ~0ull : ((1ull << (N - 64u)) - 1ull);
        SmB = __ballot(diffB != 1u) & maskB;
        WmB = __ballot(diffB >= 129u) & SmB;
        uint32_t ownB, ownWB;
        asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(ownB) : "s"(SmB));
        asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(ownWB) : "s"(WmB));
        rB = __builtin_amdgcn_mbcnt_hi((uint32_t)(SmB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)SmB, ownB + RA));
        const uint32_t wB = __builtin_amdgcn_mbcnt_hi((uint32_t)(WmB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)WmB, ownWB + WA));
        baseB = ((rB << 1) + wB) + (lane2 + (b3 + 128u));
        RB = (uint32_t)__builtin_popcountll(SmB), WB = (uint32_t)__builtin_popcountll(WmB);
        unsigned long long sv;
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\tds_write_b32 %2, %3\n\ts_mov_b64 exec, %0"
                     : "=&s"(sv) : "s"(SmB), "v"((rB << 2) + tblp), "v"((uint32_t)lane + 64u) : "memory");
    }
    const uint32_t R = RA + RB, Wd = WA + WB;
    if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(tblp + 4u * (R + 1u)), "v"(N) : "memory");   // "the record after the last" starts at item N
    wave_sync();
    // ---- bytes: values at base + 3, + 4; a record's header in front of its first value: base + 1 = zero run (or the
    // varint's second byte behind base + 0 = its first), base + 2 = count
    {
        const uint32_t nxt = *reinterpret_cast<const lds_u32 *>((uintptr_t)((rA << 2) + tblp + 4u));
        const uint32_t cnt = nxt - (uint32_t)lane;
        const uint32_t gap = diffA - 1u;
        const uint32_t h0 = (gap & 0x7Fu) | 0x80u;
        uint32_t h1;
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(h1) : "v"(gap), "v"(gap >> 7), "s"(WmA));
        const uint32_t v8 = itA >> 8;
        unsigned long long sv;
        asm volatile("s_mov_b64 %0, exec\n\t"
                     "s_mov_b64 exec, %1\n\tds_write_b8 %4, %5 offset:3\n\tds_write_b8 %4, %6 offset:4\n\t"
                     "s_mov_b64 exec, %2\n\tds_write_b8 %4, %8 offset:1\n\tds_write_b8 %4, %9 offset:2\n\t"
                     "s_mov_b64 exec, %3\n\tds_write_b8 %4, %7\n\t"
                     "s_mov_b64 exec, %0"
                     : "=&s"(sv)
                     : "s"(maskA), "s"(SmA), "s"(WmA), "v"(baseA), "v"(itA), "v"(v8), "v"(h0), "v"(h1), "v"(cnt)
                     : "memory");
    }
    uint32_t last = itA;
    if (two) {
        const uint32_t nxt = *reinterpret_cast<const lds_u32 *>((uintptr_t)((rB << 2) + tblp + 4u));
        const uint32_t cnt = nxt - ((uint32_t)lane + 64u);
        const uint32_t gap = diffB - 1u;
        const uint32_t h0 = (gap & 0x7Fu) | 0x80u;
        uint32_t h1;
        asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(h1) : "v"(gap), "v"(gap >> 7), "s"(WmB));
        const uint32_t v8 = itB >> 8;
        unsigned long long sv;
        asm volatile("s_mov_b64 %0, exec\n\t"
                     "s_mov_b64 exec, %1\n\tds_write_b8 %4, %5 offset:3\n\tds_write_b8 %4, %6 offset:4\n\t"
                     "s_mov_b64 exec, %2\n\tds_write_b8 %4, %8 offset:1\n\tds_write_b8 %4, %9 offset:2\n\t"
                     "s_mov_b64 exec, %3\n\tds_write_b8 %4, %7\n\t"
                     "s_mov_b64 exec, %0"
                     : "=&s"(sv)
                     : "s"(maskB), "s"(SmB), "s"(WmB), "v"(baseB), "v"(itB), "v"(v8), "v"(h0), "v"(h1), "v"(cnt)
                     : "memory");
        last = itB;
    }
    // closing record of the zeros behind the last non-zero: [varint zeros][0] (all 1024 when there is none: 80 08 00)
    uint32_t tot = 2u * (N + R) + Wd;
    const uint32_t pf_last = N ? ((uint32_t)__builtin_amdgcn_readlane((int)last, (int)((N - 1u) & 63u)) >> 16) : 0u;
    const uint32_t cz = 1024u - pf_last;
    {   // bytes [cz][0] or [cz & 0x7f | 0x80][cz >> 7][0] as a word, lane i < its length stores byte i
        const uint32_t wide = cz >= 128u ? 1u : 0u;
        const uint32_t word = wide ? (((cz & 0x7Fu) | 0x80u) | ((cz >> 7) << 8)) : cz;
        const uint32_t len = cz ? 2u + wide : 0u;
        if ((uint32_t)lane < len) lds_st8_at(blob + tot + (uint32_t)lane, word >> (8u * (uint32_t)lane), 0);
        tot += len;
    }
    return tot;
}

// single channel
__device__ __forceinline__ void sparse_emit(const int lane, const int (&q)[16], const SparsePlan &P, uint8_t *dst,
                                            uint32_t trash_off) {
    const int (&q1)[1][16] = reinterpret_cast<const int (&)[1][16]>(q);
    const SparsePlan (&P1)[1] = reinterpret_cast<const SparsePlan (&)[1]>(P);
    uint8_t *const d1[1] = {dst};
    const uint32_t t1[1] = {trash_off};
    sparse_emit_n<1>(lane, q1, P1, d1, t1);
}

}  // namespace flo
