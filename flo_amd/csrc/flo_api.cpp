// flo_api.cpp — C ABI of libflo_hip.so (see include/flo_hip.h): context, device-resident batches, .flo assembly.
// Host code only; the kernels live in lossy_kernels.hip / lossless_kernels.hip.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <sys/mman.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/flo_hip.h"
#include "container.hpp"
#include "analysis_kernels.hpp"
#include "container_kernels.hpp"
#include "decode_kernels.hpp"
#include "devpool.hpp"
#include "dist_engine.hpp"
#include "stager.hpp"
#include "lossless_kernels.hpp"
#include "lossy_kernels.hpp"
#include "tables.hpp"

using namespace flo;

// ------------------------------------------------------------------------------------------------ context
struct TableSet {
    LossyTablesHost host;
    void *blob = nullptr;  // one device allocation holding every table
    LossyDevTables dev{};
    const float *dev_window = nullptr;  // [2048], decode side
};

struct ProfRec {
    std::string name;
    hipEvent_t a, b;
};
struct ProfSum {   // launches already read out (their events are destroyed)
    double ms = 0;
    uint64_t n = 0;
};

struct flo_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    std::vector<TableSet *> tables;
    bool profile = false;
    std::vector<ProfRec> prof;              // bracketed launches not yet read out
    std::map<std::string, ProfSum> prof_sum;
    int force_path = 0;
    hipDeviceProp_t prop{};
    Stager *stager = nullptr;   // pinned staging ring + copy threads of the host-buffer entry points (made on first use)
    hipStream_t up_stream = nullptr, down_stream = nullptr;   // uploads / downloads of flo_encode_batch's pipeline
    int reserve_cus = -1;   // compute units the persistent chain kernel leaves free (-1: not set; see flo_ctx_reserve_cus)
    AnalysisSide an_side;   // side streams of the analysis (made on first use)
    bool an_side_ready = false;
};
static const int kDefaultReservedCus = 8;

static thread_local std::string g_create_err;

static int fail(flo_ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg;
    return code;
}
#define HIPCHK(ctx, expr)                                                                               \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(ctx, FLO_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    } while (0)

// A decoded file's PCM goes to the caller in fresh memory (flo_free = free). A fresh 60 MB of 4 KB pages is 15 000 page
// faults under the copy that fills it - 30 ms for a 3-minute file whose kernels take 1 ms - so large results are asked for
// in transparent huge pages (2 MB alignment + MADV_HUGEPAGE: a hint; where the host does not honour it nothing changes).
static void *alloc_result(size_t bytes) {
    if (bytes < ((size_t)4 << 20)) return malloc(bytes ? bytes : 1);
    void *p = nullptr;
    if (posix_memalign(&p, (size_t)2 << 20, bytes) != 0) return malloc(bytes);
    madvise(p, bytes, MADV_HUGEPAGE);
    return p;
}

extern "C" const char *flo_last_create_error(void) { return g_create_err.c_str(); }
extern "C" const char *flo_last_error(const flo_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }
extern "C" void flo_free(void *p) { free(p); }

extern "C" int flo_ctx_create(int device, flo_ctx **out) {
    if (!out) return FLO_ERR_ARG;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        g_create_err = std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "count = 0") +
                       " (libflo_hip has no CPU fallback)";
        return FLO_ERR_DEVICE;
    }
    if (device < 0 || device >= n) {
        g_create_err = "device index out of range";
        return FLO_ERR_ARG;
    }
    flo_ctx *c = new flo_ctx();
    c->device = device;
    int prev_dev = -1;
    hipGetDevice(&prev_dev);   // the calling thread's current device is left as it was found
    auto restore = [&] {
        if (prev_dev >= 0 && prev_dev != device) hipSetDevice(prev_dev);
    };
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipGetDeviceProperties(&c->prop, device)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
        g_create_err = std::string("device init failed: ") + hipGetErrorString(e);
        delete c;
        restore();
        return FLO_ERR_DEVICE;
    }
    if (std::string(c->prop.gcnArchName).find("gfx950") == std::string::npos) {
        g_create_err = std::string("device is ") + c->prop.gcnArchName + ", this library carries gfx950 code only";
        hipStreamDestroy(c->stream);
        delete c;
        restore();
        return FLO_ERR_DEVICE;
    }
    {
        std::string serr;
        c->stager = stager_create(serr);   // light: pinned buffers and copy threads appear when first needed
    }
    if (const char *e2 = getenv("FLO_RESERVE_CUS")) {   // compute units left to other kernels (RCCL's, at N > 1)
        const int v = atoi(e2);
        if (v >= 0 && v < c->prop.multiProcessorCount) c->reserve_cus = v;
    }
    restore();
    *out = c;
    return FLO_OK;
}

extern "C" void flo_ctx_destroy(flo_ctx *c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    for (auto *t : c->tables) {
        if (t->blob) hipFree(t->blob);
        delete t;
    }
    for (auto &r : c->prof) {
        hipEventDestroy(r.a);
        hipEventDestroy(r.b);
    }
    hipStreamDestroy(c->stream);
    if (c->stager) stager_destroy(c->stager);
    if (c->up_stream) hipStreamDestroy(c->up_stream);
    if (c->down_stream) hipStreamDestroy(c->down_stream);
    if (c->an_side.fork) hipEventDestroy(c->an_side.fork);
    for (int i = 0; i < 3; i++) {
        if (c->an_side.join[i]) hipEventDestroy(c->an_side.join[i]);
        if (c->an_side.st[i]) hipStreamDestroy(c->an_side.st[i]);
    }
    delete c;
}

extern "C" int flo_ctx_device_info(const flo_ctx *c, char *name, size_t cap, int *cus, uint64_t *hbm) {
    if (!c) return FLO_ERR_ARG;
    if (name && cap) snprintf(name, cap, "%s (%s)", c->prop.name, c->prop.gcnArchName);
    if (cus) *cus = c->prop.multiProcessorCount;
    if (hbm) *hbm = (uint64_t)c->prop.totalGlobalMem;
    return FLO_OK;
}
extern "C" void *flo_ctx_stream(flo_ctx *c) { return c ? (void *)c->stream : nullptr; }
extern "C" int flo_ctx_force_path(flo_ctx *c, int which) {
    if (!c || which < 0 || which > 5) return FLO_ERR_ARG;
    c->force_path = which;
    return FLO_OK;
}

// ---- profiling hooks -----------------------------------------------------------------------------------
extern "C" int flo_ctx_profile_enable(flo_ctx *c, int on) {
    if (!c) return FLO_ERR_ARG;
    c->profile = on != 0;
    return FLO_OK;
}
extern "C" int flo_ctx_profile_reset(flo_ctx *c) {
    if (!c) return FLO_ERR_ARG;
    for (auto &r : c->prof) {
        hipEventDestroy(r.a);
        hipEventDestroy(r.b);
    }
    c->prof.clear();
    c->prof_sum.clear();
    return FLO_OK;
}
// Fold every finished bracket into the per-kernel sums and destroy its events (a long profiled run keeps at most the
// launches since the last drain alive).
static int profile_drain(flo_ctx *c, bool wait) {
    size_t keep = 0;
    for (size_t i = 0; i < c->prof.size(); i++) {
        ProfRec &r = c->prof[i];
        if (!wait && hipEventQuery(r.b) != hipSuccess) {
            if (keep != i) c->prof[keep] = r;   // not finished yet: stays queued
            keep++;
            continue;
        }
        float ms = 0;
        HIPCHK(c, hipEventSynchronize(r.b));
        HIPCHK(c, hipEventElapsedTime(&ms, r.a, r.b));
        ProfSum &ps = c->prof_sum[r.name];
        ps.ms += ms;
        ps.n++;
        hipEventDestroy(r.a);
        hipEventDestroy(r.b);
    }
    c->prof.resize(keep);
    return FLO_OK;
}
extern "C" int flo_ctx_profile_query(flo_ctx *c, const char *kernel, double *total_ms, uint64_t *launches) {
    if (!c || !kernel) return FLO_ERR_ARG;
    int rc = profile_drain(c, true);
    if (rc != FLO_OK) return rc;
    auto it = c->prof_sum.find(kernel);
    if (total_ms) *total_ms = it == c->prof_sum.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == c->prof_sum.end() ? 0 : it->second.n;
    return FLO_OK;
}

template <typename F>
static int timed_launch(flo_ctx *c, const char *name, F &&launch) {
    if (!c->profile) {
        int rc = launch();
        return rc == 0 ? FLO_OK : fail(c, FLO_ERR_DEVICE, std::string("launch ") + name + " failed: " +
                                                              hipGetErrorString((hipError_t)(rc > 0 ? rc : 1)));
    }
    if (c->prof.size() >= 256) {   // bound the queue of live events
        int drc = profile_drain(c, false);
        if (drc != FLO_OK) return drc;
    }
    ProfRec r;
    r.name = name;
    HIPCHK(c, hipEventCreate(&r.a));
    HIPCHK(c, hipEventCreate(&r.b));
    HIPCHK(c, hipEventRecord(r.a, c->stream));
    int rc = launch();
    HIPCHK(c, hipEventRecord(r.b, c->stream));
    c->prof.push_back(r);
    return rc == 0 ? FLO_OK : fail(c, FLO_ERR_DEVICE, std::string("launch ") + name + " failed");
}

// ---- constant tables -----------------------------------------------------------------------------------
static int get_tables(flo_ctx *c, uint32_t sr, float quality, TableSet **out) {
    float q = quality < 0.f ? 0.f : (quality > 1.f ? 1.f : quality);
    if (quality != quality) q = 0.f;  // NaN clamps to NaN in Rust; the threshold formula then yields NaN -> treat as 0
    for (auto *t : c->tables)
        if (t->host.sample_rate == sr && t->host.quality == q) {
            *out = t;
            return FLO_OK;
        }
    TableSet *t = new TableSet();
    build_lossy_tables(sr, q, t->host);
    const LossyTablesHost &h = t->host;
    struct Part {
        const void *src;
        size_t bytes;
        size_t off;
    };
    std::vector<Part> parts;
    size_t total = 0;
    auto add = [&](const void *p, size_t b) {
        total = (total + 255) & ~(size_t)255;
        parts.push_back({p, b, total});
        total += b;
        return parts.size() - 1;
    };
    size_t i_ext = add(h.pack_ext.data(), h.pack_ext.size() * 4);
    size_t i_pack = add(h.pack.data(), h.pack.size() * 4), i_athdb = add(h.ath_db.data(), h.ath_db.size() * 4),
           i_band = add(h.band.data(), h.band.size()), i_bc = add(h.band_count.data(), h.band_count.size() * 4),
           i_s10 = add(h.s10d.data(), h.s10d.size() * 4), i_lb = add(h.lane_bnd.data(), h.lane_bnd.size() * 4),
           i_ls = add(h.lane_slot0.data(), h.lane_slot0.size() * 4), i_bs = add(h.band_slot0.data(), h.band_slot0.size() * 4),
           i_win = add(h.window.data(), h.window.size() * 4);
    hipError_t e = hipMalloc(&t->blob, total);
    if (e != hipSuccess) {
        delete t;
        return fail(c, FLO_ERR_NOMEM, std::string("hipMalloc tables: ") + hipGetErrorString(e));
    }
    std::vector<uint8_t> stage(total, 0);
    for (auto &p : parts) memcpy(stage.data() + p.off, p.src, p.bytes);
    e = hipMemcpy(t->blob, stage.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        hipFree(t->blob);
        delete t;
        return fail(c, FLO_ERR_DEVICE, std::string("hipMemcpy tables: ") + hipGetErrorString(e));
    }
    auto P = [&](size_t i) { return (const char *)t->blob + parts[i].off; };
    t->dev.pack = (const float4 *)P(i_pack);
    t->dev.pack_g = t->dev.pack;
    t->dev.pack_ext = (const float4 *)P(i_ext);
    t->dev.ath_db = (const float *)P(i_athdb);
    t->dev.band = (const uint8_t *)P(i_band);
    t->dev.band_count = (const float *)P(i_bc);
    t->dev.s10d = (const float *)P(i_s10);
    t->dev.lane_bnd = (const uint32_t *)P(i_lb);
    t->dev.lane_slot0 = (const uint32_t *)P(i_ls);
    t->dev.band_slot0 = (const uint32_t *)P(i_bs);
    t->dev_window = (const float *)P(i_win);
    t->dev.max_band_slots = h.max_band_slots;
    t->dev.dirty = h.dirty;
    t->dev.smr_thr = h.smr_threshold;
    t->dev.q_transparent = h.q_transparent;
    if (h.n_slots > kSlotCap || h.max_band_slots > 64) {   // (cannot happen: 64 lanes + 24 band edges, 64 lanes per band)
        hipFree(t->blob);
        delete t;
        return fail(c, FLO_ERR_ARG, "band segment table exceeds capacity");
    }
    c->tables.push_back(t);
    *out = t;
    return FLO_OK;
}

// ------------------------------------------------------------------------------------------------ batch
struct flo_batch {
    flo_ctx *ctx = nullptr;
    int mode = 0;
    size_t n_clips = 0;
    uint32_t sr = 0;
    uint8_t ch = 0;
    float qol = 0;
    uint8_t bit_depth = 16;
    TableSet *ts = nullptr;
    // plan (host)
    std::vector<uint64_t> n_il, clip_off, clip_nsf, clip_frame0, out_off, out_cap;
    std::vector<uint64_t> file_off, h_file_bytes;   // finished file = [file_off, file_off + 74 + 20 frames + DATA)
    std::vector<uint32_t> hops;
    uint64_t total_frames = 0, total_floats = 0, out_bytes = 0;
    // device
    float *d_pcm = nullptr;
    uint64_t *d_plan = nullptr;  // clip_off | clip_nsf | clip_frame0 | out_off
    uint32_t *d_hops = nullptr;
    uint8_t *d_out = nullptr;
    uint32_t *d_frame_size = nullptr;
    uint64_t *d_clip_bytes = nullptr;
    uint32_t *d_crc = nullptr, *d_part = nullptr, *d_next = nullptr;
    float *d_bmax = nullptr;   // band maxima of every frame (frame-parallel form: pass 1 -> pass 2)
    void *d_coef = nullptr;    // ... and, for a few long stereo clips, every frame's coefficients (8 KB per frame)
    float *d_at = nullptr, *d_sprev = nullptr;
    uint8_t *d_slots = nullptr;
    uint64_t *d_frame_off = nullptr;
    // analysis buffers (optional)
    float *d_dbg_coeffs = nullptr;
    short *d_dbg_q = nullptr;
    unsigned short *d_dbg_sfw = nullptr;
    const float *d_in_coeffs = nullptr;
    uint64_t *d_pack_plan = nullptr;
    uint64_t *pin_plan = nullptr;        // pinned: clip plan (4 n) | hops (n u32) | pack plan (3 n): read by asynchronous copies
    hipEvent_t ev_pack_plan = nullptr;
    unsigned long long *d_stamps = nullptr;
    int exact = 0;
    // results (host, valid after sync)
    bool encoded = false, synced = false, encode_failed = false;
    std::vector<uint64_t> h_clip_bytes;
    std::vector<uint32_t> h_frame_size;
    // lossless
    LosslessPlan *ll = nullptr;
};

static int auto_form(const flo_batch *b);
static int alloc_frame_scratch(flo_batch *b);
static size_t lossy_max_frame_bytes(int ch) { return 12 + 50 * (size_t)ch + (size_t)ch * (4 + 2064); }

extern "C" void flo_batch_destroy(flo_batch *b) {
    if (!b) return;
    hipSetDevice(b->ctx->device);
    hipStreamSynchronize(b->ctx->stream);
    void *ptrs[] = {b->d_pcm, b->d_plan, b->d_hops, b->d_out, b->d_frame_size, b->d_clip_bytes, b->d_crc, b->d_part, b->d_at,
                    b->d_sprev, b->d_slots, b->d_frame_off, b->d_dbg_coeffs, b->d_dbg_q, b->d_dbg_sfw, b->d_pack_plan, b->d_next, b->d_bmax, b->d_coef};
    for (void *p : ptrs)
        if (p) pool_free(p);
    if (b->ev_pack_plan) hipEventDestroy(b->ev_pack_plan);
    if (b->pin_plan) stager_pinned_put(b->ctx->stager, b->pin_plan);
    if (b->ll) lossless_plan_destroy(b->ll);
    delete b;
}

extern "C" int flo_batch_create(flo_ctx *c, int mode, size_t n_clips, const size_t *n_interleaved, uint32_t sr,
                                uint8_t ch, float qol, flo_batch **out) {
    if (!c || !out || (n_clips && !n_interleaved)) return FLO_ERR_ARG;
    *out = nullptr;
    if (ch == 0 || sr == 0) return fail(c, FLO_ERR_ARG, "sample_rate and channels must be non-zero");
    if (mode != FLO_MODE_LOSSY && mode != FLO_MODE_LOSSLESS) return fail(c, FLO_ERR_ARG, "unknown mode");
    if (mode == FLO_MODE_LOSSY && ch > kMaxLossyChannels)
        return fail(c, FLO_ERR_ARG, "lossy encode on device supports 1 to 8 channels");
    HIPCHK(c, hipSetDevice(c->device));
    flo_batch *b = new flo_batch();
    b->ctx = c;
    b->mode = mode;
    b->n_clips = n_clips;
    b->sr = sr;
    b->ch = ch;
    b->qol = qol;
    b->n_il.assign(n_interleaved, n_interleaved + n_clips);
    b->clip_off.resize(n_clips);
    b->clip_nsf.resize(n_clips);
    uint64_t off = 0;
    for (size_t i = 0; i < n_clips; i++) {
        b->clip_off[i] = off;
        b->clip_nsf[i] = n_interleaved[i] / ch;  // trailing partial sample-frame is dropped (encoder.rs:174)
        uint64_t alloc = n_interleaved[i];
        if (mode == FLO_MODE_LOSSY) {
            // every frame's 1024 new sample-frames exist in memory: the clip is followed by zeros up to
            // hops * 1024 sample-frames (the reference pads the same way, encoder.rs:177-185), so the chain kernels
            // load whole half-frames without bounds checks; one more half-frame lets them prefetch unconditionally
            // behind the last frame (the values are never used)
            const uint64_t hops = (b->clip_nsf[i] + 1024 + 1023) / 1024;
            alloc = (hops + 1) * 1024 * ch;
            if (alloc < n_interleaved[i]) alloc = n_interleaved[i];
        }
        off += (alloc + 3) & ~(uint64_t)3;
    }
    b->total_floats = off;
    int rc = FLO_OK;
    auto bail = [&](int code) {
        flo_batch_destroy(b);
        return code;
    };
#define BCHK(expr)                                                                                  \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            fail(c, e_ == hipErrorOutOfMemory ? FLO_ERR_NOMEM : FLO_ERR_DEVICE,                     \
                 std::string(#expr) + ": " + hipGetErrorString(e_));                                \
            return bail(e_ == hipErrorOutOfMemory ? FLO_ERR_NOMEM : FLO_ERR_DEVICE);                \
        }                                                                                           \
    } while (0)
    BCHK(pool_alloc(&b->d_pcm, (b->total_floats + 4) * sizeof(float)));
    if (mode == FLO_MODE_LOSSY) {   // the zero padding behind every clip (the clips themselves are written by the caller)
        if (n_clips <= 8) {
            for (size_t i = 0; i < n_clips; i++) {
                const uint64_t used = b->clip_nsf[i] * ch;
                const uint64_t end = (i + 1 < n_clips ? b->clip_off[i + 1] : b->total_floats) + (i + 1 < n_clips ? 0 : 4);
                BCHK(hipMemsetAsync(b->d_pcm + b->clip_off[i] + used, 0, (end - b->clip_off[i] - used) * sizeof(float), c->stream));
            }
        } else {
            BCHK(hipMemsetAsync(b->d_pcm, 0, (b->total_floats + 4) * sizeof(float), c->stream));
        }
    }
    if (mode == FLO_MODE_LOSSY) {
        rc = get_tables(c, sr, qol, &b->ts);
        if (rc != FLO_OK) return bail(rc);
        b->hops.resize(n_clips);
        b->clip_frame0.resize(n_clips);
        b->out_off.resize(n_clips);
        b->out_cap.resize(n_clips);
        b->file_off.resize(n_clips);
        uint64_t f = 0, o = 0;
        const size_t mfb = lossy_max_frame_bytes(ch);
        for (size_t i = 0; i < n_clips; i++) {
            uint64_t h = (b->clip_nsf[i] + 1024 + 1023) / 1024;  // encoder.rs:177-179
            b->hops[i] = (uint32_t)h;
            b->clip_frame0[i] = f;
            f += h;
            // header + TOC of the finished file sit right in front of the (16-byte aligned) DATA chunk
            const uint64_t head = 74 + 20 * h;
            o += (head + 15) & ~(uint64_t)15;
            b->out_off[i] = o;
            b->file_off[i] = o - head;
            b->out_cap[i] = ((h * mfb + 64) + 15) & ~(uint64_t)15;
            o += b->out_cap[i];
        }
        b->total_frames = f;
        b->out_bytes = o;
        std::vector<uint64_t> plan(4 * n_clips);
        for (size_t i = 0; i < n_clips; i++) {
            plan[i] = b->clip_off[i];
            plan[n_clips + i] = b->clip_nsf[i];
            plan[2 * n_clips + i] = b->clip_frame0[i];
            plan[3 * n_clips + i] = b->out_off[i];
        }
        BCHK(pool_alloc(&b->d_plan, (plan.size() + 1) * 8));
        BCHK(pool_alloc(&b->d_hops, (n_clips + 1) * 4));
        BCHK(pool_alloc(&b->d_out, b->out_bytes + 64));
        BCHK(pool_alloc(&b->d_frame_size, (b->total_frames + 1) * 4));
        BCHK(pool_alloc(&b->d_clip_bytes, (n_clips + 1) * 8));
        BCHK(pool_alloc(&b->d_crc, (n_clips + 1) * 4));
        BCHK(pool_alloc(&b->d_next, 16));
        BCHK(pool_alloc(&b->d_part, (n_clips * finish_parts_for(n_clips) + 1) * 4));
        if (n_clips) {
            // from pinned memory on the context's stream, in front of everything that will use them: a synchronous (or
            // pageable "asynchronous") copy would wait for whatever this context's other batches have in flight
            std::string perr;
            b->pin_plan = (uint64_t *)stager_pinned_get(c->stager, (8 * n_clips + 8) * 8, perr);
            if (!b->pin_plan) {
                fail(c, FLO_ERR_NOMEM, perr);
                return bail(FLO_ERR_NOMEM);
            }
            memcpy(b->pin_plan, plan.data(), plan.size() * 8);
            memcpy(b->pin_plan + 4 * n_clips, b->hops.data(), n_clips * 4);
            BCHK(hipMemcpyAsync(b->d_plan, b->pin_plan, plan.size() * 8, hipMemcpyHostToDevice, c->stream));
            BCHK(hipMemcpyAsync(b->d_hops, b->pin_plan + 4 * n_clips, n_clips * 4, hipMemcpyHostToDevice, c->stream));
        }
        if (auto_form(b) == 2 && (rc = alloc_frame_scratch(b)) != FLO_OK) return bail(rc);
    } else {
        uint8_t level = qol < 0 ? 0 : (qol > 9 ? 9 : (uint8_t)qol);  // with_compression: level.min(9)
        b->qol = level;
        std::string err;
        b->ll = lossless_plan_create(b->n_il, b->clip_off, sr, ch, level, b->d_pcm, err);
        if (!b->ll) {
            fail(c, FLO_ERR_NOMEM, "lossless plan: " + err);
            return bail(FLO_ERR_NOMEM);
        }
    }
#undef BCHK
    *out = b;
    return FLO_OK;
}

extern "C" float *flo_batch_clip_device_ptr(flo_batch *b, size_t clip) {
    if (!b || clip >= b->n_clips) return nullptr;
    return b->d_pcm + b->clip_off[clip];
}

extern "C" int flo_batch_upload(flo_batch *b, size_t clip, const float *pcm) {
    if (!b || clip >= b->n_clips || (!pcm && b->n_il[clip])) return FLO_ERR_ARG;
    flo_ctx *c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    // a trailing partial sample-frame is not part of the clip (encoder.rs:174): it must not land in the zero padding
    const uint64_t n_copy = b->mode == FLO_MODE_LOSSY ? b->clip_nsf[clip] * b->ch : b->n_il[clip];
    if (n_copy)
        HIPCHK(c, hipMemcpyAsync(b->d_pcm + b->clip_off[clip], pcm, n_copy * sizeof(float), hipMemcpyHostToDevice, c->stream));
    b->encoded = b->synced = b->encode_failed = false;
    return FLO_OK;
}

extern "C" int flo_batch_fill_synthetic(flo_batch *b, uint32_t seed, uint64_t clip_id0) {
    if (!b) return FLO_ERR_ARG;
    flo_ctx *c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (!b->n_clips) return FLO_OK;
    // plan arrays needed on device: clip_off, clip_nsf
    uint64_t *d_off = nullptr;
    std::vector<uint64_t> tmp(2 * b->n_clips);
    for (size_t i = 0; i < b->n_clips; i++) {
        tmp[i] = b->clip_off[i];
        tmp[b->n_clips + i] = b->clip_nsf[i];
    }
    HIPCHK(c, hipMalloc(&d_off, tmp.size() * 8));
    hipError_t e = hipMemcpyAsync(d_off, tmp.data(), tmp.size() * 8, hipMemcpyHostToDevice, c->stream);
    int rc = 0;
    if (e == hipSuccess)
        rc = launch_synth_fill(b->d_pcm, (const unsigned long long *)d_off, (const unsigned long long *)d_off + b->n_clips, (int)b->n_clips, b->ch, seed, clip_id0, c->stream);
    hipStreamSynchronize(c->stream);
    hipFree(d_off);
    if (e != hipSuccess || rc != 0) return fail(c, FLO_ERR_DEVICE, "synthetic fill failed");
    b->encoded = b->synced = b->encode_failed = false;
    return FLO_OK;
}

// auto selection of the lossy kernel form (flo_batch_encode with which = 0 and nothing forced)
static int auto_form(const flo_batch *b) {
    if (b->ch > 2) return 2;
    return (b->n_clips * b->ch >= 512) ? (b->ch == 2 ? 5 : 1) : 2;
}
// scratch of the frame-parallel form: per-frame masking levels, fixed-size frame slots, frame offsets
static int alloc_frame_scratch(flo_batch *b) {
    flo_ctx *c = b->ctx;
    if (b->d_at || !b->total_frames) return FLO_OK;
    const size_t n = (size_t)b->total_frames * b->ch * 32 * sizeof(float);
    HIPCHK(c, pool_alloc(&b->d_at, n));
    HIPCHK(c, pool_alloc(&b->d_sprev, n));
    HIPCHK(c, pool_alloc(&b->d_bmax, n));
    // one 3-minute clip: 63 MB that never leave the memory-side cache; a batch of thousands of clips forced into this form
    // transforms twice instead
    if (b->ch == 2 && (size_t)b->total_frames * 8192 <= ((size_t)256 << 20) && !getenv("FLO_NO_COEF_HANDOVER")) HIPCHK(c, pool_alloc(&b->d_coef, (size_t)b->total_frames * 8192));
    HIPCHK(c, pool_alloc(&b->d_slots, (size_t)b->total_frames * lossy_slot_bytes(b->ch)));
    HIPCHK(c, pool_alloc(&b->d_frame_off, (size_t)(b->total_frames + 1) * 8));
    return FLO_OK;
}

static LossyArgs make_args(flo_batch *b) {
    LossyArgs A{};
    A.T = b->ts->dev;
    A.pcm = b->d_pcm;
    const unsigned long long *plan = (const unsigned long long *)b->d_plan;
    A.clip_off = plan;
    A.clip_nsf = plan + b->n_clips;
    A.clip_frame0 = plan + 2 * b->n_clips;
    A.out_off = plan + 3 * b->n_clips;
    A.clip_hops = b->d_hops;
    A.nch = b->ch;
    A.n_clips = (int)b->n_clips;
    A.total_frames = b->total_frames;
    A.max_hops = 0;
    for (auto h : b->hops) A.max_hops = h > A.max_hops ? h : A.max_hops;
    A.out = b->d_out;
    A.frame_size = b->d_frame_size;
    A.clip_bytes = (unsigned long long *)b->d_clip_bytes;
    A.a_t = b->d_at;
    A.bmax_t = b->d_bmax;
    A.coef_t = (float4 *)b->d_coef;
    A.s_prev_out = b->d_sprev;
    A.s_prev = b->d_sprev;
    A.slots = b->d_slots;
    A.slot_bytes = lossy_slot_bytes(b->ch);
    A.frame_off = (unsigned long long *)b->d_frame_off;
    A.dbg_coeffs = b->d_dbg_coeffs;
    A.dbg_q = b->d_dbg_q;
    A.dbg_sfw = b->d_dbg_sfw;
    A.in_coeffs = b->d_in_coeffs;
    A.exact = b->exact;
    A.dbg_stamps = b->d_stamps;
    A.next_clip = b->d_next;
    A.n_cus = b->ctx->prop.multiProcessorCount - (b->ctx->reserve_cus > 0 ? b->ctx->reserve_cus : 0);
    return A;
}

static int batch_encode_launch(flo_batch *b, int which);
extern "C" int flo_batch_encode(flo_batch *b, int which) {
    if (!b) return FLO_ERR_ARG;
    flo_ctx *c = b->ctx;
    if (which < 0 || which > 5) return fail(c, FLO_ERR_ARG, "unknown kernel form");
    HIPCHK(c, hipSetDevice(c->device));
    // "encoded" is set only once every launch of this call has been accepted: after a failed encode, sync / fetch /
    // pack refuse with FLO_ERR_STATE instead of handing out stale or partial bytes
    b->encoded = false;
    b->synced = false;
    int erc = batch_encode_launch(b, which);
    b->encoded = erc == FLO_OK;
    b->encode_failed = erc != FLO_OK;
    return erc;
}

static int batch_encode_launch(flo_batch *b, int which) {
    flo_ctx *c = b->ctx;
    if (!b->n_clips) return FLO_OK;
    if (b->mode == FLO_MODE_LOSSLESS) {
        std::string err;
        int rc = lossless_encode_launch(b->ll, c->stream, c->profile ? 1 : 0, err);
        return rc == 0 ? FLO_OK : fail(c, FLO_ERR_DEVICE, "lossless encode: " + err);
    }
    if (!b->total_frames) return FLO_OK;
    // a caller who wrote all n_interleaved floats through flo_batch_clip_device_ptr left a partial sample-frame in the
    // zero padding behind the clip: it is not part of the clip (encoder.rs:174)
    for (size_t i = 0; i < b->n_clips; i++) {
        const uint64_t part = b->n_il[i] % b->ch;
        if (part) HIPCHK(c, hipMemsetAsync(b->d_pcm + b->clip_off[i] + b->clip_nsf[i] * b->ch, 0, part * sizeof(float), c->stream));
    }
    if (which == 0) which = c->force_path;
    if (which == 0) which = auto_form(b);
    if (b->ch > 2) which = 2;   // more than two channels: the generic frame-parallel kernels
    int rc;
    if (which == 3 || which == 4) which = 5;   // (earlier rounds' stereo chain forms: retired, the numbers stay valid)
    if (which == 5 && (b->exact || b->ch != 2)) which = 1;   // the exact-threshold yardstick and mono live in the one-wave-per-channel form
    if (which == 5) {   // stereo: one lock-step transform wave + one quantiser-and-packer wave per clip
#ifdef FLO_STAMPS
        if (!b->d_stamps) HIPCHK(c, pool_alloc(&b->d_stamps, b->n_clips * b->ch * 16 * 8));
#endif
        LossyArgs A = make_args(b);
        HIPCHK(c, hipMemsetAsync(b->d_next, 0, 4, c->stream));   // the batch-wide clip counter of the persistent workgroups
        rc = timed_launch(c, "lossy_chain2q", [&] { return launch_lossy_chain2q(A, c->stream); });
    } else if (which == 1) {
#ifdef FLO_STAMPS
        if (!b->d_stamps) HIPCHK(c, pool_alloc(&b->d_stamps, b->n_clips * b->ch * 16 * 8));
#endif
        LossyArgs A = make_args(b);
        rc = timed_launch(c, "lossy_chain", [&] { return launch_lossy_chain(A, c->stream); });
    } else {   // frame-parallel form
        // allocated by flo_batch_create when this form is what auto selects; only a forced form allocates here
        int arc = alloc_frame_scratch(b);
        if (arc != FLO_OK) return arc;
        LossyArgs A = make_args(b);
        if ((rc = timed_launch(c, "lossy_bands", [&] { return launch_lossy_frames_pass(A, 1, c->stream); })) != FLO_OK) return rc;
        if ((rc = timed_launch(c, "lossy_scan", [&] { return launch_lossy_scan(A, c->stream); })) != FLO_OK) return rc;
        if ((rc = timed_launch(c, "lossy_frames", [&] { return launch_lossy_frames_pass(A, 2, c->stream); })) != FLO_OK) return rc;
        rc = timed_launch(c, "lossy_compact", [&] { return launch_lossy_compact(A, c->stream); });
    }
    if (rc != FLO_OK) return rc;
    // header, TOC and CRC32 of every clip, in front of its DATA chunk (writer.rs:132-224; encoder.rs:229-238 parameters)
    FinishArgs F{};
    F.out = b->d_out;
    F.data_off = (const unsigned long long *)(b->d_plan + 3 * b->n_clips);
    F.clip_bytes = (const unsigned long long *)b->d_clip_bytes;
    F.clip_frame0 = (const unsigned long long *)(b->d_plan + 2 * b->n_clips);
    F.clip_frames = b->d_hops;
    F.frame_size = b->d_frame_size;
    F.frame_samples = nullptr;
    F.const_samples = 1024;
    F.sample_rate = b->sr;
    F.flags = (unsigned short)(0x01 | ((unsigned)b->ts->host.q_level << 8));
    F.channels = b->ch;
    F.bit_depth = 16;
    F.level = 5;
    F.n_clips = (int)b->n_clips;
    F.crc_out = b->d_crc;
    F.parts = finish_parts_for(b->n_clips);
    F.part_reg = b->d_part;
    F.max_frames = 0;
    for (auto h : b->hops) F.max_frames = h > F.max_frames ? h : F.max_frames;
    return timed_launch(c, "finish_files", [&] { return launch_finish_files(F, c->stream); });
}

static int batch_sync_impl(flo_batch *b, hipEvent_t done);
extern "C" int flo_batch_sync(flo_batch *b) {
    if (!b) return FLO_ERR_ARG;
    return batch_sync_impl(b, nullptr);
}
// done = nullptr: wait for the context's stream; else wait for that event only (recorded behind the batch's encode):
// the pipeline of flo_encode_batch must not wait for the NEXT chunk's work that is already queued on the stream
static int batch_sync_impl(flo_batch *b, hipEvent_t done) {
    flo_ctx *c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    if (done) HIPCHK(c, hipEventSynchronize(done));
    else HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!b->encoded && b->encode_failed) return fail(c, FLO_ERR_STATE, "the last flo_batch_encode on this batch failed");
    if (b->encoded && !b->synced) {
        if (b->mode == FLO_MODE_LOSSY) {
            // only the per-clip sizes come back; frame sizes stay on the device (the TOC is written there) and are
            // fetched on demand by the few host paths that want them
            b->h_clip_bytes.assign(b->n_clips, 0);
            b->h_frame_size.clear();
            if (b->n_clips && b->total_frames) {
                if (done && c->down_stream) {   // pipeline: other streams are busy, a synchronous copy would queue behind them
                    HIPCHK(c, hipMemcpyAsync(b->h_clip_bytes.data(), b->d_clip_bytes, b->n_clips * 8, hipMemcpyDeviceToHost, c->down_stream));
                    HIPCHK(c, hipStreamSynchronize(c->down_stream));
                } else {
                    HIPCHK(c, hipMemcpy(b->h_clip_bytes.data(), b->d_clip_bytes, b->n_clips * 8, hipMemcpyDeviceToHost));
                }
            }
            for (size_t i = 0; i < b->n_clips; i++)
                if (b->h_clip_bytes[i] > b->out_cap[i]) return fail(c, FLO_ERR_DEVICE, "bitstream overran its buffer");
#ifdef FLO_STAMPS
            if (b->d_stamps) {
                std::vector<unsigned long long> st(b->n_clips * b->ch * 16);
                HIPCHK(c, hipMemcpy(st.data(), b->d_stamps, st.size() * 8, hipMemcpyDeviceToHost));
                double sum[14] = {0};
                for (size_t w = 0; w < b->n_clips * b->ch; w++)
                    for (int i = 0; i < 14; i++) sum[i] += (double)st[w * 16 + i];
                double frames = (double)b->total_frames * b->ch;
                if (b->ch == 2 && (c->force_path >= 3 || c->force_path == 0)) {   // lock-step form: wave 0 = transform, wave 1 = packer
                    double t[14] = {0}, p[14] = {0};
                    for (size_t k = 0; k < b->n_clips; k++)
                        for (int i = 0; i < 14; i++) { t[i] += (double)st[(2 * k) * 16 + i]; p[i] += (double)st[(2 * k + 1) * 16 + i]; }
                    static const char *tn[] = {"fold", "prefetch", "fft", "postrot", "bandstats", "mask", "quant", "wait-consumed", "handover"};
                    static const char *pn[] = {"wait-ready", "read/quant", "pack0", "pack1", "flush", "wait-ts"};
                    fprintf(stderr, "[stamps2x] ticks (10 ns) per stereo frame | T:");
                    double tt = 0, pt = 0;
                    for (int i = 0; i < 9; i++) { fprintf(stderr, " %s=%.1f", tn[i], t[i] / b->total_frames); tt += t[i]; }
                    fprintf(stderr, " total=%.1f | P:", tt / b->total_frames);
                    for (int i = 0; i < 6; i++) { fprintf(stderr, " %s=%.1f", pn[i], p[i] / b->total_frames); pt += p[i]; }
                    fprintf(stderr, " total=%.1f item-form declined %.4f of channel-frames", pt / b->total_frames, p[6] / (2.0 * b->total_frames));
                    fprintf(stderr, " | T waits>1000: %.2f%% of frames, %.0f cyc/frame avg; >5000: %.2f%%, %.0f | P busy>12000: %.2f%%, %.0f; >20000: %.2f%%, %.0f\n",
                            100 * t[10] / b->total_frames, t[9] / b->total_frames, 100 * t[12] / b->total_frames, t[11] / b->total_frames,
                            100 * p[10] / b->total_frames, p[9] / b->total_frames, 100 * p[12] / b->total_frames, p[11] / b->total_frames);
                    // per clip: the packer's busy ticks per frame against the transform's (who waits for whom is a property
                    // of the clip's content): deciles over the clips
                    std::vector<double> pb(b->n_clips), tb(b->n_clips);
                    for (size_t k = 0; k < b->n_clips; k++) {
                        const double fr = (double)b->hops[k] > 0 ? (double)b->hops[k] : 1.0;
                        double tq = 0, pq = 0;
                        for (int i = 0; i < 9; i++) if (i != 7) tq += (double)st[(2 * k) * 16 + i];
                        for (int i = 1; i < 5; i++) pq += (double)st[(2 * k + 1) * 16 + i];
                        tb[k] = tq / fr; pb[k] = pq / fr;
                    }
                    if (const char *dump = getenv("FLO_STAMPS_DUMP")) {   // raw per-clip records for diag/stamps_clips.py
                        if (FILE *fh = fopen(dump, "wb")) {
                            fwrite(st.data(), 8, st.size(), fh);
                            fclose(fh);
                        }
                    }
                    std::vector<double> ps = pb, ts = tb;
                    std::sort(ps.begin(), ps.end()); std::sort(ts.begin(), ts.end());
                    fprintf(stderr, "[stamps2x] per-clip busy ticks per frame, deciles | P:");
                    for (int d = 0; d <= 10; d++) fprintf(stderr, " %.0f", ps[std::min(b->n_clips - 1, (size_t)(d * (b->n_clips - 1) / 10))]);
                    fprintf(stderr, " | T:");
                    for (int d = 0; d <= 10; d++) fprintf(stderr, " %.0f", ts[std::min(b->n_clips - 1, (size_t)(d * (b->n_clips - 1) / 10))]);
                    size_t pbound = 0;
                    for (size_t k = 0; k < b->n_clips; k++) pbound += pb[k] > tb[k];
                    fprintf(stderr, " | clips whose packer is busier than their transform: %.1f%%\n", 100.0 * pbound / b->n_clips);
                }
                static const char *nm[] = {"wait-loads+fold", "issue-loads", "fft", "postrot", "analyse(bands,psy,quant,plan)",
                                           "sync-tot", "emit", "sync-emit", "flush", "sync-tail"};
                fprintf(stderr, "[stamps] s_memtime ticks (100 MHz) per frame-channel:");
                for (int i = 0; i < 10; i++) fprintf(stderr, " %s=%.0f", nm[i], sum[i] / frames);
                fprintf(stderr, " | inside analyse: bands=%.0f psy=%.0f quantise=%.0f (plan = analyse rest)", sum[10] / frames, sum[11] / frames, sum[12] / frames);
                fprintf(stderr, "\n");
            }
#endif
        } else {
            std::string err;
            if (lossless_collect(b->ll, err) != 0) return fail(c, FLO_ERR_DEVICE, "lossless collect: " + err);
        }
        b->synced = true;
    }
    return FLO_OK;
}

extern "C" int flo_batch_data_bytes(flo_batch *b, uint64_t *total) {
    if (!b || !total) return FLO_ERR_ARG;
    if (!b->synced) return fail(b->ctx, FLO_ERR_STATE, "call flo_batch_encode + flo_batch_sync first");
    uint64_t t = 0;
    if (b->mode == FLO_MODE_LOSSY)
        for (auto v : b->h_clip_bytes) t += v;
    else
        t = lossless_total_bytes(b->ll);
    *total = t;
    return FLO_OK;
}

extern "C" int flo_batch_device_streams(flo_batch *b, const uint8_t **base, const uint64_t **offsets,
                                        const uint64_t **sizes) {
    if (!b) return FLO_ERR_ARG;
    if (!b->synced) return fail(b->ctx, FLO_ERR_STATE, "call flo_batch_encode + flo_batch_sync first");
    if (b->mode == FLO_MODE_LOSSY) {
        if (base) *base = b->d_out;
        if (offsets) *offsets = b->out_off.data();
        if (sizes) *sizes = b->h_clip_bytes.data();
        return FLO_OK;
    }
    return lossless_device_streams(b->ll, base, offsets, sizes) == 0 ? FLO_OK : FLO_ERR_STATE;
}

// Finished files (header + TOC + DATA, no META) as they sit in HBM after flo_batch_sync.
extern "C" int flo_batch_device_files(flo_batch *b, const uint8_t **base, const uint64_t **offsets, const uint64_t **sizes) {
    if (!b) return FLO_ERR_ARG;
    if (!b->synced) return fail(b->ctx, FLO_ERR_STATE, "call flo_batch_encode + flo_batch_sync first");
    if (b->mode == FLO_MODE_LOSSY) {
        if (b->h_file_bytes.size() != b->n_clips) b->h_file_bytes.resize(b->n_clips);
        for (size_t i = 0; i < b->n_clips; i++) b->h_file_bytes[i] = 74 + 20 * (uint64_t)b->hops[i] + b->h_clip_bytes[i];
        if (base) *base = b->d_out;
        if (offsets) *offsets = b->file_off.data();
        if (sizes) *sizes = b->h_file_bytes.data();
        return FLO_OK;
    }
    return lossless_device_files(b->ll, base, offsets, sizes) == 0 ? FLO_OK : FLO_ERR_STATE;
}

// Pack every clip's DATA chunk (files = false) or finished .flo file without META (files = true) into dst (device
// memory owned by the caller, e.g. a torch tensor), clip i at offsets[i] (16-byte aligned, offsets[n_clips] = total).
// Asynchronous on the ctx stream.
static int pack_impl(flo_batch *b, bool files, void *dst_device, size_t dst_cap, uint64_t *offsets) {
    if (!b || !offsets || (!dst_device && dst_cap)) return FLO_ERR_ARG;
    flo_ctx *c = b->ctx;
    if (!b->synced) return fail(c, FLO_ERR_STATE, "call flo_batch_encode + flo_batch_sync first");
    HIPCHK(c, hipSetDevice(c->device));
    const uint8_t *base;
    const uint64_t *offs, *sizes;
    int rc = files ? flo_batch_device_files(b, &base, &offs, &sizes) : flo_batch_device_streams(b, &base, &offs, &sizes);
    if (rc != FLO_OK) return rc;
    uint64_t pos = 0;
    for (size_t i = 0; i < b->n_clips; i++) {
        offsets[i] = pos;
        pos += (sizes[i] + 15) & ~(uint64_t)15;
    }
    offsets[b->n_clips] = pos;
    if (pos > dst_cap) return fail(c, FLO_ERR_ARG, "packed stream buffer too small");
    if (!b->n_clips || !pos) return FLO_OK;
    if (!b->d_pack_plan) {
        HIPCHK(c, pool_alloc(&b->d_pack_plan, 3 * b->n_clips * 8));
        HIPCHK(c, hipEventCreateWithFlags(&b->ev_pack_plan, hipEventDisableTiming));
    } else {
        HIPCHK(c, hipEventSynchronize(b->ev_pack_plan));   // the previous pack's copy has read the plan (long ago)
    }
    std::string perr;
    if (!b->pin_plan && !(b->pin_plan = (uint64_t *)stager_pinned_get(c->stager, (8 * b->n_clips + 8) * 8, perr)))
        return fail(c, FLO_ERR_NOMEM, perr);
    uint64_t *plan = b->pin_plan + 5 * b->n_clips;   // behind the clip plan (4 n) and the hops (n u32)
    for (size_t i = 0; i < b->n_clips; i++) {
        plan[i] = offs[i];
        plan[b->n_clips + i] = offsets[i];
        plan[2 * b->n_clips + i] = sizes[i];
    }
    HIPCHK(c, hipMemcpyAsync(b->d_pack_plan, plan, 3 * b->n_clips * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipEventRecord(b->ev_pack_plan, c->stream));
    const unsigned long long *dp = (const unsigned long long *)b->d_pack_plan;
    return timed_launch(c, "pack_streams", [&] {
        return launch_pack_streams(base, dp, dp + b->n_clips, dp + 2 * b->n_clips, (int)b->n_clips, (uint8_t *)dst_device, c->stream);
    });
}
extern "C" int flo_batch_pack_streams(flo_batch *b, void *dst_device, size_t dst_cap, uint64_t *offsets) {
    return pack_impl(b, false, dst_device, dst_cap, offsets);
}
extern "C" int flo_batch_pack_files(flo_batch *b, void *dst_device, size_t dst_cap, uint64_t *offsets) {
    return pack_impl(b, true, dst_device, dst_cap, offsets);
}

extern "C" int flo_batch_fetch(flo_batch *b, size_t clip, const uint8_t *meta, size_t meta_len, uint8_t **out,
                               size_t *out_len) {
    if (!b || clip >= b->n_clips || !out || !out_len || (meta_len && !meta)) return FLO_ERR_ARG;
    flo_ctx *c = b->ctx;
    if (!b->synced) return fail(c, FLO_ERR_STATE, "call flo_batch_encode + flo_batch_sync first");
    HIPCHK(c, hipSetDevice(c->device));
    if (b->mode == FLO_MODE_LOSSLESS) {
        std::string err;
        int rc = lossless_fetch(b->ll, clip, b->bit_depth, meta, meta_len, out, out_len, err);
        return rc == 0 ? FLO_OK : fail(c, FLO_ERR_DEVICE, "lossless fetch: " + err);
    }
    // the file was finished on the device: copy it, append META and patch meta_size (header bytes 62..69)
    const size_t head = 74 + 20 * (size_t)b->hops[clip];
    const size_t n = head + (size_t)b->h_clip_bytes[clip];
    uint8_t *f = (uint8_t *)malloc(n + meta_len ? n + meta_len : 1);
    if (!f) return fail(c, FLO_ERR_NOMEM, "malloc failed");
    hipError_t e = hipMemcpy(f, b->d_out + b->file_off[clip], n, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
        free(f);
        return fail(c, FLO_ERR_DEVICE, std::string("fetch: ") + hipGetErrorString(e));
    }
    if (meta_len) memcpy(f + n, meta, meta_len);
    for (int i = 0; i < 8; i++) f[62 + i] = (uint8_t)((uint64_t)meta_len >> (8 * i));
    *out = f;
    *out_len = n + meta_len;
    return FLO_OK;
}

// ------------------------------------------------------------------------------------------------ one-shot API
static int ctx_stager(flo_ctx *c) {
    if (c->stager && c->up_stream) return FLO_OK;
    if (c->stager) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
        HIPCHK(c, hipStreamCreateWithFlags(&c->down_stream, hipStreamNonBlocking));
        return FLO_OK;
    }
    std::string err;
    c->stager = stager_create(err);
    if (!c->stager) return fail(c, FLO_ERR_NOMEM, err);
    HIPCHK(c, hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
    HIPCHK(c, hipStreamCreateWithFlags(&c->down_stream, hipStreamNonBlocking));
    return FLO_OK;
}

// all clips of a batch from host buffers, through the pinned staging ring
static int batch_upload_all(flo_batch *b, const float *const *pcm, hipStream_t stream = nullptr) {
    flo_ctx *c = b->ctx;
    int rc = ctx_stager(c);
    if (rc != FLO_OK) return rc;
    std::vector<UploadSeg> segs;
    segs.reserve(b->n_clips);
    for (size_t i = 0; i < b->n_clips; i++) {
        // a trailing partial sample-frame is not part of the clip (encoder.rs:174): it must not land in the zero padding
        const uint64_t n_copy = b->mode == FLO_MODE_LOSSY ? b->clip_nsf[i] * b->ch : b->n_il[i];
        if (n_copy) segs.push_back({b->d_pcm + b->clip_off[i], pcm[i], n_copy * sizeof(float)});
    }
    std::string err;
    if (stager_upload(c->stager, segs, stream ? stream : c->stream, err) != 0) return fail(c, FLO_ERR_DEVICE, err);
    b->encoded = b->synced = b->encode_failed = false;
    return FLO_OK;
}

// The throughput entry point on host buffers, as a three-stage pipeline over chunks of clips: while chunk k is being
// encoded, the copy threads and the upload stream bring in chunk k + 1 and the download stream takes the finished files
// of chunk k - 1 out (one packed pinned transfer per chunk). The bytes are those of n_clips separate calls.
extern "C" int flo_encode_batch(flo_ctx *c, int mode, size_t n_clips, const float *const *pcm, const size_t *n_il,
                                uint32_t sr, uint8_t ch, float qol, uint8_t **outs, size_t *out_lens) {
    if (!c || (n_clips && (!pcm || !n_il || !outs || !out_lens))) return FLO_ERR_ARG;
    for (size_t i = 0; i < n_clips; i++)
        if (n_il[i] && !pcm[i]) return FLO_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ctx_stager(c);
    if (rc != FLO_OK) return rc;
    for (size_t i = 0; i < n_clips; i++) outs[i] = nullptr;
    const bool trace = getenv("FLO_TRACE") != nullptr;
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = tnow();
    // chunks of about 48 MB of PCM (at least one clip each)
    struct Chunk {
        size_t first = 0, count = 0;
        flo_batch *b = nullptr;
        uint8_t *d_packed = nullptr;
        uint8_t *host = nullptr;
        std::vector<uint64_t> po;
        hipEvent_t up = nullptr, enc = nullptr;
        bool encoded = false, taken = false;
    };
    std::vector<Chunk> chunks;
    {
        const uint64_t target = (uint64_t)48 << 20;
        size_t i = 0;
        while (i < n_clips) {
            Chunk k;
            k.first = i;
            uint64_t bytes = 0;
            while (i < n_clips && (k.count == 0 || bytes + n_il[i] * 4 <= target)) {
                bytes += (uint64_t)n_il[i] * 4;
                i++;
                k.count++;
            }
            chunks.push_back(k);
        }
    }
    auto cleanup = [&](int code) {
        hipStreamSynchronize(c->up_stream);
        hipStreamSynchronize(c->stream);
        hipStreamSynchronize(c->down_stream);
        for (Chunk &k : chunks) {
            if (k.d_packed) pool_free(k.d_packed);
            if (k.host) stager_pinned_put(c->stager, k.host);
            if (k.up) hipEventDestroy(k.up);
            if (k.enc) hipEventDestroy(k.enc);
            if (k.b) flo_batch_destroy(k.b);
        }
        if (code != FLO_OK)
            for (size_t i = 0; i < n_clips; i++) {
                free(outs[i]);
                outs[i] = nullptr;
            }
        return code;
    };
    // stage 3 for one chunk: sizes, pack, download (asynchronous on the download stream)
    auto take = [&](Chunk &k) -> int {
        int r = batch_sync_impl(k.b, mode == FLO_MODE_LOSSY ? k.enc : nullptr);
        if (r != FLO_OK) return r;
        const uint8_t *base;
        const uint64_t *offs, *sizes;
        if ((r = flo_batch_device_files(k.b, &base, &offs, &sizes)) != FLO_OK) return r;
        uint64_t need = 16;
        for (size_t i = 0; i < k.count; i++) need += (sizes[i] + 15) & ~(uint64_t)15;
        if (pool_alloc(&k.d_packed, need) != hipSuccess) return fail(c, FLO_ERR_NOMEM, "packed output buffer");
        k.po.resize(k.count + 1);
        if ((r = flo_batch_pack_files(k.b, k.d_packed, need, k.po.data())) != FLO_OK) return r;
        std::string err;
        k.host = (uint8_t *)stager_pinned_get(c->stager, need, err);
        if (!k.host) return fail(c, FLO_ERR_NOMEM, err);
        HIPCHK(c, hipEventRecord(k.up, c->stream));   // (the upload event has served its purpose: reused for "packed")
        HIPCHK(c, hipStreamWaitEvent(c->down_stream, k.up, 0));
        HIPCHK(c, hipMemcpyAsync(k.host, k.d_packed, k.po[k.count], hipMemcpyDeviceToHost, c->down_stream));
        for (size_t i = 0; i < k.count; i++) out_lens[k.first + i] = sizes[i];
        k.taken = true;
        return FLO_OK;
    };
    for (size_t ci = 0; ci < chunks.size(); ci++) {
        Chunk &k = chunks[ci];
        const double ta = tnow();
        if ((rc = flo_batch_create(c, mode, k.count, n_il + k.first, sr, ch, qol, &k.b)) != FLO_OK) return cleanup(rc);
        const double tb = tnow();
        if (hipEventCreateWithFlags(&k.up, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&k.enc, hipEventDisableTiming) != hipSuccess)
            return cleanup(fail(c, FLO_ERR_DEVICE, "hipEventCreate"));
        // (errors leave through cleanup(): the chunks' batches, pinned blocks and events are released, outs[] stays empty)
        auto ordered = [&](hipError_t e, const char *what) { return e == hipSuccess ? FLO_OK : fail(c, FLO_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e)); };
        // the batch's padding memset ran on the ctx stream: the uploads must land behind it
        if ((rc = ordered(hipEventRecord(k.enc, c->stream), "hipEventRecord")) != FLO_OK) return cleanup(rc);
        if ((rc = ordered(hipStreamWaitEvent(c->up_stream, k.enc, 0), "hipStreamWaitEvent")) != FLO_OK) return cleanup(rc);
        if ((rc = batch_upload_all(k.b, pcm + k.first, c->up_stream)) != FLO_OK) return cleanup(rc);
        const double tc = tnow();
        if ((rc = ordered(hipEventRecord(k.up, c->up_stream), "hipEventRecord")) != FLO_OK) return cleanup(rc);
        if ((rc = ordered(hipStreamWaitEvent(c->stream, k.up, 0), "hipStreamWaitEvent")) != FLO_OK) return cleanup(rc);
        if ((rc = flo_batch_encode(k.b, 0)) != FLO_OK) return cleanup(rc);
        if ((rc = ordered(hipEventRecord(k.enc, c->stream), "hipEventRecord")) != FLO_OK) return cleanup(rc);   // this chunk's files are finished behind this point
        k.encoded = true;
        const double td = tnow();
        if (ci > 0 && (rc = take(chunks[ci - 1])) != FLO_OK) return cleanup(rc);
        if (trace) fprintf(stderr, "  chunk %zu: create %.3f upload %.3f encode-enqueue %.3f take(prev) %.3f ms\n", ci, (tb - ta) * 1e3, (tc - tb) * 1e3, (td - tc) * 1e3, (tnow() - td) * 1e3);
    }
    if (!chunks.empty() && (rc = take(chunks.back())) != FLO_OK) return cleanup(rc);
    if (hipStreamSynchronize(c->down_stream) != hipSuccess) return cleanup(fail(c, FLO_ERR_DEVICE, "download of the finished files failed"));
    const double t1 = tnow();
    // cut the packed transfers into the per-clip buffers the caller owns (the copy threads share the work)
    std::vector<UploadSeg> cuts;
    cuts.reserve(n_clips);
    for (Chunk &k : chunks)
        for (size_t i = 0; i < k.count; i++) {
            const size_t g = k.first + i;
            outs[g] = (uint8_t *)malloc(out_lens[g] ? out_lens[g] : 1);
            if (!outs[g]) return cleanup(fail(c, FLO_ERR_NOMEM, "malloc failed"));
            cuts.push_back({outs[g], k.host + k.po[i], out_lens[g]});
        }
    stager_memcpy_many(c->stager, cuts);
    if (trace) fprintf(stderr, "[flo_encode_batch] %zu chunks: pipeline %.3f ms, cut %.3f ms\n", chunks.size(), (t1 - t0) * 1e3, (tnow() - t1) * 1e3);
    return cleanup(FLO_OK);
}

static int encode_one(flo_ctx *c, int mode, const float *pcm, size_t n, uint32_t sr, uint8_t ch, float qol,
                      uint8_t bit_depth, const uint8_t *meta, size_t meta_len, uint8_t **out, size_t *out_len) {
    if (!c || !out || !out_len || (n && !pcm) || (meta_len && !meta)) return FLO_ERR_ARG;
    static const bool trace = getenv("FLO_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = trace ? now() : 0;
    flo_batch *b = nullptr;
    int rc = flo_batch_create(c, mode, 1, &n, sr, ch, qol, &b);
    if (rc != FLO_OK) return rc;
    b->bit_depth = bit_depth;
    const double t1 = trace ? now() : 0;
    {
        const float *one[1] = {pcm};
        rc = batch_upload_all(b, one);
    }
    const double t2 = trace ? now() : 0;
    if (rc == FLO_OK) rc = flo_batch_encode(b, 0);
    const double t3 = trace ? now() : 0;
    double t4 = t3;
    bool fetched = false;
    if (rc == FLO_OK && mode == FLO_MODE_LOSSY && b->total_frames && c->stager) {
        // ONE round trip behind the kernels instead of two (sizes, then the file): the size word and a generous guess of the
        // file (768 bytes per frame; q = 0.55 makes about 420) come back together into pinned memory; a longer file fetches
        // its remainder afterwards
        const size_t head = 74 + 20 * (size_t)b->hops[0];
        size_t est = head + 768 * (size_t)b->hops[0];
        if (est > head + (size_t)b->out_cap[0]) est = head + (size_t)b->out_cap[0];
        std::string err;
        uint8_t *pin = (uint8_t *)stager_pinned_get(c->stager, 16 + est, err);
        if (pin) {
            hipError_t e = hipMemcpyAsync(pin, b->d_clip_bytes, 8, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(pin + 16, b->d_out + b->file_off[0], est, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            t4 = trace ? now() : 0;
            if (e != hipSuccess) {
                rc = fail(c, FLO_ERR_DEVICE, std::string("one-shot fetch: ") + hipGetErrorString(e));
            } else if (!b->encoded) {
                rc = fail(c, FLO_ERR_STATE, "the last flo_batch_encode on this batch failed");
            } else {
                uint64_t sz;
                memcpy(&sz, pin, 8);
                if (sz > b->out_cap[0]) {
                    rc = fail(c, FLO_ERR_DEVICE, "bitstream overran its buffer");
                } else {
                    b->h_clip_bytes.assign(1, sz);
                    b->h_frame_size.clear();
                    b->synced = true;
                    const size_t n = head + (size_t)sz;
                    uint8_t *f = (uint8_t *)malloc(n + meta_len ? n + meta_len : 1);
                    if (!f) {
                        rc = fail(c, FLO_ERR_NOMEM, "malloc failed");
                    } else {
                        memcpy(f, pin + 16, n < est ? n : est);
                        if (n > est) e = hipMemcpy(f + est, b->d_out + b->file_off[0] + est, n - est, hipMemcpyDeviceToHost);
                        if (e != hipSuccess) {
                            free(f);
                            rc = fail(c, FLO_ERR_DEVICE, std::string("one-shot fetch: ") + hipGetErrorString(e));
                        } else {
                            if (meta_len) memcpy(f + n, meta, meta_len);
                            for (int i = 0; i < 8; i++) f[62 + i] = (uint8_t)((uint64_t)meta_len >> (8 * i));   // meta_size
                            *out = f;
                            *out_len = n + meta_len;
                        }
                    }
                }
            }
            stager_pinned_put(c->stager, pin);
            fetched = true;
        }
    }
    if (!fetched) {
        if (rc == FLO_OK) rc = flo_batch_sync(b);
        t4 = trace ? now() : 0;
        if (rc == FLO_OK) rc = flo_batch_fetch(b, 0, meta, meta_len, out, out_len);
    }
    const double t5 = trace ? now() : 0;
    flo_batch_destroy(b);
    if (trace)
        fprintf(stderr, "[encode_one] create %.1f upload %.1f encode %.1f sync %.1f fetch %.1f destroy %.1f us\n", t1 - t0, t2 - t1,
                t3 - t2, t4 - t3, t5 - t4, now() - t5);
    return rc;
}

extern "C" int flo_encode_lossy(flo_ctx *c, const float *pcm, size_t n, uint32_t sr, uint8_t ch, float quality,
                                const uint8_t *meta, size_t meta_len, uint8_t **out, size_t *out_len) {
    return encode_one(c, FLO_MODE_LOSSY, pcm, n, sr, ch, quality, 16, meta, meta_len, out, out_len);
}
extern "C" int flo_encode_lossless(flo_ctx *c, const float *pcm, size_t n, uint32_t sr, uint8_t ch, uint8_t bit_depth,
                                   uint8_t level, const uint8_t *meta, size_t meta_len, uint8_t **out,
                                   size_t *out_len) {
    return encode_one(c, FLO_MODE_LOSSLESS, pcm, n, sr, ch, (float)level, bit_depth, meta, meta_len, out, out_len);
}

// ------------------------------------------------------------------------------------------------ stage entry points
extern "C" int flo_mdct_forward(flo_ctx *c, const float *frames, size_t n_frames, float *coeffs) {
    if (!c || (n_frames && (!frames || !coeffs))) return FLO_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (!n_frames) return FLO_OK;
    TableSet *ts;
    int rc = get_tables(c, 44100, 0.55f, &ts);
    if (rc != FLO_OK) return rc;
    float *d_in = nullptr, *d_out = nullptr;
    HIPCHK(c, hipMalloc(&d_in, n_frames * 2048 * sizeof(float)));
    hipError_t e = hipMalloc(&d_out, n_frames * 1024 * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, frames, n_frames * 2048 * sizeof(float), hipMemcpyHostToDevice, c->stream);
    int lrc = 0;
    if (e == hipSuccess) lrc = launch_mdct_only(ts->dev, d_in, n_frames, d_out, c->stream);
    if (e == hipSuccess && lrc == 0)
        e = hipMemcpyAsync(coeffs, d_out, n_frames * 1024 * sizeof(float), hipMemcpyDeviceToHost, c->stream);
    hipError_t e2 = hipStreamSynchronize(c->stream);
    hipFree(d_in);
    if (d_out) hipFree(d_out);
    if (e != hipSuccess || e2 != hipSuccess || lrc != 0)
        return fail(c, FLO_ERR_DEVICE, std::string("flo_mdct_forward: ") + hipGetErrorString(e != hipSuccess ? e : e2));
    return FLO_OK;
}

static int analyze_common(flo_ctx *c, const float *pcm, size_t n, const float *in_coeffs, size_t in_hops, uint32_t sr,
                          uint8_t ch, float quality, int exact, float *coeffs, int16_t *q, uint16_t *sfw, size_t *num_hops) {
    flo_batch *b = nullptr;
    size_t n_il = in_coeffs ? (in_hops ? (in_hops - 1) * 1024 * ch : 0) : n;
    if (in_coeffs && in_hops == 0) return FLO_OK;
    int rc = flo_batch_create(c, FLO_MODE_LOSSY, 1, &n_il, sr, ch, quality, &b);
    if (rc != FLO_OK) return rc;
    const size_t hops = b->hops[0];
    if (num_hops) *num_hops = hops;
    float *d_in = nullptr;
    auto done = [&](int code) {
        if (d_in) hipFree(d_in);
        flo_batch_destroy(b);
        return code;
    };
    const size_t per = hops * ch;
    if (pool_alloc(&b->d_dbg_coeffs, per * 1024 * 4 + 16) != hipSuccess || pool_alloc(&b->d_dbg_q, per * 1024 * 2 + 16) != hipSuccess ||
        pool_alloc(&b->d_dbg_sfw, per * 25 * 2 + 16) != hipSuccess)
        return done(fail(c, FLO_ERR_NOMEM, "hipMalloc analysis buffers"));
    if (in_coeffs) {
        if (hipMalloc(&d_in, per * 1024 * 4) != hipSuccess) return done(fail(c, FLO_ERR_NOMEM, "hipMalloc"));
        if (hipMemcpy(d_in, in_coeffs, per * 1024 * 4, hipMemcpyHostToDevice) != hipSuccess)
            return done(fail(c, FLO_ERR_DEVICE, "hipMemcpy"));
        b->d_in_coeffs = d_in;
        b->exact = exact ? 1 : 0;
    } else {
        rc = flo_batch_upload(b, 0, pcm);
        if (rc != FLO_OK) return done(rc);
    }
    rc = flo_batch_encode(b, c->force_path ? c->force_path : 1);
    if (rc == FLO_OK) rc = flo_batch_sync(b);
    if (rc != FLO_OK) return done(rc);
    hipError_t e = hipSuccess;
    if (coeffs && !in_coeffs) e = hipMemcpy(coeffs, b->d_dbg_coeffs, per * 1024 * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess && q) e = hipMemcpy(q, b->d_dbg_q, per * 1024 * 2, hipMemcpyDeviceToHost);
    if (e == hipSuccess && sfw) e = hipMemcpy(sfw, b->d_dbg_sfw, per * 25 * 2, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return done(fail(c, FLO_ERR_DEVICE, std::string("analysis D2H: ") + hipGetErrorString(e)));
    return done(FLO_OK);
}

extern "C" int flo_lossy_analyze(flo_ctx *c, const float *pcm, size_t n, uint32_t sr, uint8_t ch, float quality,
                                 float *coeffs, int16_t *q, uint16_t *sfw, size_t *num_hops) {
    if (!c || (n && !pcm)) return FLO_ERR_ARG;
    return analyze_common(c, pcm, n, nullptr, 0, sr, ch, quality, 0, coeffs, q, sfw, num_hops);
}
extern "C" int flo_lossy_quantize(flo_ctx *c, const float *coeffs, size_t num_hops, uint32_t sr, uint8_t ch,
                                  float quality, int exact, int16_t *q, uint16_t *sfw) {
    if (!c || (num_hops && !coeffs)) return FLO_ERR_ARG;
    return analyze_common(c, nullptr, 0, coeffs, num_hops, sr, ch, quality, exact, nullptr, q, sfw, nullptr);
}

extern "C" int flo_lossy_quantize_smr(flo_ctx *c, const float *coeffs, const float *smr, size_t n_vec, uint32_t sr, float quality,
                                      int16_t *q, float *scale_factors) {
    if (!c || (n_vec && (!coeffs || !scale_factors || (smr && !q)))) return FLO_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (!n_vec) return FLO_OK;
    TableSet *ts;
    int rc = get_tables(c, sr, quality, &ts);
    if (rc != FLO_OK) return rc;
    float *d_c = nullptr, *d_smr = nullptr, *d_sf = nullptr;
    short *d_q = nullptr;
    hipError_t e = hipMalloc(&d_c, n_vec * 4096);
    if (e == hipSuccess) e = hipMalloc(&d_sf, n_vec * 100);
    if (e == hipSuccess && smr) e = hipMalloc(&d_smr, n_vec * 4096);
    if (e == hipSuccess && smr) e = hipMalloc(&d_q, n_vec * 2048);
    if (e == hipSuccess) e = hipMemcpyAsync(d_c, coeffs, n_vec * 4096, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && smr) e = hipMemcpyAsync(d_smr, smr, n_vec * 4096, hipMemcpyHostToDevice, c->stream);
    int lrc = 0;
    if (e == hipSuccess) lrc = launch_quantise_smr(ts->dev, d_c, d_smr, n_vec, d_q, d_sf, c->stream);
    if (e == hipSuccess && lrc == 0) e = hipMemcpyAsync(scale_factors, d_sf, n_vec * 100, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && lrc == 0 && smr) e = hipMemcpyAsync(q, d_q, n_vec * 2048, hipMemcpyDeviceToHost, c->stream);
    hipError_t e2 = hipStreamSynchronize(c->stream);
    hipFree(d_c);
    if (d_smr) hipFree(d_smr);
    if (d_sf) hipFree(d_sf);
    if (d_q) hipFree(d_q);
    if (e != hipSuccess || e2 != hipSuccess || lrc != 0)
        return fail(c, FLO_ERR_DEVICE, std::string("flo_lossy_quantize_smr: ") + hipGetErrorString(e != hipSuccess ? e : e2));
    return FLO_OK;
}

extern "C" int flo_sparse_pack(flo_ctx *c, const int16_t *q, size_t n_vec, int form, uint8_t *out, size_t out_cap,
                               uint32_t *out_off) {
    if (!c || (n_vec && (!q || !out || !out_off)) || form < 0 || form > 2) return FLO_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (out_off) out_off[0] = 0;
    if (!n_vec) return FLO_OK;
    short *d_q = nullptr;
    uint8_t *d_slots = nullptr;
    uint32_t *d_sizes = nullptr;
    HIPCHK(c, hipMalloc(&d_q, n_vec * 2048));
    hipError_t e = hipMalloc(&d_slots, n_vec * 2080);
    if (e == hipSuccess) e = hipMalloc(&d_sizes, n_vec * 4);
    if (e == hipSuccess) e = hipMemcpyAsync(d_q, q, n_vec * 2048, hipMemcpyHostToDevice, c->stream);
    int lrc = 0;
    if (e == hipSuccess) lrc = launch_sparse_only(d_q, n_vec, d_slots, d_sizes, form, c->stream);
    std::vector<uint8_t> slots(n_vec * 2080);
    std::vector<uint32_t> sizes(n_vec);
    if (e == hipSuccess && lrc == 0) e = hipMemcpyAsync(slots.data(), d_slots, slots.size(), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && lrc == 0) e = hipMemcpyAsync(sizes.data(), d_sizes, n_vec * 4, hipMemcpyDeviceToHost, c->stream);
    hipError_t e2 = hipStreamSynchronize(c->stream);
    hipFree(d_q);
    if (d_slots) hipFree(d_slots);
    if (d_sizes) hipFree(d_sizes);
    if (e != hipSuccess || e2 != hipSuccess || lrc != 0) return fail(c, FLO_ERR_DEVICE, "flo_sparse_pack failed");
    size_t pos = 0;
    for (size_t i = 0; i < n_vec; i++) {
        if (pos + sizes[i] > out_cap) return fail(c, FLO_ERR_ARG, "output buffer too small");
        memcpy(out + pos, slots.data() + i * 2080, sizes[i]);
        pos += sizes[i];
        out_off[i + 1] = (uint32_t)pos;
    }
    return FLO_OK;
}

// ---- decode ---------------------------------------------------------------------------------------------------
namespace {
struct DevMem {   // frees on scope exit
    void *p = nullptr;
    ~DevMem() {
        if (p) pool_free(p);   // the owner has synchronised the stream by the time this runs
    }
    template <class T>
    T *as() const { return reinterpret_cast<T *>(p); }
};
// Declared right behind a function's DevMem objects, so that it is destroyed BEFORE them: whatever way the function is
// left (an error return in the middle included), the context's stream is idle when the blocks go back to the pool, where
// another context or thread may be handed them at once. On the normal path the stream has been synchronised already and
// this costs a few microseconds.
struct QuiesceOnExit {
    flo_ctx *c;
    explicit QuiesceOnExit(flo_ctx *ctx) : c(ctx) {}
    ~QuiesceOnExit() {
        if (c && c->stream) hipStreamSynchronize(c->stream);
    }
};
template <class T>
int upload(flo_ctx *c, DevMem &m, const std::vector<T> &v) {
    size_t bytes = v.size() * sizeof(T);
    HIPCHK(c, pool_alloc(&m.p, bytes ? bytes : 16));
    if (bytes) HIPCHK(c, hipMemcpyAsync(m.p, v.data(), bytes, hipMemcpyHostToDevice, c->stream));
    return FLO_OK;
}
}  // namespace

extern "C" int flo_probe_container(const uint8_t *flo, size_t len, flo_container_info *out, char *err, size_t err_cap) {
    if (!out || (!flo && len)) return FLO_ERR_ARG;
    memset(out, 0, sizeof *out);
    if (err && err_cap) err[0] = 0;
    ParsedFile f;
    const char *perr = "";
    if (parse_file(flo, len, f, &perr) != 0) {
        if (err && err_cap) snprintf(err, err_cap, "%s", perr);
        return FLO_ERR_FORMAT;
    }
    out->version_major = f.version_major;
    out->version_minor = f.version_minor;
    out->channels = f.channels;
    out->bit_depth = f.bit_depth;
    out->compression_level = f.compression_level;
    out->is_transform = f.is_transform ? 1 : 0;
    out->flags = f.flags;
    out->sample_rate = f.sample_rate;
    out->data_crc32 = f.data_crc32;
    out->n_frames = (uint32_t)f.frames.size();
    out->total_samples = f.total_samples;
    out->data_start = f.data_start;
    out->data_size = f.data_size;
    for (const FrameDesc &fr : f.frames) out->frame_samples_sum += fr.samples;
    return FLO_OK;
}

// Device -> caller-owned pageable memory. A copy engine writes pageable memory at a fraction of the PCIe rate, so a
// large result comes down in 8 MiB pieces through two pinned buffers: while the copy threads move piece i into the
// caller's buffer, the copy engine is already filling the other pinned buffer with piece i + 1.
static int download(flo_ctx *c, void *dst, const void *d_src, size_t bytes) {
    constexpr size_t kPieceBytes = 8u << 20;
    if (bytes <= kPieceBytes / 2) {
        HIPCHK(c, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return FLO_OK;
    }
    int rc = ctx_stager(c);
    if (rc != FLO_OK) return rc;
    std::string err;
    void *pin[2] = {stager_pinned_get(c->stager, kPieceBytes, err), stager_pinned_get(c->stager, kPieceBytes, err)};
    hipEvent_t ev[2] = {nullptr, nullptr};
    auto done = [&](int r) {
        for (int i = 0; i < 2; i++) {
            if (pin[i]) stager_pinned_put(c->stager, pin[i]);
            if (ev[i]) hipEventDestroy(ev[i]);
        }
        return r;
    };
    if (!pin[0] || !pin[1]) return done(fail(c, FLO_ERR_NOMEM, err));
    for (int i = 0; i < 2; i++)
        if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return done(fail(c, FLO_ERR_DEVICE, "hipEventCreate failed"));
    const size_t pieces = (bytes + kPieceBytes - 1) / kPieceBytes;
    auto issue = [&](size_t i) -> hipError_t {
        const size_t off = i * kPieceBytes, n = bytes - off < kPieceBytes ? bytes - off : kPieceBytes;
        hipError_t e = hipMemcpyAsync(pin[i & 1], (const char *)d_src + off, n, hipMemcpyDeviceToHost, c->stream);
        return e == hipSuccess ? hipEventRecord(ev[i & 1], c->stream) : e;
    };
    hipError_t e = issue(0);
    if (e == hipSuccess && pieces > 1) e = issue(1);
    for (size_t i = 0; i < pieces && e == hipSuccess; i++) {
        const size_t off = i * kPieceBytes, n = bytes - off < kPieceBytes ? bytes - off : kPieceBytes;
        e = hipEventSynchronize(ev[i & 1]);
        if (e != hipSuccess) break;
        stager_memcpy_many(c->stager, {{(char *)dst + off, pin[i & 1], n}});
        if (i + 2 < pieces) e = issue(i + 2);
    }
    if (e != hipSuccess) return done(fail(c, FLO_ERR_DEVICE, std::string("download: ") + hipGetErrorString(e)));
    return done(FLO_OK);
}

// The channel wrappers and frames of one or more parsed lossless files whose bytes sit in one device buffer
// (lossless/decoder.rs:21-72 per file). Output sample-frames follow each other file after file.
struct LlWork {
    std::vector<LlChannelDev> chs;
    std::vector<LlFrameDev> frs;
    unsigned long long scratch = 0, out_sf = 0;
    unsigned max_samples = 0;
    void add(const ParsedFile &f, uint64_t base, int nch) {
        for (const FrameDesc &fr : f.frames) {
            LlFrameDev fd{};
            fd.out_off = out_sf;
            fd.first_channel = (unsigned)chs.size();
            fd.n_channels = fr.n_channels;
            fd.samples = fr.samples;
            fd.mid_side = (nch == 2 && (fr.flags & 1)) ? 1u : 0u;
            for (unsigned k = 0; k < fr.n_channels; k++) {
                const ChannelDesc &cd = f.channels_desc[fr.first_channel + k];
                LlChannelDev d{};
                d.off = base + cd.off;
                d.out_off = scratch;
                d.len = cd.len;
                d.samples = fr.samples;
                d.n_coeffs = cd.n_coeffs;
                d.shift_bits = cd.shift_bits;
                d.rice_k = cd.rice_k;
                memcpy(d.coeffs, cd.coeffs, sizeof d.coeffs);
                if (k < 2) fd.scratch_off[k] = scratch;
                scratch += fr.samples;
                chs.push_back(d);
            }
            out_sf += fr.samples;
            if (fr.samples > max_samples) max_samples = fr.samples;
            frs.push_back(fd);
        }
    }
};

// Enqueue the decode of `w` on the ctx stream: integers per wrapper into a scratch, then mid/side, interleave and the
// 1/32767 scale into d_out (f32, nullable) / d_out_i32 (nullable), both out_sf * nch elements. Returns when the
// kernels have run.
static int ll_decode_device(flo_ctx *c, const LlWork &w, const uint8_t *d_bytes, int nch, float *d_out, int *d_out_i32) {
    const size_t n_out = (size_t)w.out_sf * (size_t)nch;
    if (!n_out) return FLO_OK;
    const auto t_enter = std::chrono::steady_clock::now();
    DevMem d_desc, d_scr, d_tabs, d_ent;
    QuiesceOnExit quiesce_d_desc(c);
    int rc;
    // Rice tiles per wrapper; wrappers the parallel form does not take (rice.rs k > 14; coefficient sums or shifts
    // that would leave the exact range of the f64 recurrence: it holds r * 2^shift + sum c * s, |r|, |s| < 2^31, in
    // 53 bits) go to the serial kernel
    std::vector<unsigned int> tile0(w.chs.size() + 1, 0);
    std::vector<int> serial(w.chs.size(), 0);
    std::vector<unsigned int> others;
    unsigned max_tiles = 0;
    const bool force_serial = getenv("FLO_LL_DECODE_SERIAL") != nullptr;
    for (size_t i = 0; i < w.chs.size(); i++) {
        const LlChannelDev &d = w.chs[i];
        const bool rice = d.len > 0 && (d.n_coeffs > 0 || d.shift_bits >= 128);
        long long csum = 0;
        for (unsigned q = 0; q < d.n_coeffs; q++) csum += d.coeffs[q] < 0 ? -(long long)d.coeffs[q] : (long long)d.coeffs[q];
        if (force_serial || (rice && d.rice_k > kRiceMaxK) || csum >= (1ll << 21) || (d.n_coeffs && (d.shift_bits & 63u) > 20u)) serial[i] = 1;
        if (rice && d.len > 16u * 1024u * (unsigned)kRiceTileBits) serial[i] = 1;   // the tile stages put a wrapper's tiles (four per workgroup at least) in gridDim.y (<= 65535)
        if (!(d.n_coeffs > 0 && d.n_coeffs <= 12 && d.len > 0 && d.samples > d.n_coeffs)) others.push_back((unsigned)i);   // (what ll_predict's row form does not take)
        const unsigned nt = rice && !serial[i] ? (d.len + (unsigned)kRiceTileBits / 8u - 1u) / ((unsigned)kRiceTileBits / 8u) : 0u;
        tile0[i + 1] = tile0[i] + nt;
        if (nt > max_tiles) max_tiles = nt;
    }
    if (getenv("FLO_TRACE")) {
        size_t lpc = 0, lpc8 = 0, fixed = 0, other = 0, ser = 0;
        for (size_t i = 0; i < w.chs.size(); i++) {
            const LlChannelDev &d = w.chs[i];
            if (serial[i]) ser++;
            else if (d.n_coeffs && d.len) (d.n_coeffs <= 8 ? lpc8 : lpc)++;
            else if (d.len && d.shift_bits >= 128) fixed++;
            else other++;
        }
        fprintf(stderr, "[flo] ll decode: %zu wrappers: LPC order <= 8 %zu, order 9..12 %zu, fixed %zu, raw/silent/empty %zu, serial %zu\n", w.chs.size(), lpc8,
                lpc, fixed, other, ser);
    }
    // the four descriptor arrays go up as ONE copy out of pinned memory, queued in front of the kernels (four copies out
    // of pageable vectors each held the host until the driver had staged them: 0.15 ms of an idle device per call)
    auto up256 = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_ch = 0, o_fr = up256(w.chs.size() * sizeof(LlChannelDev)), o_t0 = o_fr + up256(w.frs.size() * sizeof(LlFrameDev)),
                 o_ser = o_t0 + up256(tile0.size() * sizeof(unsigned int)), o_oth = o_ser + up256(serial.size() * sizeof(int)),
                 desc_bytes = o_oth + up256(others.size() * sizeof(unsigned int));
    if ((rc = ctx_stager(c)) != FLO_OK) return rc;
    {
        std::string perr;
        uint8_t *pin = (uint8_t *)stager_pinned(c->stager, desc_bytes, perr);
        if (!pin) return fail(c, FLO_ERR_NOMEM, perr);
        memcpy(pin + o_ch, w.chs.data(), w.chs.size() * sizeof(LlChannelDev));
        memcpy(pin + o_fr, w.frs.data(), w.frs.size() * sizeof(LlFrameDev));
        memcpy(pin + o_t0, tile0.data(), tile0.size() * sizeof(unsigned int));
        memcpy(pin + o_ser, serial.data(), serial.size() * sizeof(int));
        memcpy(pin + o_oth, others.data(), others.size() * sizeof(unsigned int));
        HIPCHK(c, pool_alloc(&d_desc.p, desc_bytes));
        HIPCHK(c, hipMemcpyAsync(d_desc.p, pin, desc_bytes, hipMemcpyHostToDevice, c->stream));   // (read before this function's final synchronise)
    }
    LlChannelDev *const d_ch = reinterpret_cast<LlChannelDev *>(d_desc.as<uint8_t>() + o_ch);
    LlFrameDev *const d_fr = reinterpret_cast<LlFrameDev *>(d_desc.as<uint8_t>() + o_fr);
    unsigned int *const d_t0 = reinterpret_cast<unsigned int *>(d_desc.as<uint8_t>() + o_t0);
    int *const d_ser = reinterpret_cast<int *>(d_desc.as<uint8_t>() + o_ser);
    const unsigned int *const d_oth = reinterpret_cast<const unsigned int *>(d_desc.as<uint8_t>() + o_oth);
    const size_t tiles = tile0.back();
    hipError_t e = pool_alloc(&d_scr.p, w.scratch ? w.scratch * sizeof(int) : 16);
    if (e == hipSuccess) e = pool_alloc(&d_tabs.p, tiles ? tiles * kRiceStates * sizeof(unsigned int) : 16);
    if (e == hipSuccess) e = pool_alloc(&d_ent.p, tiles ? tiles * sizeof(uint2) : 16);
    // the output is cleared only when some frame carries fewer channels than the file (ll_finish writes every sample
    // of every channel a frame has; the scratch needs no clearing: each wrapper's kernels write all of its samples)
    bool partial = false;
    for (const LlFrameDev &fd : w.frs)
        if ((int)fd.n_channels < nch) partial = true;
    if (e == hipSuccess && d_out && partial) e = hipMemsetAsync(d_out, 0, n_out * sizeof(float), c->stream);
    if (e == hipSuccess && d_out_i32 && partial) e = hipMemsetAsync(d_out_i32, 0, n_out * sizeof(int), c->stream);
    if (e != hipSuccess) return fail(c, FLO_ERR_NOMEM, std::string("decode buffers: ") + hipGetErrorString(e));
    LlParArgs P{d_bytes, d_ch, (unsigned)w.chs.size(), d_scr.as<int>(), d_t0, d_tabs.as<unsigned int>(), d_ent.as<uint2>(), d_ser, d_oth, (unsigned)others.size()};
    rc = timed_launch(c, "ll_decode_parallel", [&] { return launch_ll_decode_parallel(P, max_tiles, c->stream); });
    if (rc != FLO_OK) return rc;
    LlDecArgs A{d_bytes, d_ch, (unsigned)w.chs.size(), d_scr.as<int>(), d_ser};
    rc = timed_launch(c, "ll_decode", [&] { return launch_ll_decode(A, c->stream); });
    if (rc != FLO_OK) return rc;
    LlFinishArgs F{d_fr, d_ch, (unsigned)w.frs.size(), nch, d_scr.as<int>(), d_out, d_out_i32};
    rc = timed_launch(c, "ll_finish", [&] { return launch_ll_finish(F, w.max_samples, c->stream); });
    if (rc != FLO_OK) return rc;
    if (getenv("FLO_TRACE"))
        fprintf(stderr, "[flo] ll decode: host time until the last launch %.0f us\n",
                (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_enter).count() / 1e3);
    // the descriptor uploads read pageable vectors that die with this frame, and the temporaries go back to the pool
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FLO_OK;
}

// libflo::decode (lib.rs:296-315): parse on the host (a few bytes per frame), decode on the device.
static int decode_impl(flo_ctx *c, const uint8_t *flo, size_t len, float **pcm, int32_t **pcm_i32, size_t *n_interleaved,
                       uint32_t *sample_rate, uint8_t *channels) {
    if (!c || !flo || !n_interleaved || (!pcm && !pcm_i32)) return fail(c, FLO_ERR_ARG, "null argument");
    if (pcm) *pcm = nullptr;
    if (pcm_i32) *pcm_i32 = nullptr;
    *n_interleaved = 0;
    HIPCHK(c, hipSetDevice(c->device));
    ParsedFile f;
    const char *perr = "";
    if (parse_file(flo, len, f, &perr) != 0) return fail(c, FLO_ERR_FORMAT, perr);
    if (sample_rate) *sample_rate = f.sample_rate;
    if (channels) *channels = f.channels;
    const int nch = f.channels;
    DevMem d_bytes;
    QuiesceOnExit quiesce_d_bytes(c);
    HIPCHK(c, pool_alloc(&d_bytes.p, len + 32));
    {
        int rc = ctx_stager(c);
        if (rc != FLO_OK) return rc;
        std::string uerr;
        if (stager_upload(c->stager, {{d_bytes.p, flo, len}}, c->stream, uerr) != 0) return fail(c, FLO_ERR_DEVICE, uerr);
    }

    if (f.is_transform) {
        if (pcm_i32 && !pcm) return fail(c, FLO_ERR_ARG, "integer output exists for lossless files only");
        // decode_transform_file (lib.rs:325-352): frames without channels are skipped, the first decoded frame is dropped
        std::vector<unsigned long long> blob_off;
        std::vector<unsigned int> blob_len;
        for (const FrameDesc &fr : f.frames) {
            if (!fr.n_channels) continue;
            const ChannelDesc &cd = f.channels_desc[fr.first_channel];
            blob_off.push_back(cd.off);
            blob_len.push_back(cd.len);
        }
        const size_t nf = blob_off.size();
        const size_t n_out = nf > 1 ? (nf - 1) * 1024 * (size_t)nch : 0;
        float *host = (float *)alloc_result(n_out * sizeof(float));
        if (!host) return fail(c, FLO_ERR_NOMEM, "out of host memory");
        if (nf) {
            if (nch == 0) {
                free(host);
                return fail(c, FLO_ERR_FORMAT, "Failed to deserialize transform frame");
            }
            TableSet *ts;
            int rc = get_tables(c, f.sample_rate, 0.5f, &ts);
            if (rc != FLO_OK) {
                free(host);
                return rc;
            }
            DevMem d_off, d_len, d_c0, d_cn, d_co, d_out, d_err;
            QuiesceOnExit quiesce_d_off(c);
            std::vector<unsigned long long> c0{0}, co{0};
            std::vector<unsigned int> cn{(unsigned int)nf};
            std::vector<int> zero{0};
            if ((rc = upload(c, d_off, blob_off)) || (rc = upload(c, d_len, blob_len)) || (rc = upload(c, d_c0, c0)) ||
                (rc = upload(c, d_cn, cn)) || (rc = upload(c, d_co, co)) || (rc = upload(c, d_err, zero))) {
                free(host);
                return rc;
            }
            hipError_t e = pool_alloc(&d_out.p, n_out ? n_out * sizeof(float) : 16);
            // (no clearing: the decode kernel writes every sample of every output block exactly once)
            if (e != hipSuccess) {
                free(host);
                return fail(c, FLO_ERR_NOMEM, std::string("decode output: ") + hipGetErrorString(e));
            }
            LossyDecArgs A{};
            A.T = ts->dev;
            A.window = ts->dev_window;
            A.bytes = d_bytes.as<uint8_t>();
            A.blob_off = d_off.as<unsigned long long>();
            A.blob_len = d_len.as<unsigned int>();
            A.clip_frame0 = d_c0.as<unsigned long long>();
            A.clip_frames = d_cn.as<unsigned int>();
            A.clip_out = d_co.as<unsigned long long>();
            A.n_clips = 1;
            A.channels = nch;
            A.out = d_out.as<float>();
            A.error = d_err.as<int>();
            rc = timed_launch(c, "lossy_decode", [&] { return launch_lossy_decode(A, (unsigned)nf, c->stream); });
            int herr = 0;
            if (rc == FLO_OK) {
                e = hipMemcpyAsync(&herr, d_err.p, sizeof(int), hipMemcpyDeviceToHost, c->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                if (e != hipSuccess) rc = fail(c, FLO_ERR_DEVICE, std::string("lossy decode: ") + hipGetErrorString(e));
                if (rc == FLO_OK && n_out && !herr) rc = download(c, host, d_out.p, n_out * sizeof(float));
            }
            if (rc == FLO_OK && herr) rc = fail(c, FLO_ERR_FORMAT, "Failed to deserialize transform frame");
            if (rc != FLO_OK) {
                free(host);
                return rc;
            }
        }
        *pcm = host;
        *n_interleaved = n_out;
        return FLO_OK;
    }

    // lossless (lossless/decoder.rs:21-72)
    LlWork w;
    w.add(f, 0, nch);
    const size_t n_out = nch ? (size_t)w.out_sf * (size_t)nch : 0;
    float *host = pcm ? (float *)alloc_result(n_out * sizeof(float)) : nullptr;
    int32_t *host_i = pcm_i32 ? (int32_t *)alloc_result(n_out * sizeof(int32_t)) : nullptr;
    auto bail = [&](int rc) {
        free(host);
        free(host_i);
        return rc;
    };
    if ((pcm && !host) || (pcm_i32 && !host_i)) return bail(fail(c, FLO_ERR_NOMEM, "out of host memory"));
    if (n_out) {
        DevMem d_out, d_outi;
        QuiesceOnExit quiesce_d_out(c);
        int rc;
        hipError_t e = hipSuccess;
        if (host) e = pool_alloc(&d_out.p, n_out * sizeof(float));
        if (e == hipSuccess && host_i) e = pool_alloc(&d_outi.p, n_out * sizeof(int));
        if (e != hipSuccess) return bail(fail(c, FLO_ERR_NOMEM, std::string("decode buffers: ") + hipGetErrorString(e)));
        if ((rc = ll_decode_device(c, w, d_bytes.as<uint8_t>(), nch, d_out.as<float>(), d_outi.as<int>())) != FLO_OK) return bail(rc);
        if (host && (rc = download(c, host, d_out.p, n_out * sizeof(float))) != FLO_OK) return bail(rc);
        if (host_i && (rc = download(c, host_i, d_outi.p, n_out * sizeof(int))) != FLO_OK) return bail(rc);
        e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return bail(fail(c, FLO_ERR_DEVICE, std::string("lossless decode: ") + hipGetErrorString(e)));
    }
    if (pcm) *pcm = host;
    if (pcm_i32) *pcm_i32 = host_i;
    *n_interleaved = n_out;
    return FLO_OK;
}

// Lossless batches: the finished files stay where the encoder left them in HBM, and what a reader would find in them
// is known from the encoder's own frame and channel records (lossless_describe): nothing is read back or parsed, every
// wrapper of every clip is decoded in one set of launches.
static int batch_decode_lossless(flo_batch *b, float *dst, size_t dst_cap, uint64_t *offsets) {
    flo_ctx *c = b->ctx;
    std::vector<LosslessFrameInfo> fr;
    std::vector<LosslessWrapperInfo> wr;
    const uint8_t *base = nullptr;
    std::string err;
    const bool trace = getenv("FLO_TRACE") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    if (lossless_describe(b->ll, fr, wr, &base, err) != 0) return fail(c, FLO_ERR_STATE, err);
    const auto t_1 = std::chrono::steady_clock::now();
    LlWork w;
    w.chs.reserve(wr.size());
    w.frs.reserve(fr.size());
    for (size_t i = 0; i < b->n_clips; i++) offsets[i] = 0;
    uint32_t cur = 0xFFFFFFFFu;
    for (const LosslessFrameInfo &f : fr) {
        if (f.clip != cur) {   // frames are in clip order: a clip's PCM starts where its first frame's does
            cur = f.clip;
            if (cur < b->n_clips) offsets[cur] = w.out_sf * b->ch;
        }
        LlFrameDev fd{};
        fd.out_off = w.out_sf;
        fd.first_channel = (unsigned)w.chs.size();
        fd.n_channels = f.n_wrappers;
        fd.samples = f.samples;
        fd.mid_side = (b->ch == 2 && (f.flags & 1)) ? 1u : 0u;
        for (uint32_t k = 0; k < f.n_wrappers; k++) {
            const LosslessWrapperInfo &x = wr[f.first_wrapper + k];
            LlChannelDev d{};
            d.off = x.off;
            d.out_off = w.scratch;
            d.len = x.len;
            d.samples = f.samples;
            d.n_coeffs = x.n_coeffs;
            d.shift_bits = x.shift_bits;
            d.rice_k = x.rice_k;
            memcpy(d.coeffs, x.coeffs, sizeof d.coeffs);
            if (k < 2) fd.scratch_off[k] = w.scratch;
            w.scratch += f.samples;
            w.chs.push_back(d);
        }
        w.out_sf += f.samples;
        if (f.samples > w.max_samples) w.max_samples = f.samples;
        w.frs.push_back(fd);
    }
    // clips without frames (empty input) keep the offset of whatever follows them
    {
        uint64_t next = w.out_sf * b->ch;
        std::vector<char> has(b->n_clips, 0);
        for (const LosslessFrameInfo &f : fr)
            if (f.clip < b->n_clips) has[f.clip] = 1;
        for (size_t i = b->n_clips; i-- > 0;) {
            if (has[i]) next = offsets[i];
            else offsets[i] = next;
        }
    }
    const uint64_t total = w.out_sf * b->ch;
    if (total > dst_cap) return fail(c, FLO_ERR_ARG, "destination too small for the decoded batch");
    const auto t_2 = std::chrono::steady_clock::now();
    const int rc = ll_decode_device(c, w, base, b->ch, dst, nullptr);
    if (trace) {
        const auto t_3 = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b2) { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(b2 - a).count() / 1e3; };
        fprintf(stderr, "[flo] batch lossless decode: describe %.0f us, descriptors %.0f us, device %.0f us\n", us(t_0, t_1), us(t_1, t_2), us(t_2, t_3));
    }
    return rc;
}

// Decode every clip of an encoded batch from its device bitstreams (no host round trip of the payload).
extern "C" int flo_batch_decode(flo_batch *b, float *dst, size_t dst_cap, uint64_t *offsets) {
    if (!b || !offsets || (!dst && dst_cap)) return FLO_ERR_ARG;
    flo_ctx *c = b->ctx;
    if (!b->synced) return fail(c, FLO_ERR_STATE, "call flo_batch_encode + flo_batch_sync first");
    HIPCHK(c, hipSetDevice(c->device));
    if (b->mode != FLO_MODE_LOSSY) return batch_decode_lossless(b, dst, dst_cap, offsets);
    if (b->h_frame_size.size() != b->total_frames) {
        b->h_frame_size.assign(b->total_frames, 0);
        if (b->total_frames)
            HIPCHK(c, hipMemcpy(b->h_frame_size.data(), b->d_frame_size, b->total_frames * 4, hipMemcpyDeviceToHost));
    }
    std::vector<unsigned long long> blob_off(b->total_frames), c0(b->n_clips), co(b->n_clips);
    std::vector<unsigned int> blob_len(b->total_frames), cn(b->n_clips);
    uint64_t total = 0;
    unsigned max_hops = 0;
    for (size_t i = 0; i < b->n_clips; i++) {
        uint64_t off = b->out_off[i];
        for (uint32_t h = 0; h < b->hops[i]; h++) {
            const uint32_t fs = b->h_frame_size[b->clip_frame0[i] + h];
            if (fs < 10) return fail(c, FLO_ERR_STATE, "batch holds a frame shorter than its header");
            blob_off[b->clip_frame0[i] + h] = off + 10;   // [type][u32 samples][flags][u32 size] (writer.rs:236-254)
            blob_len[b->clip_frame0[i] + h] = fs - 10;
            off += fs;
        }
        c0[i] = b->clip_frame0[i];
        cn[i] = b->hops[i];
        co[i] = total;
        offsets[i] = total;
        total += b->hops[i] > 1 ? (uint64_t)(b->hops[i] - 1) * 1024 * b->ch : 0;
        if (b->hops[i] > max_hops) max_hops = b->hops[i];
    }
    if (total > dst_cap) return fail(c, FLO_ERR_ARG, "destination too small for the decoded batch");
    if (!total) return FLO_OK;
    DevMem d_off, d_len, d_c0, d_cn, d_co, d_err;
    QuiesceOnExit quiesce_d_off(c);
    std::vector<int> zero{0};
    int rc;
    if ((rc = upload(c, d_off, blob_off)) || (rc = upload(c, d_len, blob_len)) || (rc = upload(c, d_c0, c0)) ||
        (rc = upload(c, d_cn, cn)) || (rc = upload(c, d_co, co)) || (rc = upload(c, d_err, zero)))
        return rc;
    // (dst needs no clearing: the decode kernel writes every sample of every output block exactly once)
    LossyDecArgs A{};
    A.T = b->ts->dev;
    A.window = b->ts->dev_window;
    A.bytes = b->d_out;
    A.blob_off = d_off.as<unsigned long long>();
    A.blob_len = d_len.as<unsigned int>();
    A.clip_frame0 = d_c0.as<unsigned long long>();
    A.clip_frames = d_cn.as<unsigned int>();
    A.clip_out = d_co.as<unsigned long long>();
    A.n_clips = (int)b->n_clips;
    A.channels = b->ch;
    A.out = dst;
    A.error = d_err.as<int>();
    rc = timed_launch(c, "lossy_decode", [&] { return launch_lossy_decode(A, max_hops, c->stream); });
    if (rc != FLO_OK) return rc;
    int herr = 0;
    HIPCHK(c, hipMemcpyAsync(&herr, d_err.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (herr) return fail(c, FLO_ERR_FORMAT, "Failed to deserialize transform frame");
    return FLO_OK;
}

extern "C" int flo_decode(flo_ctx *c, const uint8_t *flo, size_t len, float **pcm, size_t *n_interleaved,
                          uint32_t *sample_rate, uint8_t *channels) {
    return decode_impl(c, flo, len, pcm, nullptr, n_interleaved, sample_rate, channels);
}
extern "C" int flo_decode_lossless_i32(flo_ctx *c, const uint8_t *flo, size_t len, int32_t **pcm, size_t *n_interleaved,
                                       uint32_t *sample_rate, uint8_t *channels) {
    return decode_impl(c, flo, len, nullptr, pcm, n_interleaved, sample_rate, channels);
}

// ------------------------------------------------------------------------------------------------ multi-GPU
// One process per GPU. Clips shard across ranks with no data-path collective during the encode; the one
// exchange step per batch is a variable-size gather of every rank's finished .flo files to the root, written directly
// against RCCL: ncclAllGather of one u64 per rank (the packed size), then grouped ncclSend / ncclRecv - on the fully
// connected xGMI node every peer has its own link to the root, so the seven transfers run in parallel where a ring
// collective would be bound by one link. Everything runs on a side stream of its own and is double-buffered: the
// transfer of step k overlaps the encode of step k + 1. No host synchronisation sits on that path: the sizes of step k
// travel to pinned host memory asynchronously and are only read when step k + 1 is submitted (by then they have long
// arrived), which is when the transfers of step k are posted; flo_dist_gather_flush posts and awaits the last ones.
// The ordering logic (slot parity, deferred posting, buffer growth) is the DistEngine template of dist_engine.hpp; here it
// is bound to HIP streams and RCCL. tests/native/dist_engine_test.cpp runs the same template over sockets with several
// ranks on the CPU.
#define NCCLRC(ctx, expr)                                                                               \
    do {                                                                                                \
        ncclResult_t r_ = (expr);                                                                       \
        if (r_ != ncclSuccess) return fail(ctx, FLO_ERR_DEVICE, std::string(#expr) + ": " + ncclGetErrorString(r_)); \
    } while (0)

struct RcclBackend {
    struct Buffer {
        uint8_t *p = nullptr;
        size_t cap = 0;
    };
    flo_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int world = 1;
    hipStream_t cs = nullptr;                 // communication stream
    uint64_t *d_sizes[2] = {nullptr, nullptr};   // [world] device
    uint64_t *h_sizes[2] = {nullptr, nullptr};   // [world] pinned host
    uint64_t *h_mine[2] = {nullptr, nullptr};    // pinned host: this rank's packed size
    uint64_t *d_mine[2] = {nullptr, nullptr};
    hipEvent_t ev_packed[2] = {nullptr, nullptr}, ev_sizes[2] = {nullptr, nullptr}, ev_moved[2] = {nullptr, nullptr};
    std::vector<uint64_t> pack_off;           // scratch: per-clip offsets of the last pack

    // (re)allocate a device buffer to hold `need` bytes; growing waits for the work that may still use the old one
    int reserve(Buffer &b, size_t need) {
        if (b.cap >= need) return FLO_OK;
        flo_ctx *c = ctx;
        HIPCHK(c, hipStreamSynchronize(cs));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (b.p) HIPCHK(c, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
        const size_t want = need + need / 4 + 4096;
        hipError_t e = hipMalloc(&b.p, want);
        if (e != hipSuccess) return fail(c, FLO_ERR_NOMEM, std::string("gather buffer: ") + hipGetErrorString(e));
        b.cap = want;
        return FLO_OK;
    }
    int payload_bytes(void *batch, uint64_t *need) {
        flo_batch *b = (flo_batch *)batch;
        const uint8_t *base;
        const uint64_t *offs, *sizes;
        int rc = flo_batch_device_files(b, &base, &offs, &sizes);
        if (rc != FLO_OK) return rc;
        uint64_t n = 0;
        for (size_t i = 0; i < b->n_clips; i++) n += (sizes[i] + 15) & ~(uint64_t)15;
        *need = n;
        return FLO_OK;
    }
    int pack(void *batch, Buffer &dst, uint64_t *bytes) {
        flo_batch *b = (flo_batch *)batch;
        pack_off.resize(b->n_clips + 1);
        int rc = flo_batch_pack_files(b, dst.p, dst.cap, pack_off.data());
        if (rc != FLO_OK) return rc;
        *bytes = pack_off[b->n_clips];
        return FLO_OK;
    }
    int wait_moved_before_pack(int s) {
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ev_moved[s], 0));
        return FLO_OK;
    }
    int mark_packed(int s) {
        HIPCHK(ctx, hipEventRecord(ev_packed[s], ctx->stream));
        return FLO_OK;
    }
    int sizes_exchange(int s, uint64_t mine) {
        flo_ctx *c = ctx;
        *h_mine[s] = mine;
        HIPCHK(c, hipMemcpyAsync(d_mine[s], h_mine[s], 8, hipMemcpyHostToDevice, cs));
        NCCLRC(c, ncclAllGather(d_mine[s], d_sizes[s], 1, ncclUint64, comm, cs));
        HIPCHK(c, hipMemcpyAsync(h_sizes[s], d_sizes[s], (size_t)world * 8, hipMemcpyDeviceToHost, cs));
        HIPCHK(c, hipEventRecord(ev_sizes[s], cs));
        return FLO_OK;
    }
    int sizes_wait(int s, const uint64_t **sizes) {
        HIPCHK(ctx, hipEventSynchronize(ev_sizes[s]));
        *sizes = h_sizes[s];
        return FLO_OK;
    }
    int comm_waits_for_pack(int s) {
        HIPCHK(ctx, hipStreamWaitEvent(cs, ev_packed[s], 0));
        return FLO_OK;
    }
    int copy_own(Buffer &dst, size_t off, Buffer &src, size_t n) {
        HIPCHK(ctx, hipMemcpyAsync(dst.p + off, src.p, n, hipMemcpyDeviceToDevice, cs));
        return FLO_OK;
    }
    int group_begin() {
        NCCLRC(ctx, ncclGroupStart());
        return FLO_OK;
    }
    int recv(Buffer &dst, size_t off, size_t n, int peer) {
        NCCLRC(ctx, ncclRecv(dst.p + off, n, ncclUint8, peer, comm, cs));
        return FLO_OK;
    }
    int send(Buffer &src, size_t n, int peer) {
        NCCLRC(ctx, ncclSend(src.p, n, ncclUint8, peer, comm, cs));
        return FLO_OK;
    }
    int group_end() {
        NCCLRC(ctx, ncclGroupEnd());
        return FLO_OK;
    }
    int mark_moved(int s) {
        HIPCHK(ctx, hipEventRecord(ev_moved[s], cs));
        return FLO_OK;
    }
    int drain() {
        HIPCHK(ctx, hipStreamSynchronize(cs));
        return FLO_OK;
    }
};

// Second exchange mode (flo_dist_table_*): the files stay where they were made, one ncclAllGather tells every rank where
// each file of every rank lies (offset in its owner's device buffer), how long it is and the CRC32 of its DATA chunk.
struct TableSlot {
    uint64_t *h_mine = nullptr, *d_mine = nullptr, *d_all = nullptr, *h_all = nullptr;   // pinned / device rows
    size_t words = 0;                 // capacity of one row, in u64
    hipEvent_t done = nullptr;
    bool used = false;
};
struct flo_dist {
    flo_ctx *ctx = nullptr;
    RcclBackend be;
    flo::DistEngine<RcclBackend> eng;
    TableSlot tab[2];
    uint64_t tab_steps = 0;
    size_t tab_max = 0;               // max_clips of the last submit
    bool defaulted_reserve = false;   // flo_dist_create set the context's CU reservation (and flo_dist_destroy takes it back)
};

extern "C" int flo_dist_unique_id(uint8_t *id) {
    if (!id) return FLO_ERR_ARG;
    static_assert(FLO_DIST_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId u;
    ncclResult_t r = ncclGetUniqueId(&u);
    if (r != ncclSuccess) {
        g_create_err = std::string("ncclGetUniqueId: ") + ncclGetErrorString(r);
        return FLO_ERR_DEVICE;
    }
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return FLO_OK;
}

extern "C" void flo_dist_destroy(flo_dist *d) {
    if (!d) return;
    RcclBackend &be = d->be;
    hipSetDevice(d->ctx->device);
    if (be.cs) hipStreamSynchronize(be.cs);
    for (int s = 0; s < 2; s++) {
        if (d->eng.send[s].p) hipFree(d->eng.send[s].p);
        if (d->eng.recv[s].p) hipFree(d->eng.recv[s].p);
        if (be.d_sizes[s]) hipFree(be.d_sizes[s]);
        if (be.d_mine[s]) hipFree(be.d_mine[s]);
        if (be.h_sizes[s]) hipHostFree(be.h_sizes[s]);
        if (be.h_mine[s]) hipHostFree(be.h_mine[s]);
        if (be.ev_packed[s]) hipEventDestroy(be.ev_packed[s]);
        if (be.ev_sizes[s]) hipEventDestroy(be.ev_sizes[s]);
        if (be.ev_moved[s]) hipEventDestroy(be.ev_moved[s]);
    }
    for (auto &t : d->tab) {
        if (t.h_mine) hipHostFree(t.h_mine);
        if (t.h_all) hipHostFree(t.h_all);
        if (t.d_mine) hipFree(t.d_mine);
        if (t.d_all) hipFree(t.d_all);
        if (t.done) hipEventDestroy(t.done);
    }
    if (be.comm) ncclCommDestroy(be.comm);
    if (be.cs) hipStreamDestroy(be.cs);
    if (d->defaulted_reserve) d->ctx->reserve_cus = -1;   // later single-GPU encodes on this context get every CU back
    delete d;
}

extern "C" int flo_dist_create(flo_ctx *c, const uint8_t *id, int rank, int world, int root, flo_dist **out) {
    if (!c || !id || !out || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world) return FLO_ERR_ARG;
    *out = nullptr;
    HIPCHK(c, hipSetDevice(c->device));
    flo_dist *d = new flo_dist();
    d->ctx = c;
    RcclBackend &be = d->be;
    be.ctx = c;
    be.world = world;
    d->eng.init(&be, rank, world, root);
    auto bail = [&](int rc) {
        flo_dist_destroy(d);
        return rc;
    };
    if (hipStreamCreateWithFlags(&be.cs, hipStreamNonBlocking) != hipSuccess) return bail(fail(c, FLO_ERR_DEVICE, "hipStreamCreate (communication stream)"));
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    ncclResult_t r = ncclCommInitRank(&be.comm, world, u, rank);
    if (r != ncclSuccess) return bail(fail(c, FLO_ERR_DEVICE, std::string("ncclCommInitRank: ") + ncclGetErrorString(r)));
    for (int s = 0; s < 2; s++) {
        if (hipMalloc(&be.d_sizes[s], (size_t)world * 8) != hipSuccess || hipMalloc(&be.d_mine[s], 8) != hipSuccess ||
            hipHostMalloc(&be.h_sizes[s], (size_t)world * 8) != hipSuccess || hipHostMalloc(&be.h_mine[s], 8) != hipSuccess ||
            hipEventCreateWithFlags(&be.ev_packed[s], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&be.ev_sizes[s], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&be.ev_moved[s], hipEventDisableTiming) != hipSuccess)
            return bail(fail(c, FLO_ERR_NOMEM, "flo_dist_create: buffers"));
    }
    // RCCL's send / receive are kernels: with more than one rank the persistent chain kernel leaves a few compute units
    // free for them, or the transfer of step k could not start before the encode of step k + 1 has ended
    // (flo_ctx_reserve_cus; an explicit setting or FLO_RESERVE_CUS wins)
    if (world > 1 && c->reserve_cus < 0) {
        c->reserve_cus = kDefaultReservedCus;
        d->defaulted_reserve = true;
    }
    *out = d;
    return FLO_OK;
}

extern "C" int flo_dist_table_submit(flo_dist *d, flo_batch *b, size_t max_clips) {
    if (!d || !b) return FLO_ERR_ARG;
    flo_ctx *c = d->ctx;
    if (b->ctx != c) return fail(c, FLO_ERR_ARG, "batch and communicator belong to different contexts");
    if (!b->synced) return fail(c, FLO_ERR_STATE, "call flo_batch_encode + flo_batch_sync first");
    if (b->n_clips > max_clips) return fail(c, FLO_ERR_ARG, "flo_dist_table_submit: max_clips is smaller than this rank's clip count");
    HIPCHK(c, hipSetDevice(c->device));
    RcclBackend &be = d->be;
    TableSlot &t = d->tab[d->tab_steps & 1];
    const size_t words = 1 + 3 * max_clips;
    if (t.used) HIPCHK(c, hipEventSynchronize(t.done));   // this slot's previous round trip (two steps ago): long over
    if (t.words < words) {
        if (t.h_mine) hipHostFree(t.h_mine);
        if (t.h_all) hipHostFree(t.h_all);
        if (t.d_mine) hipFree(t.d_mine);
        if (t.d_all) hipFree(t.d_all);
        t.h_mine = t.h_all = t.d_mine = t.d_all = nullptr;
        t.words = 0;
        if (hipHostMalloc(&t.h_mine, words * 8) != hipSuccess || hipHostMalloc(&t.h_all, words * 8 * (size_t)be.world) != hipSuccess ||
            hipMalloc(&t.d_mine, words * 8) != hipSuccess || hipMalloc(&t.d_all, words * 8 * (size_t)be.world) != hipSuccess)
            return fail(c, FLO_ERR_NOMEM, "flo_dist_table_submit: buffers");
        if (!t.done && hipEventCreateWithFlags(&t.done, hipEventDisableTiming) != hipSuccess) return fail(c, FLO_ERR_NOMEM, "flo_dist_table_submit: event");
        t.words = words;
    }
    const uint8_t *base;
    const uint64_t *offs, *sizes;
    int rc = flo_batch_device_files(b, &base, &offs, &sizes);
    if (rc != FLO_OK) return rc;
    memset(t.h_mine, 0, words * 8);
    t.h_mine[0] = b->n_clips;
    for (size_t i = 0; i < b->n_clips; i++) {
        t.h_mine[1 + i] = sizes[i];
        t.h_mine[1 + max_clips + i] = offs[i];
    }
    HIPCHK(c, hipMemcpyAsync(t.d_mine, t.h_mine, words * 8, hipMemcpyHostToDevice, be.cs));
    // (the batch is synced: its files, CRC fields included, are complete; nothing on the encode stream to wait for)
    if (flo::launch_table_crcs(base, (unsigned long long *)t.d_mine, b->n_clips, max_clips, be.cs) != 0)
        return fail(c, FLO_ERR_DEVICE, "flo_dist_table_submit: table kernel");
    NCCLRC(c, ncclAllGather(t.d_mine, t.d_all, words, ncclUint64, be.comm, be.cs));
    HIPCHK(c, hipMemcpyAsync(t.h_all, t.d_all, words * 8 * (size_t)be.world, hipMemcpyDeviceToHost, be.cs));
    HIPCHK(c, hipEventRecord(t.done, be.cs));
    t.used = true;
    d->tab_max = max_clips;
    d->tab_steps++;
    return FLO_OK;
}

extern "C" int flo_dist_table_flush(flo_dist *d) {
    if (!d) return FLO_ERR_ARG;
    HIPCHK(d->ctx, hipSetDevice(d->ctx->device));
    for (auto &t : d->tab)
        if (t.used) HIPCHK(d->ctx, hipEventSynchronize(t.done));
    return FLO_OK;
}

extern "C" int flo_dist_table_result(flo_dist *d, const uint64_t **rows, size_t *row_words, size_t *max_clips) {
    if (!d) return FLO_ERR_ARG;
    if (!d->tab_steps) return fail(d->ctx, FLO_ERR_STATE, "no table has been submitted yet");
    const TableSlot &t = d->tab[(d->tab_steps - 1) & 1];
    HIPCHK(d->ctx, hipEventSynchronize(t.done));
    if (rows) *rows = t.h_all;
    if (row_words) *row_words = 1 + 3 * d->tab_max;
    if (max_clips) *max_clips = d->tab_max;
    return FLO_OK;
}

extern "C" int flo_dist_gather_submit(flo_dist *d, flo_batch *b) {
    if (!d || !b) return FLO_ERR_ARG;
    flo_ctx *c = d->ctx;
    if (b->ctx != c) return fail(c, FLO_ERR_ARG, "batch and communicator belong to different contexts");
    if (!b->synced) return fail(c, FLO_ERR_STATE, "call flo_batch_encode + flo_batch_sync first");
    HIPCHK(c, hipSetDevice(c->device));
    return d->eng.submit(b);
}

extern "C" int flo_dist_gather_flush(flo_dist *d) {
    if (!d) return FLO_ERR_ARG;
    HIPCHK(d->ctx, hipSetDevice(d->ctx->device));
    return d->eng.flush();
}

extern "C" int flo_dist_gather_result(flo_dist *d, const uint8_t **base, const uint64_t **rank_offsets,
                                      const uint64_t **rank_sizes) {
    if (!d) return FLO_ERR_ARG;
    if (d->eng.rank != d->eng.root) return fail(d->ctx, FLO_ERR_STATE, "only the root holds the gathered files");
    if (d->eng.res_slot < 0) return fail(d->ctx, FLO_ERR_STATE, "nothing has been gathered yet");
    if (base) *base = d->eng.recv[d->eng.res_slot].p;
    if (rank_offsets) *rank_offsets = d->eng.res_off.data();
    if (rank_sizes) *rank_sizes = d->eng.res_size.data();
    return FLO_OK;
}

extern "C" void *flo_dist_stream(flo_dist *d) { return d ? (void *)d->be.cs : nullptr; }

extern "C" int flo_ctx_reserve_cus(flo_ctx *c, int n) {
    if (!c || n < 0 || n >= c->prop.multiProcessorCount) return FLO_ERR_ARG;
    c->reserve_cus = n;
    return FLO_OK;
}
extern "C" int flo_ctx_upload_path(flo_ctx *c, char *name, size_t cap, double *direct_gbs, double *ring_gbs) {
    if (!c) return FLO_ERR_ARG;
    int rc = ctx_stager(c);
    if (rc != FLO_OK) return rc;
    const char *n = stager_upload_choice(c->stager, direct_gbs, ring_gbs);
    if (name && cap) snprintf(name, cap, "%s", n);
    return FLO_OK;
}
extern "C" int flo_ctx_reserved_cus(flo_ctx *c) { return c ? (c->reserve_cus > 0 ? c->reserve_cus : 0) : -1; }

// ------------------------------------------------------------------------------------------------ streaming encoder
// StreamingEncoder of libflo/src/streaming/encoder.rs on the device library: samples are pushed, complete one-second
// frames are encoded (all frames a push completes go through ONE device batch, where the reference encodes them one
// after the other through temporary files), frames are pulled or assembled into a file. Frame bytes follow the
// reference's encode_frame_data / serialize_channel (encoder.rs:215-257) exactly, including its channel layout
// [rice_parameter][coefficients][residuals], which is not the container writer's.
struct StreamFrame {
    uint32_t index, timestamp_ms, samples;
    std::vector<uint8_t> data;
};
struct flo_stream {
    flo_ctx *ctx = nullptr;
    uint32_t sr = 0;
    uint8_t ch = 0, bit_depth = 16, level = 5;
    std::vector<float> buf;
    std::vector<StreamFrame> pending;
    uint64_t total_samples = 0;
    uint32_t frame_index = 0;
};

extern "C" int flo_stream_create(flo_ctx *c, uint32_t sample_rate, uint8_t channels, uint8_t bit_depth, uint8_t level,
                                 flo_stream **out) {
    if (!c || !out) return FLO_ERR_ARG;
    *out = nullptr;
    if (!sample_rate || !channels) return fail(c, FLO_ERR_ARG, "sample_rate and channels must be non-zero");
    flo_stream *s = new flo_stream();
    s->ctx = c;
    s->sr = sample_rate;
    s->ch = channels;
    s->bit_depth = bit_depth;
    s->level = level > 9 ? 9 : level;   // with_compression: level.min(9) (encoder.rs:51-56)
    *out = s;
    return FLO_OK;
}
extern "C" void flo_stream_destroy(flo_stream *s) { delete s; }
extern "C" size_t flo_stream_pending_samples(const flo_stream *s) { return s ? s->buf.size() / s->ch : 0; }
extern "C" size_t flo_stream_pending_frames(const flo_stream *s) { return s ? s->pending.size() : 0; }

// encode_frame_data for `count` chunks of `per` interleaved samples starting at `src` (encoder.rs:215-241): one
// lossless batch, then every one-frame file is parsed like Reader::read and its first frame re-serialised
static int stream_encode_chunks(flo_stream *s, const float *src, size_t count, size_t per, std::vector<std::vector<uint8_t>> &frames) {
    flo_ctx *c = s->ctx;
    frames.clear();
    if (!count) return FLO_OK;
    std::vector<size_t> n_il(count, per);
    flo_batch *b = nullptr;
    int rc = flo_batch_create(c, FLO_MODE_LOSSLESS, count, n_il.data(), s->sr, s->ch, (float)s->level, &b);
    if (rc != FLO_OK) return rc;
    b->bit_depth = s->bit_depth;
    std::vector<const float *> ptrs(count);
    for (size_t i = 0; i < count; i++) ptrs[i] = src + i * per;
    rc = batch_upload_all(b, ptrs.data());
    if (rc == FLO_OK) rc = flo_batch_encode(b, 0);
    if (rc == FLO_OK) rc = flo_batch_sync(b);
    for (size_t i = 0; i < count && rc == FLO_OK; i++) {
        uint8_t *file = nullptr;
        size_t flen = 0;
        rc = flo_batch_fetch(b, i, nullptr, 0, &file, &flen);
        if (rc != FLO_OK) break;
        ParsedFile pf;
        const char *perr = "";
        if (parse_file(file, flen, pf, &perr) != 0) {
            free(file);
            rc = fail(c, FLO_ERR_FORMAT, perr);
            break;
        }
        if (pf.frames.empty()) {
            free(file);
            rc = fail(c, FLO_ERR_FORMAT, "No frames encoded");
            break;
        }
        const FrameDesc &fr = pf.frames[0];
        std::vector<uint8_t> d;
        auto u32 = [&](uint32_t v) {
            for (int k = 0; k < 4; k++) d.push_back((uint8_t)(v >> (8 * k)));
        };
        d.push_back(fr.type);
        u32(fr.samples);
        d.push_back(fr.flags);
        for (unsigned k = 0; k < fr.n_channels; k++) {
            const ChannelDesc &cd = pf.channels_desc[fr.first_channel + k];
            if (fr.type == 0) {                        // Silence: empty
                u32(0);
            } else if (fr.type == 254 || fr.type == 253) {   // Raw / Transform: the payload as it is
                u32(cd.len);
                d.insert(d.end(), file + cd.off, file + cd.off + cd.len);
            } else {                                   // every other type: [rice_parameter][coeffs][residuals]
                u32(1u + 4u * cd.n_coeffs + cd.len);
                d.push_back(cd.rice_k);
                for (unsigned q = 0; q < cd.n_coeffs; q++) u32((uint32_t)cd.coeffs[q]);
                d.insert(d.end(), file + cd.off, file + cd.off + cd.len);
            }
        }
        free(file);
        frames.push_back(std::move(d));
    }
    flo_batch_destroy(b);
    return rc;
}

extern "C" int flo_stream_push(flo_stream *s, const float *samples, size_t n) {
    if (!s || (n && !samples)) return FLO_ERR_ARG;
    s->buf.insert(s->buf.end(), samples, samples + n);
    const size_t per = (size_t)s->sr * s->ch;
    const size_t count = s->buf.size() / per;
    if (!count) return FLO_OK;
    std::vector<std::vector<uint8_t>> frames;
    int rc = stream_encode_chunks(s, s->buf.data(), count, per, frames);
    if (rc != FLO_OK) return rc;
    for (size_t i = 0; i < count; i++) {
        StreamFrame f;
        f.index = s->frame_index;
        f.timestamp_ms = (uint32_t)((double)s->total_samples / (double)s->sr * 1000.0);
        f.samples = s->sr;
        f.data = std::move(frames[i]);
        s->pending.push_back(std::move(f));
        s->total_samples += s->sr;
        s->frame_index++;
    }
    s->buf.erase(s->buf.begin(), s->buf.begin() + count * per);
    return FLO_OK;
}

static int stream_hand_out(const StreamFrame &f, uint32_t *index, uint32_t *timestamp_ms, uint32_t *samples, uint8_t **data, size_t *len) {
    uint8_t *p = (uint8_t *)malloc(f.data.size() ? f.data.size() : 1);
    if (!p) return -1;
    memcpy(p, f.data.data(), f.data.size());
    if (index) *index = f.index;
    if (timestamp_ms) *timestamp_ms = f.timestamp_ms;
    if (samples) *samples = f.samples;
    *data = p;
    *len = f.data.size();
    return 0;
}

// 1: a frame came out (data is malloc'ed, release with flo_free); 0: none ready
extern "C" int flo_stream_next_frame(flo_stream *s, uint32_t *index, uint32_t *timestamp_ms, uint32_t *samples, uint8_t **data, size_t *len) {
    if (!s || !data || !len) return -1;
    if (s->pending.empty()) return 0;
    if (stream_hand_out(s->pending.front(), index, timestamp_ms, samples, data, len) != 0) return -1;
    s->pending.erase(s->pending.begin());
    return 1;
}

static int stream_flush_frame(flo_stream *s, StreamFrame &f) {   // 1 produced, 0 nothing buffered, < 0 error code negated
    if (s->buf.empty()) return 0;
    std::vector<std::vector<uint8_t>> frames;
    int rc = stream_encode_chunks(s, s->buf.data(), 1, s->buf.size(), frames);
    if (rc != FLO_OK) return -rc;
    const size_t spc = s->buf.size() / s->ch;
    f.index = s->frame_index;
    f.timestamp_ms = (uint32_t)((double)s->total_samples / (double)s->sr * 1000.0);
    f.samples = (uint32_t)spc;
    f.data = std::move(frames[0]);
    s->total_samples += spc;
    s->frame_index++;
    s->buf.clear();
    return 1;
}

// flush (encoder.rs:88-110): the buffered remainder as one (partial) frame, returned, not queued. 1 / 0 as next_frame
extern "C" int flo_stream_flush(flo_stream *s, uint32_t *index, uint32_t *timestamp_ms, uint32_t *samples, uint8_t **data, size_t *len) {
    if (!s || !data || !len) return -1;
    StreamFrame f;
    int r = stream_flush_frame(s, f);
    if (r != 1) return r < 0 ? -1 : 0;
    return stream_hand_out(f, index, timestamp_ms, samples, data, len) == 0 ? 1 : -1;
}

// finalize (encoder.rs:113-185): a complete .flo file from the frames that have not been pulled
extern "C" int flo_stream_finalize(flo_stream *s, const uint8_t *meta, size_t meta_len, uint8_t **out, size_t *out_len) {
    if (!s || !out || !out_len || (meta_len && !meta)) return FLO_ERR_ARG;
    {
        StreamFrame f;
        int r = stream_flush_frame(s, f);
        if (r < 0) return -r;
        if (r == 1) s->pending.push_back(std::move(f));
    }
    std::vector<uint8_t> toc, data;
    auto put = [](std::vector<uint8_t> &v, uint64_t x, int bytes) {
        for (int k = 0; k < bytes; k++) v.push_back((uint8_t)(x >> (8 * k)));
    };
    put(toc, s->pending.size(), 4);
    uint64_t off = 0, total = 0;
    for (const StreamFrame &f : s->pending) {
        put(toc, f.index, 4);
        put(toc, off, 8);
        put(toc, f.data.size(), 4);
        put(toc, f.timestamp_ms, 4);
        off += f.data.size();
        data.insert(data.end(), f.data.begin(), f.data.end());
        total += f.samples;
    }
    std::vector<uint8_t> o = {'F', 'L', 'O', '!', 1, 2, 0, 0};
    put(o, s->sr, 4);
    o.push_back(s->ch);
    o.push_back(s->bit_depth);
    put(o, total, 8);
    o.push_back(s->level);
    put(o, 0, 3);
    put(o, host_crc32(data.data(), data.size()), 4);
    put(o, 66, 8);
    put(o, toc.size(), 8);
    put(o, data.size(), 8);
    put(o, 0, 8);
    put(o, meta_len, 8);
    o.insert(o.end(), toc.begin(), toc.end());
    o.insert(o.end(), data.begin(), data.end());
    if (meta_len) o.insert(o.end(), meta, meta + meta_len);
    uint8_t *p = (uint8_t *)malloc(o.size() ? o.size() : 1);
    if (!p) return fail(s->ctx, FLO_ERR_NOMEM, "malloc failed");
    memcpy(p, o.data(), o.size());
    *out = p;
    *out_len = o.size();
    s->pending.clear();
    return FLO_OK;
}

// ------------------------------------------------------------------------------------------------ analysis metadata
// add_analysis_data_if_missing (lib.rs:219-283) for an empty input META: what libflo::encode / encode_lossy /
// encode_with_bitrate put in front of the encoders. The per-sample work runs on the device (analysis_kernels.hip); what
// is left here is scalar: the K-weighting coefficients (ebu_r128.rs:51-103), the gating of a few hundred block energies
// (:268-318), the u8 scalings of sixteen band energies (analysis.rs:320-341) and the MessagePack framing
// (rmp_serde::to_vec_named of FloMetadata, core/metadata.rs: the fields that are set, in declaration order).
namespace {
struct Mp {
    std::vector<uint8_t> b;
    void u(uint64_t v) {
        if (v < 128) b.push_back((uint8_t)v);
        else if (v < 256) { b.push_back(0xcc); b.push_back((uint8_t)v); }
        else if (v < 65536) { b.push_back(0xcd); b.push_back((uint8_t)(v >> 8)); b.push_back((uint8_t)v); }
        else if (v < 4294967296ull) { b.push_back(0xce); for (int k = 3; k >= 0; k--) b.push_back((uint8_t)(v >> (8 * k))); }
        else { b.push_back(0xcf); for (int k = 7; k >= 0; k--) b.push_back((uint8_t)(v >> (8 * k))); }
    }
    void s(const char *t) {
        const size_t n = strlen(t);
        if (n < 32) b.push_back((uint8_t)(0xa0 | n));
        else { b.push_back(0xd9); b.push_back((uint8_t)n); }
        b.insert(b.end(), t, t + n);
    }
    void f(float x) {
        uint32_t w;
        memcpy(&w, &x, 4);
        b.push_back(0xca);
        for (int k = 3; k >= 0; k--) b.push_back((uint8_t)(w >> (8 * k)));
    }
    void arr(size_t n) {
        if (n < 16) b.push_back((uint8_t)(0x90 | n));
        else if (n < 65536) { b.push_back(0xdc); b.push_back((uint8_t)(n >> 8)); b.push_back((uint8_t)n); }
        else { b.push_back(0xdd); for (int k = 3; k >= 0; k--) b.push_back((uint8_t)(n >> (8 * k))); }
    }
    void bin(const std::vector<uint8_t> &p) {
        const size_t n = p.size();
        if (n < 256) { b.push_back(0xc4); b.push_back((uint8_t)n); }
        else if (n < 65536) { b.push_back(0xc5); b.push_back((uint8_t)(n >> 8)); b.push_back((uint8_t)n); }
        else { b.push_back(0xc6); for (int k = 3; k >= 0; k--) b.push_back((uint8_t)(n >> (8 * k))); }
        b.insert(b.end(), p.begin(), p.end());
    }
};
uint8_t f32_as_u8(float v) {   // Rust `as u8`: saturating, NaN -> 0
    if (!(v == v) || v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}
float max_rust(float a, float b) {
    if (a != a) return b;
    if (b != b) return a;
    return a > b ? a : b;
}
}  // namespace

// pcm_dev: the samples are already on the device (a batch's clip): nothing is staged
static int analyze_impl(flo_ctx *c, const float *pcm, const float *pcm_dev, size_t n, uint32_t sr, uint8_t ch, uint32_t pps, float *peaks,
                        size_t peaks_cap, flo_analysis *out) {
    if (!c || !out || (n && !pcm && !pcm_dev) || !ch || !sr || !pps) return c ? fail(c, FLO_ERR_ARG, "flo_analyze: bad argument") : FLO_ERR_ARG;
    memset(out, 0, sizeof *out);
    out->sample_rate = sr;
    out->channels = ch;
    out->integrated_lufs = -23.0;
    out->loudness_range_lu = 0.0;
    out->true_peak_dbtp = -150.0;
    out->sample_peak_dbfs = -150.0;
    out->length_ms = (uint64_t)((double)(n / ch) / (double)sr * 1000.0);
    HIPCHK(c, hipSetDevice(c->device));
    AnalysisArgs A{};
    A.n = n;
    A.sample_rate = sr;
    A.channels = ch;
    A.samples_per_peak = (double)sr / (double)pps;
    if (n) {
        const double tp = std::ceil((double)n / (A.samples_per_peak * (double)ch));
        // peak windows that start inside the clip (analysis.rs:54-64: the loop breaks at the first one that does not)
        unsigned np = 0;
        const unsigned cap = tp > 0 ? (tp > 4e9 ? 4000000000u : (unsigned)tp) : 0u;
        while (np < cap && (uint64_t)((double)np * A.samples_per_peak) * ch < n) np++;
        A.n_peaks = np;
    }
    out->n_peaks = A.n_peaks;
    if (!n) return FLO_OK;
    if (peaks_cap < A.n_peaks) return fail(c, FLO_ERR_ARG, "flo_analyze: peak buffer too small");
    // K-weighting (ebu_r128.rs:51-103) and block geometry (:190-192, :236-262)
    {
        const double rate = (double)sr;
        const double f0 = 1681.974450955533, g_db = 3.999843853973347, q = 0.7071752369554196;
        const double k = std::tan(M_PI * f0 / rate), vh = std::pow(10.0, g_db / 20.0), vb = std::pow(vh, 0.4996667741545416);
        const double a0 = 1.0 + k / q + k * k;
        A.shelf[0] = (vh + vb * k / q + k * k) / a0;
        A.shelf[1] = 2.0 * (k * k - vh) / a0;
        A.shelf[2] = (vh - vb * k / q + k * k) / a0;
        A.shelf[3] = 2.0 * (k * k - 1.0) / a0;
        A.shelf[4] = (1.0 - k / q + k * k) / a0;
        const double f0h = 38.13547087602444, qh = 0.5003270373238773, kh = std::tan(M_PI * f0h / rate);
        const double a0h = 1.0 + kh / qh + kh * kh;
        A.hp[0] = 1.0;
        A.hp[1] = -2.0;
        A.hp[2] = 1.0;
        A.hp[3] = 2.0 * (kh * kh - 1.0) / a0h;
        A.hp[4] = (1.0 - kh / qh + kh * kh) / a0h;
        A.hop = (unsigned)std::llround(rate * 0.1);
    }
    const uint64_t frames = n / ch;
    std::vector<uint64_t> block_len;
    if (A.hop) {
        uint64_t start = 0;
        const uint64_t block = (uint64_t)A.hop * 4;
        while (start < frames) {
            const uint64_t end = start + block < frames ? start + block : frames;
            if (end <= start) break;
            block_len.push_back(end - start);
            if (end == frames) break;
            start += A.hop;
        }
    }
    A.n_blocks = (unsigned)block_len.size();
    // segments of the order-bound scans (analysis_kernels.hip): a block must not span more than two of them, and the
    // warm-up is a quarter of a second (the 38 Hz high-pass has decayed by exp(-59) then)
    A.seg_frames = 65536u > 8u * A.hop ? 65536u : 8u * A.hop;
    A.warm_frames = 8192u > sr / 4u ? 8192u : sr / 4u;
    {
        const uint64_t longest = (n + ch - 1) / ch;   // samples of channel 0 (a trailing partial frame counts for the FIR)
        A.n_seg = (unsigned)((longest + A.seg_frames - 1) / A.seg_frames);
        if (A.n_seg == 0) A.n_seg = 1;
    }
    // clips beyond one exact segment: two passes over short segments with the filter state handed over exactly
    // (analysis_kernels.hip, "K-weighting, long clips"); FLO_ANALYSIS_EXACT=1 keeps the one-lane walk (diagnostic)
    A.fast = (frames > 65536 && A.hop && ch <= 64 && !getenv("FLO_ANALYSIS_EXACT")) ? 1u : 0u;
    // segment length: two walks of L frames (150 ns per frame) against a scan over frames / L segments (35 ns each):
    // the power of two next to sqrt(frames / 8), between 256 and 2048
    A.kseg_frames = 256;
    while (A.kseg_frames < 2048 && (uint64_t)A.kseg_frames * A.kseg_frames * 8 < frames) A.kseg_frames *= 2;
    A.n_kseg = (unsigned)((frames + A.kseg_frames - 1) / A.kseg_frames);
    A.kq = A.hop ? A.kseg_frames / A.hop + 2 : 1;
    if (A.fast) {
        // M^L: the homogeneous system (x = 0) walked L steps from each unit state, in the kernels' own arithmetic
        for (int col = 0; col < 4; col++) {
            double v[4] = {0, 0, 0, 0};
            v[col] = 1.0;
            for (unsigned i = 0; i < A.kseg_frames; i++) {
                const double y = v[0];
                const double n1 = -A.shelf[3] * y + v[1], n2 = -A.shelf[4] * y;
                const double y2 = A.hp[0] * y + v[2];
                const double m1 = A.hp[1] * y - A.hp[3] * y2 + v[3], m2 = A.hp[2] * y - A.hp[4] * y2;
                v[0] = n1, v[1] = n2, v[2] = m1, v[3] = m2;
            }
            for (int r = 0; r < 4; r++) A.kpow[4 * r + col] = v[r];
        }
    }
    A.sq_seg = 1u << 16;
    A.n_sq_seg = (unsigned)((n + A.sq_seg - 1) / A.sq_seg);
    // beyond one segment the sum of squares is chained chunk by chunk so that it IS the sequential f32 sum (analysis_kernels.hip)
    A.sq_exact = n > A.sq_seg ? 1u : 0u;
    A.n_sq_chunks = (n + 1023) / 1024;
    if (A.sq_exact) A.n_sq_seg = 1;
    {   // compute_true_peak's filter (ebu_r128.rs:117-140): 49-tap Hann-windowed sinc, designed at 4 fs, unit sum
        const double oversample_rate = (double)sr * 4.0, cutoff = (double)sr * 0.45, center = 24.0;
        double sum = 0.0;
        for (int i = 0; i < 49; i++) {
            const double nn = (double)i - center;
            const double sinc = std::fabs(nn) < 1e-12 ? 2.0 * cutoff / oversample_rate : std::sin(2.0 * cutoff * nn / oversample_rate) / (M_PI * nn);
            const double window = 0.5 * (1.0 - std::cos(2.0 * M_PI * (double)i / 48.0));
            A.tp_coef[i] = sinc * window;
        }
        for (int i = 0; i < 49; i++) sum += A.tp_coef[i];
        for (int i = 0; i < 49; i++) A.tp_coef[i] /= sum;
    }
    A.n_chunks = (9ull + 4ull * n + 1023ull) / 1024ull;
    const uint64_t spc = n / ch;
    const uint64_t pts[3] = {spc / 4, spc / 2, spc * 3 / 4};
    for (int i = 0; i < 3; i++) {
        A.points[i] = pts[i];
        A.point_ok[i] = pts[i] + 256 < spc ? 1u : 0u;
    }
    // twiddles of the 256-point FFT: cos / sin in double, rounded to f32 (the values the oracle's FFT uses)
    // (a function-local static initialised by a lambda: thread-safe, contexts on several threads may meet here)
    struct Tw {
        float v[8 * 128 * 2];
    };
    static const Tw tw_table = [] {
        Tw t{};
        for (int s = 0; s < 8; s++)
            for (int k = 0; k < (1 << s); k++) {
                const double ang = -2.0 * M_PI * (double)k / (double)(2 << s);
                t.v[(s * 128 + k) * 2] = (float)std::cos(ang);
                t.v[(s * 128 + k) * 2 + 1] = (float)std::sin(ang);
            }
        return t;
    }();
    const float (&tw)[8 * 128 * 2] = tw_table.v;
    // device buffers: pcm | results
    DevMem d_pcm, d_res, d_cvs;
    QuiesceOnExit quiesce_d_pcm(c);
    if (!pcm_dev) HIPCHK(c, pool_alloc(&d_pcm.p, n * 4 + 64));
    const size_t o_peaks = 0, o_sumsq = o_peaks + (((size_t)A.n_peaks * 4 + 15) & ~(size_t)15),
                 o_pk = o_sumsq + (((size_t)A.n_sq_seg * 4 + 15) & ~(size_t)15), o_blocks = o_pk + 16,
                 o_tw = o_blocks + (((size_t)ch * A.n_blocks * 16 + 15) & ~(size_t)15), o_band = o_tw + sizeof tw, o_bin = o_band + 3 * 16 * 4,
                 o_kq = (o_bin + 3 * 8 * 4 + 15) & ~(size_t)15, o_kst = o_kq + (A.fast ? (size_t)ch * A.n_kseg * A.kq * 8 : 0),
                 o_sqd = o_kst + (A.fast ? (size_t)ch * A.n_kseg * 32 : 0), o_sqr = o_sqd + (A.sq_exact ? (A.n_sq_chunks + 1) * 8 : 0),
                 o_pkp = o_sqr + (A.sq_exact ? A.n_sq_chunks * 64 : 0),
                 res_bytes = o_pkp + (A.fast ? ((((n + ch - 1) / ch + 2047) / 2048) * ch * 16) : 0);
    HIPCHK(c, pool_alloc(&d_res.p, res_bytes + 64));
    HIPCHK(c, pool_alloc(&d_cvs.p, (2 * A.n_chunks + 1) * 32 + 64));
    int rc = ctx_stager(c);
    if (rc != FLO_OK) return rc;
    if (!pcm_dev) {
        std::vector<UploadSeg> segs{{d_pcm.p, pcm, n * 4}};
        std::string err;
        const auto tu0 = std::chrono::steady_clock::now();
        if (stager_upload(c->stager, segs, c->stream, err) != 0) return fail(c, FLO_ERR_DEVICE, err);
        if (getenv("FLO_TRACE")) {
            hipStreamSynchronize(c->stream);
            fprintf(stderr, "[flo] analysis: upload of %.1f MB took %.0f us (%s)\n", (double)n * 4 / 1e6,
                    (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tu0).count() / 1e3,
                    stager_upload_choice(c->stager, nullptr, nullptr));
        }
    }
    HIPCHK(c, hipMemsetAsync(d_res.p, 0, o_kst, c->stream));   // (what lies behind is fully written by its kernels)
    HIPCHK(c, hipMemcpyAsync((char *)d_res.p + o_tw, tw, sizeof tw, hipMemcpyHostToDevice, c->stream));
    A.pcm = pcm_dev ? pcm_dev : d_pcm.as<float>();
    char *rb = (char *)d_res.p;
    A.peaks = (float *)(rb + o_peaks);
    A.sumsq_part = (float *)(rb + o_sumsq);
    A.peak_bits = (unsigned long long *)(rb + o_pk);
    A.block_part = (double *)(rb + o_blocks);
    A.fft_tw = (const float *)(rb + o_tw);
    A.band_sqrt = (float *)(rb + o_band);
    A.peak_bin = (unsigned int *)(rb + o_bin);
    A.kqpart = (double *)(rb + o_kq);
    A.kstate = (double *)(rb + o_kst);
    A.sq_dsum = (double *)(rb + o_sqd);
    A.sq_rec = (double *)(rb + o_sqr);
    A.peak_part = (double *)(rb + o_pkp);
    A.cvs = d_cvs.as<unsigned int>();
    if (!c->an_side_ready && !getenv("FLO_ANALYSIS_ONE_STREAM")) {
        HIPCHK(c, hipEventCreateWithFlags(&c->an_side.fork, hipEventDisableTiming));
        for (int i = 0; i < 3; i++) {
            HIPCHK(c, hipStreamCreateWithFlags(&c->an_side.st[i], hipStreamNonBlocking));
            HIPCHK(c, hipEventCreateWithFlags(&c->an_side.join[i], hipEventDisableTiming));
        }
        c->an_side_ready = true;
    }
    rc = timed_launch(c, "analysis", [&] { return launch_analysis(A, c->stream, c->an_side_ready ? &c->an_side : nullptr); });
    if (rc != FLO_OK && c->an_side_ready)   // (a failed launch may have left a side stream unjoined: the buffers below must outlive it)
        for (int i = 0; i < 3; i++) hipStreamSynchronize(c->an_side.st[i]);
    if (rc != FLO_OK) {
        hipStreamSynchronize(c->stream);
        return rc;
    }
    std::vector<uint8_t> res(o_kst);   // (what lies behind - filter states, the chunk records of the sum of squares - stays on the device)
    uint32_t root[8];
    HIPCHK(c, hipMemcpyAsync(res.data(), d_res.p, o_kst, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(root, d_cvs.as<unsigned int>() + 2 * A.n_chunks * 8, 32, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // waveform peaks: normalise by the largest (analysis.rs:103-109)
    {
        const float *pk = (const float *)(res.data() + o_peaks);
        float mx = 0.f;
        for (unsigned i = 0; i < A.n_peaks; i++) mx = max_rust(mx, pk[i]);
        for (unsigned i = 0; i < A.n_peaks; i++) peaks[i] = mx > 0.f ? pk[i] / mx : pk[i];
    }
    // fingerprint (analysis.rs:236-356)
    {
        const double dms = (double)spc / (double)sr * 1000.0;
        const uint32_t d = dms >= 4294967295.0 ? 4294967295u : (dms <= 0 ? 0u : (uint32_t)dms);
        out->duration_ms = d < 1 ? 1 : d;
        for (int i = 0; i < 8; i++)
            for (int k = 0; k < 4; k++) out->hash[4 * i + k] = (uint8_t)(root[i] >> (8 * k));
        const float *bs = (const float *)(res.data() + o_band);
        const uint32_t *pb = (const uint32_t *)(res.data() + o_bin);
        float bands[16] = {0};
        uint8_t pk8[8] = {0};
        for (int p = 0; p < 3; p++) {
            if (!A.point_ok[p]) continue;
            for (int b = 0; b < 16; b++) bands[b] += bs[p * 16 + b];
            for (int b = 0; b < 8; b++) {
                const uint8_t v = f32_as_u8((float)pb[p * 8 + b] / 256.0f * 255.0f);
                if (v > pk8[b]) pk8[b] = v;
            }
        }
        float mx = 0.f;
        for (int b = 0; b < 16; b++) mx = max_rust(mx, bands[b]);
        for (int b = 0; b < 16; b++) out->energy_profile[b] = mx > 0.f ? f32_as_u8(bands[b] / mx * 255.0f) : 0;
        memcpy(out->frequency_peaks, pk8, 8);
        float sumsq = 0.f;   // the segments' partial sums, in order (one segment: the reference's own sequential sum)
        for (unsigned i = 0; i < A.n_sq_seg; i++) sumsq = i ? sumsq + ((const float *)(res.data() + o_sumsq))[i] : ((const float *)(res.data() + o_sumsq))[0];
        out->sum_squares = sumsq;
        if (A.sq_exact && getenv("FLO_TRACE"))
            fprintf(stderr, "[analysis] sum of squares: %llu chunks, %g walked sample by sample\n", (unsigned long long)A.n_sq_chunks,
                    (double)((const float *)(res.data() + o_sumsq))[1]);
        const float rms = sumsq / (float)n;
        float v = -20.0f * log10f(rms + 1e-10f);
        if (v == v) v = v < -60.0f ? -60.0f : (v > 0.0f ? 0.0f : v);
        out->avg_loudness = f32_as_u8(v + 60.0f);
    }
    // loudness: block energies summed over channels, the two gates, the range of the gated block loudness and the two
    // peaks (ebu_r128.rs:211-355)
    {
        const double *part = (const double *)(res.data() + o_blocks);
        std::vector<double> fastpart;
        if (A.fast) {
            // the segments' shares of every 100 ms quantum, added in segment order; a block is its four quanta, in order
            const uint64_t nq = (frames + A.hop - 1) / A.hop;
            const double *kq = (const double *)(res.data() + o_kq);
            std::vector<double> quanta((size_t)ch * nq, 0.0);
            for (unsigned cc = 0; cc < ch; cc++)
                for (uint64_t sg = 0; sg < A.n_kseg; sg++) {
                    const uint64_t f0 = sg * A.kseg_frames, f1 = std::min<uint64_t>(f0 + A.kseg_frames, frames);
                    if (f1 <= f0) continue;
                    const uint64_t q0 = f0 / A.hop, q1 = (f1 - 1) / A.hop;
                    for (uint64_t q = q0; q <= q1; q++) quanta[(size_t)cc * nq + q] += kq[((size_t)cc * A.n_kseg + sg) * A.kq + (q - q0)];
                }
            fastpart.assign((size_t)ch * A.n_blocks * 2, 0.0);
            for (unsigned cc = 0; cc < ch; cc++)
                for (unsigned k = 0; k < A.n_blocks; k++) {
                    double e = 0.0;
                    for (uint64_t q = k; q < (uint64_t)k + 4 && q < nq; q++) e += quanta[(size_t)cc * nq + q];
                    fastpart[((size_t)cc * A.n_blocks + k) * 2] = e;
                }
            part = fastpart.data();
        }
        std::vector<double> en(A.n_blocks);
        for (unsigned k = 0; k < A.n_blocks; k++) {
            double e = 0.0;
            for (unsigned cc = 0; cc < ch; cc++) {
                const double *pp = part + ((size_t)cc * A.n_blocks + k) * 2;
                e += (pp[0] + pp[1]) / (double)block_len[k];   // (a block inside one segment: x + 0.0, exact)
            }
            en[k] = e;
        }
        double lufs = -23.0, lra = 0.0;
        if (!en.empty()) {
            const double abs_gate = std::pow(10.0, (-70.0 + 0.691) / 10.0);
            double sum = 0.0;
            size_t cnt = 0;
            for (double e : en)
                if (e >= abs_gate) {
                    sum += e;
                    cnt++;
                }
            if (cnt) {
                const double ungated = -0.691 + 10.0 * std::log10(sum / (double)cnt);
                const double rel_gate = std::pow(10.0, (ungated - 10.0 + 0.691) / 10.0);
                double s2 = 0.0;
                std::vector<double> vals;
                for (double e : en)
                    if (e >= abs_gate && e >= rel_gate) {
                        s2 += e;
                        vals.push_back(e > 0.0 ? -0.691 + 10.0 * std::log10(e) : -150.0);
                    }
                lufs = !vals.empty() ? -0.691 + 10.0 * std::log10(s2 / (double)vals.size()) : ungated;
                if (vals.size() >= 2) {   // LRA: 10th - 95th percentile, linear interpolation (ebu_r128.rs:320-345)
                    std::sort(vals.begin(), vals.end());
                    const double nn = (double)vals.size();
                    auto interp = [&](double pos) {
                        const size_t i = (size_t)std::floor(pos);
                        const double frac = pos - (double)i;
                        return i + 1 < vals.size() ? vals[i] * (1.0 - frac) + vals[i + 1] * frac : vals[i];
                    };
                    lra = interp(0.95 * (nn - 1.0)) - interp(0.10 * (nn - 1.0));
                }
            }
        }
        out->integrated_lufs = lufs;
        out->loudness_range_lu = lra;
        const unsigned long long *pb = (const unsigned long long *)(res.data() + o_pk);
        double sp, tp;
        memcpy(&sp, &pb[0], 8);
        memcpy(&tp, &pb[1], 8);
        out->sample_peak_dbfs = sp > 1e-6 ? 20.0 * std::log10(sp) : -150.0;
        out->true_peak_dbtp = tp > 1e-9 ? 20.0 * std::log10(tp) : -150.0;
    }
    return FLO_OK;
}

extern "C" int flo_analyze(flo_ctx *c, const float *pcm, size_t n, uint32_t sr, uint8_t ch, uint32_t pps, float *peaks,
                           size_t peaks_cap, flo_analysis *out) {
    return analyze_impl(c, pcm, nullptr, n, sr, ch, pps, peaks, peaks_cap, out);
}

static int analysis_metadata_impl(flo_ctx *c, const float *pcm, const float *pcm_dev, size_t n, uint32_t sr, uint8_t ch, uint32_t pps,
                                  uint8_t **out, size_t *out_len) {
    if (!c || !out || !out_len) return FLO_ERR_ARG;
    *out = nullptr;
    *out_len = 0;
    if (!ch || !sr || !pps) return fail(c, FLO_ERR_ARG, "flo_analysis_metadata: bad argument");
    // (one peak per 1 / pps seconds: ceil(frames * pps / rate) of them - a vector of one float per sample frame was 32 MB of
    // zeroed fresh pages for a 3-minute clip, 4 ms of a 7 ms call)
    std::vector<float> peaks((size_t)std::ceil((double)(n / ch) * (double)pps / (double)sr) + 16);
    flo_analysis an;
    int rc = analyze_impl(c, pcm, pcm_dev, n, sr, ch, pps, peaks.data(), peaks.size(), &an);
    if (rc != FLO_OK) return rc;
    Mp m, fp;
    m.b.push_back(0x84);
    m.s("length_ms");
    m.u(an.length_ms);
    m.s("waveform_data");
    m.b.push_back(0x83);
    m.s("peaks_per_second");
    m.u(pps);
    m.s("peaks");
    m.arr(an.n_peaks);
    for (unsigned i = 0; i < an.n_peaks; i++) m.f(peaks[i]);
    m.s("channels");
    m.u(ch);
    m.s("spectrum_fingerprint");
    fp.b.push_back(0x87);
    fp.s("hash");
    fp.arr(32);
    for (int i = 0; i < 32; i++) fp.u(n ? an.hash[i] : 0);
    fp.s("duration_ms");
    fp.u(n ? an.duration_ms : 0);
    fp.s("sample_rate");
    fp.u(sr);
    fp.s("channels");
    fp.u(ch);
    fp.s("frequency_peaks");
    fp.arr(8);
    for (int i = 0; i < 8; i++) fp.u(an.frequency_peaks[i]);
    fp.s("energy_profile");
    fp.arr(16);
    for (int i = 0; i < 16; i++) fp.u(an.energy_profile[i]);
    fp.s("avg_loudness");
    fp.u(an.avg_loudness);
    m.bin(fp.b);
    m.s("loudness_profile");
    m.b.push_back(0x91);
    m.b.push_back(0x82);
    m.s("timestamp_ms");
    m.u(0);
    m.s("lufs");
    m.f((float)an.integrated_lufs);
    uint8_t *p = (uint8_t *)malloc(m.b.size());
    if (!p) return fail(c, FLO_ERR_NOMEM, "malloc failed");
    memcpy(p, m.b.data(), m.b.size());
    *out = p;
    *out_len = m.b.size();
    return FLO_OK;
}
extern "C" int flo_analysis_metadata(flo_ctx *c, const float *pcm, size_t n, uint32_t sr, uint8_t ch, uint32_t pps,
                                     uint8_t **out, size_t *out_len) {
    return analysis_metadata_impl(c, pcm, nullptr, n, sr, ch, pps, out, out_len);
}
// the same from a clip that is already on the device (uploaded into a batch that is about to be encoded): libflo::encode*
// analyse and encode the same samples (lib.rs:105-116), and they cross PCIe once
extern "C" int flo_batch_analysis_metadata(flo_batch *b, size_t clip, uint32_t pps, uint8_t **out, size_t *out_len) {
    if (!b || clip >= b->n_clips) return FLO_ERR_ARG;
    return analysis_metadata_impl(b->ctx, nullptr, b->n_il[clip] ? b->d_pcm + b->clip_off[clip] : nullptr, b->n_il[clip], b->sr, b->ch, pps, out, out_len);
}
extern "C" int flo_batch_set_bit_depth(flo_batch *b, uint8_t bit_depth) {
    if (!b) return FLO_ERR_ARG;
    b->bit_depth = bit_depth;   // (echoed into the header like flo_encode_lossless's argument: writer.rs:146)
    return FLO_OK;
}
