// stager.cpp — see stager.hpp
#include "stager.hpp"

#include <cstdlib>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>

namespace flo {

namespace {
constexpr size_t kSlotBytes = (size_t)8 << 20;   // ring slot
constexpr int kSlots = 4;
constexpr size_t kPiece = (size_t)512 << 10;     // unit of work handed to a worker thread
// Uploads up to this size always go through the runtime's own pageable path (hipMemcpyAsync straight from the caller's
// memory): waking the copy threads costs more than they save. Beyond it the faster of the two paths is used, and which
// one that is depends on the HOST (round 3 measured 43 GB/s pageable against 30 - 35 through the ring on one host and a
// factor of two the other way on another), so it is measured once per context on the first large upload (probe()).
constexpr size_t kSmallUpload = (size_t)8 << 20;
constexpr size_t kProbeBytes = (size_t)24 << 20;   // three ring slots: the ring's steady state, not its first slot
}  // namespace

class Stager {
   public:
    struct Job {
        char *dst;
        const char *src;
        size_t bytes;
    };
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<Job> jobs;
    size_t next_job = 0;
    size_t done_jobs = 0;
    bool quit = false;
    char *slot[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[kSlots] = {nullptr, nullptr, nullptr, nullptr};
    bool ev_used[kSlots] = {false, false, false, false};
    int cur = 0;
    void *pinned = nullptr;
    size_t pinned_bytes = 0;
    struct PinBlock {
        void *p = nullptr;
        size_t cap = 0;
        bool in_use = false;
    };
    std::vector<PinBlock> blocks;   // pinned download buffers, kept for the next call
    hipStream_t aux_stream = nullptr;   // the second uploading thread's stream and its "done" event
    hipEvent_t aux_ev = nullptr, aux_go = nullptr;
    int large_path = 0;                 // 0 not measured yet, 1 pageable-direct, 2 pinned ring
    double probe_direct_gbs = 0, probe_ring_gbs = 0;

    void worker() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_work.wait(lk, [&] { return quit || next_job < jobs.size(); });
            if (quit) return;
            while (next_job < jobs.size()) {
                const Job j = jobs[next_job++];
                lk.unlock();
                memcpy(j.dst, j.src, j.bytes);
                lk.lock();
                if (++done_jobs == jobs.size()) cv_done.notify_all();
            }
        }
    }
    // run the queued jobs on the workers and on the calling thread; returns when all are done
    void run(std::vector<Job> &&js) {
        if (js.empty()) return;
        if (js.size() == 1 || workers.empty()) {
            for (const Job &j : js) memcpy(j.dst, j.src, j.bytes);
            return;
        }
        std::unique_lock<std::mutex> lk(mu);
        jobs = std::move(js);
        next_job = 0;
        done_jobs = 0;
        cv_work.notify_all();
        while (next_job < jobs.size()) {   // the caller works too
            const Job j = jobs[next_job++];
            lk.unlock();
            memcpy(j.dst, j.src, j.bytes);
            lk.lock();
            ++done_jobs;
        }
        cv_done.wait(lk, [&] { return done_jobs == jobs.size(); });
        jobs.clear();
        next_job = done_jobs = 0;
    }
};

Stager *stager_create(std::string &err) {
    (void)err;
    return new Stager();   // the ring and the copy threads are made by the first large upload
}

static bool ensure_ring(Stager *s, std::string &err) {
    if (s->slot[0]) return true;
    for (int i = 0; i < kSlots; i++) {
        if (hipHostMalloc((void **)&s->slot[i], kSlotBytes) != hipSuccess ||
            hipEventCreateWithFlags(&s->ev[i], hipEventDisableTiming) != hipSuccess) {
            err = "pinned staging ring: allocation failed";
            return false;
        }
    }
    unsigned hw = std::thread::hardware_concurrency();
    int n = hw > 2 ? (int)(hw - 1) : 1;
    if (n > 5) n = 5;   // with the caller: six copying threads (four already reach the PCIe rate; more only add scheduler noise)
    for (int i = 0; i < n; i++) s->workers.emplace_back([s] { s->worker(); });
    return true;
}

void stager_destroy(Stager *s) {
    if (!s) return;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->quit = true;
    }
    s->cv_work.notify_all();
    for (auto &t : s->workers) t.join();
    for (int i = 0; i < kSlots; i++) {
        if (s->ev[i]) {
            if (s->ev_used[i]) hipEventSynchronize(s->ev[i]);
            hipEventDestroy(s->ev[i]);
        }
        if (s->slot[i]) hipHostFree(s->slot[i]);
    }
    if (s->pinned) hipHostFree(s->pinned);
    for (auto &b : s->blocks) hipHostFree(b.p);
    if (s->aux_ev) hipEventDestroy(s->aux_ev);
    if (s->aux_go) hipEventDestroy(s->aux_go);
    if (s->aux_stream) hipStreamDestroy(s->aux_stream);
    delete s;
}

void *stager_pinned_get(Stager *s, size_t bytes, std::string &err) {
    for (auto &b : s->blocks)
        if (!b.in_use && b.cap >= bytes) {
            b.in_use = true;
            return b.p;
        }
    Stager::PinBlock nb;
    nb.cap = bytes + bytes / 4 + 65536;
    if (hipHostMalloc(&nb.p, nb.cap) != hipSuccess) {
        err = "pinned download buffer: allocation failed";
        return nullptr;
    }
    nb.in_use = true;
    s->blocks.push_back(nb);
    return nb.p;
}
void stager_pinned_put(Stager *s, void *p) {
    for (auto &b : s->blocks)
        if (b.p == p) b.in_use = false;
}

void *stager_pinned(Stager *s, size_t bytes, std::string &err) {
    if (s->pinned_bytes >= bytes && s->pinned) return s->pinned;
    if (s->pinned) hipHostFree(s->pinned);
    s->pinned = nullptr;
    s->pinned_bytes = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    if (hipHostMalloc(&s->pinned, want) != hipSuccess) {
        err = "pinned download buffer: allocation failed";
        return nullptr;
    }
    s->pinned_bytes = want;
    return s->pinned;
}

void stager_memcpy_many(Stager *s, const std::vector<UploadSeg> &segs) {
    std::vector<Stager::Job> js;
    for (const UploadSeg &g : segs)
        for (size_t o = 0; o < g.bytes; o += kPiece)
            js.push_back({(char *)g.dst + o, (const char *)g.src + o, g.bytes - o < kPiece ? g.bytes - o : kPiece});
    s->run(std::move(js));
}

static int upload_direct(Stager *s, const std::vector<UploadSeg> &segs, size_t total, hipStream_t stream, std::string &err);
static int upload_ring(Stager *s, const std::vector<UploadSeg> &segs, hipStream_t stream, std::string &err);

// Which path moves a large upload faster on THIS host: both are timed once on kProbeBytes of scratch (pageable memory that
// has been touched, as a caller's buffer has) into device scratch, on a stream of the probe's own, until the data is
// on the device. About ten milliseconds, once per context. FLO_UPLOAD_PATH=direct|ring skips it.
static void probe(Stager *s) {
    if (const char *e = getenv("FLO_UPLOAD_PATH")) {
        if (!strcmp(e, "direct")) { s->large_path = 1; return; }
        if (!strcmp(e, "ring")) { s->large_path = 2; return; }
    }
    s->large_path = 2;   // what an incomplete probe leaves: the ring does not block the calling thread per copy
    char *host = (char *)malloc(kProbeBytes);
    void *dev = nullptr;
    hipStream_t st = nullptr;
    if (host && hipMalloc(&dev, kProbeBytes) == hipSuccess && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess) {
        memset(host, 1, kProbeBytes);
        std::string err;
        std::vector<UploadSeg> segs;
        for (size_t o = 0; o < kProbeBytes; o += (size_t)4 << 20) segs.push_back({(char *)dev + o, host + o, (size_t)4 << 20});
        double t[2] = {0, 0};
        bool ok = ensure_ring(s, err);
        for (int pass = 0; ok && pass < 2; pass++)        // the first pass warms both paths up (page tables, threads)
            for (int which = 0; ok && which < 2; which++) {
                const auto t0 = std::chrono::steady_clock::now();
                ok = (which ? upload_ring(s, segs, st, err) : upload_direct(s, segs, kProbeBytes, st, err)) == 0 && hipStreamSynchronize(st) == hipSuccess;
                t[which] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            }
        if (ok && t[0] > 0 && t[1] > 0) {
            s->probe_direct_gbs = kProbeBytes / t[0] / 1e9;
            s->probe_ring_gbs = kProbeBytes / t[1] / 1e9;
            s->large_path = t[0] < t[1] ? 1 : 2;
        }
    }
    if (st) hipStreamDestroy(st);
    if (dev) hipFree(dev);
    free(host);
}

const char *stager_upload_choice(Stager *s, double *direct_gbs, double *ring_gbs) {
    if (direct_gbs) *direct_gbs = s->probe_direct_gbs;
    if (ring_gbs) *ring_gbs = s->probe_ring_gbs;
    return s->large_path == 1 ? "pageable-direct" : (s->large_path == 2 ? "pinned-ring" : "not measured yet");
}

int stager_upload(Stager *s, const std::vector<UploadSeg> &segs, hipStream_t stream, std::string &err) {
    size_t total = 0;
    for (const UploadSeg &g : segs) total += g.bytes;
    static const size_t small_limit = [] {   // diagnostic override (MiB)
        const char *e = getenv("FLO_SMALL_UPLOAD_MB");
        return e ? (size_t)atoi(e) << 20 : kSmallUpload;
    }();
    if (total <= small_limit) return upload_direct(s, segs, total, stream, err);
    if (!s->large_path) probe(s);
    return s->large_path == 1 ? upload_direct(s, segs, total, stream, err) : upload_ring(s, segs, stream, err);
}

static int upload_direct(Stager *s, const std::vector<UploadSeg> &segs, size_t total, hipStream_t stream, std::string &err) {
    {
        // FLO_UPLOAD_THREADS=2: a second thread issues half of the copies on a stream of its own. Measured: 6.2 - 6.9 ms
        // against 6.8 - 6.9 for 64 x 3.5 MB alone in a process, but 12.0 against 6.5 inside bench.py's process on another
        // host: the runtime serialises most of it and the extra thread is at the scheduler's mercy. Off by default.
        static const bool two = [] {
            const char *e = getenv("FLO_UPLOAD_THREADS");
            return e && atoi(e) >= 2;
        }();
        size_t split = segs.size();
        if (two && segs.size() >= 2 && total >= ((size_t)16 << 20)) {
            size_t acc = 0;
            for (split = 0; split < segs.size() && acc < total / 2; split++) acc += segs[split].bytes;
            if (!s->aux_stream && (hipStreamCreateWithFlags(&s->aux_stream, hipStreamNonBlocking) != hipSuccess ||
                                   hipEventCreateWithFlags(&s->aux_ev, hipEventDisableTiming) != hipSuccess ||
                                   hipEventCreateWithFlags(&s->aux_go, hipEventDisableTiming) != hipSuccess))
                split = segs.size();
            // the second stream's copies land behind whatever `stream` has been ordered behind (the batch's zero fill)
            if (split < segs.size() && (hipEventRecord(s->aux_go, stream) != hipSuccess || hipStreamWaitEvent(s->aux_stream, s->aux_go, 0) != hipSuccess))
                split = segs.size();
        }
        int dev = 0;
        bool aux_ok = true;
        std::thread helper;
        if (split < segs.size()) {
            if (hipGetDevice(&dev) != hipSuccess) {
                err = "hipGetDevice failed";
                return -1;
            }
            helper = std::thread([&, dev] {
                if (hipSetDevice(dev) != hipSuccess) {
                    aux_ok = false;
                    return;
                }
                for (size_t i = split; i < segs.size(); i++)
                    if (segs[i].bytes && hipMemcpyAsync(segs[i].dst, segs[i].src, segs[i].bytes, hipMemcpyHostToDevice, s->aux_stream) != hipSuccess) aux_ok = false;
                if (hipEventRecord(s->aux_ev, s->aux_stream) != hipSuccess) aux_ok = false;
            });
        }
        bool ok = true;
        for (size_t i = 0; i < split; i++)
            if (segs[i].bytes && hipMemcpyAsync(segs[i].dst, segs[i].src, segs[i].bytes, hipMemcpyHostToDevice, stream) != hipSuccess) ok = false;
        if (helper.joinable()) {
            helper.join();
            if (!aux_ok || hipStreamWaitEvent(stream, s->aux_ev, 0) != hipSuccess) ok = false;
        }
        if (!ok) {
            err = "hipMemcpyAsync (upload) failed";
            return -1;
        }
        return 0;
    }
}

static int upload_ring(Stager *s, const std::vector<UploadSeg> &segs, hipStream_t stream, std::string &err) {
    if (!ensure_ring(s, err)) return -1;
    // walk the segments, cutting them at ring-slot boundaries: a slot is filled by the worker threads, then handed to
    // the copy engine (one asynchronous copy per piece of a segment), while the next slot is being filled
    size_t si = 0, so = 0;   // current segment and offset inside it
    while (si < segs.size()) {
        const int k = s->cur;
        s->cur = (s->cur + 1) % kSlots;
        if (s->ev_used[k] && hipEventSynchronize(s->ev[k]) != hipSuccess) {
            err = "staging ring: event wait failed";
            return -1;
        }
        std::vector<Stager::Job> js;
        struct Out {
            void *dst;
            size_t off, bytes;
        };
        std::vector<Out> outs;
        size_t fill = 0;
        while (si < segs.size() && fill < kSlotBytes) {
            const UploadSeg &g = segs[si];
            if (g.bytes == so) {   // empty or finished segment
                si++;
                so = 0;
                continue;
            }
            size_t n = g.bytes - so;
            if (n > kSlotBytes - fill) n = kSlotBytes - fill;
            outs.push_back({(char *)g.dst + so, fill, n});
            for (size_t o = 0; o < n; o += kPiece)
                js.push_back({s->slot[k] + fill + o, (const char *)g.src + so + o, n - o < kPiece ? n - o : kPiece});
            fill += (n + 63) & ~(size_t)63;
            so += n;
            if (so == g.bytes) {
                si++;
                so = 0;
            }
        }
        s->run(std::move(js));
        for (const Out &o : outs)
            if (hipMemcpyAsync(o.dst, s->slot[k] + o.off, o.bytes, hipMemcpyHostToDevice, stream) != hipSuccess) {
                err = "staging ring: hipMemcpyAsync failed";
                return -1;
            }
        if (hipEventRecord(s->ev[k], stream) != hipSuccess) {
            err = "staging ring: event record failed";
            return -1;
        }
        s->ev_used[k] = true;
    }
    return 0;
}

}  // namespace flo
