// devpool.cpp — see devpool.hpp
#include "devpool.hpp"

#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace flo {
namespace {
struct Block {
    int device;
    size_t cls;   // rounded size the block was allocated with
};
std::mutex g_mu;
std::unordered_map<void *, Block> g_live;                       // every block handed out or cached
std::map<std::pair<int, size_t>, std::vector<void *>> g_free;   // (device, class) -> cached blocks
size_t g_cached = 0;

// size classes: multiples of 256 B up to 64 KiB, then eight steps per octave (at most 12.5 % of slack)
size_t size_class(size_t n) {
    if (n < 256) n = 256;
    if (n <= (64u << 10)) return (n + 255) & ~(size_t)255;
    size_t p = 1;
    while ((p << 1) <= n) p <<= 1;      // largest power of two <= n
    const size_t step = p >> 3;
    return (n + step - 1) / step * step;
}
}  // namespace

hipError_t pool_alloc(void **p, size_t bytes) {
    *p = nullptr;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const size_t cls = size_class(bytes);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_free.find({dev, cls});
        if (it != g_free.end() && !it->second.empty()) {
            *p = it->second.back();
            it->second.pop_back();
            g_cached -= cls;
            return hipSuccess;
        }
    }
    e = hipMalloc(p, cls);
    if (e == hipErrorOutOfMemory) {   // give the cache back to the driver and try once more
        pool_trim();
        e = hipMalloc(p, cls);
    }
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_mu);
    g_live[*p] = Block{dev, cls};
    return hipSuccess;
}

void pool_free(void *p) {
    if (!p) return;
    Block b;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_live.find(p);
        if (it == g_live.end()) {   // not ours
            (void)hipFree(p);
            return;
        }
        b = it->second;
        if (b.cls <= kMaxCachedBlock && g_cached + b.cls <= kMaxCachedTotal) {
            g_free[{b.device, b.cls}].push_back(p);
            g_cached += b.cls;
            return;
        }
        g_live.erase(it);
    }
    (void)hipFree(p);
}

void pool_trim() {
    std::vector<void *> all;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto &kv : g_free)
            for (void *p : kv.second) {
                all.push_back(p);
                g_live.erase(p);
            }
        g_free.clear();
        g_cached = 0;
    }
    for (void *p : all) (void)hipFree(p);
}

}  // namespace flo
