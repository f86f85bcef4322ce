// devpool.hpp — device-memory cache behind every batch and plan object.
//
// hipMalloc / hipFree cost tens of microseconds each (and hipFree synchronises the device); a one-shot encode
// (flo_encode_lossy: what TransformEncoder::encode_to_flo binds to, lossy/encoder.rs:167-239) needs a dozen buffers,
// so allocating them per call costs more than the encode itself. Freed blocks are kept per device in size classes
// and handed out again; callers only return a block once all work that used it has completed on their stream (the
// batch objects synchronise before they release anything). Blocks above kMaxCachedBlock, or beyond kMaxCachedTotal of
// cached bytes, go straight back to the driver.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace flo {

constexpr size_t kMaxCachedBlock = (size_t)256 << 20;
constexpr size_t kMaxCachedTotal = (size_t)2 << 30;

hipError_t pool_alloc(void **p, size_t bytes);   // on the current device
void pool_free(void *p);                          // any device
void pool_trim();                                 // release every cached block of every device (tests, shutdown)
template <class T>
hipError_t pool_alloc(T **p, size_t bytes) { return pool_alloc(reinterpret_cast<void **>(p), bytes); }

}  // namespace flo
