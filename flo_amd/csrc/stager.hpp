// stager.hpp — host <-> device staging for the host-buffer entry points (flo_encode_lossy / _lossless / _batch).
//
// Callers hand over ordinary pageable buffers, as the reference's callers hand `&[f32]` slices to
// TransformEncoder::encode_to_flo / Encoder::encode (lossy/encoder.rs:167, lossless/encoder.rs:32). A copy engine
// reads pageable memory slowly on some hosts and at nearly the PCIe rate on others, so large uploads take whichever of two
// paths a one-time probe finds faster on this host: the runtime's pageable copy straight from the caller's memory, or a
// ring of pinned buffers that a few worker threads fill slice by slice (one core cannot feed PCIe gen5) while the copy
// engine drains the previous slot; a single clip always goes direct; downloads come back as one pinned transfer per batch.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace flo {

struct UploadSeg {
    void *dst;         // device
    const void *src;   // host (pageable)
    size_t bytes;
};

class Stager;
Stager *stager_create(std::string &err);
void stager_destroy(Stager *s);
// copy every segment host -> device on `stream` (asynchronous with respect to the device; the host buffers are fully
// read when the call returns)
int stager_upload(Stager *s, const std::vector<UploadSeg> &segs, hipStream_t stream, std::string &err);
// which path large uploads take on this host ("pageable-direct" / "pinned-ring" / "not measured yet") and the two rates the
// one-time probe measured (0 until it has run)
const char *stager_upload_choice(Stager *s, double *direct_gbs, double *ring_gbs);
// a pinned scratch buffer of at least `bytes` (kept for the next call); contents undefined
void *stager_pinned(Stager *s, size_t bytes, std::string &err);
// pinned buffers from a small cache (several may be out at a time); put returns one to the cache
void *stager_pinned_get(Stager *s, size_t bytes, std::string &err);
void stager_pinned_put(Stager *s, void *p);
// parallel memcpy on the worker threads (host to host)
void stager_memcpy_many(Stager *s, const std::vector<UploadSeg> &segs);

}  // namespace flo
