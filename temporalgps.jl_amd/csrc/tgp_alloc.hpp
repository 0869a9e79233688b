// A small caching allocator under the library's own hipMalloc / hipHostMalloc calls (round 5).  A hyper-parameter optimisation loop builds a new
// model -- a new handle -- every iteration (examples/exact_time_learning.jl): handle, engine tables, flags and result records are a dozen small
// device / pinned allocations, and releasing them (hipFree / hipHostFree synchronise the device) cost 0.5 ms per dropped model against 0.2 ms for
// the logpdf + gradient evaluation itself.  Blocks of up to kCacheMax bytes are parked per (device, kind, flags, size class) when they are freed
// and handed out again; larger ones (series-sized staging buffers) go straight to the runtime.  A parked block is never in use by the device: every
// call of the library returns with its stream drained.  Contents are NOT cleared on reuse (as hipMalloc's are not): whoever needs zeros writes them.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace tgp_alloc {
hipError_t dev_malloc(void** p, size_t bytes);
hipError_t dev_free(void* p);
hipError_t host_malloc(void** p, size_t bytes, unsigned flags);
hipError_t host_free(void* p);
// everything parked goes back to the runtime (tests; a process that wants its memory back)
void trim();
// (diagnostics) blocks handed out from the cache / from the runtime so far
void stats(long long* reused, long long* fresh);
}  // namespace tgp_alloc
