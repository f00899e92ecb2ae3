// engine_internal.hpp — helpers shared by the engine's translation units (engine.cpp: lifecycle, weights, blocks and the four stages;
// engine_batch.cpp: the resident-batch pipeline and its graph cache; engine_ops.cpp: the single-kernel and timing entry points of the tests
// and tools).
#pragma once
#include "engine.hpp"

namespace stn {
namespace detail {
// host array -> arena (asynchronous copy on `s`)
template <typename T>
inline T* up(Arena& ar, hipStream_t s, const T* h, size_t n) {
    T* d = static_cast<T*>(ar.alloc(n * sizeof(T)));
    STN_HIP(hipMemcpyAsync(d, h, n * sizeof(T), hipMemcpyHostToDevice, s));
    return d;
}
}  // namespace detail
}  // namespace stn
