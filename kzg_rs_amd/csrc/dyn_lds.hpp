// dyn_lds.hpp - the one place that raises a kernel's dynamic-LDS limit.  Host code; part of the translation unit kzg_capi.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <utility>
// hipFuncAttributeMaxDynamicSharedMemorySize is a MAXIMUM, kept per (kernel, device) - not per handle or stream.  Concurrent
// callers (the pipeline's per-device threads, the workers of a sharded stream, the leaders of the small-call queue) launch the
// same kernels with different dynamic sizes, so the attribute is only ever RAISED here, under one lock, and a launch passes
// just the size it wants: a set(small) that lands between another thread's set(large) and its launch can no longer fail that
// launch.  (It also keeps a runtime call out of every launch after the first.)
static hipError_t dyn_lds_ensure(const void* func, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> raised;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(mu);
    size_t& cur = raised[std::make_pair(func, dev)];
    if (bytes <= cur) return hipSuccess;
    e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) cur = bytes;
    return e;
}
#define DYN_LDS(kernel, bytes) dyn_lds_ensure(reinterpret_cast<const void*>(kernel), (size_t)(bytes))
