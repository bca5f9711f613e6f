// common.hpp -- shared host-side plumbing of libvqhip (status codes, thread-local state,
// HIP error mapping).  gfx950-only: there is no other back end and no CPU fallback.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/vqhip.h"

namespace vqhip {

struct ThreadState {
    std::string last_error;
    hipStream_t user_stream = nullptr;
    bool user_stream_set = false;
    hipStream_t own_stream = nullptr;
    int own_stream_device = -1;
    uint64_t last_rechecked = 0;
    int last_engine = 0;
};

ThreadState &tls();

// records the message and returns `code`
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// current stream of the calling thread (creates the per-thread stream lazily)
int current_stream(hipStream_t *out);

// verifies that the current device is a gfx950 part; cached per device
int require_gfx950();

int num_cus();

// Per-device "has this call site configured its kernel yet" flag (hipFuncSetAttribute is per device;
// a process may drive several GPUs through vqhip_set_device).  Idempotent under races.
struct PerDeviceOnce {
    std::atomic<uint64_t> mask{0};
    bool needed() const {
        int dev = 0;
        (void)hipGetDevice(&dev);
        return !(mask.load(std::memory_order_acquire) & (1ull << (dev & 63)));
    }
    void done() {
        int dev = 0;
        (void)hipGetDevice(&dev);
        mask.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
};

#define VQ_HIP(expr)                                                                     \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess)                                                            \
            return ::vqhip::fail(VQHIP_ERR_RUNTIME, "%s failed: %s (%s:%d)", #expr,      \
                                 hipGetErrorString(_e), __FILE__, __LINE__);             \
    } while (0)

#define VQ_TRY(expr)                 \
    do {                             \
        int _rc = (expr);            \
        if (_rc != VQHIP_OK) return _rc; \
    } while (0)

#define VQ_LAUNCH_CHECK(name)                                                            \
    do {                                                                                 \
        hipError_t _e = hipGetLastError();                                               \
        if (_e != hipSuccess)                                                            \
            return ::vqhip::fail(VQHIP_ERR_RUNTIME, "launch of %s failed: %s", name,     \
                                 hipGetErrorString(_e));                                 \
    } while (0)

// simple owning device buffer
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    int alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(VQHIP_ERR_RUNTIME, "hipMalloc(%zu) failed: %s", n, hipGetErrorString(e));
        }
        bytes = n;
        return VQHIP_OK;
    }
    int ensure(size_t n) { return (p && bytes >= n) ? VQHIP_OK : alloc(n); }
    template <class T>
    T *as() const {
        return reinterpret_cast<T *>(p);
    }
};

static inline uint32_t ceil_div_u32(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// both cosine ids share every sum; they differ in the last step only
__host__ __device__ constexpr bool vq_is_cos(int metric) { return metric == VQHIP_COSINE || metric == VQHIP_COSINE_UNCLAMPED; }
// dot, |a|, |b| (the reference's three sequential sums, norms already square-rooted) -> distance.
// VQHIP_COSINE: src/core/distance.rs:107-119 (EPSILON rule, f32::clamp keeps NaN); VQHIP_COSINE_UNCLAMPED: include/vqhip.h
__host__ __device__ inline float vq_cosine_finish(int metric, float dot, float na, float nb) {
    if (metric == VQHIP_COSINE_UNCLAMPED) {
        const float denom = na * nb;
        const float q = dot / denom;
        return 1.0f - q;
    }
    const float EPS = 1e-10f;
    if (na < EPS || nb < EPS) return 1.0f;
    const float denom = na * nb;
    const float q = dot / denom;  // correctly rounded
    const float v = 1.0f - q;
    return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
}

}  // namespace vqhip
