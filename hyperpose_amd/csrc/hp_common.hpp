// hp_common.hpp — shared host-side plumbing of libhp_hip.so: error reporting across the C ABI,
// HIP call checking, RAII device/pinned buffers.  No reference counterpart (the reference's error
// policy is print + std::exit, src/logging.hpp:31-37; the C ABI returns codes instead and the C++
// mirror classes in include/hyperpose/ restore the throw/exit behaviour on top).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/hp_hip.h"

namespace hp {

void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
const char* last_error();

#define HP_HIP_TRY(expr)                                                                                     \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) {                                                                              \
            ::hp::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);      \
            return HP_ERR_HIP;                                                                               \
        }                                                                                                    \
    } while (0)

#define HP_REQUIRE(cond, code, ...)                                                                          \
    do {                                                                                                     \
        if (!(cond)) {                                                                                       \
            ::hp::set_error(__VA_ARGS__);                                                                    \
            return (code);                                                                                   \
        }                                                                                                    \
    } while (0)

#define HP_TRY(expr)                                                                                         \
    do {                                                                                                     \
        int _rc = (expr);                                                                                    \
        if (_rc != HP_OK)                                                                                    \
            return _rc;                                                                                      \
    } while (0)

// Owning device buffer (hipMalloc / hipFree).
struct dev_buf {
    void* p = nullptr;
    size_t bytes = 0;
    dev_buf() = default;
    dev_buf(const dev_buf&) = delete;
    dev_buf& operator=(const dev_buf&) = delete;
    ~dev_buf() { release(); }
    int alloc(size_t n)
    {
        release();
        if (n == 0)
            return HP_OK;
        HP_HIP_TRY(hipMalloc(&p, n));
        bytes = n;
        return HP_OK;
    }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <typename T>
    T* as() const { return static_cast<T*>(p); }
};

// Owning pinned host buffer.
struct host_buf {
    void* p = nullptr;
    size_t bytes = 0;
    host_buf() = default;
    host_buf(const host_buf&) = delete;
    host_buf& operator=(const host_buf&) = delete;
    ~host_buf() { release(); }
    int alloc(size_t n)
    {
        release();
        if (n == 0)
            return HP_OK;
        HP_HIP_TRY(hipHostMalloc(&p, n, hipHostMallocDefault));
        bytes = n;
        return HP_OK;
    }
    void release()
    {
        if (p)
            (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <typename T>
    T* as() const { return static_cast<T*>(p); }
};

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Host worker pool for the order-dependent parser tails (PoseProposal limb selection / merge, PifPaf grow / soft-NMS): frames are
// independent, so hp_*_collect hands frame indices to a few persistent threads, the way the reference replicates its parser per
// pool thread (include/hyperpose/utility/thread_pool.hpp:21, include/hyperpose/stream/stream.hpp:139-144).  The calling thread
// takes part; run() returns when every frame is done.  `fn(frame, worker)`: worker in [0, workers()) selects per-thread scratch.
// HP_PARSER_THREADS overrides the default min(8, hardware threads).
class frame_pool {
public:
    static frame_pool& instance();
    int workers() const { return n_threads_ + 1; }
    void run(int n_frames, void (*fn)(int frame, int worker, void* ctx), void* ctx);
    ~frame_pool();

private:
    frame_pool();
    struct impl;
    impl* d_;
    int n_threads_;
};

} // namespace hp
