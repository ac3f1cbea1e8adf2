// hp_runtime.cpp — runtime entry points of the C ABI (include/hp_hip.h): device selection, error
// string, memory helpers.  Replaces the implicit CUDA context handling of the reference engine
// (src/tensorrt.cpp:106-118 `cuda_dep`).
#include "hp_common.hpp"

#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>

namespace hp {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char* last_error() { return g_err; }

struct frame_pool::impl {
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> threads;
    // one job at a time (run() holds `busy`); generation wakes the workers
    std::mutex busy;
    unsigned long generation = 0;
    bool stop = false;
    int n_frames = 0, active = 0;
    std::atomic<int> next{ 0 };
    void (*fn)(int, int, void*) = nullptr;
    void* ctx = nullptr;
};

frame_pool::frame_pool()
    : d_(new impl())
{
    int n = (int)std::thread::hardware_concurrency();
    n = n <= 0 ? 4 : (n > 8 ? 8 : n);
    if (const char* e = getenv("HP_PARSER_THREADS"))
        n = atoi(e) < 1 ? 1 : (atoi(e) > 64 ? 64 : atoi(e));
    n_threads_ = n - 1; // the caller is the n-th worker
    for (int t = 0; t < n_threads_; ++t)
        d_->threads.emplace_back([this, t] {
            unsigned long seen = 0;
            for (;;) {
                std::unique_lock<std::mutex> lk(d_->m);
                d_->cv_work.wait(lk, [&] { return d_->stop || d_->generation != seen; });
                if (d_->stop)
                    return;
                seen = d_->generation;
                lk.unlock();
                for (int f; (f = d_->next.fetch_add(1)) < d_->n_frames;)
                    d_->fn(f, t + 1, d_->ctx);
                lk.lock();
                if (--d_->active == 0)
                    d_->cv_done.notify_one();
            }
        });
}

frame_pool::~frame_pool()
{
    {
        std::lock_guard<std::mutex> lk(d_->m);
        d_->stop = true;
    }
    d_->cv_work.notify_all();
    for (auto& t : d_->threads)
        t.join();
    delete d_;
}

frame_pool& frame_pool::instance()
{
    static frame_pool pool;
    return pool;
}

void frame_pool::run(int n_frames, void (*fn)(int, int, void*), void* ctx)
{
    if (n_frames <= 0)
        return;
    if (n_frames == 1 || n_threads_ == 0) {
        for (int f = 0; f < n_frames; ++f)
            fn(f, 0, ctx);
        return;
    }
    std::lock_guard<std::mutex> job(d_->busy);
    {
        std::lock_guard<std::mutex> lk(d_->m);
        d_->n_frames = n_frames, d_->fn = fn, d_->ctx = ctx;
        d_->next.store(0);
        d_->active = n_threads_;
        ++d_->generation;
    }
    d_->cv_work.notify_all();
    for (int f; (f = d_->next.fetch_add(1)) < n_frames;)
        fn(f, 0, ctx);
    std::unique_lock<std::mutex> lk(d_->m);
    d_->cv_done.wait(lk, [&] { return d_->active == 0; });
}

} // namespace hp

extern "C" {

const char* hp_last_error(void) { return hp::last_error(); }

const char* hp_version(void) { return "hyperpose-mi355x 0.1 (gfx950)"; }

int hp_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        hp::set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return HP_ERR_NO_DEVICE;
    }
    return n;
}

int hp_init(int device)
{
    int n = hp_device_count();
    if (n <= 0) {
        if (n == 0)
            hp::set_error("no HIP device visible");
        return HP_ERR_NO_DEVICE;
    }
    HP_REQUIRE(device >= 0 && device < n, HP_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    HP_HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HP_HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        hp::set_error("device %d is %s; libhp_hip.so is built for gfx950 only", device, prop.gcnArchName);
        return HP_ERR_NO_DEVICE;
    }
    return HP_OK;
}

int hp_malloc(void** dev, size_t nbytes)
{
    HP_REQUIRE(dev, HP_ERR_INVALID, "hp_malloc: null out pointer");
    HP_HIP_TRY(hipMalloc(dev, nbytes));
    return HP_OK;
}

int hp_free(void* dev)
{
    HP_HIP_TRY(hipFree(dev));
    return HP_OK;
}

int hp_malloc_host(void** host, size_t nbytes)
{
    HP_REQUIRE(host, HP_ERR_INVALID, "hp_malloc_host: null out pointer");
    HP_HIP_TRY(hipHostMalloc(host, nbytes, hipHostMallocDefault));
    return HP_OK;
}

int hp_free_host(void* host)
{
    HP_HIP_TRY(hipHostFree(host));
    return HP_OK;
}

int hp_memcpy_h2d(void* dev, const void* host, size_t nbytes)
{
    HP_HIP_TRY(hipMemcpy(dev, host, nbytes, hipMemcpyHostToDevice));
    return HP_OK;
}

int hp_memcpy_d2h(void* host, const void* dev, size_t nbytes)
{
    HP_HIP_TRY(hipMemcpy(host, dev, nbytes, hipMemcpyDeviceToHost));
    return HP_OK;
}

int hp_device_synchronize(void)
{
    HP_HIP_TRY(hipDeviceSynchronize());
    return HP_OK;
}

int hp_stream_wait_stream(void* waiter, void* signaler)
{
    // one-shot event: recorded on `signaler`, awaited by `waiter`, released by the runtime once both have passed it
    hipEvent_t ev = nullptr;
    HP_HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, (hipStream_t)signaler);
    if (e == hipSuccess)
        e = hipStreamWaitEvent((hipStream_t)waiter, ev, 0);
    (void)hipEventDestroy(ev); // deferred by HIP until the event has completed
    HP_HIP_TRY(e);
    return HP_OK;
}

} // extern "C"
