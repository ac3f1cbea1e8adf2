// tools/storebench.hip — how fast can 256 CUs WRITE?  (epilogue design input; not part of the product)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/storebench.hip -o tools/storebench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// every wave writes `per_wave` bytes as 16 B/lane stores; seg = contiguous bytes per "pixel", stride = bytes between pixels
template <int NT>
__global__ __launch_bounds__(256) void store_pattern(u32x4* out, size_t total_pieces, int seg16, int stride16, int nwaves_total)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const u32x4 v = { 1u, 2u, 3u, (unsigned)wave };
    // wave handles a contiguous range of "pixels"; lanes: seg16 lanes per pixel
    const int ppw = 64 / seg16; // pixels per store instruction
    const size_t pix_total = total_pieces / seg16;
    const size_t pix_per_wave = (pix_total + nwaves_total - 1) / nwaves_total;
    size_t p0 = wave * pix_per_wave;
    for (size_t p = p0 + lane / seg16; p < p0 + pix_per_wave && p < pix_total; p += ppw) {
        u32x4* dst = out + p * stride16 + lane % seg16;
        if (NT)
            __builtin_nontemporal_store(v, dst);
        else
            *dst = v;
    }
}

static float time_ms(hipStream_t s, int iters, const std::function<void()>& f)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) f();
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) f();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main()
{
    hipStream_t s; CK(hipStreamCreate(&s));
    const size_t cap = 5ull << 30;
    u32x4* buf; CK(hipMalloc(&buf, cap));
    for (size_t mb : { 600 }) {
        const size_t bytes = mb << 20;
        for (int blocks : { 2048 }) {
            struct pat { int seg16, stride16; const char* name; };
            for (pat pt : { pat{ 64, 64, "1KB contiguous/instr" }, pat{ 16, 64, "256B per px, 1KB stride (25% dense)" }, pat{ 16, 16, "256B per px dense" },
                            pat{ 8, 8, "128B per px dense" }, pat{ 4, 16, "64B per px, 256B stride" }, pat{ 4, 64, "64B per px, 1KB stride" }, pat{ 8, 64, "128B per px, 1KB stride" } }) {
                // total bytes WRITTEN = bytes; footprint = bytes * stride/seg
                const size_t pieces = bytes / 16;
                if (pieces / pt.seg16 * pt.stride16 * 16 > cap) continue;
                for (int nt = 0; nt < 2; ++nt) {
                    float ms = time_ms(s, 30, [&] {
                        if (nt) hipLaunchKernelGGL(store_pattern<1>, dim3(blocks), dim3(256), 0, s, buf, pieces, pt.seg16, pt.stride16, blocks * 4);
                        else hipLaunchKernelGGL(store_pattern<0>, dim3(blocks), dim3(256), 0, s, buf, pieces, pt.seg16, pt.stride16, blocks * 4);
                    });
                    printf("%4zu MB written, %4d blocks, %-38s %s: %7.1f us  %6.2f TB/s\n", mb, blocks, pt.name, nt ? "nt " : "   ", ms * 1e3, bytes / ms / 1e9);
                }
            }
        }
    }
    return 0;
}
