// tools/microbench.hip — sanity microbenchmarks on the GPU box (not part of the product):
//   1. MFMA-only loop (v_mfma_f32_32x32x16_f16)  -> achieved TFLOP/s  (clock sanity)
//   2. streaming float4 copy                      -> GB/s
//   3. hp::launch_conv_mfma on LW-OpenPose layer shapes, long runs (clock ramped)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/microbench.hip -Lhyperpose_amd -lhp_hip -o gpurun_out/microbench
#include "../hyperpose_amd/csrc/conv_kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void mfma_only(float* out, int iters)
{
    half8 a, b;
    for (int i = 0; i < 8; ++i) a[i] = (_Float16)(threadIdx.x * 0.001f + i), b[i] = (_Float16)(i * 0.5f);
    floatx16 acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void copy4(const float4* in, float4* out, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) out[i] = in[i];
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

int main(int argc, char** argv)
{
    hipStream_t s; CK(hipStreamCreate(&s));
    float* dout; CK(hipMalloc(&dout, 256 * 2048 * 4));
    {
        const int iters = 20000, blocks = 256 * 4;
        float ms = time_ms(s, 20, [&] { hipLaunchKernelGGL(mfma_only, dim3(blocks), dim3(256), 0, s, dout, iters); });
        double fl = (double)blocks * 4 /*waves*/ * iters * 4.0 * 32 * 32 * 16 * 2;
        printf("mfma_only: %.3f ms  %.0f TFLOP/s\n", ms, fl / ms / 1e9);
        for (int bl : {256, 512, 768}) {
            float ms1 = time_ms(s, 20, [&] { hipLaunchKernelGGL(mfma_only, dim3(bl), dim3(256), 0, s, dout, iters); });
            double fl1 = (double)bl * 4 * iters * 4.0 * 32 * 32 * 16 * 2;
            printf("mfma_only %d blocks (%d wave/SIMD): %.3f ms  %.0f TFLOP/s\n", bl, bl / 256, ms1, fl1 / ms1 / 1e9);
        }
    }
    {
        size_t n = 512ull << 20; float4 *a, *b; CK(hipMalloc(&a, n)); CK(hipMalloc(&b, n));
        CK(hipMemset(a, 1, n));
        float ms = time_ms(s, 20, [&] { hipLaunchKernelGGL(copy4, dim3(256 * 8), dim3(256), 0, s, a, b, n / 16); });
        printf("copy 512MB: %.3f ms  %.0f GB/s (r+w)\n", ms, 2.0 * n / ms / 1e6);
        size_t m = 32ull << 20;
        ms = time_ms(s, 50, [&] { hipLaunchKernelGGL(copy4, dim3(256 * 8), dim3(256), 0, s, a, b, m / 16); });
        printf("copy 32MB (cache resident): %.3f ms  %.0f GB/s (r+w)\n", ms, 2.0 * m / ms / 1e6);
        CK(hipFree(a)); CK(hipFree(b));
    }
    struct cfg { int cin, cout, k, H, W, B; };
    std::vector<cfg> cfgs = { {128, 128, 3, 46, 54, 8}, {512, 512, 1, 46, 54, 8}, {128, 512, 1, 46, 54, 8}, {128, 128, 1, 46, 54, 8},
                              {128, 128, 3, 46, 54, 32}, {512, 512, 1, 46, 54, 32}, {128, 128, 3, 46, 54, 1} };
    for (auto c : cfgs) {
        hp::conv_params p{};
        const int cout_pad = c.cout > 64 ? (c.cout + 127) / 128 * 128 : 64;
        const int P = c.k / 2, Hp = c.H + 2 * P, Wp = c.W + 2 * P;
        __half *in, *out, *w; float* bias;
        size_t in_n = (size_t)c.B * Hp * Wp * c.cin, out_n = (size_t)c.B * c.H * c.W * c.cout, w_n = (size_t)c.k * c.k * cout_pad * c.cin;
        CK(hipMalloc(&in, in_n * 2)); CK(hipMalloc(&out, out_n * 2)); CK(hipMalloc(&w, w_n * 2)); CK(hipMalloc(&bias, cout_pad * 4));
        CK(hipMemset(in, 0x11, in_n * 2)); CK(hipMemset(w, 0x11, w_n * 2)); CK(hipMemset(bias, 0, cout_pad * 4));
        p.in = hp::tview{ in + ((size_t)P * Wp + P) * c.cin, c.cin, 0, Wp, Hp * Wp };
        p.B = c.B, p.H = c.H, p.W = c.W, p.OH = c.H, p.OW = c.W;
        p.Cin = c.cin, p.Cout = c.cout, p.Cout_pad = cout_pad, p.KH = p.KW = c.k, p.stride = 1, p.dil = 1, p.pad_t = p.pad_l = c.k / 2;
        p.w = w, p.bias = bias, p.alpha = nullptr, p.act = hp::ACT_RELU; hp::set_act(p);
        p.res = hp::tview{ nullptr, 0, 0, 0, 0 }, p.out = hp::tview{ out, c.cout, 0, c.W, c.H * c.W }, p.out_f32 = nullptr, p.npix = c.B * c.H * c.W;
        p.dbg = nullptr;
        float ms = time_ms(s, 300, [&] { CK(hp::launch_conv_mfma(p, s)); });
        if (c.k == 3 && c.B == 8) {
            unsigned long long* dbg; CK(hipMalloc(&dbg, 64 * 8)); CK(hipMemset(dbg, 0, 64 * 8));
            p.dbg = dbg; CK(hp::launch_conv_mfma(p, s)); CK(hipStreamSynchronize(s));
            unsigned long long h[64]; CK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
            printf("  timeline (s_memtime deltas):");
            for (int i = 1; i < 64 && h[i]; ++i) printf(" %llu", h[i] - h[i - 1]);
            printf("  total %llu\n", h[0] ? 0ull : 0ull);
            p.dbg = nullptr; CK(hipFree(dbg));
        }
        double fl = 2.0 * p.npix * c.cout * c.k * c.k * c.cin;
        printf("conv %dx%d %d->%d B=%d tile=%d: %.1f us  %.1f TFLOP/s\n", c.k, c.k, c.cin, c.cout, c.B, hp::conv_mfma_tile(p), ms * 1e3, fl / ms / 1e9);
        CK(hipFree(in)); CK(hipFree(out)); CK(hipFree(w)); CK(hipFree(bias));
    }
    return 0;
}
