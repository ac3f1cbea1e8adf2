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

// every block streams the same `bytes` region `reps` times with 16-byte loads (L2 / MALL resident read bandwidth)
__global__ __launch_bounds__(256) void read_loop(const float4* in, size_t n16, int reps, float* out)
{
    float4 acc = make_float4(0, 0, 0, 0);
    for (int r = 0; r < reps; ++r)
        for (size_t i = (size_t)blockIdx.x % 7 * 256 + threadIdx.x; i < n16; i += 256 * 8) {
            const float4 v0 = in[i], v1 = in[(i + 256 * 2) % n16], v2 = in[(i + 256 * 4) % n16], v3 = in[(i + 256 * 6) % n16];
            acc.x += v0.x + v1.x + v2.x + v3.x, acc.y += v0.y + v1.y + v2.y + v3.y, acc.z += v0.z + v1.z, acc.w += v2.w + v3.w;
        }
    if (acc.x == 12345.f) out[0] = acc.y + acc.z + acc.w;
}

// dw-like access pattern without compute: block = 8x8 pixel tile x 64 channels (128 B per pixel at C*2-byte stride)
__global__ __launch_bounds__(256) void tile_copy(const __half* in, __half* out, int B, int H, int W, int C, int tiles_x, int tiles_y, int cgroups)
{
    int t = blockIdx.x; const int cgi = t % cgroups; t /= cgroups; const int tx = t % tiles_x; t /= tiles_x; const int ty = t % tiles_y, b = t / tiles_y;
    const int chunk = threadIdx.x & 7;
    for (int pass = 0; pass < 2; ++pass) {
        const int pix = (threadIdx.x >> 3) + pass * 32, y = ty * 8 + pix / 8, x = tx * 8 + pix % 8;
        if (y < H && x < W) {
            const size_t off = (((size_t)b * H + y) * W + x) * C + cgi * 64 + chunk * 8;
            *reinterpret_cast<float4*>(out + off) = *reinterpret_cast<const float4*>(in + off);
        }
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
    for (size_t mb : {1, 2, 16, 128}) {
        const size_t bytes = mb << 20; float4* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
        const int reps = mb >= 16 ? 2 : 16, blocks = 256 * 4;
        float ms = time_ms(s, 20, [&] { hipLaunchKernelGGL(read_loop, dim3(blocks), dim3(256), 0, s, buf, bytes / 16, reps, dout); });
        const double rd = (double)blocks * reps * (bytes / 2.0); // each block reads 4 of every 8 256-vector groups
        printf("read_loop %zu MB region: %.3f ms  %.1f TB/s  = %.1f B/clk/CU @2.1GHz\n", mb, ms, rd / ms / 1e9, rd / ms / 1e3 / 256 / 2.1e9 * 1e3);
        CK(hipFree(buf));
    }
    {   // depthwise 3x3 at LW-OpenPose shapes vs a plain 16-byte copy of the same tensor
        for (int C : {512, 128}) {
            const int B = 8, H = 46, W = 54, P = 1, Hp = H + 2, Wp = W + 2;
            __half *in, *out, *w; float* bias;
            size_t n_in = (size_t)B * Hp * Wp * C;
            CK(hipMalloc(&in, n_in * 2)); CK(hipMalloc(&out, n_in * 2)); CK(hipMalloc(&w, 9 * C * 2)); CK(hipMalloc(&bias, C * 4));
            CK(hipMemset(in, 0x11, n_in * 2)); CK(hipMemset(w, 0x11, 9 * C * 2)); CK(hipMemset(bias, 0, C * 4));
            hp::dw_params p{};
            p.in = hp::tview{ in + ((size_t)P * Wp + P) * C, C, 0, Wp, Hp * Wp };
            p.out = hp::tview{ out + ((size_t)P * Wp + P) * C, C, 0, Wp, Hp * Wp };
            p.B = B, p.H = H, p.W = W, p.OH = H, p.OW = W, p.C = C, p.stride = 1, p.dil = 1, p.pad_t = p.pad_l = 1, p.halo = 1;
            p.w = w, p.bias = bias, p.act = hp::ACT_RELU, p.act_param = 0;
            float ms = time_ms(s, 300, [&] { CK(hp::launch_dwconv3x3(p, s)); });
            double bytes = 2.0 * B * H * W * C * 2;
            printf("dwconv3x3 C=%d B=8 46x54: %.1f us  %.0f GB/s (r+w once)\n", C, ms * 1e3, bytes / ms / 1e6);
            {
                const int tx_ = (W + 7) / 8, ty_ = (H + 7) / 8, cg_ = C / 64;
                float ms3 = time_ms(s, 300, [&] { hipLaunchKernelGGL(tile_copy, dim3(tx_ * ty_ * cg_ * B), dim3(256), 0, s, in, out, B, H, W, C, tx_, ty_, cg_); });
                printf("  tile-pattern copy (8x8 px x 64 ch blocks): %.1f us  %.0f GB/s\n", ms3 * 1e3, bytes / ms3 / 1e6);
            }
            float ms2 = time_ms(s, 300, [&] { hipLaunchKernelGGL(copy4, dim3(256 * 8), dim3(256), 0, s, (const float4*)in, (float4*)out, n_in * 2 / 16); });
            printf("  plain copy of the same %.1f MB tensor: %.1f us  %.0f GB/s\n", n_in * 2 / 1e6, ms2 * 1e3, 2.0 * n_in * 2 / ms2 / 1e6);
            CK(hipFree(in)); CK(hipFree(out)); CK(hipFree(w)); CK(hipFree(bias));
        }
    }
    {   // fused depthwise + pointwise block at the LW-OpenPose backbone shape
        for (int C : {512, 128}) {
            const int B = 8, H = 46, W = 54, P = 1, Hp = H + 2, Wp = W + 2, cout = C;
            __half *in, *out, *dww, *pww; float *dwb, *pwb;
            size_t n_in = (size_t)B * Hp * Wp * C, n_out = (size_t)B * H * W * cout;
            CK(hipMalloc(&in, n_in * 2)); CK(hipMalloc(&out, n_out * 2)); CK(hipMalloc(&dww, 9 * C * 2)); CK(hipMalloc(&pww, (size_t)cout * C * 2));
            CK(hipMalloc(&dwb, C * 4)); CK(hipMalloc(&pwb, cout * 4));
            CK(hipMemset(in, 0x11, n_in * 2)); CK(hipMemset(dww, 0x11, 9 * C * 2)); CK(hipMemset(pww, 0x11, (size_t)cout * C * 2));
            CK(hipMemset(dwb, 0, C * 4)); CK(hipMemset(pwb, 0, cout * 4));
            hp::sep_params p{};
            p.in = hp::tview{ in + ((size_t)P * Wp + P) * C, C, 0, Wp, Hp * Wp };
            p.B = B, p.H = H, p.W = W, p.OH = H, p.OW = W, p.C = C, p.stride = 1, p.dil = 1, p.pad_t = p.pad_l = 1, p.halo = 1;
            p.dw_w = dww, p.dw_bias = dwb, p.dw_slope = 0.f, p.dw_hi = 6.f;
            auto& q = p.pw;
            q.w = pww, q.bias = pwb, q.alpha = nullptr, q.act = hp::ACT_RELU; hp::set_act(q);
            q.B = B, q.H = H, q.W = W, q.OH = H, q.OW = W, q.Cin = C, q.Cout = cout, q.Cout_pad = cout, q.KH = q.KW = 1, q.stride = 1, q.dil = 1;
            q.res = hp::tview{ nullptr, 0, 0, 0, 0 }, q.out = hp::tview{ out, cout, 0, W, H * W }, q.out_f32 = nullptr, q.npix = B * H * W, q.dbg = nullptr;
            float ms = time_ms(s, 200, [&] { CK(hp::launch_sepconv(p, s)); });
            unsigned long long* dbg; CK(hipMalloc(&dbg, 128 * 8)); CK(hipMemset(dbg, 0, 128 * 8));
            q.dbg = dbg; CK(hp::launch_sepconv(p, s)); CK(hipStreamSynchronize(s));
            unsigned long long h[128]; CK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
            printf("sepconv C=%d->%d variant %d: %.1f us\n  timeline:", C, cout, hp::sepconv_variant(p), ms * 1e3);
            for (int i = 1; i < 40 && h[i]; ++i) printf(" %llu", h[i] - h[i - 1]);
            printf("\n  epilogue:");
            for (int i = 41; i < 64 && h[i]; ++i) printf(" %llu", h[i] - h[i - 1]);
            printf("\n");
            CK(hipFree(dbg)); CK(hipFree(in)); CK(hipFree(out)); CK(hipFree(dww)); CK(hipFree(pww)); CK(hipFree(dwb)); CK(hipFree(pwb));
        }
    }
    {   // fused two-layer head 128 -> 512 -> 38
        const int B = 8, H = 46, W = 54, K1 = 128, C2 = 38;
        __half *in, *w1, *w2, *out; float *b1, *b2, *of;
        CK(hipMalloc(&in, (size_t)B * H * W * K1 * 2)); CK(hipMalloc(&w1, 512 * K1 * 2)); CK(hipMalloc(&w2, 64 * 512 * 2));
        CK(hipMalloc(&out, (size_t)B * H * W * 64 * 2)); CK(hipMalloc(&b1, 512 * 4)); CK(hipMalloc(&b2, 64 * 4)); CK(hipMalloc(&of, (size_t)B * C2 * H * W * 4));
        CK(hipMemset(in, 0x11, (size_t)B * H * W * K1 * 2)); CK(hipMemset(w1, 0x11, 512 * K1 * 2)); CK(hipMemset(w2, 0x11, 64 * 512 * 2));
        CK(hipMemset(b1, 0, 512 * 4)); CK(hipMemset(b2, 0, 64 * 4));
        hp::head_params p{};
        p.in = hp::tview{ in, K1, 0, W, H * W };
        p.B = B, p.H = H, p.W = W, p.K1 = K1, p.w1 = w1, p.b1 = b1, p.hi1 = 1e30f, p.w2 = w2;
        auto& q = p.pw;
        q.bias = b2, q.alpha = nullptr, q.act = hp::ACT_NONE; hp::set_act(q);
        q.B = B, q.OH = H, q.OW = W, q.Cout = C2, q.Cout_pad = 64, q.out = hp::tview{ out, 64, 0, W, H * W }, q.out_f32 = of, q.dbg = nullptr;
        float ms = time_ms(s, 200, [&] { CK(hp::launch_mlp_head(p, s)); });
        unsigned long long* dbg; CK(hipMalloc(&dbg, 64 * 8)); CK(hipMemset(dbg, 0, 64 * 8));
        q.dbg = dbg; CK(hp::launch_mlp_head(p, s)); CK(hipStreamSynchronize(s));
        unsigned long long h[64]; CK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
        printf("mlp_head 128->512->38: %.1f us\n  timeline:", ms * 1e3);
        for (int i = 1; i < 20 && h[i]; ++i) printf(" %llu", h[i] - h[i - 1]);
        printf("\n");
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
        p.w_layout = hp::conv_weight_layout(p);
        float ms = time_ms(s, 300, [&] { CK(hp::launch_conv_mfma(p, s)); });
        if (c.B == 8) {
            unsigned long long* dbg; CK(hipMalloc(&dbg, 64 * 8)); CK(hipMemset(dbg, 0, 64 * 8));
            p.dbg = dbg; CK(hp::launch_conv_mfma(p, s)); CK(hipStreamSynchronize(s));
            unsigned long long h[64]; CK(hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost));
            printf("  timeline (s_memtime deltas):");
            for (int i = 1; i < 40 && h[i]; ++i) printf(" %llu", h[i] - h[i - 1]);
            printf("\n  epilogue stamps:");
            for (int i = 41; i < 64 && h[i]; ++i) printf(" %llu", h[i] - h[i - 1]);
            printf("  total %llu\n", h[0] ? 0ull : 0ull);
            p.dbg = nullptr; CK(hipFree(dbg));
        }
        double fl = 2.0 * p.npix * c.cout * c.k * c.k * c.cin;
        printf("conv %dx%d %d->%d B=%d tile=%d: %.1f us  %.1f TFLOP/s\n", c.k, c.k, c.cin, c.cout, c.B, hp::conv_mfma_tile(p), ms * 1e3, fl / ms / 1e9);
        CK(hipFree(in)); CK(hipFree(out)); CK(hipFree(w)); CK(hipFree(bias));
    }
    return 0;
}
