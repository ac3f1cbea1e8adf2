// tools/pkfma_hazard.hip — do packed-fp32 vector instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) give the results of their scalar forms
// while wavefronts of ANOTHER kernel on the same CU issue matrix instructions?  (round 6)
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pkfma_hazard.hip -o /tmp/pkfma_hazard && /tmp/pkfma_hazard
//
// Victim (stream 1): every lane evaluates the same chain twice per round - once with packed instructions on float2 halves, once with the scalar
// instructions - on operands that come either from registers or from LDS, and counts the rounds in which the two disagree bit for bit (they are
// the same IEEE operations: a correct machine counts zero).  Aggressor (stream 2): a kernel that keeps the matrix pipe busy with
// v_mfma_f32_32x32x8_f16 / v_mfma_f32_32x32x16_f16 / v_mfma_f32_32x32x2_f32, or nothing.  Found with hp::first_conv32_kernel: isolated output
// values differed between runs whenever fp16-MFMA kernels of another stream shared its CUs; compiled without packed FMAs it is exact.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));                   \
            std::exit(1);                                                               \
        }                                                                               \
    } while (0)

// OP: 0 = fma, 1 = mul, 2 = add;  LDS: operands re-read from LDS every round
template <int OP, bool LDS>
__global__ __launch_bounds__(256) void victim(unsigned* bad, int rounds, float seed)
{
    __shared__ float s_x[256 * 8];
    const int t = threadIdx.x;
    for (int i = 0; i < 8; ++i)
        s_x[t * 8 + i] = seed * (float)(1 + ((t * 8 + i) * 2654435761u >> 20) % 1000) * 1e-3f;
    __syncthreads();
    unsigned nbad = 0;
    f32x2 acc = { 0.25f, -0.5f };
    float a0 = 0.25f, a1 = -0.5f;
    for (int r = 0; r < rounds; ++r) {
        f32x2 x, w;
        if (LDS) {
            const int k = (t + r) & 255;
            x = f32x2{ s_x[k * 8 + 0], s_x[k * 8 + 1] };
            w = f32x2{ s_x[k * 8 + 2], s_x[k * 8 + 3] };
        } else {
            x = f32x2{ 0.001f * (float)((t + r) & 1023), 0.002f * (float)((t * 3 + r) & 511) };
            w = f32x2{ 1.0f - 0.0005f * (float)(r & 255), 0.75f + 0.0007f * (float)(t & 127) };
        }
        float s0, s1;
        f32x2 pk;
        if (OP == 0) {
            pk = __builtin_elementwise_fma(x, w, acc);
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s0) : "v"(x.x), "v"(w.x), "v"(a0));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s1) : "v"(x.y), "v"(w.y), "v"(a1));
        } else if (OP == 1) {
            pk = x * w;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s0) : "v"(x.x), "v"(w.x));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(s1) : "v"(x.y), "v"(w.y));
        } else {
            pk = x + w;
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(s0) : "v"(x.x), "v"(w.x));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(s1) : "v"(x.y), "v"(w.y));
        }
        // keep the packed form packed: the compiler may not split a value it cannot see through
        asm volatile("" : "+v"(pk));
        nbad += (__float_as_uint(pk.x) != __float_as_uint(s0)) + (__float_as_uint(pk.y) != __float_as_uint(s1));
        // carry a bounded state so that the chain is a chain
        acc = f32x2{ pk.x * 0.5f, pk.y * 0.5f };
        a0 = s0 * 0.5f, a1 = s1 * 0.5f;
    }
    if (nbad)
        atomicAdd(bad, nbad);
}

// KIND: 0 = v_mfma_f32_32x32x8_f16, 1 = v_mfma_f32_32x32x16_f16, 2 = v_mfma_f32_32x32x2_f32
template <int KIND>
__global__ __launch_bounds__(256) void aggressor(float* sink, int rounds)
{
    floatx16 acc;
    for (int i = 0; i < 16; ++i)
        acc[i] = 0.f;
    const float v = 1e-3f * (float)(threadIdx.x & 63);
    for (int r = 0; r < rounds; ++r) {
        if (KIND == 0) {
            const half4 a = { (_Float16)v, (_Float16)0.5f, (_Float16)0.25f, (_Float16)1.f };
            acc = __builtin_amdgcn_mfma_f32_32x32x8f16(a, a, acc, 0, 0, 0);
        } else if (KIND == 1) {
            const half8 a = { (_Float16)v, (_Float16)0.5f, (_Float16)0.25f, (_Float16)1.f, (_Float16)v, (_Float16)0.5f, (_Float16)0.25f, (_Float16)1.f };
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc, 0, 0, 0);
        } else
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v, 0.5f, acc, 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i)
        s += acc[i];
    if (s == 12345.678f)
        sink[0] = s;
}

template <int OP, bool LDS>
static unsigned run(int aggr, hipStream_t sv, hipStream_t sa, unsigned* d_bad, float* d_sink)
{
    CHECK(hipMemset(d_bad, 0, 4));
    for (int it = 0; it < 40; ++it) {
        if (aggr == 0)
            hipLaunchKernelGGL(aggressor<0>, dim3(1024), dim3(256), 0, sa, d_sink, 4000);
        else if (aggr == 1)
            hipLaunchKernelGGL(aggressor<1>, dim3(1024), dim3(256), 0, sa, d_sink, 4000);
        else if (aggr == 2)
            hipLaunchKernelGGL(aggressor<2>, dim3(1024), dim3(256), 0, sa, d_sink, 2000);
        hipLaunchKernelGGL((victim<OP, LDS>), dim3(2048), dim3(256), 0, sv, d_bad, 2000, 1.0f + 0.01f * it);
    }
    CHECK(hipStreamSynchronize(sv));
    CHECK(hipStreamSynchronize(sa));
    unsigned h = 0;
    CHECK(hipMemcpy(&h, d_bad, 4, hipMemcpyDeviceToHost));
    return h;
}

int main()
{
    hipStream_t sv, sa;
    CHECK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    unsigned* d_bad;
    float* d_sink;
    CHECK(hipMalloc(&d_bad, 4));
    CHECK(hipMalloc(&d_sink, 4));
    const char* aggr_name[4] = { "v_mfma_f32_32x32x8_f16 ", "v_mfma_f32_32x32x16_f16", "v_mfma_f32_32x32x2_f32 ", "(no second kernel)     " };
    std::printf("packed fp32 results that differ from the scalar instruction's, of 40 launches x 2048 blocks x 256 lanes x 2000 rounds x 2 halves = 8.4e10 each:\n");
    for (int aggr = 0; aggr < 4; ++aggr) {
        std::printf("  next to %s  v_pk_fma_f32: registers %u, LDS operands %u | v_pk_mul_f32: %u, %u | v_pk_add_f32: %u, %u\n", aggr_name[aggr],
            run<0, false>(aggr, sv, sa, d_bad, d_sink), run<0, true>(aggr, sv, sa, d_bad, d_sink), run<1, false>(aggr, sv, sa, d_bad, d_sink),
            run<1, true>(aggr, sv, sa, d_bad, d_sink), run<2, false>(aggr, sv, sa, d_bad, d_sink), run<2, true>(aggr, sv, sa, d_bad, d_sink));
    }
    return 0;
}
