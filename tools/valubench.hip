// tools/valubench.hip — issue cost (cycles per wave-instruction, one wave per SIMD) of the VALU forms a depthwise tap can use
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/valubench.hip -o tools/valubench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define REP8(X) X X X X X X X X
#define BODY(NAME, ASM)                                                                                     \
    __global__ __launch_bounds__(64) void NAME(float* out, unsigned long long* cyc, int iters)              \
    {                                                                                                       \
        float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;                     \
        unsigned x = 0x3c003c00u + threadIdx.x, w = 0x38003800u;                                            \
        float xf = 1.0001f, wf = 0.999f;                                                                    \
        float __attribute__((ext_vector_type(2))) p0 = { 1, 2 }, p1 = { 3, 4 }, p2 = { 5, 6 }, p3 = { 7, 8 }, px = { 1.0001f, 0.9999f }, pw = { 0.999f, 1.001f }; \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                               \
        for (int i = 0; i < iters; ++i) {                                                                   \
            REP8(ASM)                                                                                       \
        }                                                                                                   \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                               \
        out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[1] + p2[0] + p3[1]; \
        if (threadIdx.x == 0 && blockIdx.x == 0)                                                            \
            cyc[0] = t1 - t0;                                                                               \
    }

BODY(k_fma_mix, asm volatile("v_fma_mix_f32 %0, %8, %9, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n v_fma_mix_f32 %1, %8, %9, %1 op_sel:[1,1,0] op_sel_hi:[1,1,0]\n"
    "v_fma_mix_f32 %2, %8, %9, %2 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n v_fma_mix_f32 %3, %8, %9, %3 op_sel:[1,1,0] op_sel_hi:[1,1,0]\n"
    "v_fma_mix_f32 %4, %8, %9, %4 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n v_fma_mix_f32 %5, %8, %9, %5 op_sel:[1,1,0] op_sel_hi:[1,1,0]\n"
    "v_fma_mix_f32 %6, %8, %9, %6 op_sel:[0,0,0] op_sel_hi:[1,1,0]\n v_fma_mix_f32 %7, %8, %9, %7 op_sel:[1,1,0] op_sel_hi:[1,1,0]\n"
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));)
BODY(k_fma_f32, asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
    "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(xf), "v"(wf));)
BODY(k_pk_fma_f32, asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n"
    "v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n"
    : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(px), "v"(pw));)
BODY(k_cvt_f32_f16, asm volatile("v_cvt_f32_f16 %0, %8\n v_cvt_f32_f16 %1, %8\n v_cvt_f32_f16 %2, %8\n v_cvt_f32_f16 %3, %8\n"
    "v_cvt_f32_f16 %4, %9\n v_cvt_f32_f16 %5, %9\n v_cvt_f32_f16 %6, %9\n v_cvt_f32_f16 %7, %9\n"
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));)
BODY(k_pk_fma_f16, asm volatile("v_pk_fma_f16 %0, %8, %9, %0\n v_pk_fma_f16 %1, %8, %9, %1\n v_pk_fma_f16 %2, %8, %9, %2\n v_pk_fma_f16 %3, %8, %9, %3\n"
    "v_pk_fma_f16 %4, %8, %9, %4\n v_pk_fma_f16 %5, %8, %9, %5\n v_pk_fma_f16 %6, %8, %9, %6\n v_pk_fma_f16 %7, %8, %9, %7\n"
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));)
BODY(k_dot2_f32_f16, asm volatile("v_dot2_f32_f16 %0, %8, %9, %0\n v_dot2_f32_f16 %1, %8, %9, %1\n v_dot2_f32_f16 %2, %8, %9, %2\n v_dot2_f32_f16 %3, %8, %9, %3\n"
    "v_dot2_f32_f16 %4, %8, %9, %4\n v_dot2_f32_f16 %5, %8, %9, %5\n v_dot2_f32_f16 %6, %8, %9, %6\n v_dot2_f32_f16 %7, %8, %9, %7\n"
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(w));)
BODY(k_med3, asm volatile("v_med3_f32 %0, %0, %8, %9\n v_med3_f32 %1, %1, %8, %9\n v_med3_f32 %2, %2, %8, %9\n v_med3_f32 %3, %3, %8, %9\n"
    "v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_med3_f32 %6, %6, %8, %9\n v_med3_f32 %7, %7, %8, %9\n"
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(xf), "v"(wf));)

int main()
{
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 1024 * 64 * 4)); CK(hipMalloc(&cyc, 8));
    const int iters = 2000;
    auto run = [&](const char* name, void (*k)(float*, unsigned long long*, int), int waves_per_cu_mult) {
        for (int blocks : { 1, 1024, 2048, 4096 }) { // 1 wave total; 1 / 2 / 4 waves per SIMD chip-wide
            hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
            CK(hipDeviceSynchronize());
            unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
            printf("%-16s %5d waves: %6.2f cycles per wave-instruction (wave 0's view)\n", name, blocks, (double)c / (iters * 64.0));
        }
    };
    run("v_fma_mix_f32", k_fma_mix, 1);
    run("v_fma_f32", k_fma_f32, 1);
    run("v_pk_fma_f32", k_pk_fma_f32, 1);
    run("v_cvt_f32_f16", k_cvt_f32_f16, 1);
    run("v_pk_fma_f16", k_pk_fma_f16, 1);
    run("v_dot2_f32_f16", k_dot2_f32_f16, 1);
    run("v_med3_f32", k_med3, 1);
    return 0;
}
