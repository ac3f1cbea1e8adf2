// tools/fillbench.hip — what a vector instruction costs BESIDE the matrix pipe (measurement probe, not part of the product).
// One block of NW wavefronts per CU (NW = 4: one per SIMD, 8: two per SIMD); every wavefront runs ITERS x { 6 MFMAs 32x32x16 f16 on six
// accumulators, K filler instructions of one kind after each }.  Prints s_memtime ticks per MFMA for every (kind, K).
// Kinds: 0 none; 1 v_fma_mix_f32 (VOP3P, fp16 operands: what the depthwise taps of sepconv_pipe*_kernel use); 2 v_fma_f32 (plain VOP3);
//        3 v_cvt_f32_f16 + v_fma_f32 alternating (the "convert at use" form); 4 v_pk_fma_f32; 5 ds_read_b64 + 4 x v_fma_mix (tap-like mix)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fillbench.hip -o gpurun_out/fillbench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int KIND, int K>
__global__ __launch_bounds__(512) void fill_kernel(float* out, unsigned long long* ticks, int iters)
{
    __shared__ float lds[4096];
    half8 a, b;
    for (int i = 0; i < 8; ++i)
        a[i] = (_Float16)(threadIdx.x * 0.001f + i), b[i] = (_Float16)(i * 0.5f);
    floatx16 acc[6] = {};
    float v[12];
    unsigned xh[4], wh[4];
    float xf[4], wf[4];
    for (int i = 0; i < 12; ++i)
        v[i] = threadIdx.x * 0.5f + i;
    for (int i = 0; i < 4; ++i)
        xh[i] = 0x3c003c00u + threadIdx.x + i, wh[i] = 0x38003800u + i, xf[i] = 1.f + i, wf[i] = 0.5f + threadIdx.x * 1e-3f;
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
            if (KIND == 5 && K > 0) {
                uint2 r = *reinterpret_cast<const uint2*>(&lds[(threadIdx.x * 2 + j * 64) & 4094]);
                xh[j & 3] ^= r.x & 1u;
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int d = (j * K + k) % 12, s = (j + k) & 3;
                if (KIND == 1 || KIND == 5)
                    asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(v[d]) : "v"(xh[s]), "v"(wh[s]));
                else if (KIND == 2)
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[d]) : "v"(xf[s]), "v"(wf[s]));
                else if (KIND == 3) {
                    if (k & 1)
                        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[d]) : "v"(xf[s]), "v"(wf[s]));
                    else
                        asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(xf[s]) : "v"(xh[s]));
                } else if (KIND == 4) {
                    f32x2 t = { v[d], v[(d + 1) % 12] }, x2 = { xf[s], xf[(s + 1) & 3] }, w2 = { wf[s], wf[(s + 1) & 3] };
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(t) : "v"(x2), "v"(w2));
                    v[d] = t[0], v[(d + 1) % 12] = t[1];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int j = 0; j < 6; ++j)
        for (int r = 0; r < 16; ++r)
            s += acc[j][r];
    for (int i = 0; i < 12; ++i)
        s += v[i];
    for (int i = 0; i < 4; ++i)
        s += xf[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0)
        ticks[0] = t1 - t0;
}

template <int KIND, int K>
static void run(int nw, float* out, unsigned long long* ticks, const char* name)
{
    const int iters = 2000;
    hipLaunchKernelGGL((fill_kernel<KIND, K>), dim3(256), dim3(nw * 64), 0, 0, out, ticks, iters);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((fill_kernel<KIND, K>), dim3(256), dim3(nw * 64), 0, 0, out, ticks, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long t = 0;
    CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
    const double per = (double)t / (iters * 6.0);
    // ns per MFMA slot of ONE wavefront; with two wavefronts per SIMD the SIMD retires two MFMAs per slot
    printf("%-28s K=%d  waves/SIMD=%d  ticks per MFMA of a wavefront %7.1f   ns per MFMA %6.2f   (SIMD: %5.1f ns per MFMA)\n", name, K, nw / 4, per,
        ms * 1e6 / (iters * 6.0), ms * 1e6 / (iters * 6.0) / (nw / 4));
}

int main()
{
    float* out;
    unsigned long long* ticks;
    CK(hipMalloc(&out, 256 * 512 * 4));
    CK(hipMalloc(&ticks, 64));
    for (int nw : { 4, 8 }) {
        run<0, 0>(nw, out, ticks, "MFMA only");
        run<1, 4>(nw, out, ticks, "v_fma_mix_f32");
        run<1, 6>(nw, out, ticks, "v_fma_mix_f32");
        run<1, 8>(nw, out, ticks, "v_fma_mix_f32");
        run<2, 4>(nw, out, ticks, "v_fma_f32");
        run<2, 6>(nw, out, ticks, "v_fma_f32");
        run<2, 8>(nw, out, ticks, "v_fma_f32");
        run<3, 6>(nw, out, ticks, "v_cvt_f32_f16 / v_fma_f32");
        run<3, 8>(nw, out, ticks, "v_cvt_f32_f16 / v_fma_f32");
        run<3, 10>(nw, out, ticks, "v_cvt_f32_f16 / v_fma_f32");
        run<4, 3>(nw, out, ticks, "v_pk_fma_f32");
        run<4, 4>(nw, out, ticks, "v_pk_fma_f32");
        run<5, 4>(nw, out, ticks, "ds_read_b64 + v_fma_mix");
        run<5, 6>(nw, out, ticks, "ds_read_b64 + v_fma_mix");
    }
    return 0;
}
