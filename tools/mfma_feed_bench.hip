// tools/mfma_feed_bench.hip — what FEEDING the matrix pipe costs (measurement probe, not part of the product).
// The MFMA loops of conv_direct_kernel / conv_chain_kernel run at ~0.7 of the rate the pipe sustains from registers.
// One block of 8 wavefronts per CU (two per SIMD, as those kernels run); every wavefront repeats a "k16 step": 6 MFMAs 32x32x16 f16 on six
// accumulators whose operands were fetched during the previous step - NB B fragments by ds_read_b128 (swizzled rows of an LDS tile, as the
// kernels read them) and NA A fragments by global_load_dwordx4 (1 KB per wavefront and fragment out of a 2 MB buffer that stays in L2).
// Prints ns per MFMA and SIMD for each (NB, NA, how many steps ahead the A fragments are requested); 16 ns = the pipe's rate from registers
// at the clock it holds.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_feed_bench.hip -o gpurun_out/mfma_feed_bench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NB, int NA, int AHEAD>
__global__ __launch_bounds__(512, 1) void feed_kernel(const u32x4* __restrict__ wbuf, float* out, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[96 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 96 * 1024 / 16; i += 512)
        reinterpret_cast<u32x4*>(lds)[i] = u32x4{ 0x3c003c00u + (unsigned)i, 0x38003800u, 0x3c003c00u, 0x34003400u };
    __syncthreads();
    floatx16 acc[6] = {};
    u32x4 fb[2][6], fa[8][2]; // B fragments one step ahead, A fragments AHEAD steps ahead (a ring of eight steps)
    for (int j = 0; j < 6; ++j)
        fb[0][j] = fb[1][j] = u32x4{ 0x3c003c00u + (unsigned)lane, 0x38003800u, 0x3c003c00u, 0x34003400u };
    for (int q = 0; q < 8; ++q)
        fa[q][0] = fa[q][1] = u32x4{ 0x3c003c00u, 0x38003800u + (unsigned)lane, 0x3c003c00u, 0x34003400u };
    const int row = lane & 31, fk = lane >> 5;
    const u32x4* wp = wbuf + (size_t)(blockIdx.x % 32) * 2048 + wave * 256 + lane; // (<= 31 * 2048 + 7 * 256 + 63 + 7 * 128 + 64 < 128 K entries)
#pragma unroll 1
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            // operands of later steps
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int r = j * 32 + row, chunk = (2 * (st & 3) + fk) ^ (r & 15);
                fb[(st + 1) & 1][j] = *reinterpret_cast<const u32x4*>(lds + (size_t)((r + 7 * ((it + st) & 15)) % 352) * 256 + chunk * 16);
            }
#pragma unroll
            for (int a = 0; a < NA; ++a)
                fa[(st + AHEAD) & 7][a] = wp[((it * 2 + st) & 7) * 64 * 2 + a * 64];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                half8 x, y;
                __builtin_memcpy(&x, &fa[st][NA ? j % NA : 0], 16);
                __builtin_memcpy(&y, &fb[st & 1][NB ? j % NB : 0], 16);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[j], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (j < NB)
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if (NA)
                __builtin_amdgcn_sched_group_barrier(0x020, NA, 0);
        }
    }
    float s = 0;
    for (int j = 0; j < 6; ++j)
        for (int r = 0; r < 16; ++r)
            s += acc[j][r];
    out[blockIdx.x * 512 + tid] = s;
}

template <int NB, int NA, int AHEAD>
static void run(const u32x4* wbuf, float* out)
{
    const int iters = 4000;
    hipLaunchKernelGGL((feed_kernel<NB, NA, AHEAD>), dim3(256), dim3(512), 0, 0, wbuf, out, iters);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((feed_kernel<NB, NA, AHEAD>), dim3(256), dim3(512), 0, 0, wbuf, out, iters);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    // a wavefront issues iters / 4 * 48 = iters * 12 MFMAs; two wavefronts per SIMD
    printf("B fragments from LDS per step %d, A fragments from L2 per step %d (requested %d steps ahead):  %6.2f ns per MFMA and SIMD  (%.0f TFLOP/s on the chip)\n", NB, NA, AHEAD,
        ms * 1e6 / (iters * 12.0) / 2, 256.0 * 4 * 2 * iters * 12.0 * 32768 / (ms * 1e-3) / 1e12);
}

int main()
{
    u32x4* wbuf;
    float* out;
    CK(hipMalloc(&wbuf, 2u << 20));
    CK(hipMemset(wbuf, 0x3c, 2u << 20));
    CK(hipMalloc(&out, 256 * 512 * 4));
    run<0, 0, 1>(wbuf, out);
    run<3, 0, 1>(wbuf, out);
    run<6, 0, 1>(wbuf, out);
    run<0, 1, 1>(wbuf, out);
    run<0, 1, 4>(wbuf, out);
    run<0, 2, 4>(wbuf, out);
    run<6, 1, 1>(wbuf, out);
    run<6, 1, 4>(wbuf, out);
    run<6, 1, 7>(wbuf, out);
    run<3, 2, 4>(wbuf, out);
    run<3, 2, 7>(wbuf, out);
    return 0;
}
