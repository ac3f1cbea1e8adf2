// tools/clock_probe.hip — the shader clock the chip SUSTAINS under matrix-pipe load, measured from inside the kernel:
// every wavefront issues N independent-accumulator MFMAs back to back (4 accumulators: the issue rate, not the dependent latency, bounds
// it).  An MFMA occupies its SIMD's matrix pipe for a fixed number of shader cycles (v_mfma_f32_32x32x16_f16: 32;
// v_mfma_f32_32x32x2_f32: 64 - MI355X_MICROARCH.md), so
//     shader clock = N * cycles_per_mfma * (wavefronts per SIMD) / the kernel's duration on the EVENT clock.
// The kernel also brackets its loop with s_memtime and prints ticks / cycle: on gfx950 the tick IS the shader cycle (1.000 in every case,
// loaded or not - MI355X_MICROARCH.md says the same), NOT a constant 100 MHz reference: an s_memtime difference measures cycles, never
// time, and cannot by itself say anything about the clock (DESIGN.md section 7.5 of round 4 assumed it could).
// The operands here are constants (few toggling bits): this is the clock of the matrix pipe issuing at full rate with minimal data
// power - the UPPER bound of what a real kernel sustains; bench.py's sampler (amdgpu hwmon freq1_input / power1_input every 20 ms
// during each timed region) gives the clocks of the real workloads.
// The nominal peaks (2.5 PFLOP/s fp16, 157.3 TFLOP/s fp32) assume 2.4 GHz; the fractions in bench.py / DESIGN.md are of those nominal peaks,
// and this tool says how much of the gap is clock.  Cases: one wavefront on an otherwise idle chip, 1 wavefront per SIMD on every CU,
// 2 per SIMD; each for ~20 ms after a 50 ms ramp of the same load.  Also checks s_memtime's rate against the host's event clock.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/clock_probe.hip -o gpurun_out/clock_probe && gpurun_out/clock_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <bool F32>
__global__ __launch_bounds__(256) void mfma_chain(int n_iter, unsigned long long* ticks, float* sink)
{
    floatx16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r)
            acc[i][r] = 0.001f * (i + 1) * (r + 1); // four DIFFERENT chains: identical ones would be merged into one
    half8 a, b;
    for (int e = 0; e < 8; ++e)
        a[e] = (_Float16)(0.001f * (threadIdx.x & 7)), b[e] = (_Float16)0.5f;
    const float af = 0.001f * (threadIdx.x & 7), bf = 0.5f;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
    for (int it = 0; it < n_iter; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (F32)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[i], 0, 0, 0);
                else
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
            }
    }
    // the last MFMA of every chain has delivered before the clock is read (a VALU read of its result waits for it)
    float s = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];
    asm volatile("v_mov_b32 %0, %0\n\ts_nop 0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(s), "=s"(t1)::"memory");
    for (int i = 0; i < 4; ++i)
        for (int r = 1; r < 16; ++r)
            s += acc[i][r];
    if ((threadIdx.x & 63) == 0)
        ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    if (s == 12345.678f)
        sink[0] = s;
}

template <bool F32>
static void run_case(const char* what, int blocks, int threads, int wps, double target_ms, double ticks_per_us, hipStream_t st, unsigned long long* dticks, float* dsink)
{
    const int cyc = F32 ? 64 : 32, per_iter = 32;
    // calibrate the iteration count on the nominal clock, then ramp and measure
    const int n_iter = (int)(target_ms * 1e-3 * 2.4e9 / (per_iter * cyc) / wps);
    const int waves = blocks * threads / 64;
    std::vector<unsigned long long> h(waves);
    for (int rep = 0; rep < 3; ++rep) // ~3 x target of ramp
        hipLaunchKernelGGL(mfma_chain<F32>, dim3(blocks), dim3(threads), 0, st, n_iter, dticks, dsink);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(mfma_chain<F32>, dim3(blocks), dim3(threads), 0, st, n_iter, dticks, dsink);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), dticks, waves * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double n_mfma = (double)n_iter * per_iter;
    (void)ticks_per_us;
    printf("%-58s %8.3f ms  shader clock %7.1f MHz   s_memtime ticks per MFMA cycle: median %.4f, slowest wave %.4f   (%d waves, %d per SIMD)\n", what, ms,
        n_mfma * cyc * wps / (ms * 1e3), h[waves / 2] / (n_mfma * cyc * wps), h[waves - 1] / (n_mfma * cyc * wps), waves, wps);
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreate(&st));
    unsigned long long* dticks;
    float* dsink;
    CK(hipMalloc(&dticks, 256 * 8 * 8 * 8));
    CK(hipMalloc(&dsink, 64));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("%s: %d CUs, clockRate %d kHz (the driver's nominal)\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    // s_memtime rate against the event clock: one long single-wave kernel
    double ticks_per_us = 100.0;
    {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(mfma_chain<false>, dim3(1), dim3(64), 0, st, 200000, dticks, dsink);
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(mfma_chain<false>, dim3(1), dim3(64), 0, st, 200000, dticks, dsink);
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long t;
        CK(hipMemcpy(&t, dticks, 8, hipMemcpyDeviceToHost));
        printf("s_memtime: %llu ticks inside a one-wavefront kernel of %.0f MFMA cycles that the event clock times at %.3f ms -> %.2f ticks/us\n", t, 200000.0 * 32 * 32, ms,
            t / (ms * 1e3));
    }
    const int cus = prop.multiProcessorCount;
    run_case<false>("fp16 MFMA 32x32x16, ONE wavefront, chip otherwise idle", 1, 64, 1, 20, ticks_per_us, st, dticks, dsink);
    run_case<false>("fp16 MFMA 32x32x16, 1 wavefront per SIMD on every CU", cus, 256, 1, 20, ticks_per_us, st, dticks, dsink);
    run_case<false>("fp16 MFMA 32x32x16, 2 wavefronts per SIMD on every CU", cus * 2, 256, 2, 20, ticks_per_us, st, dticks, dsink);
    run_case<true>("fp32 MFMA 32x32x2, ONE wavefront, chip otherwise idle", 1, 64, 1, 20, ticks_per_us, st, dticks, dsink);
    run_case<true>("fp32 MFMA 32x32x2, 1 wavefront per SIMD on every CU", cus, 256, 1, 20, ticks_per_us, st, dticks, dsink);
    run_case<true>("fp32 MFMA 32x32x2, 2 wavefronts per SIMD on every CU", cus * 2, 256, 2, 20, ticks_per_us, st, dticks, dsink);
    // a longer soak: 1 s of the full fp16 load, the clock of the last 20 ms
    for (int i = 0; i < 50; ++i)
        hipLaunchKernelGGL(mfma_chain<false>, dim3(cus * 2), dim3(256), 0, st, (int)(20e-3 * 2.4e9 / (32 * 32) / 2), dticks, dsink);
    run_case<false>("fp16 MFMA, 2 per SIMD, after 1 s of the same load", cus * 2, 256, 2, 20, ticks_per_us, st, dticks, dsink);
    for (int i = 0; i < 50; ++i)
        hipLaunchKernelGGL(mfma_chain<true>, dim3(cus * 2), dim3(256), 0, st, (int)(20e-3 * 2.4e9 / (32 * 64) / 2), dticks, dsink);
    run_case<true>("fp32 MFMA, 2 per SIMD, after 1 s of the same load", cus * 2, 256, 2, 20, ticks_per_us, st, dticks, dsink);
    return 0;
}
