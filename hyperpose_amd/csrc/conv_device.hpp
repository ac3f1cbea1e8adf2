// conv_device.hpp — device-side vocabulary shared by the convolution translation units (conv_kernels.hip, conv_chain.hip):
// native vector types, the halo-view address, the LDS-only barrier and the launch macro that lets
// hp_engine_profile_sequence bracket a kernel with its own begin / end timestamps.
#pragma once
#include "conv_kernels.hpp"

#include <hip/hip_ext.h>

// Every launch goes through HP_LAUNCH: normally a plain launch; while hp::prof_start / prof_stop are set
// (hp_engine_profile_sequence) the launch carries the two events, which then hold the kernel's OWN begin / end timestamps
// (what rocprofv3's kernel trace reports) without putting extra packets between the kernels.
#define HP_LAUNCH(kernel, grid, block, lds, stream, ...)                                                            \
    do {                                                                                                            \
        if (hp::prof_start)                                                                                         \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, hp::prof_start, hp::prof_stop, 0, __VA_ARGS__); \
        else                                                                                                        \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                      \
    } while (0)

namespace hp {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4))); // native 16-byte vector (HIP's uint4 struct defeats SROA here)

__device__ __forceinline__ long tv_off(const tview& t, int b, int y, int x)
{
    return ((long)b * t.img + (long)y * t.wp + x) * t.cs + t.coff;
}

// Workgroup barrier that only waits for this wave's LDS traffic: __syncthreads() also drains vmcnt, i.e. every global
// prefetch in flight (measured: 1.5k cycles per K-chunk in sepconv_kernel when the weight prefetch crosses a barrier).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

} // namespace hp
