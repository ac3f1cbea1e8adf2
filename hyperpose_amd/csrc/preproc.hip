// preproc.hip — replaces hyperpose::nhwc_images_append_nchw_batch (reference src/data.cpp:21-51), the scalar
// push_back loop the reference marks "TODO: Parallel".  u8 HWC -> f32 CHW, value = (float)((double)u8 * factor),
// channel order {2,1,0} when flip_rb.  HBM-bound: 3 B read + 12 B written per pixel; each thread converts
// 4 consecutive pixels (12 contiguous input bytes, one float4 store per plane).
// Inside the engine this conversion is fused into the first convolution's load instead; this kernel serves
// the float-buffer entry point (tensorrt::inference(const std::vector<float>&, size_t), src/tensorrt.cpp:364).
#include "hp_common.hpp"

namespace {

__global__ __launch_bounds__(256) void preproc_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int n,
    int plane, double factor, int flip_rb)
{
    const int quads = (plane + 3) / 4;
    const long long total = (long long)n * quads;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(i / quads), q = (int)(i - (long long)b * quads);
        const int p0 = q * 4;
        const uint8_t* src = in + ((size_t)b * plane + p0) * 3;
        float* dst = out + (size_t)b * 3 * plane;
        const int cnt = min(4, plane - p0);
        uint8_t px[12];
        if (cnt == 4 && ((((size_t)src) & 3) == 0)) {
            const uint32_t* s32 = reinterpret_cast<const uint32_t*>(src);
            uint32_t w0 = s32[0], w1 = s32[1], w2 = s32[2];
            memcpy(px, &w0, 4), memcpy(px + 4, &w1, 4), memcpy(px + 8, &w2, 4);
        } else {
            for (int j = 0; j < cnt * 3; ++j)
                px[j] = src[j];
        }
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            const int c = flip_rb ? 2 - ci : ci;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[j] = (float)((double)px[j * 3 + c] * factor);
            float* d = dst + (size_t)ci * plane + p0;
            if (cnt == 4 && ((((size_t)d) & 15) == 0))
                *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
            else
                for (int j = 0; j < cnt; ++j)
                    d[j] = v[j];
        }
    }
}

} // namespace

extern "C" int hp_preproc_u8hwc_to_f32nchw(const uint8_t* dev_hwc, int n, int h, int w, double factor, int flip_rb,
    float* dev_nchw, void* stream)
{
    HP_REQUIRE(dev_hwc && dev_nchw, HP_ERR_INVALID, "hp_preproc: null pointer");
    HP_REQUIRE(n >= 0 && h > 0 && w > 0, HP_ERR_INVALID, "hp_preproc: bad shape n=%d h=%d w=%d", n, h, w);
    if (n == 0)
        return HP_OK; // reference: `if (images.empty()) return;` (data.cpp:23-24)
    const int plane = h * w;
    const long long total = (long long)n * ((plane + 3) / 4);
    const int blocks = (int)std::min<long long>((total + 255) / 256, 256 * 8);
    hipLaunchKernelGGL(preproc_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dev_hwc, dev_nchw, n, plane, factor, flip_rb);
    HP_HIP_TRY(hipGetLastError());
    return HP_OK;
}
