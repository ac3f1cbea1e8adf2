// resize.hip — the stream front-end's per-frame geometry on gfx950: cv::resize (INTER_LINEAR, CV_8UC3) and
// hyperpose::non_scaling_resize (reference src/data.cpp:53-69, used at src/stream.cpp:89-103 and
// src/tensorrt.cpp:448), plus resume_ratio (include/hyperpose/utility/human.hpp:44-58) for the way back.
//
// One thread = one output pixel.  Every output pixel re-derives its own source coordinates and 11-bit fixed-point
// coefficients with exactly OpenCV's arithmetic (double scale, float fractional part, round-half-even to short,
// integer horizontal pass, ((b*(S>>4))>>16) vertical pass, 2x2 down-scale rerouted to the area average, equal sizes
// copied) - see oracle/resize_oracle.cpp for the derivation; the two agree bit for bit (tests/test_resize_gpu.py).
// Frames are a few hundred KB: the kernels are latency-trivial next to the conv stack; what matters is that the
// frames never go back to the host between decode and parse.
#include "hp_common.hpp"

#include <cmath>

namespace {

struct rz_params {
    const uint8_t* src;
    int sw, sh, src_stride;
    uint8_t* dst;
    int dw, dh, dst_stride; // full destination frame
    int iw, ih;             // resized region (top-left); the rest of the frame gets the border colour
    int mode;               // 0 linear, 1 area 2x2, 2 copy
    double scale_x, scale_y;
    int bg[3];
};

__device__ __forceinline__ short sat_short_rn(float v)
{
    const int r = __float2int_rn(v);
    return (short)min(max(r, -32768), 32767);
}

__global__ __launch_bounds__(256) void resize_u8c3_kernel(const rz_params p)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= p.dw || y >= p.dh)
        return;
    uint8_t* d = p.dst + (size_t)y * p.dst_stride + x * 3;
    if (x >= p.iw || y >= p.ih) {
        d[0] = (uint8_t)p.bg[0], d[1] = (uint8_t)p.bg[1], d[2] = (uint8_t)p.bg[2];
        return;
    }
    if (p.mode == 2) {
        const uint8_t* s = p.src + (size_t)y * p.src_stride + x * 3;
        d[0] = s[0], d[1] = s[1], d[2] = s[2];
        return;
    }
    if (p.mode == 1) {
        const uint8_t *s0 = p.src + (size_t)(2 * y) * p.src_stride + (2 * x) * 3, *s1 = s0 + p.src_stride;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            d[c] = (uint8_t)((s0[c] + s0[3 + c] + s1[c] + s1[3 + c] + 2) >> 2);
        return;
    }
    float fx = (float)((x + 0.5) * p.scale_x - 0.5);
    int sx = (int)floorf(fx);
    fx -= sx;
    if (sx < 0)
        fx = 0.f, sx = 0;
    const bool two_tap = sx + 1 < p.sw; // dx < xmax
    if (sx >= p.sw - 1)
        fx = 0.f, sx = p.sw - 1;
    const int a0 = sat_short_rn((1.f - fx) * 2048.f), a1 = sat_short_rn(fx * 2048.f);
    float fy = (float)((y + 0.5) * p.scale_y - 0.5);
    const int sy = (int)floorf(fy);
    fy -= sy;
    const int b0 = sat_short_rn((1.f - fy) * 2048.f), b1 = sat_short_rn(fy * 2048.f);
    const int y0 = min(max(sy, 0), p.sh - 1), y1 = min(max(sy + 1, 0), p.sh - 1);
    const uint8_t *r0 = p.src + (size_t)y0 * p.src_stride + sx * 3, *r1 = p.src + (size_t)y1 * p.src_stride + sx * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int h0 = two_tap ? r0[c] * a0 + r0[3 + c] * a1 : r0[c] * 2048;
        const int h1 = two_tap ? r1[c] * a0 + r1[3 + c] * a1 : r1[c] * 2048;
        d[c] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
    }
}

int launch_resize(const uint8_t* src, int sw, int sh, int src_stride, uint8_t* dst, int dw, int dh, int dst_stride, int iw, int ih,
    const int bg[3], hipStream_t s)
{
    HP_REQUIRE(src && dst && sw > 0 && sh > 0 && dw > 0 && dh > 0 && iw >= 0 && ih >= 0 && iw <= dw && ih <= dh, HP_ERR_INVALID, "resize: bad geometry");
    HP_REQUIRE(src_stride >= sw * 3 && dst_stride >= dw * 3, HP_ERR_INVALID, "resize: row stride smaller than a row");
    rz_params p;
    p.src = src, p.sw = sw, p.sh = sh, p.src_stride = src_stride, p.dst = dst, p.dw = dw, p.dh = dh, p.dst_stride = dst_stride;
    p.iw = iw, p.ih = ih, p.bg[0] = bg[0], p.bg[1] = bg[1], p.bg[2] = bg[2];
    p.mode = 0, p.scale_x = 1, p.scale_y = 1;
    if (iw > 0 && ih > 0) {
        const double inv_scale_x = (double)iw / sw, inv_scale_y = (double)ih / sh;
        p.scale_x = 1. / inv_scale_x, p.scale_y = 1. / inv_scale_y;
        const int iscale_x = (int)std::lrint(p.scale_x), iscale_y = (int)std::lrint(p.scale_y);
        const bool is_area_fast = std::abs(p.scale_x - iscale_x) < 2.220446049250313e-16 && std::abs(p.scale_y - iscale_y) < 2.220446049250313e-16;
        if (sw == iw && sh == ih)
            p.mode = 2;
        else if (is_area_fast && iscale_x == 2 && iscale_y == 2)
            p.mode = 1;
    }
    hipLaunchKernelGGL(resize_u8c3_kernel, dim3(hp::ceil_div(dw, 32), hp::ceil_div(dh, 8)), dim3(256), 0, s, p);
    HP_HIP_TRY(hipGetLastError());
    return HP_OK;
}

} // namespace

extern "C" {

int hp_resize_u8c3(const uint8_t* dev_src, int sw, int sh, int src_stride, uint8_t* dev_dst, int dw, int dh, int dst_stride, void* stream)
{
    const int bg[3] = { 0, 0, 0 };
    return launch_resize(dev_src, sw, sh, src_stride, dev_dst, dw, dh, dst_stride, dw, dh, bg, (hipStream_t)stream);
}

void hp_letterbox_inner(int sw, int sh, int dw, int dh, int* iw, int* ih)
{
    // src/data.cpp:57-64, doubles truncated by cv::Size(int, int)
    const double h1 = dw * (sh / (double)sw);
    const double w2 = dh * (sw / (double)sh);
    if (h1 <= dh)
        *iw = dw, *ih = (int)h1;
    else
        *iw = (int)w2, *ih = dh;
}

int hp_letterbox_u8c3(const uint8_t* dev_src, int sw, int sh, int src_stride, uint8_t* dev_dst, int dw, int dh, int dst_stride, int b, int g,
    int r, void* stream)
{
    HP_REQUIRE(sw > 0 && sh > 0, HP_ERR_INVALID, "letterbox: empty source");
    int iw = 0, ih = 0;
    hp_letterbox_inner(sw, sh, dw, dh, &iw, &ih);
    const int bg[3] = { b, g, r };
    return launch_resize(dev_src, sw, sh, src_stride, dev_dst, dw, dh, dst_stride, iw, ih, bg, (hipStream_t)stream);
}

void hp_resume_ratio(hp_human* humans, int n, int src_w, int src_h, int dst_w, int dst_h)
{
    // include/hyperpose/utility/human.hpp:44-58 (float *= double: promote, multiply, narrow)
    if (!humans)
        return;
    if ((long)src_h * dst_w > (long)src_w * dst_h) {
        const double xratio = (double)dst_w * src_h / ((double)dst_h * src_w);
        for (int i = 0; i < n; ++i)
            for (auto& part : humans[i].parts)
                part.x = (float)(part.x * xratio);
    } else {
        const double yratio = (double)dst_h * src_w / ((double)dst_w * src_h);
        for (int i = 0; i < n; ++i)
            for (auto& part : humans[i].parts)
                part.y = (float)(part.y * yratio);
    }
}

} // extern "C"
