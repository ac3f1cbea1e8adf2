// oracle/resize_oracle.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// CPU restatement of what the reference's stream front-end does to every frame before inference:
//   cv::resize(input, output, size)                 (default INTER_LINEAR; src/stream.cpp:93,101)
//   hyperpose::non_scaling_resize(input, size)      (src/data.cpp:53-69: resize keeping the aspect ratio into the
//                                                    top-left corner + cv::copyMakeBorder(BORDER_CONSTANT, bgcolor))
// for 8-bit, 3-channel images.  OpenCV is a third-party dependency that is absent here (the reference pins none;
// its Dockerfile installs the distribution's libopencv-dev, 4.2 on Ubuntu 20.04): the algorithm below is OpenCV 4's
// published imgproc/resize.cpp for CV_8UC3 —
//   * scale factors in double, source coordinate fx = (float)((dx + 0.5) * scale - 0.5), sx = floor(fx), fx -= sx;
//     horizontally sx < 0 -> (0, fx = 0) and sx >= w-1 -> (w-1, fx = 0); vertically rows are only clipped;
//   * 11-bit fixed-point coefficients saturate_cast<short>(c * 2048) (round half to even);
//   * horizontal pass in int: S[sx]*a0 + S[sx+1]*a1 (S[sx]*2048 from the first dx whose sx+1 leaves the row);
//   * vertical pass: (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
//   * an exact 2x2 down-scale is rerouted to INTER_AREA: (a + b + c + d + 2) >> 2;
//   * equal sizes copy.
// PARITY UNPINNED: no OpenCV build and no reference fixture for this step exist in the container; the restatement
// is checked against a float bilinear model (+-1 LSB) and against its own invariants in tests/test_resize.py.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

inline short sat_short_round(float v)
{
    long r = std::lrintf(v); // cvRound: round half to even (default rounding mode)
    return (short)std::min<long>(std::max<long>(r, -32768), 32767);
}

} // namespace

extern "C" {

// src [sh][sw][3] (row stride src_stride bytes) -> dst [dh][dw][3] (row stride dst_stride bytes)
void oracle_resize_linear_u8c3(const uint8_t* src, int sw, int sh, int src_stride, uint8_t* dst, int dw, int dh, int dst_stride)
{
    if (sw == dw && sh == dh) {
        for (int y = 0; y < sh; ++y)
            memcpy(dst + (size_t)y * dst_stride, src + (size_t)y * src_stride, (size_t)sw * 3);
        return;
    }
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    const int iscale_x = (int)std::lrint(scale_x), iscale_y = (int)std::lrint(scale_y); // saturate_cast<int>(double)
    const bool is_area_fast = std::abs(scale_x - iscale_x) < 2.220446049250313e-16 && std::abs(scale_y - iscale_y) < 2.220446049250313e-16;
    if (is_area_fast && iscale_x == 2 && iscale_y == 2) { // INTER_LINEAR -> INTER_AREA (fast) for an exact 2x2 down-scale
        for (int y = 0; y < dh; ++y) {
            const uint8_t *s0 = src + (size_t)(2 * y) * src_stride, *s1 = s0 + src_stride;
            uint8_t* d = dst + (size_t)y * dst_stride;
            for (int x = 0; x < dw; ++x)
                for (int c = 0; c < 3; ++c)
                    d[x * 3 + c] = (uint8_t)((s0[(2 * x) * 3 + c] + s0[(2 * x + 1) * 3 + c] + s1[(2 * x) * 3 + c] + s1[(2 * x + 1) * 3 + c] + 2) >> 2);
        }
        return;
    }
    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(2 * dw), ibeta(2 * dh);
    int xmax = dw;
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)std::floor(fx);
        fx -= sx;
        if (sx < 0)
            fx = 0, sx = 0;
        if (sx + 1 >= sw) {
            xmax = std::min(xmax, dx);
            if (sx >= sw - 1)
                fx = 0, sx = sw - 1;
        }
        xofs[dx] = sx;
        ialpha[2 * dx] = sat_short_round((1.f - fx) * 2048.f);
        ialpha[2 * dx + 1] = sat_short_round(fx * 2048.f);
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        const int sy = (int)std::floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[2 * dy] = sat_short_round((1.f - fy) * 2048.f);
        ibeta[2 * dy + 1] = sat_short_round(fy * 2048.f);
    }
    std::vector<int> row0((size_t)dw * 3), row1((size_t)dw * 3);
    auto hresize = [&](int sy, std::vector<int>& out) {
        const uint8_t* S = src + (size_t)sy * src_stride;
        for (int dx = 0; dx < dw; ++dx) {
            const int sx = xofs[dx];
            for (int c = 0; c < 3; ++c)
                out[dx * 3 + c] = dx < xmax ? S[sx * 3 + c] * ialpha[2 * dx] + S[(sx + 1) * 3 + c] * ialpha[2 * dx + 1] : S[sx * 3 + c] * 2048;
        }
    };
    auto clip = [](int v, int lo, int hi) { return v < lo ? lo : (v >= hi ? hi - 1 : v); };
    for (int dy = 0; dy < dh; ++dy) {
        hresize(clip(yofs[dy], 0, sh), row0);
        hresize(clip(yofs[dy] + 1, 0, sh), row1);
        const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        uint8_t* d = dst + (size_t)dy * dst_stride;
        for (int i = 0; i < dw * 3; ++i)
            d[i] = (uint8_t)((((b0 * (row0[i] >> 4)) >> 16) + ((b1 * (row1[i] >> 4)) >> 16) + 2) >> 2);
    }
}

// inner size of non_scaling_resize (src/data.cpp:57-64): doubles truncated by cv::Size(int, int)
void oracle_letterbox_inner(int sw, int sh, int dw, int dh, int* iw, int* ih)
{
    const double h1 = dw * (sh / (double)sw);
    const double w2 = dh * (sw / (double)sh);
    if (h1 <= dh)
        *iw = dw, *ih = (int)h1;
    else
        *iw = (int)w2, *ih = dh;
}

// non_scaling_resize: dst [dh][dw][3] packed
void oracle_letterbox_u8c3(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, int b, int g, int r)
{
    int iw, ih;
    oracle_letterbox_inner(sw, sh, dw, dh, &iw, &ih);
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x) {
            uint8_t* p = dst + ((size_t)y * dw + x) * 3;
            p[0] = (uint8_t)b, p[1] = (uint8_t)g, p[2] = (uint8_t)r;
        }
    if (iw > 0 && ih > 0)
        oracle_resize_linear_u8c3(src, sw, sh, sw * 3, dst, iw, ih, dw * 3);
}

// resume_ratio (include/hyperpose/utility/human.hpp:44-58) on n (x, y) pairs, in place
void oracle_resume_ratio(float* xy, int n, int src_w, int src_h, int dst_w, int dst_h)
{
    if (src_h * dst_w > src_w * dst_h) {
        const double xratio = (double)dst_w * src_h / (dst_h * src_w);
        for (int i = 0; i < n; ++i)
            xy[2 * i] *= xratio;
    } else {
        const double yratio = (double)dst_h * src_w / (dst_w * src_h);
        for (int i = 0; i < n; ++i)
            xy[2 * i + 1] *= yratio;
    }
}

} // extern "C"
