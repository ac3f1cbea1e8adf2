// hyperpose::human_t / body_part_t — same names and layout as the reference's
// include/hyperpose/utility/human.hpp:10-58 (18 parts x {bool,f32,f32,f32} + f32 score = 292 bytes, the layout
// of hp_human in include/hp_hip.h); draw_human restates src/human.cpp:7-39.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <utility>

#include "cv_min.hpp"

namespace hyperpose {

constexpr int COCO_N_PARTS = 18;
constexpr int COCO_N_PAIRS = 19;

struct body_part_t {
    bool has_value = false;
    float x = 0;
    float y = 0;
    float score = 0;
};

template <size_t J>
struct human_t_ {
    std::array<body_part_t, J> parts;
    float score;
};

using human_t = human_t_<COCO_N_PARTS>;
static_assert(sizeof(human_t) == 292, "human_t must match hp_human");

// reference human.hpp:44-58
template <size_t J>
inline void resume_ratio(human_t_<J>& human, cv::Size src, cv::Size dst)
{
    if (src.height * dst.width > src.width * dst.height) {
        double xratio = (double)dst.width * src.height / (dst.height * src.width);
        for (auto& par : human.parts)
            par.x *= xratio;
    } else {
        double yratio = (double)dst.height * src.width / (dst.width * src.height);
        for (auto& par : human.parts)
            par.y *= yratio;
    }
}

// ---- draw_human (reference include/hyperpose/utility/human.hpp:36-42, src/human.cpp:7-39, colours src/color.hpp:16-36, limb table
// src/coco.hpp:32-51): limbs as lines, parts as filled circles, thickness = max(1, int(sqrt(bbox area in pixels)) / 32).
// With OpenCV the drawing calls are the reference's (cv::line / cv::circle); without it (cv_min.hpp) a plain rasteriser draws the same
// geometry - a capsule of the same thickness and a disc of the same radius - which is visualisation, not part of any parity claim.
namespace detail {
    constexpr uint8_t coco_colors_rgb[19][3] = { { 255, 0, 0 }, { 255, 85, 0 }, { 255, 170, 0 }, { 255, 255, 0 }, { 170, 255, 0 }, { 85, 255, 0 },
        { 0, 255, 0 }, { 0, 255, 85 }, { 0, 255, 170 }, { 0, 255, 255 }, { 0, 170, 255 }, { 0, 85, 255 }, { 0, 0, 255 }, { 85, 0, 255 },
        { 170, 0, 255 }, { 255, 0, 255 }, { 255, 0, 170 }, { 255, 0, 85 }, { 127, 127, 127 } };
    constexpr int coco_pairs[COCO_N_PAIRS][2] = { { 1, 2 }, { 1, 5 }, { 2, 3 }, { 3, 4 }, { 5, 6 }, { 6, 7 }, { 1, 8 }, { 8, 9 }, { 9, 10 },
        { 1, 11 }, { 11, 12 }, { 12, 13 }, { 1, 0 }, { 0, 14 }, { 14, 16 }, { 0, 15 }, { 15, 17 }, { 2, 16 }, { 5, 17 } };
#ifndef HYPERPOSE_USE_OPENCV
    inline void put(cv::Mat& img, int x, int y, const uint8_t* rgb)
    {
        if (x < 0 || y < 0 || x >= img.cols || y >= img.rows)
            return;
        uint8_t* px = img.data() + ((size_t)y * img.cols + x) * 3;
        px[0] = rgb[2], px[1] = rgb[1], px[2] = rgb[0]; // BGR
    }
    inline void disc(cv::Mat& img, int cx, int cy, int r, const uint8_t* rgb)
    {
        for (int y = cy - r; y <= cy + r; ++y)
            for (int x = cx - r; x <= cx + r; ++x)
                if ((x - cx) * (x - cx) + (y - cy) * (y - cy) <= r * r)
                    put(img, x, y, rgb);
    }
    inline void capsule(cv::Mat& img, int x0, int y0, int x1, int y1, int thickness, const uint8_t* rgb)
    {
        const float half = std::max(0.5f, thickness * 0.5f);
        const int pad = (int)std::ceil(half);
        const float dx = float(x1 - x0), dy = float(y1 - y0), len2 = dx * dx + dy * dy;
        for (int y = std::min(y0, y1) - pad; y <= std::max(y0, y1) + pad; ++y)
            for (int x = std::min(x0, x1) - pad; x <= std::max(x0, x1) + pad; ++x) {
                float t = len2 > 0 ? ((x - x0) * dx + (y - y0) * dy) / len2 : 0.f;
                t = std::min(1.f, std::max(0.f, t));
                const float ex = x - (x0 + t * dx), ey = y - (y0 + t * dy);
                if (ex * ex + ey * ey <= half * half)
                    put(img, x, y, rgb);
            }
    }
#endif
} // namespace detail

inline void draw_human(cv::Mat& img, const human_t& human)
{
    float n = 1, s = 0, w = 1, e = 0;
    for (const auto& p : human.parts)
        if (p.has_value) {
            n = std::min(n, p.y);
            s = std::max(s, p.y);
            w = std::min(w, p.x);
            e = std::max(e, p.x);
        }
    const int thickness = std::max(1, static_cast<int>(std::sqrt((e - w) * (s - n) * img.size().area())) / 32);
    for (int pair_id = 0; pair_id < COCO_N_PAIRS; ++pair_id) {
        const auto p1 = human.parts[detail::coco_pairs[pair_id][0]];
        const auto p2 = human.parts[detail::coco_pairs[pair_id][1]];
        const uint8_t* c = detail::coco_colors_rgb[pair_id];
        if (p1.has_value && p2.has_value) {
#ifdef HYPERPOSE_USE_OPENCV
            cv::line(img, cv::Point(p1.x * img.cols, p1.y * img.rows), cv::Point(p2.x * img.cols, p2.y * img.rows), cv::Scalar(c[2], c[1], c[0]), thickness);
#else
            detail::capsule(img, int(p1.x * img.cols), int(p1.y * img.rows), int(p2.x * img.cols), int(p2.y * img.rows), thickness, c);
#endif
        }
    }
    for (int part_idx = 0; part_idx < COCO_N_PARTS; ++part_idx) {
        const uint8_t* c = detail::coco_colors_rgb[part_idx];
        const auto p = human.parts[part_idx];
        if (p.has_value) {
#ifdef HYPERPOSE_USE_OPENCV
            cv::circle(img, cv::Point(p.x * img.cols, p.y * img.rows), thickness, cv::Scalar(c[2], c[1], c[0]), cv::FILLED);
#else
            detail::disc(img, int(p.x * img.cols), int(p.y * img.rows), thickness, c);
#endif
        }
    }
}

} // namespace hyperpose
