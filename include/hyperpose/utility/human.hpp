// hyperpose::human_t / body_part_t — same names and layout as the reference's
// include/hyperpose/utility/human.hpp:10-58 (18 parts x {bool,f32,f32,f32} + f32 score = 292 bytes, the layout
// of hp_human in include/hp_hip.h).  draw_human (visualisation) is out of scope.
#pragma once
#include <array>
#include <cstddef>

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

} // namespace hyperpose
