// oracle/ref_wrap.cpp — TEST INFRASTRUCTURE ONLY.
//
// extern "C" entry points around the REFERENCE's own parser classes, so that Python tests can call the
// real reference code.  This file is compiled together with the reference translation units
//   src/pose_proposal.cpp, src/pifpaf.cpp, src/pifpaf_decoder/{openpifpaf_postprocessor,math_helpers}.cpp,
//   src/logging.cpp
// taken directly from /root/reference (never copied into this repo) into oracle/_ref/libhp_ref.so by
// oracle/Makefile.  It only exists in containers where /root/reference is mounted; the prebuilt .so
// travels to the GPU box (oracle/_ref/ is git-ignored, not gpurun-ignored).
//
// The one reference symbol defined here rather than compiled from the reference is the
// feature_map_t constructor (src/data.cpp:5-10): data.cpp cannot be compiled without real OpenCV
// (it calls cv::resize / cv::copyMakeBorder), and the constructor is three member moves.
#include <cstring>
#include <memory>
#include <vector>

#include <hyperpose/operator/parser/pifpaf.hpp>
#include <hyperpose/operator/parser/proposal_network.hpp>

#include "oracle_common.h"

namespace hyperpose {
feature_map_t::feature_map_t(std::string name, std::unique_ptr<char[]>&& tensor, std::vector<int> shape)
    : m_name(std::move(name))
    , m_data(std::move(tensor))
    , m_shape(std::move(shape))
{
}
} // namespace hyperpose

namespace {

hyperpose::feature_map_t make_map(const char* name, const float* data, std::vector<int> shape)
{
    size_t n = 1;
    for (int s : shape)
        n *= (size_t)s;
    std::unique_ptr<char[]> buf(new char[n * sizeof(float)]);
    std::memcpy(buf.get(), data, n * sizeof(float));
    return hyperpose::feature_map_t(name, std::move(buf), std::move(shape));
}

int emit(const std::vector<hyperpose::human_t>& humans, o_human* out, int cap)
{
    static_assert(sizeof(hyperpose::human_t) == sizeof(o_human), "human_t layout");
    int n = 0;
    for (const auto& h : humans) {
        if (out && n < cap) {
            o_human o;
            std::memset(&o, 0, sizeof(o));
            o.score = h.score;
            for (int i = 0; i < O_COCO_N_PARTS; ++i) {
                o.parts[i].has_value = h.parts[i].has_value ? 1 : 0;
                o.parts[i].x = h.parts[i].x;
                o.parts[i].y = h.parts[i].y;
                o.parts[i].score = h.parts[i].score;
            }
            out[n] = o;
        }
        ++n;
    }
    return n;
}

} // namespace

extern "C" {

/* hyperpose::parser::pose_proposal::process (reference src/pose_proposal.cpp:68-337).
 * Tensors: 6 x [K,gh,gw] + edge [E,nh,nw,gh,gw]. */
int ref_ppn_process(int net_w, int net_h, float point_thresh, float limb_thresh, float nms_thresh,
    const float* conf_point, const float* conf_iou, const float* x, const float* y, const float* w, const float* h,
    const float* edge, int K, int gh, int gw, int E, int nh, int nw, o_human* out, int cap)
{
    hyperpose::parser::pose_proposal parser(cv::Size(net_w, net_h), point_thresh, limb_thresh, nms_thresh);
    const std::vector<int> s3{ K, gh, gw };
    auto humans = parser.process(make_map("conf_point", conf_point, s3), make_map("conf_iou", conf_iou, s3),
        make_map("x", x, s3), make_map("y", y, s3), make_map("w", w, s3), make_map("h", h, s3),
        make_map("edge", edge, { E, nh, nw, gh, gw }));
    return emit(humans, out, cap);
}

/* hyperpose::parser::pifpaf::process (reference src/pifpaf.cpp:7-95): arg0 = PAF/CAF [19,9,h,w],
 * arg1 = PIF/CIF [17,5,h,w] (the .cpp parameter order is authoritative). */
int ref_pifpaf_process(int net_h, int net_w, float thresh, const float* paf, const float* pif, int fh, int fw,
    o_human* out, int cap)
{
    hyperpose::parser::pifpaf parser(net_h, net_w, thresh);
    auto humans = parser.process(make_map("paf", paf, { 19, 9, fh, fw }), make_map("pif", pif, { 17, 5, fh, fw }));
    return emit(humans, out, cap);
}

} // extern "C"
