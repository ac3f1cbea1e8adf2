// hyperpose::parser::pifpaf over libhp_hip.so — reference include/hyperpose/operator/parser/pifpaf.hpp:8-26.
// process(arg0, arg1): arg0 = PAF/CAF [19,9,h,w], arg1 = PIF/CIF [17,5,h,w] (the reference's .cpp parameter order,
// src/pifpaf.cpp:7; its header names them the other way round).  The engine also emits them as [171,h,w] / [85,h,w].
#pragma once
#include <cassert>
#include <cstdlib>
#include <iostream>

#include "../../../hp_hip.h"
#include "../../utility/data.hpp"

namespace hyperpose::parser {

class pifpaf {
public:
    inline explicit pifpaf(int h, int w, float thresh = 0.1)
        : m_net_h(h), m_net_w(w), m_keypoint_thresh(thresh) {}
    pifpaf(const pifpaf& p)
        : m_net_h(p.m_net_h), m_net_w(p.m_net_w), m_keypoint_thresh(p.m_keypoint_thresh) {}
    ~pifpaf() { hp_pifpaf_destroy(m_h); }
    hp_parser_desc stream_desc() const
    {
        hp_parser_desc d{};
        d.kind = HP_PARSER_PIFPAF;
        d.thresh[0] = m_keypoint_thresh;
        d.res_w = d.res_h = -1;
        return d;
    }

    std::vector<human_t> process(const feature_map_t& paf, const feature_map_t& pif)
    {
        if (!m_h && hp_pifpaf_create(&m_h, m_net_h, m_net_w, m_keypoint_thresh, 1) != HP_OK)
            fatal(hp_last_error());
        const int fh = pif.shape()[pif.shape().size() - 2], fw = pif.shape().back(); // src/pifpaf.cpp:23-24
        std::vector<hp_human> out(CAP);
        int n = 0;
        if (hp_pifpaf_process_batch(m_h, 1, paf.view<float>(), pif.view<float>(), fh, fw, 0, out.data(), CAP, &n) != HP_OK)
            fatal(hp_last_error());
        std::vector<human_t> ret(n);
        for (int i = 0; i < n; ++i) {
            ret[i].score = out[i].score;
            for (int k = 0; k < COCO_N_PARTS; ++k)
                ret[i].parts[k] = body_part_t{ out[i].parts[k].has_value != 0, out[i].parts[k].x, out[i].parts[k].y, out[i].parts[k].score };
        }
        return ret;
    }
    template <typename C>
    std::vector<human_t> process(C&& feature_map_containers)
    {
        assert(feature_map_containers.size() == 2);
        return process(feature_map_containers[0], feature_map_containers[1]);
    }

private:
    static constexpr int CAP = 128;
    [[noreturn]] static void fatal(const char* msg)
    {
        std::cerr << "[HyperPose::ERROR  ] " << msg << "\n";
        std::exit(-1);
    }
    int m_net_h, m_net_w;
    float m_keypoint_thresh;
    hp_pifpaf* m_h = nullptr;
};

} // namespace hyperpose::parser
