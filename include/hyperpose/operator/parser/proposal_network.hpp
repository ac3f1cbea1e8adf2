// hyperpose::parser::pose_proposal over libhp_hip.so — same constructor, process() overloads and setters as the
// reference class (include/hyperpose/operator/parser/proposal_network.hpp:17-81, src/pose_proposal.cpp).
#pragma once
#include <cassert>
#include <cstdlib>
#include <iostream>

#include "../../../hp_hip.h"
#include "../../utility/data.hpp"

namespace hyperpose {
namespace parser {

    class pose_proposal {
    public:
        explicit pose_proposal(cv::Size net_resolution, float point_thresh = 0.10, float limb_thresh = 0.05, float mns_thresh = 0.3)
            : m_net_resolution(net_resolution), m_point_thresh(point_thresh), m_limb_thresh(limb_thresh), m_nms_thresh(mns_thresh)
        {
        }
        pose_proposal(const pose_proposal& p)
            : m_net_resolution(p.m_net_resolution), m_point_thresh(p.m_point_thresh), m_limb_thresh(p.m_limb_thresh), m_nms_thresh(p.m_nms_thresh)
        {
        }
        ~pose_proposal() { hp_ppn_destroy(m_h); }

        std::vector<human_t> process(const feature_map_t& conf_point, const feature_map_t& conf_iou, const feature_map_t& x,
            const feature_map_t& y, const feature_map_t& w, const feature_map_t& h, const feature_map_t& edge)
        {
            assert(conf_point.shape().size() == 3); // reference asserts, src/pose_proposal.cpp:76-80
            const int cs[3] = { conf_point.shape()[0], conf_point.shape()[1], conf_point.shape()[2] };
            int es[5] = { 17, 9, 9, cs[1], cs[2] };
            if (edge.shape().size() == 5)
                for (int i = 0; i < 5; ++i)
                    es[i] = edge.shape()[i];
            else if (edge.shape().size() == 3) // [L*9*9, h, w], the layout the engine emits
                es[0] = edge.shape()[0] / 81;
            // maps that still lie in their engine's device buffers: the whole batch in one launch when its first frame is asked for
            // (operator/parser/paf.hpp has the reasoning; utility/data.hpp the record)
            const feature_map_t* maps[7] = { &conf_point, &conf_iou, &x, &y, &w, &h, &edge };
            const detail::device_batch* b = conf_point.device_batch();
            bool on_dev = b && !std::getenv("HP_MIRROR_HOST_MAPS");
            for (int k = 1; k < 7 && on_dev; ++k)
                on_dev = maps[k]->device_batch() == b && maps[k]->batch_frame() == conf_point.batch_frame();
            if (on_dev) {
                if (m_cached.lock().get() != b || m_cached_gen != b->gen) {
                    ensure(b->n);
                    const float* t[7];
                    for (int k = 0; k < 7; ++k)
                        t[k] = b->outs[maps[k]->batch_output()].dev;
                    std::vector<hp_human> out((size_t)b->n * CAP);
                    std::vector<int> cnt(b->n);
                    if (hp_ppn_process_batch(m_h, b->n, t, cs, es, 1, out.data(), CAP, cnt.data()) != HP_OK)
                        fatal(hp_last_error());
                    m_cache.assign(b->n, {});
                    for (int f = 0; f < b->n; ++f)
                        m_cache[f] = convert(out.data() + (size_t)f * CAP, cnt[f]);
                    m_cached = conf_point.batch_handle(), m_cached_gen = b->gen;
                }
                return m_cache[conf_point.batch_frame()];
            }
            ensure(1);
            const float* t[7] = { conf_point.view<float>(), conf_iou.view<float>(), x.view<float>(), y.view<float>(), w.view<float>(),
                h.view<float>(), edge.view<float>() };
            std::vector<hp_human> out(CAP);
            int n = 0;
            if (hp_ppn_process_batch(m_h, 1, t, cs, es, 0, out.data(), CAP, &n) != HP_OK)
                fatal(hp_last_error());
            return convert(out.data(), n);
        }
        inline std::vector<human_t> process(const std::vector<feature_map_t>& l)
        {
            assert(l.size() == 7);
            return this->process(l.at(0), l.at(1), l.at(2), l.at(3), l.at(4), l.at(5), l.at(6));
        }
        hp_parser_desc stream_desc() const
        {
            hp_parser_desc d{};
            d.kind = HP_PARSER_PPN;
            d.thresh[0] = m_point_thresh, d.thresh[1] = m_limb_thresh, d.thresh[2] = m_nms_thresh;
            d.res_w = d.res_h = -1;
            return d;
        }
        void set_point_thresh(float thresh) { m_point_thresh = thresh, push(); }
        void set_limb_thresh(float thresh) { m_limb_thresh = thresh, push(); }
        void set_nms_thresh(float thresh) { m_nms_thresh = thresh, push(); }

    private:
        static constexpr int CAP = 128;
        [[noreturn]] static void fatal(const char* msg)
        {
            std::cerr << "[HyperPose::ERROR  ] " << msg << "\n";
            std::exit(-1);
        }
        void push()
        {
            m_cached.reset();
            if (m_h)
                hp_ppn_set_thresholds(m_h, m_point_thresh, m_limb_thresh, m_nms_thresh);
        }
        static std::vector<human_t> convert(const hp_human* out, int n)
        {
            std::vector<human_t> ret(n);
            for (int i = 0; i < n; ++i) {
                ret[i].score = out[i].score;
                for (int k = 0; k < COCO_N_PARTS; ++k)
                    ret[i].parts[k] = body_part_t{ out[i].parts[k].has_value != 0, out[i].parts[k].x, out[i].parts[k].y, out[i].parts[k].score };
            }
            return ret;
        }
        void ensure(int batch) // a parser handle that takes `batch` frames per call (rebuilt when a larger batch arrives)
        {
            if (m_h && batch <= m_handle_batch)
                return;
            hp_ppn_destroy(m_h);
            m_h = nullptr;
            if (hp_ppn_create(&m_h, m_net_resolution.width, m_net_resolution.height, m_point_thresh, m_limb_thresh, m_nms_thresh, batch) != HP_OK)
                fatal(hp_last_error());
            m_handle_batch = batch;
        }
        cv::Size m_net_resolution;
        float m_point_thresh, m_limb_thresh, m_nms_thresh;
        hp_ppn* m_h = nullptr;
        int m_handle_batch = 0;
        std::weak_ptr<detail::device_batch> m_cached;
        uint64_t m_cached_gen = 0;
        std::vector<std::vector<human_t>> m_cache;
    };

} // namespace parser
} // namespace hyperpose
