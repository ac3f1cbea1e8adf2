// hyperpose::parser::pifpaf over libhp_hip.so — reference include/hyperpose/operator/parser/pifpaf.hpp:8-26.
// process(arg0, arg1): arg0 = PAF/CAF [19,9,h,w], arg1 = PIF/CIF [17,5,h,w] (the reference's .cpp parameter order,
// src/pifpaf.cpp:7; its header names them the other way round).  The engine also emits them as [171,h,w] / [85,h,w].
#pragma once
#include <cassert>
#include <cstdlib>
#include <iostream>
#include <memory>
#include <vector>

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
        const int fh = pif.shape()[pif.shape().size() - 2], fw = pif.shape().back(); // src/pifpaf.cpp:23-24
        // maps that still lie in their engine's device buffers: the whole batch is decoded in one launch when its first frame is asked for
        // (operator/parser/paf.hpp has the reasoning; utility/data.hpp the record)
        const detail::device_batch* b = paf.device_batch();
        if (b && b == pif.device_batch() && paf.batch_frame() == pif.batch_frame() && !std::getenv("HP_MIRROR_HOST_MAPS")) {
            if (m_cached.lock().get() != b || m_cached_gen != b->gen) {
                ensure(b->n);
                std::vector<hp_human> out((size_t)b->n * CAP);
                std::vector<int> cnt(b->n);
                if (hp_pifpaf_process_batch(m_h, b->n, b->outs[paf.batch_output()].dev, b->outs[pif.batch_output()].dev, fh, fw, 1, out.data(), CAP, cnt.data()) != HP_OK)
                    fatal(hp_last_error());
                m_cache.assign(b->n, {});
                for (int f = 0; f < b->n; ++f)
                    m_cache[f] = convert(out.data() + (size_t)f * CAP, cnt[f]);
                m_cached = paf.batch_handle(), m_cached_gen = b->gen;
            }
            return m_cache[paf.batch_frame()];
        }
        ensure(1);
        std::vector<hp_human> out(CAP);
        int n = 0;
        if (hp_pifpaf_process_batch(m_h, 1, paf.view<float>(), pif.view<float>(), fh, fw, 0, out.data(), CAP, &n) != HP_OK)
            fatal(hp_last_error());
        return convert(out.data(), n);
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
    void ensure(int batch) // a decoder handle that takes `batch` frames per call (rebuilt when a larger batch arrives)
    {
        if (m_h && batch <= m_handle_batch)
            return;
        hp_pifpaf_destroy(m_h);
        m_h = nullptr;
        if (hp_pifpaf_create(&m_h, m_net_h, m_net_w, m_keypoint_thresh, batch) != HP_OK)
            fatal(hp_last_error());
        m_handle_batch = batch;
    }
    int m_net_h, m_net_w;
    float m_keypoint_thresh;
    hp_pifpaf* m_h = nullptr;
    int m_handle_batch = 0;
    std::weak_ptr<detail::device_batch> m_cached;
    uint64_t m_cached_gen = 0;
    std::vector<std::vector<human_t>> m_cache;
};

} // namespace hyperpose::parser
