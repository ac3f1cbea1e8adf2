// hyperpose::parser::paf over libhp_hip.so — same constructor, process() overloads, setters and copy semantics
// as the reference class (include/hyperpose/operator/parser/paf.hpp:17-93, src/paf.cpp:284-387).
// Header-only; all arithmetic happens in the HIP kernels behind hp_paf_* (include/hp_hip.h).
#pragma once
#include <cstdlib>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <vector>

#include "../../../hp_hip.h"
#include "../../utility/data.hpp"

namespace hyperpose {
namespace parser {

    class paf {
    public:
        explicit paf(float conf_thresh = 0.05, float paf_thresh = 0.05, cv::Size resolution_size = cv::Size(-1, -1), int max_batch = 8)
            : m_conf_thresh(conf_thresh), m_paf_thresh(paf_thresh), m_resolution_size(resolution_size), m_max_batch(max_batch)
        {
        }
        // "This copy constructor will only copy the parameters" (paf.hpp:72-74): device scratch is per object.
        paf(const paf& p)
            : m_conf_thresh(p.m_conf_thresh), m_paf_thresh(p.m_paf_thresh), m_resolution_size(p.m_resolution_size), m_max_batch(p.m_max_batch)
        {
        }
        ~paf() { hp_paf_destroy(m_h); }

        std::vector<human_t> process(const feature_map_t& conf, const feature_map_t& paf_map)
        {
            // reference: error() prints and std::exit(-1) (src/paf.cpp:305-306, src/logging.hpp:31-37)
            if (conf.shape().size() != 3 || paf_map.shape().size() != 3)
                fatal("Input of PAF::PROCESS didn't meet requirements: [conf, paf], tensor.dims() == 3\n");
            const int cs[3] = { conf.shape()[0], conf.shape()[1], conf.shape()[2] };
            const int ps[3] = { paf_map.shape()[0], paf_map.shape()[1], paf_map.shape()[2] };
            // Maps that still lie in their engine's device buffers (utility/data.hpp, detail::device_batch) are read from there - no D2H + H2D
            // round trip - and, because the reference's callers ask for the frames of a batch one after the other
            // (examples/operator_api_batched_images_paf.example.cpp:70-73), the WHOLE batch is parsed in one launch the first time one of its
            // frames is asked for; the other frames' calls return what that launch found.  Per frame the result is process()'s own: the kernels
            // are the same and frames are independent (tests/cpp/operator_api_paf.cpp compares the two paths).
            const detail::device_batch* bc = conf.device_batch();
            if (bc && bc == paf_map.device_batch() && conf.batch_frame() == paf_map.batch_frame() && bc->n <= m_max_batch && !std::getenv("HP_MIRROR_HOST_MAPS")) {
                if (m_cached.lock().get() != bc || m_cached_gen != bc->gen) {
                    m_cache = run(bc->n, bc->outs[conf.batch_output()].dev, cs, bc->outs[paf_map.batch_output()].dev, ps, 1);
                    m_cached = conf.batch_handle(), m_cached_gen = bc->gen;
                }
                return m_cache[conf.batch_frame()];
            }
            std::vector<std::vector<human_t>> r = run(1, conf.view<float>(), cs, paf_map.view<float>(), ps, 0);
            return std::move(r[0]);
        }
        template <typename C>
        std::vector<human_t> process(C&& feature_map_containers) { return process(feature_map_containers[0], feature_map_containers[1]); }

        // MI355X addition: n frames whose conf/paf maps are already in HBM (e.g. dnn::tensorrt outputs).
        std::vector<std::vector<human_t>> process_device(int n, const float* dev_conf, const int conf_shape[3], const float* dev_paf, const int paf_shape[3])
        {
            return run(n, dev_conf, conf_shape, dev_paf, paf_shape, 1);
        }

        // the constructor arguments in the form the GPU stream operator takes (hp_pipeline_create_ex)
        hp_parser_desc stream_desc() const
        {
            hp_parser_desc d{};
            d.kind = HP_PARSER_PAF;
            d.thresh[0] = m_conf_thresh, d.thresh[1] = m_paf_thresh;
            d.res_w = m_resolution_size.width, d.res_h = m_resolution_size.height;
            return d;
        }

        void set_paf_thresh(float thresh)
        {
            m_cached.reset();
            m_paf_thresh = thresh;
            if (m_h)
                hp_paf_set_paf_thresh(m_h, thresh);
        }
        void set_conf_thresh(float thresh)
        {
            m_cached.reset();
            m_conf_thresh = thresh;
            if (m_h)
                hp_paf_set_conf_thresh(m_h, thresh);
        }

    private:
        static constexpr int CAP = 128;
        [[noreturn]] static void fatal(const char* msg)
        {
            std::cerr << "[HyperPose::ERROR  ] " << msg;
            std::exit(-1);
        }
        std::vector<std::vector<human_t>> run(int n, const float* conf, const int cs[3], const float* pafm, const int ps[3], int on_device)
        {
            if (!m_h && hp_paf_create(&m_h, m_conf_thresh, m_paf_thresh, m_resolution_size.width, m_resolution_size.height, m_max_batch) != HP_OK)
                fatal(hp_last_error());
            std::vector<hp_human> out((size_t)n * CAP);
            std::vector<int> cnt(n);
            if (hp_paf_process_batch(m_h, n, conf, cs, pafm, ps, on_device, out.data(), CAP, cnt.data()) != HP_OK)
                fatal(hp_last_error());
            std::vector<std::vector<human_t>> res(n);
            for (int f = 0; f < n; ++f)
                for (int i = 0; i < cnt[f]; ++i) {
                    const hp_human& h = out[(size_t)f * CAP + i];
                    human_t hu;
                    hu.score = h.score;
                    for (int k = 0; k < COCO_N_PARTS; ++k)
                        hu.parts[k] = body_part_t{ h.parts[k].has_value != 0, h.parts[k].x, h.parts[k].y, h.parts[k].score };
                    res[f].push_back(hu);
                }
            return res;
        }
        float m_conf_thresh, m_paf_thresh;
        cv::Size m_resolution_size;
        int m_max_batch;
        hp_paf* m_h = nullptr;
        std::weak_ptr<detail::device_batch> m_cached; // the batch m_cache holds the humans of
        uint64_t m_cached_gen = 0;
        std::vector<std::vector<human_t>> m_cache;
    };

} // namespace parser
} // namespace hyperpose
