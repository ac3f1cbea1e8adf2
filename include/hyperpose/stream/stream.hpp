// hyperpose::hip_stream — the GPU form of the reference's stream operator
// (include/hyperpose/stream/stream.hpp:119-145: `make_stream(engine, parser)`, `stream.async() << frames`,
// `stream.sync() >> writer`).  The reference wires four CPU threads and three queues (resize -> inference -> parse ->
// write); here a batch of frames of any size is handed to `hp_pipeline_submit` and stays on the device from the H2D copy
// to the parsed humans (cv::resize / non_scaling_resize, conv stack, PAF parser on one HIP stream per in-flight batch).
// push() = `async() << frames` for one batch, pop() = the oldest batch's pose set in submission order, with
// resume_ratio already applied when keep_ratio is set (src/stream.cpp:120-124).
#pragma once
#include "../../hp_hip.h"
#include "../operator/dnn/hip_engine.hpp"
#include "../utility/cv_min.hpp"
#include "../utility/human.hpp"

#include <stdexcept>
#include <vector>

namespace hyperpose {

class hip_stream {
public:
    using pose_set = std::vector<human_t>; // one frame

    hip_stream(const dnn::builtin_model& model, cv::Size input_size, int max_batch_size = 8, bool keep_ratio = false,
        int n_pipes = 4, cv::Size max_frame = cv::Size(1920, 1080), float conf_thresh = 0.05f, float paf_thresh = 0.05f,
        double factor = 1. / 255, bool flip_rgb = true)
        : m_max_batch(max_batch_size)
    {
        hp_model* m = nullptr;
        check(hp_model_build(&m, model.arch.c_str(), input_size.width, input_size.height));
        std::vector<float> w = model.weights;
        if (w.empty()) {
            w.resize(hp_model_num_weights(m));
            check(hp_model_init_weights(m, model.seed, w.data(), w.size()));
        }
        hp_engine_desc d{};
        const hp_layer* layers = nullptr;
        const hp_output_desc* outs = nullptr;
        int nl = 0, no = 0;
        check(hp_model_layers(m, &layers, &nl));
        check(hp_model_outputs(m, &outs, &no));
        check(hp_model_preproc(m, d.mean, d.inv_std));
        d.in_w = input_size.width, d.in_h = input_size.height, d.max_batch = max_batch_size, d.factor = factor, d.flip_rb = flip_rgb ? 1 : 0;
        d.layers = layers, d.n_layers = nl, d.outputs = outs, d.n_outputs = no, d.weights = w.data(), d.n_weights = w.size();
        const int rc = hp_pipeline_create(&m_pl, &d, n_pipes, keep_ratio ? 1 : 0, conf_thresh, paf_thresh, (size_t)max_frame.area() * 3);
        hp_model_destroy(m);
        check(rc);
        m_out.resize((size_t)max_batch_size * CAP);
        m_n.resize(max_batch_size);
    }
    hip_stream(const hip_stream&) = delete;
    ~hip_stream() { hp_pipeline_destroy(m_pl); }

    size_t in_flight() const { return (size_t)hp_pipeline_in_flight(m_pl); }

    // `stream.async() << frames`: one batch (<= max_batch_size frames, any sizes); throws when every pipe is busy
    void push(const std::vector<cv::Mat>& frames)
    {
        std::vector<const uint8_t*> ptrs;
        std::vector<int> ws, hs;
        for (const auto& f : frames)
            ptrs.push_back(f.data()), ws.push_back(f.cols), hs.push_back(f.rows);
        check(hp_pipeline_submit(m_pl, ptrs.data(), ws.data(), hs.data(), (int)frames.size()));
    }

    // the oldest batch's humans, one pose_set per frame
    std::vector<pose_set> pop()
    {
        int nf = 0;
        check(hp_pipeline_collect(m_pl, reinterpret_cast<hp_human*>(m_out.data()), CAP, m_n.data(), &nf));
        std::vector<pose_set> r(nf);
        for (int i = 0; i < nf; ++i)
            r[i].assign(m_out.begin() + (size_t)i * CAP, m_out.begin() + (size_t)i * CAP + m_n[i]);
        return r;
    }

private:
    static constexpr int CAP = 128;
    static void check(int rc)
    {
        if (rc != HP_OK)
            throw std::runtime_error(hp_last_error());
    }
    hp_pipeline* m_pl = nullptr;
    int m_max_batch;
    std::vector<human_t> m_out;
    std::vector<int> m_n;
};

} // namespace hyperpose
