// hyperpose::dnn::hip_engine over libhp_hip.so — the Engine concept the stream API and user code expect
// (include/hyperpose/stream/stream.hpp:136,139,265,338-339): input_size(), max_batch_size(),
// inference(std::vector<cv::Mat>) -> std::vector<internal_t>, with the constructor argument order of
// dnn::tensorrt (include/hyperpose/operator/dnn/tensorrt.hpp:44-74): (model, cv::Size input_size,
// int max_batch_size = 8, bool keep_ratio = false, double factor = 1./255, bool flip_rgb = true).
// `using tensorrt = hip_engine;` under HYPERPOSE_TENSORRT_COMPAT keeps existing call sites compiling.
// Model descriptors: dnn::onnx{path} as in the reference (utility/model.hpp:23-25), a serialized engine file, or a built-in
// topology + a flat fp32 weight blob (UFF files are TensorRT-only and not read).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <string>

#include "../../../hp_hip.h"
#include "../../utility/data.hpp"

namespace hyperpose {
namespace dnn {

    struct serialized_model { // hyperpose::dnn::tensorrt_serialized (utility/model.hpp:30-32)
        std::string model_path;
    };
    struct onnx { // hyperpose::dnn::onnx (utility/model.hpp:23-25)
        std::string model_path;
    };
    struct builtin_model {
        std::string arch;            // see hp_model_archs()
        std::vector<float> weights;  // empty: deterministic synthetic weights (seed below)
        uint64_t seed = 20241;
    };

    class hip_engine {
    public:
        explicit hip_engine(const builtin_model& model, cv::Size input_size, int max_batch_size = 8, bool keep_ratio = false,
            double factor = 1. / 255, bool flip_rgb = true)
            : m_inp_size(input_size), m_max_batch_size(max_batch_size), m_keep_ratio(keep_ratio)
        {
            if (hp_model_build(&m_model, model.arch.c_str(), input_size.width, input_size.height) != HP_OK)
                fatal(hp_last_error());
            std::vector<float> w = model.weights;
            if (w.empty()) {
                w.resize(hp_model_num_weights(m_model));
                hp_model_init_weights(m_model, model.seed, w.data(), w.size());
            }
            if (hp_engine_create_from_model(&m_engine, m_model, max_batch_size, factor, flip_rgb ? 1 : 0, w.data(), w.size()) != HP_OK)
                fatal(hp_last_error());
        }
        // tensorrt(const onnx&, cv::Size input_size, int max_batch_size = 8, bool keep_ratio = false, data_type, double factor = 1./255,
        // bool flip_rgb = true) (include/hyperpose/operator/dnn/tensorrt.hpp:53-62; the data_type argument has no meaning here:
        // activations are fp16 with fp32 accumulation)
        explicit hip_engine(const onnx& onnx_model, cv::Size input_size, int max_batch_size = 8, bool keep_ratio = false,
            double factor = 1. / 255, bool flip_rgb = true)
            : m_inp_size(input_size), m_max_batch_size(max_batch_size), m_keep_ratio(keep_ratio)
        {
            if (hp_model_from_onnx_file(&m_model, onnx_model.model_path.c_str(), input_size.width, input_size.height) != HP_OK)
                fatal(hp_last_error());
            if (hp_engine_create_from_model(&m_engine, m_model, max_batch_size, factor, flip_rgb ? 1 : 0, nullptr, 0) != HP_OK)
                fatal(hp_last_error());
        }
        // tensorrt(const tensorrt_serialized&, ...) (include/hyperpose/operator/dnn/tensorrt.hpp:72-74, utility/model.hpp:27-32)
        explicit hip_engine(const serialized_model& model, cv::Size input_size, int max_batch_size = 8, bool keep_ratio = false)
            : m_inp_size(input_size), m_max_batch_size(max_batch_size), m_keep_ratio(keep_ratio)
        {
            if (hp_engine_load(&m_engine, model.model_path.c_str(), max_batch_size) != HP_OK)
                fatal(hp_last_error());
            int w = 0, h = 0;
            hp_engine_input_size(m_engine, &w, &h);
            if (w != input_size.width || h != input_size.height)
                fatal("serialized engine was built for another input size");
        }
        // tensorrt::save (tensorrt.hpp:121-123)
        void save(const std::string path)
        {
            if (hp_engine_save(m_engine, path.c_str()) != HP_OK)
                fatal(hp_last_error());
        }
        hip_engine(const hip_engine&) = delete;
        ~hip_engine()
        {
            hp_engine_destroy(m_engine);
            hp_model_destroy(m_model);
        }

        inline int max_batch_size() noexcept { return m_max_batch_size; }
        inline cv::Size input_size() noexcept { return m_inp_size; }

        // tensorrt::inference(std::vector<cv::Mat>) (src/tensorrt.cpp:436-461).  Frames must already be
        // network-sized u8 BGR (cv::resize to the same size is a copy); keep_ratio letter-boxing is a later row.
        std::vector<internal_t> inference(std::vector<cv::Mat> inputs)
        {
            if (inputs.size() > (size_t)m_max_batch_size) // src/tensorrt.cpp:439-443
                throw std::logic_error("Input batch size overflow: Yours@" + std::to_string(inputs.size()) + " Max@" + std::to_string(m_max_batch_size));
            const size_t frame = (size_t)m_inp_size.width * m_inp_size.height * 3;
            std::vector<uint8_t> batch(frame * inputs.size());
            for (size_t i = 0; i < inputs.size(); ++i) {
                if (inputs[i].size() != m_inp_size)
                    fatal("hip_engine::inference: frames must be network-sized\n");
                std::memcpy(batch.data() + i * frame, inputs[i].data(), frame);
            }
            if (hp_engine_infer_u8(m_engine, batch.data(), (int)inputs.size(), 0, nullptr) != HP_OK)
                fatal(hp_last_error());
            return collect(inputs.size());
        }
        // tensorrt::inference(const std::vector<float>&, size_t) (src/tensorrt.cpp:364-434)
        std::vector<internal_t> inference(const std::vector<float>& nchw, size_t batch_size)
        {
            if (batch_size > (size_t)m_max_batch_size)
                throw std::logic_error("Input batch size overflow");
            if (hp_engine_infer_f32(m_engine, nchw.data(), (int)batch_size, 0, nullptr) != HP_OK)
                fatal(hp_last_error());
            return collect(batch_size);
        }
        // MI355X addition: frames already in HBM, outputs stay in HBM (feed parser::paf::process_device).
        void inference_device(const uint8_t* dev_hwc_bgr, int n, void* stream = nullptr)
        {
            if (hp_engine_infer_u8(m_engine, dev_hwc_bgr, n, 1, stream) != HP_OK)
                fatal(hp_last_error());
        }
        hp_engine* handle() { return m_engine; }

    private:
        [[noreturn]] static void fatal(const char* msg)
        {
            std::cerr << "[HyperPose::ERROR  ] " << msg << "\n";
            std::exit(-1);
        }
        std::vector<internal_t> collect(size_t n)
        {
            std::vector<internal_t> ret(n);
            const int no = hp_engine_num_outputs(m_engine);
            for (int i = 0; i < no; ++i) { // already sorted by tensor name (src/tensorrt.cpp:405)
                const char* name = nullptr;
                int shape[3];
                hp_engine_output(m_engine, i, &name, shape, nullptr);
                const size_t per = (size_t)shape[0] * shape[1] * shape[2];
                std::vector<float> host(per * n);
                if (hp_engine_output_to_host(m_engine, i, (int)n, host.data()) != HP_OK)
                    fatal(hp_last_error());
                for (size_t j = 0; j < n; ++j) {
                    std::unique_ptr<char[]> data{ new char[per * sizeof(float)] };
                    std::memcpy(data.get(), host.data() + j * per, per * sizeof(float));
                    ret[j].emplace_back(name, std::move(data), std::vector<int>{ shape[0], shape[1], shape[2] });
                }
            }
            return ret;
        }
        const cv::Size m_inp_size;
        const int m_max_batch_size;
        const bool m_keep_ratio;
        hp_model* m_model = nullptr;
        hp_engine* m_engine = nullptr;
    };

#ifdef HYPERPOSE_TENSORRT_COMPAT
    using tensorrt = hip_engine;
#endif

} // namespace dnn
} // namespace hyperpose
