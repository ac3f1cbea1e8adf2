// hyperpose::dnn::tensorrt over libhp_hip.so — the class name, constructor signatures, member functions and error behaviour of the
// reference engine (include/hyperpose/operator/dnn/tensorrt.hpp:14-19, 33-141; src/tensorrt.cpp), so that reference call sites such as
// examples/operator_api_batched_images_paf.example.cpp:36-56 compile and run unchanged:
//     tensorrt(onnx{file}, {w, h}, batch)          tensorrt(uff{...}, {w, h}, batch)          tensorrt(tensorrt_serialized{file}, {w, h}, batch)
//     engine.inference(std::vector<cv::Mat>)       engine.inference(std::vector<float> nchw, n)       engine.save(path)
// Header-only on top of the C ABI (include/hp_hip.h); the network runs as hand-written gfx950 kernels.  `data_type` selects the
// arithmetic as it does in the reference (src/tensorrt.cpp:327,353): kFLOAT - the default, as there - is fp32 storage and fp32
// matrix-pipe arithmetic (HP_DTYPE_F32), kHALF the fused fp16 kernels with fp32 accumulation (HP_DTYPE_F16, the fast path); the integer
// types have no meaning for these networks and are refused like an engine-build failure.  Frames of any size are resized on
// the DEVICE exactly as the reference does on the host: cv::resize (INTER_LINEAR) or, with keep_ratio, non_scaling_resize
// (src/tensorrt.cpp:446-451, src/data.cpp:53-69) through hp_resize_u8c3 / hp_letterbox_u8c3.
#pragma once
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "../../../hp_hip.h"
#include "../../utility/data.hpp"
#include "../../utility/model.hpp"

namespace hyperpose {

/// Data type related to TensorRT data type (include/hyperpose/operator/dnn/tensorrt.hpp:14-30).
struct data_type {
    static constexpr int kFLOAT = 0;
    static constexpr int kHALF = 1;
    static constexpr int kINT8 = 2;
    static constexpr int kINT32 = 3;
    static constexpr int kBOOL = 4;
    int val = kFLOAT;
    inline data_type(int v)
        : val(v)
    {
    }
    /// HP_DTYPE_* of the C ABI, or -1 for the types no engine is built in
    inline int hp_dtype() const { return val == kFLOAT ? HP_DTYPE_F32 : val == kHALF ? HP_DTYPE_F16 : -1; }
};

namespace detail {
    // 8-bit BGR pixels of a cv::Mat for both the bundled cv_min.hpp and real OpenCV (cv::Mat::data is a member there and the matrix
    // may be non-continuous or of another type)
    inline const uint8_t* mat_bytes(const cv::Mat& m, std::vector<uint8_t>& scratch)
    {
#ifdef HYPERPOSE_USE_OPENCV
        if (m.type() != CV_8UC3)
            throw std::logic_error("hyperpose: frames must be 8-bit 3-channel (CV_8UC3)");
        if (m.isContinuous())
            return m.data;
        scratch.resize((size_t)m.rows * m.cols * 3);
        for (int r = 0; r < m.rows; ++r)
            std::memcpy(scratch.data() + (size_t)r * m.cols * 3, m.ptr(r), (size_t)m.cols * 3);
        return scratch.data();
#else
        (void)scratch;
        return m.data();
#endif
    }
} // namespace detail

namespace dnn {

    class tensorrt {
    public:
        /// UFF is TensorRT's own graph format and cannot be read here: like every unrecoverable engine error of the reference
        /// (src/tensorrt.cpp:141-158) this logs and exits.
        explicit tensorrt(const uff& uff_model, cv::Size input_size, int max_batch_size = 8, bool keep_ratio = false,
            data_type dtype = data_type::kFLOAT, double factor = 1. / 255, bool flip_rgb = true)
            : m_inp_size(input_size), m_max_batch_size(max_batch_size), m_keep_ratio(keep_ratio), m_factor(factor), m_flip_rgb(flip_rgb)
        {
            (void)dtype;
            fatal(("UFF models (" + uff_model.model_path + ") are a TensorRT format; export the model to ONNX (dnn::onnx) instead").c_str());
        }

        explicit tensorrt(const onnx& onnx_model, cv::Size input_size, int max_batch_size = 8, bool keep_ratio = false,
            data_type dtype = data_type::kFLOAT, double factor = 1. / 255, bool flip_rgb = true)
            : m_inp_size(input_size), m_max_batch_size(max_batch_size), m_keep_ratio(keep_ratio), m_factor(factor), m_flip_rgb(flip_rgb)
        {
            if (dtype.hp_dtype() < 0)
                fatal("hyperpose::dnn::tensorrt: only data_type::kFLOAT and data_type::kHALF engines can be built");
            if (hp_model_from_onnx_file(&m_model, onnx_model.model_path.c_str(), input_size.width, input_size.height) != HP_OK)
                fatal(hp_last_error());
            if (hp_engine_create_from_model_dtype(&m_engine, m_model, max_batch_size, factor, flip_rgb ? 1 : 0, nullptr, 0, dtype.hp_dtype()) != HP_OK)
                fatal(hp_last_error());
            after_create();
        }

        explicit tensorrt(const tensorrt_serialized& serialized_model, cv::Size input_size, int max_batch_size = 8, bool keep_ratio = false,
            double factor = 1. / 255, bool flip_rgb = true)
            : m_inp_size(input_size), m_max_batch_size(max_batch_size), m_keep_ratio(keep_ratio), m_factor(factor), m_flip_rgb(flip_rgb)
        {
            // (factor / flip_rgb are stored in the file by `save`, as TensorRT bakes them into the plan it serializes)
            if (hp_engine_load(&m_engine, serialized_model.model_path.c_str(), max_batch_size) != HP_OK)
                fatal(hp_last_error());
            int w = 0, h = 0;
            hp_engine_input_size(m_engine, &w, &h);
            if (w != input_size.width || h != input_size.height)
                fatal("serialized engine was built for another input size");
            after_create();
        }

        /// Addition: a built-in topology (no model file needed).
        explicit tensorrt(const builtin_model& model, cv::Size input_size, int max_batch_size = 8, bool keep_ratio = false,
            data_type dtype = data_type::kFLOAT, double factor = 1. / 255, bool flip_rgb = true)
            : m_inp_size(input_size), m_max_batch_size(max_batch_size), m_keep_ratio(keep_ratio), m_factor(factor), m_flip_rgb(flip_rgb)
        {
            if (dtype.hp_dtype() < 0)
                fatal("hyperpose::dnn::tensorrt: only data_type::kFLOAT and data_type::kHALF engines can be built");
            if (hp_model_build(&m_model, model.arch.c_str(), input_size.width, input_size.height) != HP_OK)
                fatal(hp_last_error());
            std::vector<float> w = model.weights;
            if (w.empty()) {
                w.resize(hp_model_num_weights(m_model));
                hp_model_init_weights(m_model, model.seed, w.data(), w.size());
            }
            if (hp_engine_create_from_model_dtype(&m_engine, m_model, max_batch_size, factor, flip_rgb ? 1 : 0, w.data(), w.size(), dtype.hp_dtype()) != HP_OK)
                fatal(hp_last_error());
            after_create();
        }

        /// The builtin_model constructor (an addition of this library) took (model, size, batch, keep_ratio, factor, flip_rgb) before it learned
        /// `dtype`; data_type(int) is implicit - as in the reference - so that old positional call would still compile, with factor truncated
        /// into a data_type.  A floating-point argument in the dtype position is a compile error instead.
        template <class F, class = typename std::enable_if<std::is_floating_point<F>::value>::type>
        tensorrt(const builtin_model&, cv::Size, int, bool, F, bool = true) = delete;

        tensorrt(const tensorrt&) = delete;
        tensorrt& operator=(const tensorrt&) = delete;
        ~tensorrt()
        {
            retire_last_batch(); // maps of the last call that are still alive get their host copy before the buffers go
            if (m_host_net)
                hp_free_host(m_host_net);
            if (m_dev_raw)
                hp_free(m_dev_raw);
            if (m_dev_net)
                hp_free(m_dev_net);
            hp_engine_destroy(m_engine);
            hp_model_destroy(m_model);
        }

        inline int max_batch_size() noexcept { return m_max_batch_size; }
        inline cv::Size input_size() noexcept { return m_inp_size; }

        /// src/tensorrt.cpp:436-461: every image is brought to the network's size (cv::resize, or non_scaling_resize when keep_ratio)
        /// and the batch is inferred; throws std::logic_error on an over-size batch (:439-443).
        std::vector<internal_t> inference(std::vector<cv::Mat> inputs)
        {
            if (inputs.size() > (size_t)m_max_batch_size)
                throw std::logic_error("Input batch size overflow: Yours@" + std::to_string(inputs.size()) + " Max@" + std::to_string(m_max_batch_size));
            if (inputs.empty())
                return {};
            const size_t net_frame = (size_t)m_inp_size.width * m_inp_size.height * 3;
            if (!m_dev_net && hp_malloc((void**)&m_dev_net, net_frame * m_max_batch_size) != HP_OK)
                fatal(hp_last_error());
            if (!m_host_net && hp_malloc_host((void**)&m_host_net, net_frame * m_max_batch_size) != HP_OK)
                fatal(hp_last_error());
            retire_last_batch(); // (the engine is idle here: the previous call was synchronised before it returned)
            std::vector<uint8_t> scratch;
            bool all_net_sized = true;
            for (const cv::Mat& f : inputs) {
                if (f.empty())
                    fatal("hyperpose::dnn::tensorrt::inference: empty image");
                all_net_sized = all_net_sized && f.cols == m_inp_size.width && f.rows == m_inp_size.height;
            }
            if (all_net_sized) {
                // resize to the same size is a copy (src/tensorrt.cpp:446-451): the frames are gathered in ONE pinned buffer and go up in one
                // asynchronous copy per half-batch in front of the network's launches (the reference converts every frame to an f32 NCHW
                // staging vector on the host and copies 4 x the bytes, src/tensorrt.cpp:380-383)
                for (size_t i = 0; i < inputs.size(); ++i)
                    std::memcpy(m_host_net + i * net_frame, detail::mat_bytes(inputs[i], scratch), net_frame);
                if (hp_engine_infer_u8(m_engine, m_host_net, (int)inputs.size(), 0, nullptr) != HP_OK)
                    fatal(hp_last_error());
                return collect(inputs.size());
            }
            for (size_t i = 0; i < inputs.size(); ++i) {
                const cv::Mat& f = inputs[i];
                const uint8_t* src = detail::mat_bytes(f, scratch);
                const size_t bytes = (size_t)f.cols * f.rows * 3;
                uint8_t* dst = m_dev_net + i * net_frame;
                if (f.cols == m_inp_size.width && f.rows == m_inp_size.height) { // resize to the same size is a copy
                    if (hp_memcpy_h2d(dst, src, bytes) != HP_OK)
                        fatal(hp_last_error());
                    continue;
                }
                if (bytes > m_raw_bytes) {
                    if (m_dev_raw)
                        hp_free(m_dev_raw);
                    m_dev_raw = nullptr, m_raw_bytes = 0;
                    if (hp_malloc((void**)&m_dev_raw, bytes) != HP_OK)
                        fatal(hp_last_error());
                    m_raw_bytes = bytes;
                }
                if (hp_memcpy_h2d(m_dev_raw, src, bytes) != HP_OK)
                    fatal(hp_last_error());
                const int rc = m_keep_ratio
                    ? hp_letterbox_u8c3(m_dev_raw, f.cols, f.rows, f.cols * 3, dst, m_inp_size.width, m_inp_size.height, m_inp_size.width * 3, 0, 0, 0,
                          hp_engine_stream(m_engine))
                    : hp_resize_u8c3(m_dev_raw, f.cols, f.rows, f.cols * 3, dst, m_inp_size.width, m_inp_size.height, m_inp_size.width * 3,
                          hp_engine_stream(m_engine));
                if (rc != HP_OK || hp_engine_synchronize(m_engine) != HP_OK) // m_dev_raw is re-used by the next frame
                    fatal(hp_last_error());
            }
            if (hp_engine_infer_u8(m_engine, m_dev_net, (int)inputs.size(), 1, nullptr) != HP_OK)
                fatal(hp_last_error());
            return collect(inputs.size());
        }

        /// src/tensorrt.cpp:364-434: plain NCHW float buffers, no scaling / channel swap.
        std::vector<internal_t> inference(const std::vector<float>& float_buffer, size_t batch_size)
        {
            if (batch_size > (size_t)m_max_batch_size)
                throw std::logic_error("Input batch size overflow: Yours@" + std::to_string(batch_size) + " Max@" + std::to_string(m_max_batch_size));
            if (float_buffer.size() < batch_size * 3 * (size_t)m_inp_size.area())
                throw std::logic_error("Input float buffer is smaller than batch_size x 3 x H x W");
            retire_last_batch();
            if (hp_engine_infer_f32(m_engine, float_buffer.data(), (int)batch_size, 0, nullptr) != HP_OK)
                fatal(hp_last_error());
            return collect(batch_size);
        }

        /// tensorrt::save (tensorrt.hpp:121-123, src/tensorrt.cpp:463-471)
        void save(const std::string path)
        {
            if (hp_engine_save(m_engine, path.c_str()) != HP_OK)
                fatal(hp_last_error());
        }

        // ---- additions for device-resident use (the stream operator and the parsers' process_device forms)
        void inference_device(const uint8_t* dev_hwc_bgr, int n, void* stream = nullptr)
        {
            retire_last_batch();
            if (hp_engine_infer_u8(m_engine, dev_hwc_bgr, n, 1, stream) != HP_OK)
                fatal(hp_last_error());
        }
        hp_engine* handle() { return m_engine; }
        bool keep_ratio() const { return m_keep_ratio; }

    private:
        [[noreturn]] static void fatal(const char* msg)
        {
            std::cerr << "[HyperPose::ERROR  ] " << msg << "\n";
            std::exit(-1);
        }
        // one batch in flight per call, like the reference's synchronous inference: kFLOAT engines run it as two half-batches side by side
        // (hp_engine_set_concurrency; bit-identical outputs, measured 2490 -> 1962 us per batch of 8 LW-OpenPose frames)
        void after_create()
        {
            m_calls = std::make_shared<std::atomic<uint64_t>>(0);
            if (!std::getenv("HP_MIRROR_ONE_STREAM"))
                hp_engine_set_concurrency(m_engine, 2);
        }
        // the maps of the previous call lose their device buffers now: those still alive copy themselves to the host first
        void retire_last_batch()
        {
            if (auto last = m_last.lock())
                last->materialize();
            m_last.reset();
            if (m_calls)
                ++*m_calls;
        }
        /// src/tensorrt.cpp:400-433: one internal_t per image, maps sorted by tensor name (:405).  The maps are views of the engine's device
        /// buffers with a host copy made on demand (utility/data.hpp, detail::device_batch).
        std::vector<internal_t> collect(size_t n)
        {
            if (hp_engine_synchronize(m_engine) != HP_OK)
                fatal(hp_last_error());
            auto rec = std::make_shared<detail::device_batch>();
            rec->engine = m_engine, rec->live = m_calls, rec->gen = m_calls->load(), rec->n = (int)n;
            const int no = hp_engine_num_outputs(m_engine);
            for (int i = 0; i < no; ++i) { // already sorted by tensor name (src/tensorrt.cpp:405)
                const char* name = nullptr;
                int shape[3];
                const float* dev = nullptr;
                hp_engine_output(m_engine, i, &name, shape, &dev);
                detail::device_batch::out o;
                o.name = name, o.shape = { shape[0], shape[1], shape[2] }, o.dev = dev, o.per = (size_t)shape[0] * shape[1] * shape[2];
                rec->outs.push_back(std::move(o));
            }
            std::vector<internal_t> ret(n);
            for (size_t j = 0; j < n; ++j)
                for (int i = 0; i < no; ++i)
                    ret[j].emplace_back(rec, i, (int)j);
            m_last = rec;
            if (std::getenv("HP_MIRROR_EAGER_HOST_COPY")) // (tests: the reference's behaviour, every map on the host before the call returns)
                rec->materialize();
            return ret;
        }
        const cv::Size m_inp_size; // w, h
        const int m_max_batch_size;
        const bool m_keep_ratio;
        const double m_factor;
        const bool m_flip_rgb;
        hp_model* m_model = nullptr;
        hp_engine* m_engine = nullptr;
        uint8_t* m_dev_raw = nullptr; // one camera-sized frame
        size_t m_raw_bytes = 0;
        uint8_t* m_dev_net = nullptr; // the batch at network size
        uint8_t* m_host_net = nullptr; // ... and its pinned staging copy on the host
        std::shared_ptr<std::atomic<uint64_t>> m_calls; // bumped whenever the engine's output buffers are about to be overwritten
        std::weak_ptr<detail::device_batch> m_last;
    };

} // namespace dnn
} // namespace hyperpose
