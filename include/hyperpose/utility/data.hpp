// hyperpose::feature_map_t / internal_t — reference include/hyperpose/utility/data.hpp:17-67.
// A feature_map_t owns a HOST copy (API compatibility with the reference, src/tensorrt.cpp:423-428).  The
// MI355X fast path keeps feature maps in HBM (dnn::tensorrt::inference_device + parser::paf::process_device).
#pragma once
#include <atomic>
#include <cstring>
#include <memory>
#include <mutex>
#include <ostream>
#include <array>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../hp_hip.h"
#include "human.hpp"

namespace hyperpose {

namespace detail {
    // The output tensors of ONE dnn::tensorrt::inference call while they still lie in the engine's device buffers.  The reference copies
    // every tensor of every image to the host before it returns (src/tensorrt.cpp:423-428) and the parser copies nothing back; here the
    // maps a call returns are views of this record: a parser mirror that is handed such a map reads the batch straight from HBM (and
    // parses ALL its frames in one launch the first time one of them is asked for), and the HOST copy the reference's feature_map_t
    // promises is made when somebody looks at it - view<T>() - or, at the latest, just before the engine overwrites the buffers (its
    // next inference call, or its destruction).  Semantics are the reference's; the 4.5 MB D2H + H2D round trip per LW-OpenPose batch
    // happens only for callers that really read the maps.
    struct device_batch {
        hp_engine* engine = nullptr;
        std::shared_ptr<std::atomic<uint64_t>> live; // the engine's call counter: == gen while the device buffers still hold this batch
        uint64_t gen = 0;
        int n = 0;
        struct out {
            std::string name;
            std::vector<int> shape; // {C, H, W}
            const float* dev = nullptr; // [n][C][H][W]
            size_t per = 0;         // floats per frame
        };
        std::vector<out> outs;
        std::vector<std::unique_ptr<char[]>> host; // per output, [n][per] floats, once materialised
        std::mutex m;
        bool device_valid() const { return live && live->load() == gen; }
        // the host copy of output i (made on first use, while the device buffers are valid; the engine calls it for every output before it
        // re-uses them)
        const char* host_of(int i)
        {
            std::lock_guard<std::mutex> lk(m);
            if (host.empty())
                host.resize(outs.size());
            if (!host[i]) {
                if (!device_valid())
                    throw std::logic_error("hyperpose: feature map outlived its engine's buffers without a host copy (engine bug)");
                std::unique_ptr<char[]> h{ new char[outs[i].per * n * sizeof(float)] };
                if (hp_engine_output_to_host(engine, i, n, reinterpret_cast<float*>(h.get())) != HP_OK)
                    throw std::runtime_error(hp_last_error());
                host[i] = std::move(h);
            }
            return host[i].get();
        }
        void materialize()
        {
            for (size_t i = 0; i < outs.size(); ++i)
                (void)host_of((int)i);
        }
    };
} // namespace detail

struct feature_map_t {
public:
    feature_map_t(std::string name, std::unique_ptr<char[]>&& tensor, std::vector<int> shape)
        : m_name(std::move(name)), m_data(std::move(tensor)), m_shape(std::move(shape)) {}
    /// (this library's engine: output `out` of frame `frame` of a batch that still lives on the device)
    feature_map_t(std::shared_ptr<detail::device_batch> batch, int out, int frame)
        : m_name(batch->outs[out].name), m_shape(batch->outs[out].shape), m_batch(std::move(batch)), m_out(out), m_frame(frame) {}
    friend std::ostream& operator<<(std::ostream& out, const feature_map_t& map)
    {
        out << map.m_name << ":[";
        for (auto& s : map.m_shape)
            out << s << ", ";
        return out << ']';
    }
    inline const std::string& name() const { return m_name; }
    inline const std::vector<int>& shape() const { return m_shape; }
    template <typename T>
    inline const T* view() const
    {
        if (m_batch)
            return reinterpret_cast<const T*>(m_batch->host_of(m_out) + (size_t)m_frame * m_batch->outs[m_out].per * sizeof(float));
        return reinterpret_cast<T*>(m_data.get());
    }
    // ---- additions: where the tensor lies on the device, if it still does (nullptr otherwise)
    inline const detail::device_batch* device_batch() const { return m_batch && m_batch->device_valid() ? m_batch.get() : nullptr; }
    inline const std::shared_ptr<detail::device_batch>& batch_handle() const { return m_batch; }
    inline int batch_output() const { return m_out; }
    inline int batch_frame() const { return m_frame; }

private:
    std::string m_name;
    std::unique_ptr<char[]> m_data;
    std::vector<int> m_shape;
    std::shared_ptr<detail::device_batch> m_batch;
    int m_out = 0, m_frame = 0;
};

using internal_t = std::vector<feature_map_t>;

// ---- free functions of the reference's data.hpp (:58-67), evaluated by the same device code as the engine's own pre-processing
namespace detail {
    struct dev_ptr { // scoped hp_malloc
        void* p = nullptr;
        explicit dev_ptr(size_t n) { hp_malloc(&p, n); }
        ~dev_ptr() { hp_free(p); }
        dev_ptr(const dev_ptr&) = delete;
    };
    inline const uint8_t* mat_data(const cv::Mat& m)
    {
#ifdef HYPERPOSE_USE_OPENCV
        return m.data;
#else
        return m.data();
#endif
    }
} // namespace detail

/// nhwc_images_append_nchw_batch (include/hyperpose/utility/data.hpp:58-64, src/data.cpp:21-51): u8 HWC images -> f32 CHW appended to
/// `data`, every value multiplied by `factor`, channels {2,1,0} when flip_rb.  hp_preproc_u8hwc_to_f32nchw does the arithmetic.
inline void nhwc_images_append_nchw_batch(std::vector<float>& data, std::vector<cv::Mat> images, double factor = 1.0, bool flip_rb = false)
{
    for (const auto& im : images) {
        const size_t n = (size_t)im.rows * im.cols * 3;
        if (!n)
            continue;
        detail::dev_ptr din(n), dout(n * sizeof(float));
        const size_t at = data.size();
        data.resize(at + n);
        if (!din.p || !dout.p || hp_memcpy_h2d(din.p, detail::mat_data(im), n) != HP_OK
            || hp_preproc_u8hwc_to_f32nchw((const uint8_t*)din.p, 1, im.rows, im.cols, factor, flip_rb ? 1 : 0, (float*)dout.p, nullptr) != HP_OK
            || hp_device_synchronize() != HP_OK || hp_memcpy_d2h(data.data() + at, dout.p, n * sizeof(float)) != HP_OK)
            throw std::runtime_error(hp_last_error());
    }
}

/// non_scaling_resize (data.hpp:67, src/data.cpp:53-69): aspect-preserving resize into the top-left corner of a `size` image filled with
/// `bgcolor`; hp_letterbox_u8c3 does the arithmetic (bit-equal to cv::resize INTER_LINEAR on the resized region).
inline cv::Mat non_scaling_resize(const cv::Mat& input, const cv::Size& size, const std::array<int, 3>& bgcolor = { 0, 0, 0 })
{
    cv::Mat out(size.height, size.width, CV_8UC3);
    const size_t ni = (size_t)input.rows * input.cols * 3, no = (size_t)size.area() * 3;
    detail::dev_ptr din(ni), dout(no);
    if (!din.p || !dout.p || hp_memcpy_h2d(din.p, detail::mat_data(input), ni) != HP_OK
        || hp_letterbox_u8c3((const uint8_t*)din.p, input.cols, input.rows, input.cols * 3, (uint8_t*)dout.p, size.width, size.height, size.width * 3,
               bgcolor[0], bgcolor[1], bgcolor[2], nullptr) != HP_OK
        || hp_device_synchronize() != HP_OK || hp_memcpy_d2h(const_cast<uint8_t*>(detail::mat_data(out)), dout.p, no) != HP_OK)
        throw std::runtime_error(hp_last_error());
    return out;
}

} // namespace hyperpose
