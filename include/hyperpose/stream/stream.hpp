// hyperpose::stream / hyperpose::make_stream on the GPU — the surface of the reference's stream operator
// (include/hyperpose/stream/stream.hpp:121-319):
//     auto stream = hyperpose::make_stream(engine, parser [, use_original_resolution, keep_ratio, parser_cnt, queue_max_size]);
//     stream.async() << frames;        // std::vector<cv::Mat>, one cv::Mat, (with OpenCV) a cv::VideoCapture
//     stream.sync() >> sink;           // blocks until every ingested frame has come out, in input order
// for all three parsers (paf, pose_proposal, pifpaf).  The reference wires four CPU threads and three mutex-guarded queues
// (resize -> inference -> parse -> write, stream.hpp:326-385, src/stream.cpp) with a host round trip between each; here ONE feeder thread
// cuts the input into batches and hands them to hp_pipeline_* (include/hp_hip.h): the frames stay on the device from the H2D copy to
// the parsed humans (cv::resize / non_scaling_resize, conv stack and parser kernels on one HIP stream per batch, several batches in
// flight), and resume_ratio is applied on the way out when keep_ratio is set (src/stream.cpp:120-124).
// Sinks: with OpenCV (HYPERPOSE_USE_OPENCV) a cv::VideoWriter or a name generator `std::string()` exactly as in the reference
// (draw_human + write / imwrite); in any build a `std::vector<std::vector<human_t>>` (one pose set per frame, appended in order) or a
// callable `void(size_t index, const cv::Mat& frame, const std::vector<human_t>& poses)`.
// `hip_stream` (push / pop of whole batches) is the thin form underneath.
#pragma once
#include "../../hp_hip.h"
#include "../operator/dnn/tensorrt.hpp"
#include "../operator/parser/paf.hpp"
#include "../utility/cv_min.hpp"
#include "../utility/human.hpp"

#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

namespace hyperpose {

namespace detail {
    inline void hp_check(int rc)
    {
        if (rc != HP_OK)
            throw std::runtime_error(hp_last_error());
    }
    inline human_t to_human(const hp_human& h)
    {
        human_t r;
        r.score = h.score;
        for (int k = 0; k < COCO_N_PARTS; ++k)
            r.parts[k] = body_part_t{ h.parts[k].has_value != 0, h.parts[k].x, h.parts[k].y, h.parts[k].score };
        return r;
    }
} // namespace detail

// Whole batches in, whole batches out (submission order); at most n_pipes batches in flight.
class hip_stream {
public:
    using pose_set = std::vector<human_t>; // one frame

    hip_stream(const dnn::builtin_model& model, cv::Size input_size, int max_batch_size = 8, bool keep_ratio = false,
        int n_pipes = 4, cv::Size max_frame = cv::Size(1920, 1080), float conf_thresh = 0.05f, float paf_thresh = 0.05f,
        double factor = 1. / 255, bool flip_rgb = true, data_type dtype = data_type::kFLOAT)
        : m_max_batch(max_batch_size)
    {
        dnn::tensorrt engine(model, input_size, max_batch_size, keep_ratio, dtype, factor, flip_rgb);
        hp_parser_desc pd{};
        pd.kind = HP_PARSER_PAF, pd.thresh[0] = conf_thresh, pd.thresh[1] = paf_thresh, pd.res_w = pd.res_h = -1;
        init(engine.handle(), pd, max_batch_size, keep_ratio, n_pipes, max_frame);
    }
    // replicate an existing engine (any model source) behind any of the three parsers
    hip_stream(hp_engine* engine, const hp_parser_desc& parser, int max_batch_size, bool keep_ratio, int n_pipes = 4,
        cv::Size max_frame = cv::Size(1920, 1080))
        : m_max_batch(max_batch_size)
    {
        init(engine, parser, max_batch_size, keep_ratio, n_pipes, max_frame);
    }
    hip_stream(const hip_stream&) = delete;
    ~hip_stream() { hp_pipeline_destroy(m_pl); }

    size_t in_flight() const { return (size_t)hp_pipeline_in_flight(m_pl); }
    size_t truncated() const { return m_truncated.load(); }
    int n_pipes() const { return m_pipes; }
    int max_batch() const { return m_max_batch; }

    // one batch (<= max_batch_size frames, any sizes); throws when every pipe is busy
    void push(const std::vector<cv::Mat>& frames)
    {
        std::vector<const uint8_t*> ptrs;
        std::vector<int> ws, hs;
        m_scratch.resize(frames.size());
        for (size_t i = 0; i < frames.size(); ++i) {
            ptrs.push_back(detail::mat_bytes(frames[i], m_scratch[i]));
            ws.push_back(frames[i].cols), hs.push_back(frames[i].rows);
        }
        detail::hp_check(hp_pipeline_submit(m_pl, ptrs.data(), ws.data(), hs.data(), (int)frames.size()));
    }

    // the oldest batch's humans, one pose_set per frame
    std::vector<pose_set> pop()
    {
        int nf = 0;
        const int rc = hp_pipeline_collect(m_pl, m_out.data(), CAP, m_n.data(), &nf);
        if (rc != HP_ERR_CAPACITY)
            detail::hp_check(rc);
        // HP_ERR_CAPACITY: a frame exceeded a hard list limit of the parser - its pose list is cut, the batch is otherwise complete.  The
        // frames concerned are the ones whose list comes back full (the parser stops at the limit).
        std::vector<pose_set> r(nf);
        size_t cut = 0;
        for (int i = 0; i < nf; ++i) {
            cut += m_n[i] >= CAP;
            for (int k = 0; k < m_n[i] && k < CAP; ++k)
                r[i].push_back(detail::to_human(m_out[(size_t)i * CAP + k]));
        }
        if (rc == HP_ERR_CAPACITY)
            m_truncated += cut ? cut : 1; // frames, not batches (at least the one the parser reported)
        return r;
    }

private:
    static constexpr int CAP = 128;
    void init(hp_engine* engine, const hp_parser_desc& parser, int max_batch_size, bool keep_ratio, int n_pipes, cv::Size max_frame)
    {
        hp_engine_desc d{};
        detail::hp_check(hp_engine_describe(engine, &d));
        d.max_batch = max_batch_size;
        detail::hp_check(hp_pipeline_create_ex(&m_pl, &d, &parser, n_pipes, keep_ratio ? 1 : 0, (size_t)max_frame.area() * 3));
        m_pipes = n_pipes;
        m_out.resize((size_t)max_batch_size * CAP);
        m_n.resize(max_batch_size);
    }
    hp_pipeline* m_pl = nullptr;
    int m_max_batch, m_pipes = 0;
    std::atomic<size_t> m_truncated{ 0 };
    std::vector<hp_human> m_out;
    std::vector<int> m_n;
    std::vector<std::vector<uint8_t>> m_scratch;
};

template <typename DNNEngine, typename Parser>
class stream {
public:
    using pose_set = std::vector<human_t>;

    /// Same parameters as the reference (stream.hpp:136): `parser_cnt` (CPU parser replicas there) is the number of batches kept in
    /// flight here (0: 4), `queue_max_size` bounds BOTH the frames waiting for a batch slot and the finished frames waiting for a sink
    /// (the reference bounds every one of its queues with it, src/stream.cpp:7-15).  `max_frame_size`: the largest source frame
    /// the device staging buffers are sized for (an addition: the reference resizes on the host and has no such limit).
    explicit stream(DNNEngine& engine, Parser& parser, bool use_original_resolution = false, bool keep_ratio = false, size_t parser_cnt = 0,
        size_t queue_max_size = 128, cv::Size max_frame_size = cv::Size(1920, 1080))
        : m_engine_ref(engine), m_main_parser_ref(parser), m_use_original_resolution(use_original_resolution), m_keep_ratio(keep_ratio)
        , m_queue_max(queue_max_size ? queue_max_size : 1)
        , m_gpu(engine.handle(), parser.stream_desc(), engine.max_batch_size(), keep_ratio, parser_cnt == 0 ? 4 : (int)std::min<size_t>(parser_cnt, 16),
              max_frame_size)
    {
        m_worker = std::thread([this] { run(); });
        m_input_worker = std::thread([this] { run_inputs(); });
    }
    stream(const stream&) = delete;
    ~stream()
    {
        {
            std::lock_guard<std::mutex> lk(m_mu);
            m_shutdown = true;
        }
        notify_everyone();
        {
            std::lock_guard<std::mutex> lk(m_mu_sinks);
            m_async_sinks.clear(); // joins the sink threads (they leave on m_shutdown)
        }
        if (m_input_worker.joinable())
            m_input_worker.join();
        if (m_worker.joinable())
            m_worker.join();
    }

    class async_handler {
        stream& m_stream;

    public:
        async_handler(stream& s)
            : m_stream(s)
        {
        }
        template <typename S>
        async_handler& operator<<(S&& source)
        {
            m_stream.add_input_stream(std::forward<S>(source));
            return *this;
        }
        // (asynchronous output as in the reference: the sink is drained on another thread; it must outlive the stream or be awaited
        // through a later sync() call)
        template <typename S>
        async_handler& operator>>(S&& sink)
        {
            // (the sink thread must not let an exception escape - std::terminate -: a failure of the stream is kept for the next call
            // that can report it)
            std::lock_guard<std::mutex> lk(m_stream.m_mu_sinks);
            m_stream.m_async_sinks.emplace_back([&s = m_stream, &sink] {
                try {
                    s.write_to(sink);
                } catch (const std::exception& e) {
                    std::lock_guard<std::mutex> lk2(s.m_mu);
                    if (s.m_error.empty())
                        s.m_error = e.what();
                }
            });
            return *this;
        }
    };
    class sync_handler {
        stream& m_stream;

    public:
        sync_handler(stream& s)
            : m_stream(s)
        {
        }
        // As in the reference (stream.hpp:211-215: the future of the input job is dropped), "synchronous" input is still ingested by
        // the single input thread: `stream.sync() << frames; stream.sync() >> sink;` cannot dead-lock on the bounded queues.
        template <typename S>
        sync_handler& operator<<(S&& source)
        {
            m_stream.add_input_stream(std::forward<S>(source));
            return *this;
        }
        template <typename S>
        sync_handler& operator>>(S&& sink)
        {
            m_stream.write_to(sink);
            return *this;
        }
    };
    async_handler async() { return *this; }
    sync_handler sync() { return *this; }

    /// (the reference prints queue lengths periodically, src/stream.cpp add_queue_monitor; kept as a no-op hook)
    void add_monitor(size_t) {}
    size_t processed_num() const noexcept { return m_ingest.load(); }
    /// frames whose pose list was cut at a hard capacity of the device parser (reported, the stream keeps running)
    size_t truncated_num() const noexcept { return m_gpu.truncated(); }

private:
    struct item {
        cv::Mat frame;
        pose_set poses;
    };

    // ---- input side (src/stream.cpp:18-66).  Like the reference's one-thread `m_mpsc_worker` (stream.hpp:248-254) a single input
    // thread ingests one source at a time, so `<<` returns at once and the bounded queues exert back-pressure on that thread, not on
    // the caller.  Sources are held by value where that is a handle copy (cv::Mat, vectors of them) and by reference for a
    // cv::VideoCapture (which must outlive the ingestion, as in the reference).
    void post_input(std::function<void()> job)
    {
        {
            std::lock_guard<std::mutex> lk(m_mu);
            if (m_shutdown) { // the input thread is gone: the source would be queued and never read
                rethrow_or_ignore();
                throw std::runtime_error("hyperpose::stream: the stream is closed, the source was not accepted");
            }
            ++m_pending_inputs;
            m_input_jobs.push_back(std::move(job));
        }
        m_cv_jobs.notify_one();
    }
    void rethrow_or_ignore()
    {
        if (!m_error.empty())
            throw std::runtime_error(m_error);
    }
    void add_input_stream(const std::vector<cv::Mat>& frames)
    {
        post_input([this, frames] {
            for (const auto& f : frames)
                if (!enqueue(f))
                    return;
        });
    }
    void add_input_stream(std::vector<cv::Mat>& frames) { add_input_stream(static_cast<const std::vector<cv::Mat>&>(frames)); }
    void add_input_stream(std::vector<cv::Mat>&& frames)
    {
        post_input([this, frames = std::move(frames)] {
            for (const auto& f : frames)
                if (!enqueue(f))
                    return;
        });
    }
    void add_input_stream(const cv::Mat& f)
    {
        post_input([this, f] { enqueue(f); });
    }
    void add_input_stream(cv::Mat& f) { add_input_stream(static_cast<const cv::Mat&>(f)); }
    void add_input_stream(cv::Mat&& f) { add_input_stream(static_cast<const cv::Mat&>(f)); }
#ifdef HYPERPOSE_USE_OPENCV
    void add_input_stream(cv::VideoCapture& cap)
    {
        post_input([this, &cap] {
            while (cap.isOpened()) {
                cv::Mat mat;
                cap >> mat;
                if (mat.empty() || !enqueue(mat))
                    break;
            }
        });
    }
#endif
    void run_inputs()
    {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lk(m_mu);
                m_cv_jobs.wait(lk, [this] { return !m_input_jobs.empty() || m_shutdown; });
                if (m_shutdown)
                    return;
                job = std::move(m_input_jobs.front());
                m_input_jobs.pop_front();
            }
            try {
                job();
            } catch (const std::exception& e) {
                fail(e.what());
            }
            {
                std::lock_guard<std::mutex> lk(m_mu);
                --m_pending_inputs;
            }
            m_cv_out.notify_all(); // a sink waiting for "everything ingested so far" re-checks
        }
    }
    // false once the stream has shut down (error or destruction): the source stops feeding
    bool enqueue(const cv::Mat& f)
    {
        if (f.empty())
            return true;
        std::unique_lock<std::mutex> lk(m_mu);
        m_cv_space.wait(lk, [this] { return m_in.size() < m_queue_max || m_shutdown; });
        if (m_shutdown)
            return false;
        m_in.push_back(f);
        ++m_remaining;
        ++m_ingest;
        m_cv_in.notify_one();
        return true;
    }
    void notify_everyone()
    {
        m_cv_in.notify_all();
        m_cv_out.notify_all();
        m_cv_space.notify_all();
        m_cv_out_space.notify_all();
        m_cv_jobs.notify_all();
    }
    void fail(const char* what)
    {
        {
            std::lock_guard<std::mutex> lk(m_mu);
            if (m_error.empty())
                m_error = what;
            m_shutdown = true;
        }
        notify_everyone(); // producers blocked on a full queue and sinks waiting for results both wake up and see the error
    }

    // ---- feeder: batches -> hp_pipeline -> ordered results (the reference's resize / inference / parse stages, stream.hpp:326-385)
    void run()
    {
        std::deque<std::vector<cv::Mat>> inflight;
        for (;;) {
            std::vector<cv::Mat> batch;
            {
                std::unique_lock<std::mutex> lk(m_mu);
                if (inflight.empty())
                    m_cv_in.wait(lk, [this] { return !m_in.empty() || m_shutdown; });
                if (m_shutdown)
                    return;
                while (!m_in.empty() && (int)batch.size() < m_gpu.max_batch()) {
                    batch.push_back(std::move(m_in.front()));
                    m_in.pop_front();
                }
            }
            m_cv_space.notify_all();
            try {
                if (!batch.empty()) {
                    if ((int)m_gpu.in_flight() == m_gpu.n_pipes())
                        if (!deliver(inflight))
                            return;
                    m_gpu.push(batch);
                    inflight.push_back(std::move(batch));
                } else if (!inflight.empty()) {
                    if (!deliver(inflight))
                        return;
                }
            } catch (const std::exception& e) {
                fail(e.what());
                return;
            }
        }
    }
    // false when the stream shut down while waiting for room in the output queue
    bool deliver(std::deque<std::vector<cv::Mat>>& inflight)
    {
        auto poses = m_gpu.pop();
        std::vector<cv::Mat> frames = std::move(inflight.front());
        inflight.pop_front();
        std::unique_lock<std::mutex> lk(m_mu);
        // bounded like the reference's m_pose_sets_queue: with no sink attached the finished frames do not pile up without limit
        m_cv_out_space.wait(lk, [this] { return m_out.size() < m_queue_max || m_shutdown; });
        if (m_shutdown)
            return false;
        for (size_t i = 0; i < frames.size(); ++i)
            m_out.push_back(item{ std::move(frames[i]), i < poses.size() ? std::move(poses[i]) : pose_set{} });
        lk.unlock();
        m_cv_out.notify_all();
        return true;
    }

    // ---- output side: blocks until everything handed to `<<` so far has been written (src/stream.cpp:114-147)
    template <typename F>
    void drain(F&& emit)
    {
        for (;;) {
            item it;
            {
                std::unique_lock<std::mutex> lk(m_mu);
                m_cv_out.wait(lk, [this] { return !m_out.empty() || (m_remaining == 0 && m_pending_inputs == 0) || m_shutdown; });
                if (!m_error.empty())
                    throw std::runtime_error(m_error);
                if (m_out.empty()) {
                    if ((m_remaining == 0 && m_pending_inputs == 0) || m_shutdown)
                        return;
                    continue;
                }
                it = std::move(m_out.front());
                m_out.pop_front();
                --m_remaining;
            }
            m_cv_out_space.notify_all();
            emit(m_written++, it);
        }
    }
    void write_to(std::vector<pose_set>& sink)
    {
        drain([&](size_t, item& it) { sink.push_back(std::move(it.poses)); });
    }
    template <typename F>
    auto write_to(F& fn) -> decltype(fn(size_t(0), std::declval<const cv::Mat&>(), std::declval<const pose_set&>()), void())
    {
        drain([&](size_t idx, item& it) { fn(idx, static_cast<const cv::Mat&>(it.frame), static_cast<const pose_set&>(it.poses)); });
    }

public:
    /// What the writers draw (src/stream.cpp:114-147).  hp_pipeline_collect has already applied resume_ratio against the ORIGINAL
    /// frame; when the output is the letter-boxed network-sized image (use_original_resolution = false, keep_ratio = true) the
    /// reference's `resume_ratio(pose, raw_image.size() == input_size, input_size)` is the identity, so the ratio is undone here
    /// before drawing.  `network_sized` = the frame resized / letter-boxed to the network size by the caller (OpenCV builds: rendered()).
    static void poses_for_network_sized_image(pose_set& poses, cv::Size original, cv::Size network, bool keep_ratio)
    {
        if (!keep_ratio)
            return;
        for (auto& h : poses) {
            if (original.height * network.width > original.width * network.height) {
                const double xratio = (double)network.width * original.height / ((double)network.height * original.width);
                for (auto& par : h.parts)
                    par.x = (float)(par.x / xratio);
            } else {
                const double yratio = (double)network.height * original.width / ((double)network.width * original.height);
                for (auto& par : h.parts)
                    par.y = (float)(par.y / yratio);
            }
        }
    }

private:
#ifdef HYPERPOSE_USE_OPENCV
    cv::Mat rendered(item& it)
    {
        cv::Mat img = it.frame;
        if (!m_use_original_resolution) {
            cv::Mat r;
            if (m_keep_ratio)
                r = non_scaling_resize(img, m_engine_ref.input_size());
            else
                cv::resize(img, r, m_engine_ref.input_size());
            poses_for_network_sized_image(it.poses, img.size(), m_engine_ref.input_size(), m_keep_ratio);
            img = r;
        }
        for (auto&& pose : it.poses)
            draw_human(img, pose);
        return img;
    }
    void write_to(cv::VideoWriter& writer)
    {
        drain([&](size_t, item& it) { writer << rendered(it); });
    }
    template <typename NameGetter>
    auto write_to(NameGetter& name_getter) -> std::enable_if_t<std::is_convertible_v<decltype(name_getter()), std::string>>
    {
        drain([&](size_t, item& it) { cv::imwrite(name_getter(), rendered(it)); });
    }
#endif

    DNNEngine& m_engine_ref;
    Parser& m_main_parser_ref;
    const bool m_use_original_resolution, m_keep_ratio;
    const size_t m_queue_max;
    hip_stream m_gpu;

    std::mutex m_mu, m_mu_sinks; // (m_mu_sinks: the list of sink threads alone, never held together with a wait)
    std::condition_variable m_cv_in, m_cv_out, m_cv_space, m_cv_out_space, m_cv_jobs;
    std::deque<cv::Mat> m_in;
    std::deque<item> m_out;
    std::deque<std::function<void()>> m_input_jobs;
    size_t m_remaining = 0, m_written = 0, m_pending_inputs = 0;
    std::atomic<size_t> m_ingest{ 0 };
    bool m_shutdown = false;
    std::string m_error;
    std::thread m_worker, m_input_worker;
    struct joining_thread {
        std::thread t;
        template <typename F>
        explicit joining_thread(F&& f)
            : t(std::forward<F>(f))
        {
        }
        joining_thread(joining_thread&&) = default;
        ~joining_thread()
        {
            if (t.joinable())
                t.join();
        }
    };
    std::vector<joining_thread> m_async_sinks;
};

/// include/hyperpose/stream/stream.hpp:307-315
template <typename DNNEngine, typename Parser, typename... Others>
auto make_stream(DNNEngine&& engine, Parser&& parser, Others&&... others)
{
    return stream<std::remove_reference_t<DNNEngine>, std::remove_reference_t<Parser>>(
        std::forward<DNNEngine>(engine), std::forward<Parser>(parser), std::forward<Others>(others)...);
}

} // namespace hyperpose
