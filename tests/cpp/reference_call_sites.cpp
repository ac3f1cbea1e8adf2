// Reference call sites, VERBATIM, against the mirror headers: the engine-construction lambda of
// examples/operator_api_batched_images_paf.example.cpp:36-56 (onnx / uff / tensorrt_serialized descriptors, positional constructor
// arguments), its inference + parser loop (:58-74), and the stream lines of examples/stream_api_video_paf.example.cpp:56-78
// (make_stream(engine, parser, flag), add_monitor, async() <<, sync() >>) with a frame vector in place of cv::VideoCapture and a
// pose-set vector in place of cv::VideoWriter (no OpenCV in this image).  Only the gflags variables and two helpers of
// examples/utils.hpp are provided here.  Prints "OK ..." and is run by tests/test_cpp_mirror.py on the GPU box.
#include <hyperpose/hyperpose.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <iostream>
#include <sstream>
#include <string_view>

static std::string FLAGS_model_file;
static std::string FLAGS_input_name = "image";
static std::string FLAGS_output_name_list = "outputs/conf,outputs/paf";
static int FLAGS_input_height = 64, FLAGS_input_width = 48, FLAGS_max_batch_size = 4;
static bool FLAGS_original_resolution = false;

static std::vector<std::string> split(const std::string& text, char sep) // examples/utils.hpp
{
    std::vector<std::string> tokens;
    std::stringstream ss(text);
    std::string item;
    while (std::getline(ss, item, sep))
        tokens.push_back(item);
    return tokens;
}
static std::ostream& example_log() { return std::cout << "[HyperPose::EXAMPLE] "; }

static std::vector<cv::Mat> make_frames(int n, int w, int h, int salt)
{
    std::vector<cv::Mat> v;
    for (int i = 0; i < n; ++i) {
        cv::Mat m(h, w);
        for (size_t k = 0; k < m.total() * 3; ++k)
            m.data()[k] = (uint8_t)((k * 29 + i * 11 + salt) & 255);
        v.push_back(m);
    }
    return v;
}

int main(int argc, char** argv)
{
    if (argc < 2)
        return 90;
    if (hp_init(0) != HP_OK) {
        std::printf("NO_DEVICE %s\n", hp_last_error());
        return 2;
    }
    FLAGS_model_file = argv[1];
    std::vector<cv::Mat> batch = make_frames(3, FLAGS_input_width, FLAGS_input_height, 0);
    namespace hp = hyperpose;

    // ---- examples/operator_api_batched_images_paf.example.cpp:36-56, verbatim
    auto engine = [&] {
        using namespace hp::dnn;
        constexpr std::string_view onnx_suffix = ".onnx";
        constexpr std::string_view uff_suffix = ".uff";

        if (std::equal(onnx_suffix.crbegin(), onnx_suffix.crend(), FLAGS_model_file.crbegin()))
            return tensorrt(onnx{ FLAGS_model_file }, { FLAGS_input_width, FLAGS_input_height }, batch.size());

        if (std::equal(uff_suffix.crbegin(), uff_suffix.crend(), FLAGS_model_file.crbegin()))
            return tensorrt(
                uff{ FLAGS_model_file, FLAGS_input_name, split(FLAGS_output_name_list, ',') },
                { FLAGS_input_width, FLAGS_input_height },
                batch.size());

        example_log() << "Your model file's suffix is not [.onnx | .uff]. Your model file path: " << FLAGS_model_file;
        example_log() << "Trying to be viewed as a serialized TensorRT model.";

        return tensorrt(tensorrt_serialized{ FLAGS_model_file }, { FLAGS_input_width, FLAGS_input_height }, batch.size());
    }();

    hp::parser::paf parser{};

    using clk_t = std::chrono::high_resolution_clock;
    auto beg = clk_t::now();
    size_t n_humans = 0;
    {
        // ---- :62-74, verbatim
        // * TensorRT Inference.
        auto feature_map_packets = engine.inference(batch);
        for (const auto& packet : feature_map_packets)
            for (const auto& feature_map : packet)
                example_log() << feature_map << std::endl;

        // * Paf.
        std::vector<std::vector<hp::human_t>> pose_vectors;
        pose_vectors.reserve(feature_map_packets.size());
        for (auto&& packet : feature_map_packets) {
            pose_vectors.push_back(parser.process(packet[0], packet[1]));
        }

        std::cout << batch.size() << " images got processed. FPS = "
                  << 1000. * batch.size() / std::chrono::duration<double, std::milli>(clk_t::now() - beg).count()
                  << '\n';
        if (pose_vectors.size() != batch.size())
            return 3;
        for (auto& v : pose_vectors)
            n_humans += v.size();
    }

    // ---- the other constructor forms of include/hyperpose/operator/dnn/tensorrt.hpp:44-74 with every positional argument
    {
        using namespace hp::dnn;
        tensorrt full(onnx{ FLAGS_model_file }, { FLAGS_input_width, FLAGS_input_height }, 2, /*keep_ratio*/ true, hp::data_type::kHALF, 1. / 255, true);
        const std::string saved = std::string(argv[1]) + ".hpeng.tmp";
        full.save(saved);
        tensorrt again(tensorrt_serialized{ saved }, { FLAGS_input_width, FLAGS_input_height }, 2, /*keep_ratio*/ true, 1. / 255, true);
        // frames of another size: resized / letter-boxed on the device (src/tensorrt.cpp:446-451)
        auto big = make_frames(2, 200, 120, 5);
        auto a = full.inference(big), b = again.inference(big);
        if (a.size() != 2 || b.size() != 2 || a[0].size() != 2)
            return 4;
        for (size_t i = 0; i < a.size(); ++i)
            for (size_t k = 0; k < a[i].size(); ++k) {
                size_t n = 1;
                for (int d : a[i][k].shape())
                    n *= d;
                if (std::memcmp(a[i][k].view<float>(), b[i][k].view<float>(), n * sizeof(float)) != 0)
                    return 5; // a reloaded engine is bit-identical
            }
        std::remove(saved.c_str());
    }

    // ---- examples/stream_api_video_paf.example.cpp:74-88 (make_stream, add_monitor, async() <<, sync() >>)
    size_t stream_frames = 0, stream_humans = 0;
    {
        auto stream = hp::make_stream(engine, parser, FLAGS_original_resolution);

        stream.add_monitor(1000);

        std::vector<cv::Mat> capture = make_frames(11, 160, 100, 3); // 11 frames: several batches of <= 3, any size
        std::vector<std::vector<hp::human_t>> writer;

        stream.async() << capture;

        stream.sync() >> writer;

        if (writer.size() != capture.size() || stream.processed_num() != capture.size())
            return 6;
        // the same frames through the operator API give the same pose sets, in order
        for (size_t i = 0; i < capture.size(); ++i) {
            auto packets = engine.inference({ capture[i] });
            auto poses = parser.process(packets[0][0], packets[0][1]);
            if (poses.size() != writer[i].size())
                return 7;
            for (size_t h = 0; h < poses.size(); ++h)
                if (std::memcmp(&poses[h], &writer[i][h], sizeof(hp::human_t)) != 0)
                    return 8;
            stream_humans += poses.size();
        }
        stream_frames = writer.size();
        // a second round on the same stream, one frame at a time, callable sink
        size_t seen = 0;
        for (int i = 0; i < 5; ++i)
            stream.async() << capture[i];
        auto sink = [&](size_t, const cv::Mat& frame, const std::vector<hp::human_t>& poses) {
            seen += (frame.cols == 160) + 0 * poses.size();
        };
        stream.sync() >> sink;
        if (seen != 5 || stream.processed_num() != capture.size() + 5)
            return 9;
    }
    // ---- the other two parsers behind make_stream
    {
        hp::dnn::tensorrt ppn_engine(hp::dnn::builtin_model{ "pose_proposal_resnet50", {}, 3 }, cv::Size(160, 128), 2);
        hp::parser::pose_proposal ppn(cv::Size(160, 128));
        auto s1 = hp::make_stream(ppn_engine, ppn);
        std::vector<std::vector<hp::human_t>> out1;
        s1.async() << make_frames(5, 210, 170, 1);
        s1.sync() >> out1;
        hp::dnn::tensorrt pp_engine(hp::dnn::builtin_model{ "pifpaf_resnet50", {}, 4 }, cv::Size(97, 97), 2);
        hp::parser::pifpaf pp(97, 97);
        auto s2 = hp::make_stream(pp_engine, pp, false, true);
        std::vector<std::vector<hp::human_t>> out2;
        s2.async() << make_frames(3, 120, 90, 2);
        s2.sync() >> out2;
        if (out1.size() != 5 || out2.size() != 3)
            return 10;
    }
    bool threw = false;
    try {
        engine.inference(std::vector<cv::Mat>(5, batch[0]));
    } catch (const std::logic_error&) {
        threw = true;
    }
    std::printf("OK %zu %zu %zu %d\n", n_humans, stream_frames, stream_humans, (int)threw);
    return threw ? 0 : 11;
}
