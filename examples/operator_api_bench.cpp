// examples/operator_api_bench.cpp — the reference's operator-API loop, timed (VERDICT r5 missing #3 / item 3c).
//
// The loop is examples/operator_api_batched_images_paf.example.cpp:60-76 of the reference, statement for statement:
//     auto feature_map_packets = engine.inference(batch);
//     for (auto&& packet : feature_map_packets) pose_vectors.push_back(parser.process(packet[0], packet[1]));
// with host cv::Mat frames in and std::vector<human_t> out, one batch in flight - a synchronous caller, no pipelining by the caller.
// Behind the two calls: frames gathered in pinned memory -> one asynchronous H2D per half-batch -> the conv stack as two half-batches
// side by side (kFLOAT) -> the batch's maps parsed in ONE launch from HBM when the first frame is asked for -> humans on the host.
//
//     operator_api_bench [arch = lw_openpose_mobilenet] [w = 432] [h = 368] [batch = 8] [seconds = 2] [dtype = f32 | f16]
// prints one JSON line: {"operator_api_fps": ..., "ms_per_batch": ..., "batches": ..., "humans_per_batch": ..., ...}
#include <hyperpose/hyperpose.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

int main(int argc, char** argv)
{
    namespace hp = hyperpose;
    const std::string arch = argc > 1 ? argv[1] : "lw_openpose_mobilenet";
    const int w = argc > 2 ? std::atoi(argv[2]) : 432, h = argc > 3 ? std::atoi(argv[3]) : 368, batch = argc > 4 ? std::atoi(argv[4]) : 8;
    const double seconds = argc > 5 ? std::atof(argv[5]) : 2.0;
    const bool half = argc > 6 && std::strcmp(argv[6], "f16") == 0;
    if (hp_init(0) != HP_OK) {
        std::printf("{\"error\": \"%s\"}\n", hp_last_error());
        return 2;
    }
    using clk = std::chrono::steady_clock;
    hp::dnn::tensorrt engine(hp::dnn::builtin_model{ arch, {}, 20241 }, cv::Size(w, h), batch, false, half ? hp::data_type::kHALF : hp::data_type::kFLOAT);
    hp::parser::paf parser{};
    std::vector<cv::Mat> frames;
    uint32_t x = 12345;
    for (int i = 0; i < batch; ++i) {
        cv::Mat m(h, w);
        for (size_t k = 0; k < m.total() * 3; ++k) {
            x = x * 1664525u + 1013904223u;
            m.data()[k] = (uint8_t)(x >> 24);
        }
        frames.push_back(m);
    }
    double t_inf = 0, t_parse = 0; // where a step's time goes: engine.inference() | the parser.process() calls
    auto step = [&]() {
        size_t humans = 0;
        const auto a = clk::now();
        auto feature_map_packets = engine.inference(frames);
        const auto b = clk::now();
        std::vector<std::vector<hp::human_t>> pose_vectors;
        pose_vectors.reserve(feature_map_packets.size());
        for (auto&& packet : feature_map_packets)
            pose_vectors.push_back(parser.process(packet[0], packet[1]));
        for (auto& v : pose_vectors)
            humans += v.size();
        const auto c = clk::now();
        t_inf += std::chrono::duration<double>(b - a).count(), t_parse += std::chrono::duration<double>(c - b).count();
        return humans;
    };
    const auto warm = clk::now();
    while (std::chrono::duration<double>(clk::now() - warm).count() < 0.5)
        step();
    size_t batches = 0, humans = 0;
    t_inf = t_parse = 0;
    const auto t0 = clk::now();
    double dt = 0;
    do {
        humans += step();
        ++batches;
        dt = std::chrono::duration<double>(clk::now() - t0).count();
    } while (dt < seconds);
    std::printf("{\"operator_api_fps\": %.1f, \"ms_per_batch\": %.4f, \"batches\": %zu, \"batch\": %d, \"humans_per_batch\": %.2f, \"inference_ms\": %.4f, \"parse_ms\": %.4f, \"arch\": \"%s\", \"dtype\": \"%s\", "
                "\"engine_concurrency\": %d, \"loop\": \"engine.inference(std::vector<cv::Mat>) then parser.process(packet[0], packet[1]) per frame, one batch in flight\"}\n",
        batch * batches / dt, dt / batches * 1e3, batches, batch, (double)humans / batches, t_inf / batches * 1e3, t_parse / batches * 1e3, arch.c_str(), half ? "f16" : "f32", hp_engine_concurrency(engine.handle()));
    return 0;
}
