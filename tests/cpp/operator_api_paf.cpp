// The reference's operator-API example (examples/operator_api_batched_images_paf.example.cpp:58-74) written
// against the mirror headers: engine.inference(batch) -> for each packet parser.process(packet[0], packet[1]).
// Prints "OK <n_outputs> <humans>"; run by tests/test_cpp_mirror.py on the GPU box.
#define HYPERPOSE_TENSORRT_COMPAT
#include <hyperpose/hyperpose.hpp>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static bool same_humans(const std::vector<hyperpose::human_t>& a, const std::vector<hyperpose::human_t>& b)
{
    if (a.size() != b.size())
        return false;
    for (size_t i = 0; i < a.size(); ++i) {
        if (a[i].score != b[i].score)
            return false;
        for (int k = 0; k < hyperpose::COCO_N_PARTS; ++k)
            if (a[i].parts[k].has_value != b[i].parts[k].has_value || a[i].parts[k].x != b[i].parts[k].x || a[i].parts[k].y != b[i].parts[k].y
                || a[i].parts[k].score != b[i].parts[k].score)
                return false;
    }
    return true;
}

int main(int argc, char** argv)
{
    if (hp_init(0) != HP_OK) {
        std::printf("NO_DEVICE %s\n", hp_last_error());
        return 2;
    }
    namespace hp = hyperpose;
    hp::dnn::tensorrt engine(hp::dnn::builtin_model{ "lw_openpose_mobilenet", {}, 7 }, cv::Size(96, 80), 4);
    hp::parser::paf parser{};
    std::vector<cv::Mat> batch;
    for (int i = 0; i < 3; ++i) {
        cv::Mat m(80, 96);
        for (size_t k = 0; k < m.total() * 3; ++k)
            m.data()[k] = (uint8_t)((k * 31 + i * 7) & 255);
        batch.push_back(m);
    }
    auto packets = engine.inference(batch);
    size_t humans = 0;
    for (auto& packet : packets) {
        if (packet.size() != 2 || packet[0].name() != "conf" || packet[1].name() != "paf")
            return 3;
        hp::parser::paf p(parser); // copy-construct like the stream API does (stream.hpp:139)
        humans += p.process(packet[0], packet[1]).size();
    }
    {   // the maps above never left the device (utility/data.hpp, detail::device_batch): the batch was parsed in one launch at the first
        // process() call.  The reference's form - every map copied to the host, every frame uploaded again and parsed by itself - must give
        // the same humans, and a map's host view must be what hp_engine_output_to_host returns
        setenv("HP_MIRROR_HOST_MAPS", "1", 1);
        size_t again = 0;
        for (size_t f = 0; f < packets.size(); ++f) {
            hp::parser::paf p(parser);
            const auto a = p.process(packets[f][0], packets[f][1]);
            unsetenv("HP_MIRROR_HOST_MAPS");
            hp::parser::paf q(parser);
            const auto b = q.process(packets[f][0], packets[f][1]);
            setenv("HP_MIRROR_HOST_MAPS", "1", 1);
            if (a.size() != b.size())
                return 20;
            for (size_t i = 0; i < a.size(); ++i) {
                if (a[i].score != b[i].score)
                    return 21;
                for (int k = 0; k < hp::COCO_N_PARTS; ++k)
                    if (a[i].parts[k].has_value != b[i].parts[k].has_value || a[i].parts[k].x != b[i].parts[k].x || a[i].parts[k].y != b[i].parts[k].y
                        || a[i].parts[k].score != b[i].parts[k].score)
                        return 21;
            }
            again += a.size();
        }
        unsetenv("HP_MIRROR_HOST_MAPS");
        if (again != humans)
            return 22;
        const auto& cm = packets[1][0];
        std::vector<float> direct((size_t)3 * cm.shape()[0] * cm.shape()[1] * cm.shape()[2]);
        if (hp_engine_output_to_host(engine.handle(), 0, 3, direct.data()) != HP_OK)
            return 23;
        const size_t per = direct.size() / 3;
        if (std::memcmp(cm.view<float>(), direct.data() + per, per * sizeof(float)) != 0)
            return 24;
        // a map that outlives the next inference call keeps the values of ITS call (the host copy is made just before the buffers are re-used)
        auto first = engine.inference({ batch[0] });
        std::vector<float> keep(first[0][1].view<float>(), first[0][1].view<float>() + (size_t)first[0][1].shape()[0] * first[0][1].shape()[1] * first[0][1].shape()[2]);
        auto held = engine.inference({ batch[1] });          // `held`'s maps stay on the device ...
        auto other = engine.inference({ batch[0] });         // ... until this call retires them
        if (std::memcmp(first[0][1].view<float>(), keep.data(), keep.size() * sizeof(float)) != 0)
            return 25;
        if (std::memcmp(other[0][1].view<float>(), keep.data(), keep.size() * sizeof(float)) != 0) // same frame, same values
            return 26;
        if (std::memcmp(held[0][1].view<float>(), keep.data(), keep.size() * sizeof(float)) == 0)  // another frame: other values, and its own
            return 27;
        hp::parser::paf late(parser);
        (void)late.process(held[0][0], held[0][1]); // maps whose batch has left the device are parsed from their host copies
    }
    {   // the other two parsers through their mirrors, fed by their engines (reduced sizes)
        hp::dnn::tensorrt ppn_engine(hp::dnn::builtin_model{ "pose_proposal_resnet50", {}, 3 }, cv::Size(160, 128), 2);
        hp::parser::pose_proposal ppn(cv::Size(160, 128));
        cv::Mat m(128, 160);
        for (size_t k = 0; k < m.total() * 3; ++k)
            m.data()[k] = (uint8_t)((k * 13) & 255);
        auto out = ppn_engine.inference({ m, m });
        if (out[0].size() != 7)
            return 5;
        const auto dev0 = ppn.process(out[0]), dev1 = ppn.process(out[1]); // from the device, both frames in one launch
        setenv("HP_MIRROR_HOST_MAPS", "1", 1);
        hp::parser::pose_proposal ppn_host(cv::Size(160, 128));
        const auto host0 = ppn_host.process(out[0]), host1 = ppn_host.process(out[1]); // the reference's form: host maps, frame by frame
        unsetenv("HP_MIRROR_HOST_MAPS");
        if (!same_humans(dev0, host0) || !same_humans(dev1, host1) || !same_humans(dev0, dev1))
            return 30;
        humans += dev0.size();
        hp::dnn::tensorrt pp_engine(hp::dnn::builtin_model{ "pifpaf_resnet50", {}, 4 }, cv::Size(97, 97), 2);
        hp::parser::pifpaf pp(97, 97);
        cv::Mat m2(97, 97);
        for (size_t k = 0; k < m2.total() * 3; ++k)
            m2.data()[k] = (uint8_t)((k * 7) & 255);
        auto out2 = pp_engine.inference({ m2, m2 });
        if (out2[0].size() != 2)
            return 6;
        const auto pdev0 = pp.process(out2[0]), pdev1 = pp.process(out2[1]);
        setenv("HP_MIRROR_HOST_MAPS", "1", 1);
        hp::parser::pifpaf pp_host(97, 97);
        const auto phost0 = pp_host.process(out2[0]), phost1 = pp_host.process(out2[1]);
        unsetenv("HP_MIRROR_HOST_MAPS");
        if (!same_humans(pdev0, phost0) || !same_humans(pdev1, phost1))
            return 31;
        humans += pdev0.size();
    }
    {   // the stream operator (stream.hpp:119-145): frames of any size in, pose sets out in submission order
        hp::hip_stream stream(hp::dnn::builtin_model{ "lw_openpose_mobilenet", {}, 7 }, cv::Size(96, 80), 4, /*keep_ratio*/ true, /*n_pipes*/ 2,
            cv::Size(640, 480));
        std::vector<cv::Mat> big;
        for (int i = 0; i < 3; ++i) {
            cv::Mat m(120 + 40 * i, 200 - 30 * i);
            for (size_t k = 0; k < m.total() * 3; ++k)
                m.data()[k] = (uint8_t)((k * 17 + i) & 255);
            big.push_back(m);
        }
        stream.push(big);
        stream.push({ big[1] });
        if (stream.in_flight() != 2)
            return 7;
        const auto first = stream.pop(), second = stream.pop();
        if (first.size() != 3 || second.size() != 1 || stream.in_flight() != 0)
            return 8;
    }
    if (argc > 1) { // dnn::onnx{path}: the reference's model descriptor (examples/operator_api_batched_images_paf.example.cpp:40-47)
        hp::dnn::tensorrt onnx_engine(hp::dnn::onnx{ argv[1] }, cv::Size(48, 64), 2);
        hp::parser::paf onnx_parser{};
        cv::Mat m(64, 48);
        for (size_t k = 0; k < m.total() * 3; ++k)
            m.data()[k] = (uint8_t)((k * 29) & 255);
        auto out = onnx_engine.inference({ m, m });
        if (out.size() != 2 || out[0].size() != 2 || out[0][0].name() != "conf" || out[0][1].name() != "paf")
            return 9;
        if (out[0][0].shape() != std::vector<int>{ 5, 16, 12 } || out[0][1].shape() != std::vector<int>{ 6, 16, 12 })
            return 10;
    }
    {   // data_type (include/hyperpose/operator/dnn/tensorrt.hpp:14-21,48): the default is kFLOAT, as in the reference; kHALF is the fused
        // fp16 engine.  Same model, same frame: the two agree to fp16 accuracy and are not the same numbers.
        if (hp_engine_dtype(engine.handle()) != HP_DTYPE_F32)
            return 11;
        hp::dnn::tensorrt half_engine(hp::dnn::builtin_model{ "lw_openpose_mobilenet", {}, 7 }, cv::Size(96, 80), 4, false, hp::data_type::kHALF);
        if (hp_engine_dtype(half_engine.handle()) != HP_DTYPE_F16)
            return 12;
        auto full = engine.inference({ batch[0] }), half = half_engine.inference({ batch[0] });
        const auto &a = full[0][0], &b = half[0][0];
        if (a.shape() != b.shape())
            return 13;
        size_t n = 1;
        for (int d : a.shape())
            n *= (size_t)d;
        const float *x = a.view<float>(), *y = b.view<float>();
        float scale = 0, err = 0;
        bool same = true;
        for (size_t k = 0; k < n; ++k) {
            scale = std::max(scale, std::fabs(x[k])), err = std::max(err, std::fabs(x[k] - y[k]));
            same = same && x[k] == y[k];
        }
        if (same || err > 2e-2f * scale + 1e-3f)
            return 14;
    }
    bool threw = false;
    try {
        engine.inference(std::vector<cv::Mat>(5, batch[0]));
    } catch (const std::logic_error&) {
        threw = true;
    }
    std::printf("OK %zu %zu %d\n", packets.size(), humans, (int)threw);
    return threw ? 0 : 4;
}
