// tests/cpp/reference_stream_unchanged.cpp — the reference's stream operator, UNCHANGED, over this repo's engine and parser.
//
// the reference's `hyperpose::stream<Engine, Parser>` is a duck-typed template
// (include/hyperpose/stream/stream.hpp:119-319 + the out-of-line stages in src/stream.cpp, src/thread_pool.cpp): it needs
// `engine.input_size()`, `engine.max_batch_size()`, `engine.inference(std::vector<cv::Mat>) -> std::vector<internal_t>`, a copyable
// parser with `process(internal_t)` and the data types of utility/data.hpp / human.hpp.  tests/test_cpp_mirror.py builds this file
// against an include tree of SYMLINKS (nothing is copied into the repo):
//     hyperpose/stream/stream.hpp, hyperpose/utility/{thread_pool,thread_safe_queue,logging}.hpp   -> /root/reference/include/...
//     hyperpose/utility/{data,human,model,cv_min}.hpp, hyperpose/operator/, hp_hip.h                 -> this repo's include/
//     opencv2/opencv.hpp                                                                              -> tests/cpp/cv_stream_shim.hpp
// together with the reference's own src/stream.cpp, src/thread_pool.cpp and src/logging.cpp.  So the four reference threads (resize,
// inference, parse on the reference's thread pool with parser replicas, write) and their bounded queues run as shipped, and every
// engine / parser call they make lands in libhp_hip.so.  (The repo's own stream mirror keeps the frames on the device instead.)
#include <hyperpose/operator/dnn/tensorrt.hpp>
#include <hyperpose/operator/parser/paf.hpp>
#include <hyperpose/stream/stream.hpp> // the reference's header

#include <cstdio>
#include <cstring>

extern "C" void oracle_resize_linear_u8c3(const uint8_t* src, int sw, int sh, int sstep, uint8_t* dst, int dw, int dh, int dstep);

namespace cv {
void resize(const Mat& src, Mat& dst, Size size)
{
    Mat out(size.height, size.width, CV_8UC3);
    if (src.size() == size)
        std::memcpy(out.data(), src.data(), (size_t)size.area() * 3);
    else
        oracle_resize_linear_u8c3(src.data(), src.cols, src.rows, src.cols * 3, out.data(), size.width, size.height, size.width * 3);
    dst = out; // (src and dst may be the same object, src/stream.cpp:93)
}
} // namespace cv

int main(int argc, char** argv)
{
    if (argc < 2) {
        std::fprintf(stderr, "usage: %s model.onnx\n", argv[0]);
        return 2;
    }
    namespace hp = hyperpose;
    hp::dnn::tensorrt engine(hp::dnn::onnx{ argv[1] }, { 64, 48 }, 4, false);
    hp::parser::paf parser{};
    std::vector<cv::Mat> frames;
    unsigned s = 7;
    // 70 frames: the reference's read_from(const std::vector<cv::Mat>&) (src/stream.cpp:18-29) loops `while (distance(it, end) <=
    // step_size)` with step_size = capacity / 2 = 64 and reads past the end of a shorter vector (the condition is inverted), and throws
    // above the queue capacity of 128 - 65 .. 128 frames is the range in which the shipped code is well defined.
    for (int i = 0; i < 70; ++i) { // sizes other than the network's: the reference's resize stage has work to do
        cv::Mat m(60 + 4 * (i % 3), 90, CV_8UC3);
        for (size_t k = 0; k < m.total() * 3; ++k)
            s = s * 1664525u + 1013904223u, m.data()[k] = (uint8_t)(s >> 24);
        frames.push_back(m);
    }
    cv::VideoWriter writer;
    {
        auto stream = hp::make_stream(engine, parser, /*use_original_resolution=*/true, /*keep_ratio=*/false);
        stream.async() << frames;
        stream.sync() >> writer;
        std::printf("processed %zu\n", stream.processed_num());
    }
    bool sizes_ok = writer.frames.size() == frames.size();
    for (size_t i = 0; sizes_ok && i < frames.size(); ++i)
        sizes_ok = writer.frames[i].size() == frames[i].size();
    std::printf("%s %zu %d\n", sizes_ok ? "OK" : "FAIL", writer.frames.size(), (int)sizes_ok);
    return sizes_ok ? 0 : 1;
}
