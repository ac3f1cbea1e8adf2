// include/hyperpose/utility/cv_min.hpp — the few OpenCV value types HyperPose's public headers mention
// (cv::Size, cv::Rect, cv::Mat as an 8-bit HWC image view), for builds WITHOUT OpenCV (this image has none).
// Define HYPERPOSE_USE_OPENCV to use the real <opencv2/opencv.hpp> instead.  No OpenCV arithmetic is needed by
// the hot path: frames are expected network-sized (cv::resize with equal sizes is a copy).
#pragma once
#ifdef HYPERPOSE_USE_OPENCV
#include <opencv2/opencv.hpp>
#else
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

constexpr int CV_8UC3 = 16; // (a macro at global scope in OpenCV)
namespace cv {
struct Size {
    int width = 0, height = 0;
    Size() = default;
    Size(int w, int h) : width(w), height(h) {}
    int area() const { return width * height; }
    bool operator==(const Size& o) const { return width == o.width && height == o.height; }
    bool operator!=(const Size& o) const { return !(*this == o); }
};
// Minimal continuous 8-bit 3-channel image (rows x cols x 3, BGR), shared ownership like cv::Mat.
class Mat {
public:
    int rows = 0, cols = 0;
    Mat() = default;
    Mat(int r, int c, int type = CV_8UC3) : rows(r), cols(c), m_data(new uint8_t[(size_t)r * c * 3](), std::default_delete<uint8_t[]>()) { (void)type; }
    Mat(int r, int c, int type, void* external) : rows(r), cols(c), m_ext((uint8_t*)external) { (void)type; }
    Size size() const { return Size(cols, rows); }
    int type() const { return CV_8UC3; }
    bool isContinuous() const { return true; }
    size_t total() const { return (size_t)rows * cols; }
    bool empty() const { return rows == 0 || cols == 0; }
    uint8_t* data() { return m_ext ? m_ext : m_data.get(); }
    const uint8_t* data() const { return m_ext ? m_ext : m_data.get(); }
private:
    std::shared_ptr<uint8_t> m_data;
    uint8_t* m_ext = nullptr;
};
} // namespace cv
#endif
