// oracle/shim/opencv2/opencv.hpp — TEST INFRASTRUCTURE ONLY.
//
// Type-only stand-in for the handful of OpenCV value types that the reference's public headers
// mention (include/hyperpose/utility/{data,human}.hpp, operator/parser/proposal_network.hpp), so that
// the reference's src/pose_proposal.cpp, src/pifpaf.cpp and src/pifpaf_decoder/*.cpp can be compiled
// verbatim FROM /root/reference into oracle/_ref/ (see oracle/Makefile).  None of those three translation units
// calls any OpenCV arithmetic.
//
// src/paf.cpp + src/post_process.hpp (also compiled verbatim, oracle/ref_paf_wrap.cpp) make exactly two arithmetic
// OpenCV calls on single-channel float images: cv::resize(..., INTER_AREA) (post_process.hpp:50) and
// cv::GaussianBlur(k x k, sigma) (post_process.hpp:66).  OpenCV 4.4.0 (Dockerfile:32) is not in this image; the
// two calls are declared here and FORWARD to the restatements of their published algorithms in
// oracle/paf_oracle.cpp (oracle_resize_area_1ch / oracle_gaussian_blur_1ch) - the only part of the PAF path
// that is not reference code ("parity unpinned" for these two calls only; tests/test_paf_envelope.py bounds
// what a real OpenCV build could change).
//
// cv::Rect::operator& follows OpenCV's documented semantics (core/types.hpp Rect_<T>& operator&=):
// intersection; an empty intersection yields the all-zero rectangle.
#pragma once
#include <algorithm>
#include <array>
#include <cassert>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <ostream>
#include <string>
#include <vector>

namespace cv {

struct Size {
    int width = 0, height = 0;
    Size() = default;
    Size(int w, int h)
        : width(w)
        , height(h)
    {
    }
    int area() const { return width * height; }
    bool operator==(const Size& o) const { return width == o.width && height == o.height; }
    bool operator!=(const Size& o) const { return !(*this == o); }
};

struct Point {
    int x = 0, y = 0;
    Point() = default;
    Point(int x_, int y_)
        : x(x_)
        , y(y_)
    {
    }
};

struct Rect {
    int x = 0, y = 0, width = 0, height = 0;
    Rect() = default;
    Rect(int x_, int y_, int w_, int h_)
        : x(x_)
        , y(y_)
        , width(w_)
        , height(h_)
    {
    }
    int area() const { return width * height; }
};

inline Rect operator&(const Rect& a, const Rect& b)
{
    Rect r;
    const int x1 = std::max(a.x, b.x);
    const int y1 = std::max(a.y, b.y);
    r.width = std::min(a.x + a.width, b.x + b.width) - x1;
    r.height = std::min(a.y + a.height, b.y + b.height) - y1;
    r.x = x1;
    r.y = y1;
    if (r.width <= 0 || r.height <= 0)
        r = Rect();
    return r;
}

struct Scalar {
    double val[4] = { 0, 0, 0, 0 };
    Scalar() = default;
    Scalar(double a, double b = 0, double c = 0, double d = 0)
        : val{ a, b, c, d }
    {
    }
};

// A borrowed single-plane image: what `cv::Mat(size, type, data)` (user-allocated data, no copy) is to the two
// calls below.  Everything else in the reference only names cv::Mat in declarations the oracle never calls.
template <typename T>
struct DataType;
template <>
struct DataType<float> {
    static constexpr int type = 5; // CV_32FC1
};

class Mat {
public:
    Mat() = default;
    Mat(Size size, int type, void* data)
        : m_size(size)
        , m_type(type)
        , m_data(data)
    {
    }
    Size size() const { return m_size; }
    int type() const { return m_type; }
    void* ptr() const { return m_data; }

private:
    Size m_size;
    int m_type = -1;
    void* m_data = nullptr;
};

enum InterpolationFlags { INTER_NEAREST = 0, INTER_LINEAR = 1, INTER_CUBIC = 2, INTER_AREA = 3 };

// defined in oracle/ref_paf_wrap.cpp (forwarders to oracle/paf_oracle.cpp)
void resize(const Mat& src, Mat& dst, Size dsize, double fx = 0, double fy = 0, int interpolation = INTER_LINEAR);
void GaussianBlur(const Mat& src, Mat& dst, Size ksize, double sigmaX, double sigmaY = 0, int borderType = 4);

} // namespace cv
