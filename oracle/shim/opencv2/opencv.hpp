// oracle/shim/opencv2/opencv.hpp — TEST INFRASTRUCTURE ONLY.
//
// Type-only stand-in for the handful of OpenCV value types that the reference's public headers
// mention (include/hyperpose/utility/{data,human}.hpp, operator/parser/proposal_network.hpp), so that
// the reference's src/pose_proposal.cpp, src/pifpaf.cpp and src/pifpaf_decoder/*.cpp can be compiled
// verbatim FROM /root/reference into oracle/_ref/ (see oracle/Makefile).  No OpenCV arithmetic is
// provided: none of those three translation units calls any.
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

class Mat; // opaque: only named in declarations that the oracle never calls

} // namespace cv
