// tests/cpp/cv_stream_shim.hpp — TEST INFRASTRUCTURE ONLY: what the reference's UNCHANGED include/hyperpose/stream/stream.hpp and
// src/stream.cpp need from <opencv2/opencv.hpp> beyond the value types of the mirror's cv_min.hpp (this image has no OpenCV):
// a host cv::resize for 8-bit BGR frames (forwarded to the oracle's restatement of OpenCV's fixed-point bilinear resize -
// oracle/resize_oracle.cpp; this is test code, the product resizes on the device), and cv::VideoCapture / cv::VideoWriter /
// cv::imwrite stand-ins: the capture is never opened, the writer keeps the frames it is handed so that the test can count them.
#pragma once
#include <hyperpose/utility/cv_min.hpp>

#include <string>
#include <vector>

namespace cv {

enum { CAP_PROP_POS_FRAMES = 1, CAP_PROP_FRAME_WIDTH = 3, CAP_PROP_FRAME_HEIGHT = 4, CAP_PROP_FPS = 5, CAP_PROP_FOURCC = 6, CAP_PROP_FRAME_COUNT = 7 };

class VideoCapture {
public:
    double get(int) const { return -1; }
    bool isOpened() const { return false; }
    VideoCapture& operator>>(Mat&) { return *this; }
};

class VideoWriter {
public:
    std::vector<Mat> frames;
    VideoWriter& operator<<(const Mat& m)
    {
        frames.push_back(m);
        return *this;
    }
};

void resize(const Mat& src, Mat& dst, Size size); // tests/cpp/reference_stream_unchanged.cpp
inline bool imwrite(const std::string&, const Mat&) { return true; }

} // namespace cv
