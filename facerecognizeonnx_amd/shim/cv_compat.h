// cv_compat.h — the handful of OpenCV value types the reference's public API mentions
// (cv::Mat as an image view, cv::Rect, cv::Point2f, cv::Size), for builds without OpenCV.
// When real OpenCV is available define FACEHIP_USE_OPENCV and <opencv2/core.hpp> is used instead,
// so reference callers (src/main.cpp) compile unchanged against face_detector.h / face_recognizer.h.
#pragma once
#if defined(FACEHIP_USE_OPENCV)
#include <opencv2/core.hpp>
#include <opencv2/imgcodecs.hpp>
#else
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>

extern "C" {                                          // include/facehip.h (libfacehip.so)
int fh_imread(const char* path, unsigned char** bgr, int* rows, int* cols);
void fh_image_free(unsigned char* bgr);
}

#ifndef CV_8UC3
#define CV_8UC3 16
#endif

namespace cv {

struct Point2f {
    float x = 0.f, y = 0.f;
    Point2f() = default;
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};

struct Size {
    int width = 0, height = 0;
    Size() = default;
    Size(int w, int h) : width(w), height(h) {}
};

struct Rect {
    int x = 0, y = 0, width = 0, height = 0;
    Rect() = default;
    Rect(int x_, int y_, int w, int h) : x(x_), y(y_), width(w), height(h) {}
};

// 8-bit, 3-channel, row-major image: either a view on caller memory or an owning buffer.
class Mat {
  public:
    int rows = 0, cols = 0;
    uint8_t* data = nullptr;
    size_t step = 0;                                  // bytes per row

    Mat() = default;
    Mat(int r, int c, int type, void* ptr, size_t step_bytes = 0)
        : rows(r), cols(c), data(static_cast<uint8_t*>(ptr)), step(step_bytes ? step_bytes : (size_t)c * 3) { (void)type; }
    Mat(int r, int c, int type) : rows(r), cols(c), step((size_t)c * 3) {
        (void)type;
        own_.reset(new uint8_t[(size_t)r * c * 3]());
        data = own_.get();
    }
    static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
    bool empty() const { return data == nullptr || rows <= 0 || cols <= 0; }
    int type() const { return CV_8UC3; }
    uint8_t* ptr(int r) { return data + (size_t)r * step; }
    const uint8_t* ptr(int r) const { return data + (size_t)r * step; }

  private:
    std::shared_ptr<uint8_t[]> own_;
};

// cv::imread(path) with its default flag (IMREAD_COLOR): 8-bit BGR, or an empty Mat when the file cannot be read /
// decoded.  Decoding is the library's own host code (csrc/image_io.cpp: JPEG, PNG, BMP, PPM).
inline Mat imread(const std::string& path) {
    unsigned char* p = nullptr;
    int r = 0, c = 0;
    if (fh_imread(path.c_str(), &p, &r, &c) != 0 || !p) return Mat();
    Mat m(r, c, CV_8UC3);
    std::memcpy(m.data, p, (size_t)r * c * 3);
    fh_image_free(p);
    return m;
}

}  // namespace cv
#endif
