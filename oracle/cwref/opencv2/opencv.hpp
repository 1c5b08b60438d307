// ORACLE build shim (test infrastructure; NOT OpenCV and not part of the product).
//
// The reference's only implementation of Cluster-Weighted NMS is its C++ edge demo
// (examples/YOLO-Master-Cross-Platform-Edge-Deployment/cpp/src/common.cpp:128-215), which includes
// <opencv2/opencv.hpp> but, on the decode + NMS path, uses nothing beyond cv::Rect_ arithmetic.  OpenCV is not in
// this image, so oracle/cwref/build.py compiles that source file IN PLACE against this header: the geometry types
// below follow OpenCV's documented semantics (Rect_ x/y/width/height, area(), operator& = intersection, empty
// when the boxes do not overlap); the image / drawing entry points the same file references elsewhere
// (resize, rectangle, putText, Mat) are inert stand-ins that are never executed by the oracle.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <string>
#include <vector>

#define CV_8UC4 24

inline int cvRound(double v) { return (int)std::lrint(v); }

namespace cv {

template <typename T>
struct Point_ {
    T x{}, y{};
    Point_() = default;
    Point_(T x_, T y_) : x(x_), y(y_) {}
};
typedef Point_<int> Point;

template <typename T>
struct Size_ {
    T width{}, height{};
    Size_() = default;
    Size_(T w, T h) : width(w), height(h) {}
};
typedef Size_<int> Size;

template <typename T>
struct Rect_ {
    T x{}, y{}, width{}, height{};
    Rect_() = default;
    Rect_(T x_, T y_, T w, T h) : x(x_), y(y_), width(w), height(h) {}
    T area() const { return width * height; }
};
// intersection, as cv::Rect_::operator& : the overlap rectangle, or an all-zero rectangle when there is none
template <typename T>
inline Rect_<T> operator&(const Rect_<T>& a, const Rect_<T>& b) {
    const T x1 = std::max(a.x, b.x), y1 = std::max(a.y, b.y);
    const T w = std::min(a.x + a.width, b.x + b.width) - x1, h = std::min(a.y + a.height, b.y + b.height) - y1;
    if (w <= 0 || h <= 0) return Rect_<T>();
    return Rect_<T>(x1, y1, w, h);
}
typedef Rect_<int> Rect;
typedef Rect_<float> Rect2f;
typedef Rect_<double> Rect2d;

struct Scalar {
    double v[4];
    Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : v{a, b, c, d} {}
};

// inert image stand-in: the oracle never touches pixels
struct Mat {
    int rows = 0, cols = 0, type_ = 0;
    std::vector<uint8_t> buf;
    Mat() = default;
    Mat(int r, int c, int t, const Scalar& = Scalar()) : rows(r), cols(c), type_(t), buf((size_t)std::max(r, 0) * std::max(c, 0) * 4) {}
    int type() const { return type_; }
    bool empty() const { return rows == 0 || cols == 0; }
    Mat operator()(const Rect&) const { return *this; }
    void copyTo(Mat) const {}
    template <typename T> T* ptr(int r = 0) { return reinterpret_cast<T*>(buf.data() + (size_t)r * cols * 4); }
};

enum { FONT_HERSHEY_SIMPLEX = 0, FILLED = -1 };
inline void resize(const Mat&, Mat& dst, Size s) { dst = Mat(s.height, s.width, 0); }
inline void rectangle(Mat&, Rect, const Scalar&, int = 1) {}
inline Size getTextSize(const std::string&, int, double, int, int* base) { if (base) *base = 0; return Size(0, 0); }
inline void putText(Mat&, const std::string&, Point, int, double, const Scalar&, int = 1) {}

}  // namespace cv
