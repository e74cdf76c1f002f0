// Stand-in (see ../../README.md): the OpenCV value types the reference's extractor touches.
#ifndef SVGPU_SHIM_OPENCV_TYPES_HPP
#define SVGPU_SHIM_OPENCV_TYPES_HPP
// the standard headers OpenCV's own headers pull in, which the reference relies on
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>
#include "opencv2/core/fast_math.hpp"
typedef unsigned char uchar;
namespace cv {
class Mat;
template <class T>
struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T x_, T y_) : x(x_), y(y_) {}
};
template <class T>
static inline Point_<T>& operator*=(Point_<T>& a, float b) {  // types.hpp: saturate_cast<T>(a.x * b)
    a.x = (T)(a.x * b);
    a.y = (T)(a.y * b);
    return a;
}
typedef Point_<float> Point2f;
typedef Point_<int> Point2i;
typedef Point2i Point;
struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
};
struct Scalar {
    double val[4];
    Scalar(double v0 = 0) : val{v0, 0, 0, 0} {}
};
struct KeyPoint {  // 28 bytes, the layout of cv::KeyPoint
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
    KeyPoint() : pt(0, 0), size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(float x, float y, float size_, float angle_ = -1, float response_ = 0, int octave_ = 0, int class_id_ = -1)
        : pt(x, y), size(size_), angle(angle_), response(response_), octave(octave_), class_id(class_id_) {}
};
float fastAtan2(float y, float x);  // -> the oracle's restatement (oracle/orb_oracle.c)
}  // namespace cv
#endif
