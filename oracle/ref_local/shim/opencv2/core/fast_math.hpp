// Stand-in (see ../../README.md): the documented roundings of OpenCV core/fast_math.hpp.
#ifndef SVGPU_SHIM_OPENCV_FAST_MATH_HPP
#define SVGPU_SHIM_OPENCV_FAST_MATH_HPP
#include <cmath>
static inline int cvFloor(float value) {
    const int i = (int)value;
    return i - (i > value);
}
static inline int cvFloor(double value) {
    const int i = (int)value;
    return i - (i > value);
}
static inline int cvCeil(double value) {
    const int i = (int)value;
    return i + (i < value);
}
// round to nearest, ties to even (cvtss2si / lrint in the default rounding mode)
static inline int cvRound(float value) { return (int)lrintf(value); }
static inline int cvRound(double value) { return (int)lrint(value); }
static inline int cvRound(int value) { return value; }
#endif
