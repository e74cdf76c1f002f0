// Stand-in (see ../README.md): match/stereo.cc includes the umbrella header.
#ifndef SVGPU_SHIM_OPENCV_CORE_HPP
#define SVGPU_SHIM_OPENCV_CORE_HPP
#include <limits>
#include "opencv2/core/mat.hpp"
#endif
