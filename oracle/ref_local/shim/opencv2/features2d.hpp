// Stand-in (see ../README.md): cv::FAST (TYPE_9_16), delegating to the oracle's restatement.
#ifndef SVGPU_SHIM_OPENCV_FEATURES2D_HPP
#define SVGPU_SHIM_OPENCV_FEATURES2D_HPP
#include <vector>
#include "opencv2/core/mat.hpp"
namespace cv {
void FAST(const Mat& image, std::vector<KeyPoint>& keypoints, int threshold, bool nonmax_suppression);
}
#endif
