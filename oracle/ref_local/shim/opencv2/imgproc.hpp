// Stand-in (see ../README.md): the three imgproc calls of orb_extractor.cc, delegating to the oracle's restatements.
#ifndef SVGPU_SHIM_OPENCV_IMGPROC_HPP
#define SVGPU_SHIM_OPENCV_IMGPROC_HPP
#include "opencv2/core/mat.hpp"
namespace cv {
enum { INTER_LINEAR = 1, BORDER_REFLECT_101 = 4, LINE_AA = 16 };
void resize(const Mat& src, Mat& dst, Size dsize, double fx, double fy, int interpolation);
void GaussianBlur(const Mat& src, Mat& dst, Size ksize, double sigma_x, double sigma_y, int border);
void rectangle(Mat& img, Point2i pt1, Point2i pt2, const Scalar& color, int thickness, int line_type);
}  // namespace cv
#endif
