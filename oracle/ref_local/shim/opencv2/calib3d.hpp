// Stand-in (see ../README.md): the two undistortion solvers the camera sources call, delegating to the oracle's restatements.
#ifndef SVGPU_SHIM_OPENCV_CALIB3D_HPP
#define SVGPU_SHIM_OPENCV_CALIB3D_HPP
#include "opencv2/core/mat.hpp"
namespace cv {
struct TermCriteria {
    enum { COUNT = 1, MAX_ITER = COUNT, EPS = 2 };
    int type, maxCount;
    double epsilon;
    TermCriteria(int t = 0, int n = 0, double e = 0) : type(t), maxCount(n), epsilon(e) {}
};
// points: n x 2 floats, in place or not; K: 3 x 3 float, D: 5 x 1 (k1 k2 p1 p2 k3) / 4 x 1 (fisheye) float; R empty, P = K
void undistortPoints(const Mat& src, Mat& dst, const Mat& K, const Mat& D, const Mat& R, const Mat& P, TermCriteria criteria = TermCriteria());
namespace fisheye {
void undistortPoints(const Mat& distorted, Mat& undistorted, const Mat& K, const Mat& D, const Mat& R = Mat(), const Mat& P = Mat(),
                     TermCriteria criteria = TermCriteria());
}
}  // namespace cv
#endif
