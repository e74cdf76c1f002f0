// cv::undistortPoints / cv::fisheye::undistortPoints for the stand-in headers of shim/: both delegate to the oracle's restatements
// (oracle/frame_oracle.c), so the camera library built here differs from the oracle only by running the reference's own camera code
// around them.  Test infrastructure only.
#include <cstring>
#include <vector>

#include "opencv2/calib3d.hpp"

extern "C" {
typedef struct {
    int32_t model;
    int32_t pad_;
    double cols, rows;
    double fx, fy, cx, cy;
    double dist[5];
    double focal_x_baseline;
    float min_x, max_x, min_y, max_y;
} orc_camera;
void orc_undistort_keypoints(const orc_camera* c, int n, const float* xy, float* out);
}

namespace {
void run(int model, const cv::Mat& src, cv::Mat& dst, const cv::Mat& K, const cv::Mat& D, int nd) {
    orc_camera c;
    memset(&c, 0, sizeof(c));
    c.model = model;
    c.fx = K.at<float>(0, 0), c.fy = K.at<float>(1, 1), c.cx = K.at<float>(0, 2), c.cy = K.at<float>(1, 2);
    for (int i = 0; i < nd; ++i) c.dist[i] = D.at<float>(i, 0);
    const int n = src.rows;  // n x 2 floats (one channel) or n x 1 (two channels): the same bytes
    std::vector<float> in(2 * (size_t)n), out(2 * (size_t)n);
    for (int i = 0; i < n; ++i) in[2 * i] = src.at<float>(i, 0), in[2 * i + 1] = src.at<float>(i, 1);
    orc_undistort_keypoints(&c, n, in.data(), out.data());
    if (dst.data != src.data) dst.create(n, 2, CV_32F);
    for (int i = 0; i < n; ++i) dst.at<float>(i, 0) = out[2 * i], dst.at<float>(i, 1) = out[2 * i + 1];
}
}  // namespace

namespace cv {
void undistortPoints(const Mat& src, Mat& dst, const Mat& K, const Mat& D, const Mat&, const Mat&, TermCriteria) { run(0, src, dst, K, D, 5); }
namespace fisheye {
void undistortPoints(const Mat& src, Mat& dst, const Mat& K, const Mat& D, const Mat&, const Mat&, TermCriteria) { run(1, src, dst, K, D, 4); }
}  // namespace fisheye
}  // namespace cv
