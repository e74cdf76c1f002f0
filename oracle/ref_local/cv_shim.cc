// The six OpenCV functions the reference's extractor calls, for the stand-in headers of shim/: every one of them delegates to
// the oracle's restatement (oracle/orb_oracle.c, linked in), so the library built here differs from the oracle ONLY by running the
// reference's own code around them.  Test infrastructure only.
#include <stdexcept>
#include <vector>

#include "opencv2/core/mat.hpp"
#include "opencv2/features2d.hpp"
#include "opencv2/imgproc.hpp"

extern "C" {
float orc_fast_atan2(float y, float x);
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride);
void orc_gaussian_blur7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
int orc_fast9_16(const uint8_t* img, int cols, int rows, int step, int threshold, int nms, int* out_xys, int cap);
}

namespace cv {

float fastAtan2(float y, float x) { return orc_fast_atan2(y, x); }

void resize(const Mat& src, Mat& dst, Size dsize, double fx, double fy, int interpolation) {
    if (interpolation != INTER_LINEAR || fx != 0 || fy != 0) throw std::runtime_error("shim: only resize(.., dsize, 0, 0, INTER_LINEAR)");
    dst.create(dsize.height, dsize.width, CV_8UC1);
    orc_resize_linear_u8(src.data, src.cols, src.rows, (int)src.step, dst.data, dst.cols, dst.rows, (int)dst.step);
}

void GaussianBlur(const Mat& src, Mat& dst, Size ksize, double sigma_x, double sigma_y, int border) {
    if (ksize.width != 7 || ksize.height != 7 || sigma_x != 2 || sigma_y != 2 || border != BORDER_REFLECT_101)
        throw std::runtime_error("shim: only GaussianBlur(.., Size(7, 7), 2, 2, BORDER_REFLECT_101)");
    dst.create(src.rows, src.cols, CV_8UC1);
    orc_gaussian_blur7_u8(src.data, src.cols, src.rows, (int)src.step, dst.data, (int)dst.step);
}

// filled (thickness < 0) axis-aligned rectangle with both corners inclusive, clipped to the image
void rectangle(Mat& img, Point2i pt1, Point2i pt2, const Scalar& color, int thickness, int /*line_type*/) {
    if (thickness >= 0) throw std::runtime_error("shim: only filled rectangles");
    const int x0 = pt1.x < pt2.x ? pt1.x : pt2.x, x1 = pt1.x < pt2.x ? pt2.x : pt1.x;
    const int y0 = pt1.y < pt2.y ? pt1.y : pt2.y, y1 = pt1.y < pt2.y ? pt2.y : pt1.y;
    for (int y = y0 < 0 ? 0 : y0; y <= y1 && y < img.rows; ++y)
        for (int x = x0 < 0 ? 0 : x0; x <= x1 && x < img.cols; ++x) img.at<uchar>(y, x) = (uchar)color.val[0];
}

void FAST(const Mat& image, std::vector<KeyPoint>& keypoints, int threshold, bool nonmax_suppression) {
    keypoints.clear();
    if (image.rows < 7 || image.cols < 7) return;
    std::vector<int> xys(3 * (size_t)image.rows * image.cols);
    const int n = orc_fast9_16(image.data, image.cols, image.rows, (int)image.step, threshold, nonmax_suppression ? 1 : 0, xys.data(),
                               image.rows * image.cols);
    keypoints.reserve(n);
    for (int i = 0; i < n; ++i) keypoints.push_back(KeyPoint((float)xys[3 * i], (float)xys[3 * i + 1], 7.f, -1, (float)xys[3 * i + 2]));
}

}  // namespace cv
