// Stand-in (see ../../README.md): single-channel cv::Mat headers (CV_8U, and CV_32F for the 11 x 11 patches of match/stereo.cc) over
// reference-counted storage, with the view semantics the reference relies on (rowRange / colRange share the data; assigning
// Mat::zeros(...) to a header of the same size writes INTO its data, which is how orb_extractor.cc:105-106 clears a row range of the
// descriptor matrix; convertTo to another depth re-allocates, which is how stereo.cc:208 turns a view into its own float patch).
#ifndef SVGPU_SHIM_OPENCV_MAT_HPP
#define SVGPU_SHIM_OPENCV_MAT_HPP
#include <cassert>
#include <cstring>
#include <memory>
#include "opencv2/core/types.hpp"
#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
namespace cv {
struct MatZeros {
    int rows, cols;
};
struct MatConst {  // alpha * Mat::ones(rows, cols, CV_32F)
    int rows, cols;
    double alpha;
};
static inline MatConst operator*(double s, const MatConst& m) { return MatConst{m.rows, m.cols, s * m.alpha}; }
class Mat {
public:
    struct Step {
        size_t v = 0;
        operator size_t() const { return v; }
    };
    int rows = 0, cols = 0;
    uchar* data = nullptr;
    Step step;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int /*type*/, const Scalar& s) {
        create(r, c, CV_8UC1);
        memset(data, (int)s.val[0], (size_t)r * c);
    }
    Mat(int r, int c, int type, void* ext, size_t step_bytes) : rows(r), cols(c), data((uchar*)ext), esz_(type == CV_32F ? 4 : 1) { step.v = step_bytes; }  // user data, not owned
    void create(int r, int c, int type) {
        const int esz = type == CV_32F ? 4 : 1;
        if (data && rows == r && cols == c && esz == esz_) return;  // cv::Mat::create keeps a buffer of the right size and type
        store_.reset(new std::vector<uchar>((size_t)r * c * esz));
        rows = r;
        cols = c;
        esz_ = esz;
        data = store_->data();
        step.v = (size_t)c * esz;
    }
    static MatZeros zeros(int r, int c, int /*type*/) { return MatZeros{r, c}; }
    static MatConst ones(int r, int c, int /*type*/) { return MatConst{r, c, 1.0}; }
    void convertTo(Mat& dst, int rtype) const {  // 8U -> 32F (dst may be *this: the result is built first)
        Mat tmp;
        tmp.create(rows, cols, rtype);
        for (int y = 0; y < rows; ++y)
            for (int x = 0; x < cols; ++x) {
                const float v = esz_ == 4 ? at<float>(y, x) : (float)at<uchar>(y, x);
                if (rtype == CV_32F) tmp.at<float>(y, x) = v;
                else tmp.at<uchar>(y, x) = (uchar)v;
            }
        dst = tmp;
    }
    Mat& operator-=(const MatConst& e) {  // float matrices only: every element minus saturate_cast<float>(alpha)
        const float a = (float)e.alpha;
        for (int y = 0; y < rows; ++y)
            for (int x = 0; x < cols; ++x) at<float>(y, x) -= a;
        return *this;
    }
    int depth() const { return esz_ == 4 ? CV_32F : CV_8U; }
    Mat reshape(int /*cn*/) const { return *this; }  // n x 2 one-channel <-> n x 1 two-channel: the same bytes, which is all the callers use
    Mat& operator=(const MatZeros& z) {
        create(z.rows, z.cols, CV_8UC1);
        for (int y = 0; y < rows; ++y) memset(data + (size_t)y * step.v, 0, cols);
        return *this;
    }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return step.v == (size_t)cols * esz_; }
    int type() const { return depth(); }
    size_t step1() const { return step.v; }
    void release() {
        store_.reset();
        data = nullptr;
        rows = cols = 0;
    }
    Mat rowRange(int a, int b) const {
        Mat m(*this);
        m.data = data + (size_t)a * step.v;
        m.rows = b - a;
        return m;
    }
    Mat colRange(int a, int b) const {
        Mat m(*this);
        m.data = data + (size_t)a * esz_;
        m.cols = b - a;
        return m;
    }
    template <class T>
    T& at(int y, int x) { return *reinterpret_cast<T*>(data + (size_t)y * step.v + (size_t)x * sizeof(T)); }
    template <class T>
    const T& at(int y, int x) const { return *reinterpret_cast<const T*>(data + (size_t)y * step.v + (size_t)x * sizeof(T)); }
    uchar* ptr(int y = 0) { return data + (size_t)y * step.v; }
    const uchar* ptr(int y = 0) const { return data + (size_t)y * step.v; }
    template <class T>
    T* ptr(int y = 0) { return reinterpret_cast<T*>(data + (size_t)y * step.v); }
    template <class T>
    const T* ptr(int y = 0) const { return reinterpret_cast<const T*>(data + (size_t)y * step.v); }
    Mat row(int y) const { return rowRange(y, y + 1); }
    Mat clone() const {
        Mat m;
        m.create(rows, cols, CV_8UC1);
        for (int y = 0; y < rows; ++y) memcpy(m.data + (size_t)y * m.step.v, data + (size_t)y * step.v, cols);
        return m;
    }

private:
    std::shared_ptr<std::vector<uchar>> store_;
    int esz_ = 1;
};
// (cv::Mat_<float>(r, c) << a, b, ...): row-major comma initialiser of a float matrix
template <class T>
class Mat_ : public Mat {
public:
    Mat_(int r, int c) { create(r, c, CV_32F); }
    struct Comma {
        Mat_& m;
        int k;
        Comma& operator,(double x) {
            m.template at<T>(k / m.cols, k % m.cols) = (T)x;
            ++k;
            return *this;
        }
        operator Mat() const { return m; }
    };
    Comma operator<<(double x) {
        this->template at<T>(0, 0) = (T)x;
        return Comma{*this, 1};
    }
};
enum { NORM_L1 = 2 };
static inline double norm(const Mat& a, const Mat& b, int /*NORM_L1*/) {  // float matrices; the sum is accumulated in double as cv::norm does
    double s = 0.0;
    for (int y = 0; y < a.rows; ++y)
        for (int x = 0; x < a.cols; ++x) {
            const float d = a.at<float>(y, x) - b.at<float>(y, x);
            s += (double)(d < 0 ? -d : d);
        }
    return s;
}
class _InputArray {
public:
    _InputArray() {}
    _InputArray(const Mat& m) : m_(m) {}
    bool empty() const { return m_.empty(); }
    Mat getMat() const { return m_; }

protected:
    mutable Mat m_;
};
class _OutputArray : public _InputArray {
public:
    _OutputArray(Mat& m) : _InputArray(m), out_(&m) {}
    void create(int r, int c, int type) const {
        out_->create(r, c, type);
        m_ = *out_;
    }
    void release() const {
        out_->release();
        m_ = *out_;
    }

private:
    Mat* out_;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
}  // namespace cv
#endif
