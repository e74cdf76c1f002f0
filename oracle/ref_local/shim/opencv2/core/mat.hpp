// Stand-in (see ../../README.md): 8-bit single-channel cv::Mat headers over reference-counted storage, with the view semantics
// the reference relies on (rowRange / colRange share the data; assigning Mat::zeros(...) to a header of the same size writes
// INTO its data, which is how orb_extractor.cc:105-106 clears a row range of the descriptor matrix).
#ifndef SVGPU_SHIM_OPENCV_MAT_HPP
#define SVGPU_SHIM_OPENCV_MAT_HPP
#include <cassert>
#include <cstring>
#include <memory>
#include "opencv2/core/types.hpp"
#define CV_8U 0
#define CV_8UC1 0
namespace cv {
struct MatZeros {
    int rows, cols;
};
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
    Mat(int r, int c, int /*type*/) { create(r, c, CV_8UC1); }
    Mat(int r, int c, int /*type*/, const Scalar& s) {
        create(r, c, CV_8UC1);
        memset(data, (int)s.val[0], (size_t)r * c);
    }
    Mat(int r, int c, int /*type*/, void* ext, size_t step_bytes) : rows(r), cols(c), data((uchar*)ext) { step.v = step_bytes; }  // user data, not owned
    void create(int r, int c, int /*type*/) {
        if (data && rows == r && cols == c) return;  // cv::Mat::create keeps a buffer of the right size
        store_.reset(new std::vector<uchar>((size_t)r * c));
        rows = r;
        cols = c;
        data = store_->data();
        step.v = (size_t)c;
    }
    static MatZeros zeros(int r, int c, int /*type*/) { return MatZeros{r, c}; }
    Mat& operator=(const MatZeros& z) {
        create(z.rows, z.cols, CV_8UC1);
        for (int y = 0; y < rows; ++y) memset(data + (size_t)y * step.v, 0, cols);
        return *this;
    }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return CV_8UC1; }
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
        m.data = data + a;
        m.cols = b - a;
        return m;
    }
    template <class T>
    T& at(int y, int x) { return *reinterpret_cast<T*>(data + (size_t)y * step.v + x); }
    template <class T>
    const T& at(int y, int x) const { return *reinterpret_cast<const T*>(data + (size_t)y * step.v + x); }
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
};
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
