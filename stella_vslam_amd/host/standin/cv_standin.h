// Minimal stand-ins for the few OpenCV types the adaptor signatures mention, used ONLY when the real
// OpenCV headers are not available (this container).  With OpenCV present, compile with
// -DSVGPU_WITH_OPENCV and the adaptors use <opencv2/core.hpp> instead; layouts are identical where it
// matters (cv::KeyPoint is 28 bytes: pt.x pt.y size angle response octave class_id).
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#define CV_8U 0    // macros, as in OpenCV's cvdef.h: the adaptors spell the depth the way code written against OpenCV does
#define CV_8UC1 0
namespace cv {


struct Point2f {
    float x = 0, y = 0;
};

struct KeyPoint {
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

// 2-D 8-bit matrix view/owner: rows x cols, `step` bytes per row
class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    uint8_t* data = nullptr;
    Mat() = default;
    Mat(int r, int c, int /*type*/) { create(r, c, CV_8U); }
    Mat(int r, int c, int /*type*/, void* ext, size_t stp) : rows(r), cols(c), step(stp), data((uint8_t*)ext) {}
    void create(int r, int c, int /*type*/) {
        if (r == rows && c == cols && own_) return;
        own_ = std::shared_ptr<uint8_t>(new uint8_t[(size_t)r * c], std::default_delete<uint8_t[]>());
        rows = r;
        cols = c;
        step = (size_t)c;
        data = own_.get();
    }
    void release() {
        own_.reset();
        rows = cols = 0;
        step = 0;
        data = nullptr;
    }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return step == (size_t)cols; }
    int type() const { return CV_8UC1; }
    uint8_t* ptr(int r = 0) { return data + (size_t)r * step; }
    const uint8_t* ptr(int r = 0) const { return data + (size_t)r * step; }
    Mat row(int r) const { return Mat(1, cols, CV_8U, data + (size_t)r * step, step); }

private:
    std::shared_ptr<uint8_t> own_;
};

// The reference passes images as cv::_InputArray / cv::_OutputArray proxies; a Mat reference is enough here.
class _InputArray {
public:
    _InputArray() = default;
    _InputArray(const Mat& m) : m_(&m) {}
    bool empty() const { return m_ == nullptr || m_->empty(); }
    Mat getMat() const { return m_ ? *m_ : Mat(); }

private:
    const Mat* m_ = nullptr;
};
class _OutputArray {
public:
    _OutputArray(Mat& m) : m_(&m) {}
    void create(int r, int c, int t) const { m_->create(r, c, t); }
    void release() const { m_->release(); }
    Mat getMat() const { return *m_; }
    Mat& getMatRef() const { return *m_; }

private:
    Mat* m_;
};
using InputArray = const _InputArray&;
using OutputArray = const _OutputArray&;

}  // namespace cv
