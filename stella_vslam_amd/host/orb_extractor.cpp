#include "orb_extractor.h"

#include <cmath>
#include <cstring>

namespace stella_vslam_hip {
namespace feature {

namespace {
void check(svgpu_ctx* ctx, int rc, const char* where) {
    if (rc != SVGPU_OK) throw std::runtime_error(std::string(where) + ": " + svgpu_status_string(rc) + " (" + svgpu_last_error(ctx) + ")");
}
}  // namespace

orb_params::orb_params(const std::string& name, float scale_factor, unsigned int num_levels, unsigned int ini_fast_thr,
                       unsigned int min_fast_thr)
    : name_(name), scale_factor_(scale_factor), log_scale_factor_(std::log(scale_factor)), num_levels_(num_levels),
      ini_fast_thr_(ini_fast_thr), min_fast_thr_(min_fast_thr), scale_factors_(num_levels), inv_scale_factors_(num_levels),
      level_sigma_sq_(num_levels), inv_level_sigma_sq_(num_levels) {
    if (svgpu_orb_scale_tables(scale_factor, (int)num_levels, scale_factors_.data(), inv_scale_factors_.data(), level_sigma_sq_.data(),
                               inv_level_sigma_sq_.data())
        != SVGPU_OK)
        throw std::runtime_error("orb_params: invalid number of levels");
}

orb_extractor::orb_extractor(const orb_params* orb_params, unsigned int min_area, descriptor_type desc_type,
                             const std::vector<std::vector<float>>& mask_rects, int device)
    : orb_params_(orb_params), mask_rects_(mask_rects), min_area_(min_area), desc_type_(desc_type) {
    image_pyramid_.resize(orb_params_->num_levels_);
    if (desc_type_ != descriptor_type::ORB) throw std::runtime_error("Invalid descriptor_type");  // orb_extractor.cc:117-125
    const int rc = svgpu_create(device, &ctx_);
    if (rc != SVGPU_OK) throw std::runtime_error(std::string("svgpu_create: ") + svgpu_status_string(rc));  // no CPU fallback
}

orb_extractor::~orb_extractor() { svgpu_destroy(ctx_); }

void orb_extractor::configure(int cols, int rows) {
    if (cols == cols_ && rows == rows_) return;
    check(ctx_, svgpu_orb_configure(ctx_, cols, rows, 1, orb_params_->scale_factor_, (int)orb_params_->num_levels_,
                                    (int)orb_params_->ini_fast_thr_, (int)orb_params_->min_fast_thr_, min_area_),
          "svgpu_orb_configure");
    cols_ = cols;
    rows_ = rows;
    kp_buf_.resize((size_t)std::max(svgpu_orb_max_keypoints(ctx_), 1));
    mask_is_initialized_ = false;
    rect_mask_.release();
}

void orb_extractor::create_rectangle_mask(unsigned int cols, unsigned int rows) {
    if (rect_mask_.empty()) {
        rect_mask_.create((int)rows, (int)cols, CV_8UC1);
        for (unsigned y = 0; y < rows; ++y) std::memset(rect_mask_.ptr((int)y), 255, cols);
    }
    for (const auto& r : mask_rects_) {  // cv::rectangle(..., -1): both corner points are inside the filled area
        const unsigned x_min = (unsigned)std::round(cols * r.at(0)), x_max = (unsigned)std::round(cols * r.at(1));
        const unsigned y_min = (unsigned)std::round(rows * r.at(2)), y_max = (unsigned)std::round(rows * r.at(3));
        for (unsigned y = y_min; y <= y_max && y < rows; ++y)
            for (unsigned x = x_min; x <= x_max && x < cols; ++x) rect_mask_.ptr((int)y)[x] = 0;
    }
}

void orb_extractor::extract(const cv::_InputArray& in_image, const cv::_InputArray& in_image_mask, std::vector<cv::KeyPoint>& keypts,
                            const cv::_OutputArray& out_descriptors) {
    if (in_image.empty()) return;  // orb_extractor.cc:30-32
    const cv::Mat image = in_image.getMat();
    configure(image.cols, image.rows);
    image_pyramid_.at(0) = image;  // level 0 aliases the caller's image (orb_extractor.cc:154)
    if (!mask_is_initialized_ && !mask_rects_.empty()) {
        create_rectangle_mask((unsigned)image.cols, (unsigned)image.rows);
        mask_is_initialized_ = true;
    }
    const uint8_t* mask = nullptr;
    int mask_stride = 0;
    cv::Mat image_mask;
    if (!in_image_mask.empty()) {  // the image mask wins over the rectangle mask (orb_extractor.cc:49-63)
        image_mask = in_image_mask.getMat();
        mask = image_mask.ptr();
        mask_stride = (int)image_mask.step;
    }
    else if (!rect_mask_.empty()) {
        mask = rect_mask_.ptr();
        mask_stride = (int)rect_mask_.step;
    }
    const int cap = (int)kp_buf_.size();
    std::vector<uint8_t> desc((size_t)cap * 32);
    int n = 0;
    check(ctx_, svgpu_orb_extract(ctx_, image.ptr(), (int)image.step, mask, mask_stride, kp_buf_.data(), desc.data(), cap, &n, nullptr),
          "svgpu_orb_extract");
    keypts.clear();
    if (n == 0) {
        out_descriptors.release();
        return;
    }
    out_descriptors.create(n, 32, CV_8U);
    cv::Mat d = out_descriptors.getMat();
    for (int i = 0; i < n; ++i) std::memcpy(d.ptr(i), desc.data() + (size_t)i * 32, 32);
    keypts.resize((size_t)n);
    static_assert(sizeof(cv::KeyPoint) == sizeof(svgpu_keypoint), "KeyPoint layout");
    std::memcpy((void*)keypts.data(), kp_buf_.data(), (size_t)n * sizeof(svgpu_keypoint));
}

void orb_extractor::sync_image_pyramid() {
    for (unsigned l = 1; l < orb_params_->num_levels_; ++l) {
        int w = 0, h = 0;
        check(ctx_, svgpu_orb_level_size(ctx_, (int)l, &w, &h), "svgpu_orb_level_size");
        image_pyramid_.at(l).create(h, w, CV_8UC1);
        check(ctx_, svgpu_orb_pyramid_download(ctx_, 0, (int)l, image_pyramid_.at(l).ptr(), (int)image_pyramid_.at(l).step),
              "svgpu_orb_pyramid_download");
    }
}

}  // namespace feature
}  // namespace stella_vslam_hip
