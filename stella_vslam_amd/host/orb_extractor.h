// Drop-in C++ adaptor with the signature of stella_vslam::feature::orb_extractor
// (reference: src/stella_vslam/feature/orb_extractor.h:46-71, orb_params.h) on top of the C ABI (include/svgpu.h).
//
// tracking_module / system.cc keep calling
//     extractor->extract(img, mask, keypts, descriptors);
// and keep reading the public members orb_params_, mask_rects_, image_pyramid_ unchanged.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#ifdef SVGPU_WITH_OPENCV
#include <opencv2/core/mat.hpp>
#include <opencv2/core/types.hpp>
#else
#include "standin/cv_standin.h"
#endif

#include "svgpu.h"

namespace stella_vslam_hip {
namespace feature {

// feature/orb_params.h: same public members; tables from the reference's fp32 recurrences (orb_params.cc:41-71)
struct orb_params {
    orb_params() = delete;
    explicit orb_params(const std::string& name) : orb_params(name, 1.2f, 8, 20, 7) {}
    orb_params(const std::string& name, float scale_factor, unsigned int num_levels, unsigned int ini_fast_thr,
               unsigned int min_fast_thr);
    std::string name_;
    float scale_factor_;
    float log_scale_factor_;
    unsigned int num_levels_;
    unsigned int ini_fast_thr_;
    unsigned int min_fast_thr_;
    std::vector<float> scale_factors_, inv_scale_factors_, level_sigma_sq_, inv_level_sigma_sq_;
};

enum class descriptor_type { ORB, HASH_SIFT };  // feature/orb_extractor.h:18-21

class orb_extractor {
public:
    orb_extractor() = delete;
    orb_extractor(const orb_params* orb_params, unsigned int min_area, descriptor_type desc_type = descriptor_type::ORB,
                  const std::vector<std::vector<float>>& mask_rects = {}, int device = 0);
    virtual ~orb_extractor();
    orb_extractor(const orb_extractor&) = delete;
    orb_extractor& operator=(const orb_extractor&) = delete;

    //! Extract keypoints and each descriptor of them (orb_extractor.cc:28-136)
    void extract(const cv::_InputArray& in_image, const cv::_InputArray& in_image_mask, std::vector<cv::KeyPoint>& keypts,
                 const cv::_OutputArray& out_descriptors);

    const orb_params* orb_params_;
    std::vector<std::vector<float>> mask_rects_;
    //! Image pyramid of the last frame; levels >= 1 are downloaded from the GPU on demand by sync_image_pyramid()
    std::vector<cv::Mat> image_pyramid_;
    void sync_image_pyramid();

    svgpu_ctx* context() const { return ctx_; }

private:
    void configure(int cols, int rows);
    void create_rectangle_mask(unsigned int cols, unsigned int rows);  // orb_extractor.cc:138-151

    unsigned int min_area_;
    descriptor_type desc_type_;
    svgpu_ctx* ctx_ = nullptr;
    int cols_ = 0, rows_ = 0;
    bool mask_is_initialized_ = false;
    cv::Mat rect_mask_;
    std::vector<svgpu_keypoint> kp_buf_;
};

}  // namespace feature
}  // namespace stella_vslam_hip
