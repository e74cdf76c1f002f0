// Stand-in (see ../../README.md) for data/landmark.h: plain data + the accessors the matcher sources call.  predict_scale_level restates
// data/landmark.cc:336-353 (the oracle's can_observe does the same and is compared with the device).
#ifndef SVGPU_SHIM_STELLA_DATA_LANDMARK_H
#define SVGPU_SHIM_STELLA_DATA_LANDMARK_H
#include <atomic>
#include <memory>
#include <opencv2/core/mat.hpp>
#include "stella_vslam/type.h"
namespace stella_vslam {
namespace data {
class keyframe;
class landmark : public std::enable_shared_from_this<landmark> {
public:
    using observations_t = std::map<std::weak_ptr<keyframe>, unsigned int, id_less<std::weak_ptr<keyframe>>>;
    landmark(unsigned int id, const Vec3_t& pos_w) : id_(id), pos_w_(pos_w) {}
    Vec3_t get_pos_in_world() const { return pos_w_; }
    Vec3_t get_obs_mean_normal() const { return mean_normal_; }
    float get_min_valid_distance() const { return min_valid_dist_; }
    float get_max_valid_distance() const { return max_valid_dist_; }
    cv::Mat get_descriptor() const { return descriptor_; }
    bool has_observation() const { return has_observation_; }
    unsigned int num_observations() const { return has_observation_ ? 1 : 0; }
    bool will_be_erased() { return will_be_erased_; }
    bool is_observed_in_keyframe(const std::shared_ptr<keyframe>& keyfrm) const;
    int get_index_in_keyframe(const std::shared_ptr<keyframe>& keyfrm) const;
    // data/landmark.h:88-92
    bool is_inside_in_orb_scale(const float cam_to_lm_dist, const float margin_far, const float margin_near) const {
        const float max_dist = margin_far * max_valid_dist_;
        const float min_dist = margin_near * min_valid_dist_;
        return (min_dist <= cam_to_lm_dist && cam_to_lm_dist <= max_dist);
    }
    void increase_num_observable(unsigned int num_observable = 1) { num_observable_ += num_observable; }  // :416-419
    observations_t get_observations() const { return observations_; }
    unsigned int predict_scale_level(const float cam_to_lm_dist, float num_scale_levels, float log_scale_factor) const {
        const float ratio = max_valid_dist_ / cam_to_lm_dist;
        const auto pred_scale_level = static_cast<int>(std::ceil(std::log(ratio) / log_scale_factor));
        if (pred_scale_level < 0) return 0;
        else if (num_scale_levels <= static_cast<unsigned int>(pred_scale_level)) return num_scale_levels - 1;
        else return static_cast<unsigned int>(pred_scale_level);
    }
    unsigned int id_;
    Vec3_t pos_w_, mean_normal_;
    float min_valid_dist_ = 0, max_valid_dist_ = 0;
    cv::Mat descriptor_;
    bool has_observation_ = true;
    bool will_be_erased_ = false;
    unsigned int num_observable_ = 1;
    observations_t observations_;
    // observations as (keyframe id -> keypoint index): enough for is_observed_in_keyframe / get_index_in_keyframe
    std::map<unsigned int, unsigned int> obs_by_keyfrm_id_;
};
}  // namespace data
}  // namespace stella_vslam
#endif
