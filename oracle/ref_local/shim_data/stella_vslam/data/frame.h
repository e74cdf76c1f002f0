// Stand-in (see ../../README.md) for data/frame.h: plain data + the accessors the matcher sources call, plus what module/frame_tracker.cc and the
// tracker fixture (ref_trk_exports.cc) use: erase_landmark_with_index, the reference's constructor, ref_keyfrm_, and can_observe -- a restatement of
// data/frame.cc:59-85 over the stand-in camera (whose reprojection is the oracle's; the real frame.cc is pinned in libsvref_frm.so).
#ifndef SVGPU_SHIM_STELLA_DATA_FRAME_H
#define SVGPU_SHIM_STELLA_DATA_FRAME_H
#include <cmath>
#include <memory>
#include <unordered_map>
#include "stella_vslam/camera/base.h"
#include "stella_vslam/data/bow_vocabulary.h"
#include "stella_vslam/data/common.h"
#include "stella_vslam/data/frame_observation.h"
#include "stella_vslam/data/landmark.h"
#include "stella_vslam/feature/orb_params.h"
namespace stella_vslam {
namespace data {
class keyframe;
struct marker2d {};
class frame {
public:
    frame() = default;
    frame(unsigned int id, camera::base* camera, const feature::orb_params* orb_params) : id_(id), camera_(camera), orb_params_(orb_params) {}
    // data/frame.cc:15-20
    frame(const unsigned int frame_id, const double timestamp, camera::base* camera, feature::orb_params* orb_params, const frame_observation frm_obs,
          const std::unordered_map<unsigned int, marker2d>& markers_2d)
        : id_(frame_id), timestamp_(timestamp), camera_(camera), orb_params_(orb_params), frm_obs_(frm_obs), markers_2d_(markers_2d),
          landmarks_(frm_obs_.undist_keypts_.size(), nullptr) {}
    void set_pose_cw(const Mat44_t& pose_cw) {
        pose_cw_ = pose_cw;
        pose_is_valid_ = true;
    }
    Mat44_t get_pose_cw() const { return pose_cw_; }
    Mat33_t get_rot_cw() const { return pose_cw_.block<3, 3>(0, 0); }
    Vec3_t get_trans_cw() const { return pose_cw_.block<3, 1>(0, 3); }
    Vec3_t get_trans_wc() const { return -(get_rot_cw().transpose()) * get_trans_cw(); }
    bool pose_is_valid() const { return pose_is_valid_; }
    void add_landmark(const std::shared_ptr<landmark>& lm, const unsigned int idx) { landmarks_.at(idx) = lm; }
    std::shared_ptr<landmark> get_landmark(const unsigned int idx) const { return landmarks_.at(idx); }
    std::vector<std::shared_ptr<landmark>> get_landmarks() const { return landmarks_; }
    bool has_landmark(const std::shared_ptr<landmark>& lm) const {
        for (const auto& l : landmarks_)
            if (l == lm) return true;
        return false;
    }
    void set_landmarks(const std::vector<std::shared_ptr<landmark>>& lms) { landmarks_ = lms; }
    void erase_landmarks() { std::fill(landmarks_.begin(), landmarks_.end(), nullptr); }
    void erase_landmark_with_index(const unsigned int idx) { landmarks_.at(idx) = nullptr; }  // data/frame.cc:101-105
    // data/frame.cc:59-85
    bool can_observe(const std::shared_ptr<landmark>& lm, const float ray_cos_thr, Vec2_t& reproj, float& x_right, unsigned int& pred_scale_level) const {
        const Vec3_t pos_w = lm->get_pos_in_world();
        const bool in_image = camera_->reproject_to_image(get_rot_cw(), get_trans_cw(), pos_w, reproj, x_right);
        if (!in_image) return false;
        const Vec3_t cam_to_lm_vec = pos_w - get_trans_wc();
        const auto cam_to_lm_dist = cam_to_lm_vec.norm();
        const auto margin_far = 1.3;
        const auto margin_near = 1.0 / margin_far;
        if (!lm->is_inside_in_orb_scale(cam_to_lm_dist, margin_far, margin_near)) return false;
        const Vec3_t obs_mean_normal = lm->get_obs_mean_normal();
        const auto ray_cos = cam_to_lm_vec.dot(obs_mean_normal) / cam_to_lm_dist;
        if (ray_cos < ray_cos_thr) return false;
        pred_scale_level = lm->predict_scale_level(cam_to_lm_dist, this->orb_params_->num_levels_, this->orb_params_->log_scale_factor_);
        return true;
    }
    std::vector<unsigned int> get_keypoints_in_cell(const float ref_x, const float ref_y, const float margin, const int min_level = -1,
                                                    const int max_level = -1) const {
        return data::get_keypoints_in_cell(camera_, frm_obs_, ref_x, ref_y, margin, min_level, max_level);
    }
    unsigned int id_ = 0;
    double timestamp_ = 0.0;
    camera::base* camera_ = nullptr;
    const feature::orb_params* orb_params_ = nullptr;
    frame_observation frm_obs_;
    std::unordered_map<unsigned int, marker2d> markers_2d_;
    bow_vector bow_vec_;
    bow_feature_vector bow_feat_vec_;
    std::shared_ptr<keyframe> ref_keyfrm_ = nullptr;
    std::vector<std::shared_ptr<landmark>> landmarks_;

private:
    Mat44_t pose_cw_ = Mat44_t::Identity();
    bool pose_is_valid_ = false;
};
}  // namespace data
}  // namespace stella_vslam
#endif
